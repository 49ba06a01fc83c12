import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
x = torch.randn(1000, device='cuda')
try:
    g = torch.cuda.CUDAGraph(keep_graph=True)
except TypeError as e:
    print('no keep_graph', e); sys.exit(0)
with torch.cuda.graph(g):
    y = x * 2
    z = y + 1
    w = z.sin()
raw = g.raw_cuda_graph()
print('raw', raw)
hip = ctypes.CDLL('libamdhip64.so')
n = ctypes.c_size_t(0)
rc = hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n))
print('rc', rc, 'nodes', n.value)
g.replay(); torch.cuda.synchronize(); print(float(w.sum()))
