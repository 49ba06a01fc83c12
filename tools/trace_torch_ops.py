"""Which lines of the host code launch torch (non-libstabstitch) kernels inside one clip step: a TorchDispatchMode logs
every aten op that touches a device tensor, with the innermost stabstitch2_amd caller.   python tools/trace_torch_ops.py"""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stabstitch2_amd import synth, pipeline, _hip
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device('cuda:0')
_hip.lib(); torch.set_grad_enabled(False)
nets, sds = bench.build_nets(dev)
hr, lr = synth.make_clip_device(32, 720, 1280, seed=0, views=2, device=dev)
step = lambda: pipeline.run_two_view(hr[0], hr[1], lr[0], lr[1], nets, 'NORMAL', 'AVERAGE')
for _ in range(2): step()
torch.cuda.synchronize()
VIEW = ('view', 'reshape', 'permute', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'transpose', 'as_strided', 'alias',
        'detach', 't.default', '_unsafe_view', 'unbind', 'split', 'empty', 'sym_', 'narrow', 'size', 'stride', 'is_', 'numel', 'item', '_local_scalar')
counts = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEW):
            site = '?'
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'stabstitch2_amd' in fr.filename:
                    site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno); break
            counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, site), n in sorted(counts.items(), key=lambda kv: (-kv[1], kv[0])):
    print('%4d  %-40s %s' % (n, name, site))
print('total device-touching aten calls per clip:', sum(counts.values()))
