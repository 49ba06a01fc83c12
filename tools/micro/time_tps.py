import sys, ctypes, torch
dev = torch.device('cuda:0')
ys, xs = torch.meshgrid(torch.linspace(-1, 1, 7), torch.linspace(-1, 1, 9), indexing='ij')
rigid = torch.stack([xs, ys], -1).reshape(1, 63, 2)
g = torch.Generator().manual_seed(3)
n = 2
src = (rigid + 0.05 * torch.randn((n, 63, 2), generator=g)).to(dev).contiguous()
tgt = rigid.repeat(n, 1, 1).to(dev).contiguous()
T = torch.empty((n, 2, 66), device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
ref = None
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.ss_tps_solve.restype = ctypes.c_int
    lib.ss_tps_solve.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
    f = lambda: lib.ss_tps_solve(P(src), P(tgt), P(T), n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = T.clone()
    print('%-40s %.2f us   same as first: %s' % (path, e0.elapsed_time(e1) * 10, torch.equal(T, ref)))
