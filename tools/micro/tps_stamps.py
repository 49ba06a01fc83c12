import sys, ctypes, torch, numpy as np
dev = torch.device('cuda:0')
ys, xs = torch.meshgrid(torch.linspace(-1, 1, 7), torch.linspace(-1, 1, 9), indexing='ij')
rigid = torch.stack([xs, ys], -1).reshape(1, 63, 2)
g = torch.Generator().manual_seed(3)
n = 2
src = (rigid + 0.05 * torch.randn((n, 63, 2), generator=g)).to(dev).contiguous()
tgt = rigid.repeat(n, 1, 1).to(dev).contiguous()
NW = int(sys.argv[2]) if len(sys.argv) > 2 else 4
T = torch.zeros((1024 + NW * 68 * 4 * 2 + 64,), device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib = ctypes.CDLL(sys.argv[1])
lib.ss_tps_solve.restype = ctypes.c_int
lib.ss_tps_solve.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]
for _ in range(3):
    lib.ss_tps_solve(P(src), P(tgt), P(T), n, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
st = T[1024:1024 + NW * 68 * 4 * 2].cpu().numpy().view(np.uint64).astype(np.int64).reshape(NW, 68, 4)
t0 = st[0, 0, 0]
print('total ticks first stamp -> last: %d' % (st[st > 0].max() - t0))
print('step | per wave: arrive-at-barrier, leave barrier(+), state read (+), search done (+) [ticks relative to previous step leave]')
for col in list(range(0, 8)) + list(range(15, 21)) + list(range(48, 54)) + [63, 64, 65]:
    row = []
    for q in range(NW):
        a = st[q, col]
        row.append('w%d: arr %6d bar +%4d rd +%4d srch %s' % (q, a[0] - t0, a[1] - a[0], a[2] - a[1], ('+%4d' % (a[3] - a[2])) if a[3] > a[2] else '    -'))
    print('%2d  ' % col + ' | '.join(row))
lv = st[:, :66, 1].max(axis=0)
d = np.diff(lv)
print('leave-barrier to leave-barrier per step: median %d, mean %.1f, first 17: %s' % (np.median(d), d.mean(), d[:17].tolist()))
print('steps 17..33: %s' % d[17:34].tolist())
print('steps 51..65: %s' % d[50:].tolist())
