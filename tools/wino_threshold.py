import sys, os, torch
sys.path.insert(0, '/root/repo')
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n, h, w, cin, cout) in ((2,90,120,64,64),(4,90,120,64,64),(8,90,120,64,64),(2,45,60,128,128),(4,45,60,128,128),(8,45,60,128,128),(16,45,60,128,128),
                             (2,23,30,256,256),(8,23,30,256,256),(16,23,30,256,256),(32,23,30,256,256),(1,45,60,160,64),(2,45,60,124,64),(4,45,60,124,64)):
    x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 1, 3, 3, cin, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    out = torch.empty(n, h, w, cout, device=dev)
    tw = t(lambda: ops.conv_winograd(x, wt, b, None, relu=True, out=out))
    ops.WINOGRAD = False
    ti = t(lambda: ops.conv(x, wt, b, None, relu=True, out=out))
    ops.WINOGRAD = True
    import math
    tiles = math.ceil(math.ceil(h/2)/8)*math.ceil(math.ceil(w/2)/4)
    print('n %2d %3dx%3d %3d->%3d: winograd %.1f us, igemm %.1f us  (wino wgs ~%d, uses=%d)' % (n, h, w, cin, cout, tw, ti, n*tiles*(cout//64),
          ops._uses_winograd(1,3,3,1,(0,1,1),cin,cout,h,w,n)))
