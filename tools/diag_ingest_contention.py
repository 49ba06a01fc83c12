"""Which copy slows the uint8 ingest kernel down when it runs beside it: the H2D (SDMA) or the D2H (blit kernel)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stabstitch2_amd  # noqa
import torch
from stabstitch2_amd import ops
dev = torch.device('cuda:0')
n = 32
src_h = torch.randint(0, 255, (n, 720, 1280, 3), dtype=torch.uint8).pin_memory()
f = src_h.to(dev)
dbuf = torch.empty_like(f)
dout = torch.randint(0, 255, (n, 740, 1882, 3), dtype=torch.uint8, device=dev)
hout = torch.empty(tuple(dout.shape), dtype=torch.uint8).pin_memory()
lr = torch.empty((n, 3, 360, 480), device=dev)
s_up, s_dn, s_c = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def ingest_ms(pre):
    torch.cuda.synchronize()
    pre()
    time.sleep(0.0005)
    with torch.cuda.stream(s_c):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.ingest_u8(f, want_hr=False, lr_out=lr)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def h2d():
    with torch.cuda.stream(s_up):
        dbuf.copy_(src_h, non_blocking=True)


def d2h():
    with torch.cuda.stream(s_dn):
        hout.copy_(dout, non_blocking=True)


for name, pre in (('alone', lambda: None), ('beside H2D', h2d), ('beside D2H', d2h), ('beside both', lambda: (h2d(), d2h()))):
    ingest_ms(pre)
    print('%-12s ingest %s ms' % (name, ['%.3f' % ingest_ms(pre) for _ in range(5)]))
t0 = time.perf_counter(); d2h(); torch.cuda.synchronize(); print('d2h alone %.2f ms' % ((time.perf_counter() - t0) * 1e3))
for var in ('GPU_FORCE_BLIT_COPY_SIZE', 'HSA_ENABLE_SDMA', 'GPU_BLIT_ENGINE_TYPE'):
    print(var, os.environ.get(var))
