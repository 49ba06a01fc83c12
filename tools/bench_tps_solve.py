"""ss_tps_solve_shared_target: time per launch for 2 systems (a streaming push) and 64 (a 32-frame clip), residual against an
fp64 numpy solve of the same system.      python tools/bench_tps_solve.py"""
import torch, time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stabstitch2_amd import ops, pipeline
dev=torch.device('cuda:0')
torch.manual_seed(0)
nr = pipeline.norm_rigid_mesh(720,1280,dev)
for n in (2,64):
    src = (nr.view(1,63,2) + 0.05*torch.randn(n,63,2,device=dev)).contiguous()
    for _ in range(3): T=ops.tps_solve_shared(src, nr)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): T=ops.tps_solve_shared(src, nr)
    e1.record(); torch.cuda.synchronize()
    print('n=%d: %.1f us per launch'%(n, e0.elapsed_time(e1)*1e3/20))
    # residual check in fp64 on the host
    import numpy as np
    s=src[0].double().cpu().numpy(); t=nr.view(63,2).double().cpu().numpy()
    d2=((s[:,None,:]-s[None,:,:])**2).sum(-1); K=d2*np.log(d2+1e-6)
    L=np.zeros((66,66)); L[:63,0]=1; L[:63,1:3]=s; L[:63,3:]=K; L[63,3:]=1; L[64:,3:]=s.T
    rhs=np.zeros((66,2)); rhs[:63]=t
    ref=np.linalg.solve(L,rhs).T
    print('  max |T - fp64 numpy| =', np.abs(T[0].double().cpu().numpy()-ref).max(), ' |T|max', np.abs(ref).max())
