import sys; sys.path.insert(0,'/root/repo')
import torch
from stabstitch2_amd import ops, _hip
dev=torch.device('cuda:0')
for r in (5,3):
    for (n,h,w) in ((3,45,60),(2,9,12),(1,23,30)):
        a=torch.randn(n,h,w,128,device=dev); b=torch.randn(n,h,w,128,device=dev)
        _hip.lib().ss_cost_volume_set_tile(4); o4=ops.cost_volume(a,b,r); ob4=ops.cost_volume_bidir(a,b,r)
        _hip.lib().ss_cost_volume_set_tile(8); o8=ops.cost_volume(a,b,r); ob8=ops.cost_volume_bidir(a,b,r)
        print(r,(n,h,w), torch.equal(o4,o8), torch.equal(ob4,ob8))
