import sys; sys.path.insert(0,'/root/repo')
import torch
from adversarial_gene_expression_amd import _dev,_lib
lib=_lib.load(); dev=_dev.device(); out=_dev.zeros(4); st=_dev.stream()
for blocks,n in [(1,100),(1,1000),(256,100),(256,1000),(1024,1000),(1024,20000)]:
    for rep in range(3):
        lib.agx_prof_clock(st,blocks,n,out.data_ptr()); torch.cuda.synchronize()
    print(blocks,n,out.cpu().numpy()[:2])
# after sustained load
for rep in range(200): lib.agx_prof_clock(st,1024,2000,out.data_ptr())
torch.cuda.synchronize(); print('sustained',out.cpu().numpy()[:2])
lib.agx_prof_clock(st,1,100,out.data_ptr()); torch.cuda.synchronize(); print('small after sustained',out.cpu().numpy()[:2])
