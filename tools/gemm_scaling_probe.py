import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tools'))
import torch
from small_kernel_bench import gemm_case
for (M,N,K,ta,tb) in [(18161,256,96,1,0),(18161,32,96,1,0),(18161,256,32,1,0),(18161,256,8,1,0),(4540,256,96,1,0),(18161,256,96,0,0),(18161,256,96,0,1),
                      (32,18154,256,0,0),(32,18154,64,0,0),(32,4538,256,0,0),(96,256,18161,0,0),(96,256,4540,0,0),(96,32,18161,0,0),(32,256,18161,0,0)]:
    for cfg in (3,2,1):
        t,keep=gemm_case(M,N,K,ta,tb,cfg,0); del keep
        print('M=%5d N=%5d K=%5d ta=%d tb=%d cfg=%d  %7.2f us'%(M,N,K,ta,tb,cfg,t),flush=True)
