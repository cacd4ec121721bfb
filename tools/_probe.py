import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case
for (M,N,K,ta,tb,cfg,sk) in [(18161,256,8,1,0,8,0),(18161,256,8,1,0,3,0),(18161,256,8,1,0,0,0),(18161,256,32,1,0,8,0),(18161,256,96,1,0,8,0),(18161,256,192,1,0,8,0),(18161,256,384,1,0,8,0),
   (18176,256,96,1,0,8,0),(16384,256,96,1,0,8,0),(4096,256,96,1,0,8,0),(8192,256,96,1,0,8,0),(18161,64,96,1,0,8,0),(18161,128,96,1,0,8,0)]:
    t, keep = gemm_case(M,N,K,ta,tb,cfg,sk)
    print('M=%5d N=%5d K=%5d ta=%d tb=%d cfg=%2d sk=%3d  %8.2f us' % (M,N,K,ta,tb,cfg,sk,t), flush=True)
