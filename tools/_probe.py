import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case
for (M,N,K,ta,tb) in [(16384,256,1536,1,0),(16384,256,6144,1,0),(18161,256,1536,1,0),(4096,4096,4096,0,0),(512,18154,256,0,1),(1024,256,18161,0,0),(3072,18154,256,0,0)]:
  for cfg,sk in ((-1,0),(0,0),(4,0),(1,0)):
    t, keep = gemm_case(M,N,K,ta,tb,cfg,sk)
    print('M=%5d N=%5d K=%5d ta=%d tb=%d cfg=%2d  %8.2f us  %.1f TF' % (M,N,K,ta,tb,cfg,t, 2.0*M*N*K/t/1e6), flush=True)
