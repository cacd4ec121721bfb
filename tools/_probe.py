import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case
for (M,K) in [(128,32768),(128,65536),(1024,32768),(1024,65536),(16384,6144),(16384,12288)]:
  for cfg in (23,0):
    t, keep = gemm_case(M,128,K,1,0,cfg,1)
    print('M=%5d N=128 K=%5d cfg=%2d  %8.2f us  per K-tile %.3f us' % (M,K,cfg,t, t/(K/32)), flush=True)
