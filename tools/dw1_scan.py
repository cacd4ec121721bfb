"""K scan of the critic's layer-1 weight-gradient shape (TN 18161 x 256): where does its time go?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case
for M in (18161, 9080, 4540):
    for K in (8, 32, 64, 96, 192, 384):
        for cfg in (1, 2):
            t, keep = gemm_case(M, 256, K, 1, 0, cfg, 0)
            print('TN M=%5d N=256 K=%3d cfg=%d  %7.2f us' % (M, K, cfg, t), flush=True)
            del keep
