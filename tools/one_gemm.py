"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes).  Measurement tool only.
usage: one_gemm.py M N K ta tb cfg sk [iters]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
import torch  # noqa: E402
from small_kernel_bench import gemm_case  # noqa: E402

M, N, K, ta, tb, cfg, sk = [int(v) for v in sys.argv[1:8]]
t, keep = gemm_case(M, N, K, ta, tb, cfg, sk)
print('M=%d N=%d K=%d ta=%d tb=%d cfg=%d sk=%d: %.2f us' % (M, N, K, ta, tb, cfg, sk, t))
