"""Sweep kernel configs over the batch-32 GTEx GEMM shapes (back-to-back GPU time).  Measurement tool only.
usage: skinny_bench.py [B]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G, H, Dd = 18154, 256, 18161
CASES = [
    ('disc_dw_in TN', Dd, H, 3 * B, 1, 0, [(-1, 0), (0, 0), (1, 0), (2, 0), (3, 0), (5, 0), (8, 0)]),
    ('gen_dw_out TN', H, G, B, 1, 0, [(-1, 0), (0, 0), (1, 0), (2, 0), (3, 0), (8, 0)]),
    ('disc_dx    NT', B, G, H, 0, 1, [(-1, 0), (1, 0), (2, 0), (3, 0), (8, 0)]),
    ('gen_fwd_out NN', 6 * B, G, H, 0, 0, [(-1, 0), (0, 0), (1, 0), (2, 0), (3, 0), (5, 0)]),
    ('disc_fwd_in NN', 2 * B, H, Dd, 0, 0, [(-1, 0), (3, 16), (3, 32), (3, 48), (3, 64), (1, 32), (1, 64), (2, 32), (2, 64)]),
    ('disc_fwd_in1 NN', B, H, Dd, 0, 0, [(-1, 0), (3, 32), (3, 64), (3, 96)]),
    ('gp2_in     NN', B, H, G, 0, 0, [(-1, 0), (3, 32), (3, 48), (3, 64), (3, 96), (2, 32), (2, 64)]),
    ('gen_bwd_out NT', B, H, G, 0, 1, [(-1, 0), (3, 24), (3, 32), (3, 48), (3, 64), (2, 32)]),
]
for name, M, N, K, ta, tb, variants in CASES:
    for cfg, sk in variants:
        t, keep = gemm_case(M, N, K, ta, tb, cfg, sk)
        print('%-15s M=%5d N=%5d K=%5d cfg=%2d sk=%3d  %8.2f us' % (name, M, N, K, cfg, sk, t), flush=True)
        del keep
