"""Sweep kernel configs over the 512-per-GPU GTEx GEMM shapes (back-to-back GPU time).  Measurement tool only."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
from small_kernel_bench import gemm_case  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G, H, Dd = 18154, 256, 18161
V = [(-1, 0), (0, 0), (5, 0), (6, 0), (7, 0), (1, 0)]
CASES = [
    ('disc_dw_in TN', Dd, H, 3 * B, 1, 0, V),
    ('disc_dw_in head', 16384, H, 3 * B, 1, 0, [(-1, 0), (0, 0)]),
    ('disc_dw_in tail', Dd - 16384, H, 3 * B, 1, 0, [(-1, 0), (0, 9), (0, 18), (0, 6)]),
    ('gen_dw_out TN', H, G, B, 1, 0, V),
    ('disc_dx    NT', B, G, H, 0, 1, V),
    ('gen_fwd_out NN', 6 * B, G, H, 0, 0, [(-1, 0), (0, 0), (5, 0)]),
    ('disc_fwd_in NN', 2 * B, H, Dd, 0, 0, [(-1, 0), (0, 16), (0, 32), (0, 24), (5, 16), (5, 32), (1, 8), (1, 16)]),
    ('gp2_in     NN', B, H, G, 0, 0, [(-1, 0), (0, 32), (0, 64), (5, 32), (1, 16), (1, 32)]),
    ('gen_bwd_out NT', B, H, G, 0, 1, [(-1, 0), (0, 32), (0, 64), (1, 16)]),
    ('hidden NN 3B', 3 * B, H, H, 0, 0, [(-1, 0), (0, 0), (1, 0), (3, 0)]),
    ('hidden TN 3B', H, H, 3 * B, 1, 0, [(-1, 0), (0, 0), (1, 0), (1, 4), (3, 8)]),
]
for name, M, N, K, ta, tb, variants in CASES:
    for cfg, sk in variants:
        t, keep = gemm_case(M, N, K, ta, tb, cfg, sk)
        tf = 2.0 * M * N * K / t / 1e6
        print('%-15s M=%5d N=%5d K=%5d cfg=%2d sk=%3d  %8.2f us  %6.1f TF (%4.1f%%)' % (name, M, N, K, cfg, sk, t, tf, tf / 1.573), flush=True)
        del keep
