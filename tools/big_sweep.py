"""Config sweep of agx_gemm_f32 over the 512-rows-per-GPU GTEx shapes (back-to-back GPU time).
Measurement tool only.  usage: big_sweep.py [B]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from adversarial_gene_expression_amd import _dev, _lib  # noqa: E402

lib = _lib.load()
dev = _dev.device()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G, H, Dd = 18154, 256, 18161


def case(M, N, K, ta, tb, cfg, sk, n=20):
    A = torch.randn((K, _dev.r4(M)) if ta else (M, _dev.r4(K)), device=dev)
    Bm = torch.randn((N, _dev.r4(K)) if tb else (K, _dev.r4(N)), device=dev)
    out = _dev.empty((M, _dev.r4(N)))
    ws = _dev.empty(max(1, max(sk, 64) * M * N) if M * N <= (1 << 20) else 1)
    g = _lib.GemmDesc()
    g.A, g.B, g.C = A.data_ptr(), Bm.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), Bm.stride(0), out.stride(0)
    g.ta, g.tb, g.alpha = ta, tb, 1.0
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    g.force_cfg, g.force_splitk = cfg, sk
    st = _dev.stream()
    for _ in range(2):
        _lib.check(lib.agx_gemm_f32(st, g))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        lib.agx_gemm_f32(st, g)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [
    ('disc_fwd_in NN', 2 * B, H, Dd, 0, 0, [(-1, 0)] + [(c, s) for c in (0, 4, 5, 6, 7, 1) for s in (8, 16, 32)]),
    ('gp2_in      NN', B, H, G, 0, 0, [(-1, 0)] + [(c, s) for c in (0, 4, 5, 6, 1) for s in (16, 32, 64)]),
    ('gen_bwd_out NT', B, H, G, 0, 1, [(-1, 0)] + [(c, s) for c in (0, 4, 6, 1) for s in (16, 32, 64)]),
    ('disc_dx     NT', B, G, H, 0, 1, [(c, 1) for c in (-1, 0, 1, 4, 5, 6, 7)]),
    ('gen_fwd_out NN', 6 * B, G, H, 0, 0, [(c, 1) for c in (-1, 0, 1, 4, 5, 6, 7)]),
    ('disc_dw_in  TN', Dd, H, 3 * B, 1, 0, [(c, 1) for c in (-1, 0, 1, 4, 5, 6, 7)]),
    ('gen_dw_out  TN', H, G, B, 1, 0, [(c, 1) for c in (-1, 0, 1, 4, 5, 6, 7)]),
]
for name, M, N, K, ta, tb, variants in CASES:
    for cfg, sk in variants:
        t = case(M, N, K, ta, tb, cfg, sk)
        tf = 2.0 * M * N * K / t / 1e6
        print('%-15s M=%5d N=%5d K=%5d cfg=%2d sk=%3d  %8.2f us  %6.1f TF (%4.1f%%)' % (name, M, N, K, cfg, sk, t, tf, tf / 1.573), flush=True)
