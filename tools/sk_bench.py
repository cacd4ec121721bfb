"""Stream-K A/B: the TN weight-gradient product with and without the stream-K hand-off words (same box, same process)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adversarial_gene_expression_amd import _dev, _lib, ops

lib = _lib.load()
shapes = [(18161, 256, 1536), (256, 18154, 512), (18161, 512, 1536)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
for M, N, K in shapes:
    A = torch.randn(K, _dev.r4(M), device='cuda')
    B = torch.randn(K, _dev.r4(N), device='cuda')
    out = _dev.empty((M, _dev.r4(N)))
    flags = torch.zeros(int(lib.agx_gemm_sk_flag_words()), dtype=torch.int32, device='cuda')
    for name, kw in (('static', {}), ('static128', dict(force_cfg=0)), ('stream-K', dict(sk_flags=flags))):
        for _ in range(3):
            ops.gemm(A, B, M, N, K, ta=1, out=out, **kw)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 20
        ev[0].record()
        for _ in range(n):
            ops.gemm(A, B, M, N, K, ta=1, out=out, **kw)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / n * 1e3
        print('%-9s %6d x %5d x %5d: %8.1f us  %6.1f TFLOP/s  flags %d' % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6,
                                                                           int(flags.abs().sum())))
