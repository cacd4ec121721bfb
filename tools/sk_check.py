"""Stream-K TN product against float64 at the chunk shapes of the data-parallel path (measurement / diagnosis tool)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adversarial_gene_expression_amd import _dev, _lib, ops

lib = _lib.load()
torch.manual_seed(0)
for M, N, K in [(18161, 256, 1280), (18161, 256, 1536), (4540, 256, 1280), (4540, 256, 1536), (4541, 256, 1280), (4544, 256, 1280), (4540, 256, 1024)]:
    ld = 18168
    A = torch.randn(K, ld, device='cuda')
    B = torch.randn(K, _dev.r4(N), device='cuda')
    want = (A[:, 4:4 + M].double().t() @ B[:, :N].double())
    flags = torch.zeros(int(lib.agx_gemm_sk_flag_words()), dtype=torch.int32, device='cuda')
    for name, kw in (('static', {}), ('stream-K', dict(sk_flags=flags))):
        out = torch.full((M, _dev.r4(N)), float('nan'), device='cuda')
        ops.gemm(A[:, 4:], B, M, N, K, ta=1, out=out, **kw)
        torch.cuda.synchronize()
        err = (out[:, :N].double() - want).abs()
        print('%-9s %6d x %4d x %5d  max err %.3e  rows with err > 1e-2: %d  flags %d' % (
            name, M, N, K, float(err.max()), int((err.max(1).values > 1e-2).sum()), int(flags.abs().sum())), flush=True)
