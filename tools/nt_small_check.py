import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adversarial_gene_expression_amd import _dev, ops
torch.manual_seed(0)
for (M, N, K, lda, ldb, ldc) in [(16, 8, 37, 44, 37, 8), (16, 8, 37, 44, 40, 8), (40, 8, 37, 44, 37, 8), (24, 16, 130, 140, 132, 16)]:
    A = torch.randn(M, lda, device='cuda'); B = torch.randn(N, ldb, device='cuda')
    out = torch.full((M, ldc), float('nan'), device='cuda')
    ops.gemm(A[:, :K], B[:, :K], M, N, K, ta=0, tb=1, out=out)
    want = A[:, :K].double() @ B[:, :K].double().t()
    print('NT', M, N, K, lda, ldb, 'err %.3e' % float((out[:, :N].double() - want).abs().max()))
    # NN with mask, B operand ld = N
    A2 = torch.randn(40, 16, device='cuda'); B2 = torch.randn(16, 8, device='cuda'); mk = torch.randn(40, 8, device='cuda')
    out2 = torch.full((40, 8), float('nan'), device='cuda')
    ops.gemm(A2, B2, 40, 8, 16, out=out2, mask=mk)
    want2 = (A2.double() @ B2.double()) * (mk > 0).double()
    print('NN mask err %.3e' % float((out2.double() - want2).abs().max()))
