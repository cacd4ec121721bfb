"""Floor for small streaming kernels in a back-to-back launch sequence (same harness as gemm_case)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
from small_kernel_bench import timed, gemm_case
for rows in (4540, 9080, 18161, 36322):
    a = torch.empty(rows, 256, device='cuda'); b = torch.randn(rows, 256, device='cuda'); c = torch.randn(rows, 256, device='cuda')
    print('rows %5d (%.1f MB): fill %.2f us  copy %.2f us  add3 %.2f us  addcmul_ %.2f us' % (
        rows, rows * 1024 / 1e6, timed(lambda: a.fill_(1.0)), timed(lambda: a.copy_(b)),
        timed(lambda: torch.add(b, c, out=a)), timed(lambda: a.addcmul_(b, c))), flush=True)
e = torch.empty(1, device='cuda')
print('empty-ish kernel: %.2f us' % timed(lambda: e.fill_(0.0)))
