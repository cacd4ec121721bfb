"""Config sweep of agx_gemm_f32 at the batch-32 generator output layer (192 x 18154 x 256) and friends."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import torch
from adversarial_gene_expression_amd import _dev, _lib
lib = _lib.load(); dev = _dev.device()
def case(M, N, K, ta, tb, cfg, sk, n=50, bias=True):
    A = torch.randn((K, _dev.r4(M)) if ta else (M, _dev.r4(K)), device=dev)
    Bm = torch.randn((N, _dev.r4(K)) if tb else (K, _dev.r4(N)), device=dev)
    out = _dev.empty((M, _dev.r4(N)))
    b = torch.randn(_dev.r4(N), device=dev)
    ws = _dev.empty(max(1, max(sk, 64) * M * N) if M * N <= (1 << 20) else 1)
    g = _lib.GemmDesc()
    g.A, g.B, g.C = A.data_ptr(), Bm.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), Bm.stride(0), out.stride(0)
    g.ta, g.tb, g.alpha = ta, tb, 1.0
    if bias: g.bias = b.data_ptr()
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    g.force_cfg, g.force_splitk = cfg, sk
    st = _dev.stream()
    for _ in range(3):
        rc = lib.agx_gemm_f32(st, g)
        if rc != 0: return float('nan')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.agx_gemm_f32(st, g)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K, ta, tb in [('gen_fwd_out b32', 192, 18154, 256, 0, 0), ('gen_fwd_out ecoli', 192, 4297, 128, 0, 0), ('gen_fwd_out wide', 192, 20000, 1024, 0, 0)]:
    for cfg in [-1, 0, 1, 2, 3] + list(range(4, 16)):
        t = case(M, N, K, ta, tb, cfg, 1)
        print('%-18s M=%5d N=%5d K=%5d cfg=%2d  %8.2f us  %6.1f TF  %6.0f GB/s' % (name, M, N, K, cfg, t, 2.0 * M * N * K / t / 1e6, 4.0 * (M * K + K * N + M * N) / t / 1e3), flush=True)
