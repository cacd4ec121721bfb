"""One agx_gemm_f32 shape repeated (for rocprofv3 --pmc / --kernel-trace).  usage: one_case.py M N K ta tb cfg sk [reps] [rowsq]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from adversarial_gene_expression_amd import _dev, _lib
lib = _lib.load(); dev = _dev.device()
M, N, K, ta, tb, cfg, sk = [int(v) for v in sys.argv[1:8]]
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
PAD = int(os.environ.get('LDPAD', '4'))     # leading dimensions rounded up to this many floats
rp = lambda n: (n + PAD - 1) // PAD * PAD
A = torch.randn((K, rp(M)) if ta else (M, rp(K)), device=dev)
B = torch.randn((N, rp(K)) if tb else (K, rp(N)), device=dev)
out = _dev.empty((M, rp(N)))
ws = _dev.empty(max(1, max(sk, 64) * M * N) if M * N <= (1 << 20) else 1)
g = _lib.GemmDesc()
g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
g.M, g.N, g.K = M, N, K
g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
g.ta, g.tb, g.alpha = ta, tb, 1.0
g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
g.force_cfg, g.force_splitk = cfg, sk
st = _dev.stream()
for _ in range(3):
    _lib.check(lib.agx_gemm_f32(st, g))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    lib.agx_gemm_f32(st, g)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e3
print('M=%d N=%d K=%d ta=%d tb=%d cfg=%d sk=%d: %.2f us  %.1f TF' % (M, N, K, ta, tb, cfg, sk, t, 2.0 * M * N * K / t / 1e6))
