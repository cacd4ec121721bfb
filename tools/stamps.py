"""Per-workgroup phase timeline of gemm_kernel from the diagnostic build (tools/ab/libagx_stamps.so, -DAGX_STAMPS).
usage: stamps.py M N K ta tb cfg sk   (stamps: start | first tile staged | main loop done | epilogue issued | drained)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
from stamps_common import report
import torch
from adversarial_gene_expression_amd import _dev, _lib
_lib.LIB_PATH = os.path.join(R, 'tools', 'ab', 'libagx_stamps.so')
lib = _lib.load(); dev = _dev.device()
M, N, K, ta, tb, cfg, sk = [int(v) for v in sys.argv[1:8]]
A = torch.randn((K, _dev.r4(M)) if ta else (M, _dev.r4(K)), device=dev)
B = torch.randn((N, _dev.r4(K)) if tb else (K, _dev.r4(N)), device=dev)
out = _dev.empty((M, _dev.r4(N)))
ws = _dev.empty(max(1, max(sk, 64) * M * N) if M * N <= (1 << 20) else 1)
NB = 1 << 15
st_buf = torch.zeros(NB * 8, dtype=torch.int64, device=dev)
lib.agx_dbg_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]; lib.agx_dbg_stamps.restype = C.c_int
assert lib.agx_dbg_stamps(st_buf.data_ptr(), -1, -1) == 0
g = _lib.GemmDesc()
g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
g.M, g.N, g.K = M, N, K
g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
g.ta, g.tb, g.alpha = ta, tb, 1.0
g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
g.force_cfg, g.force_splitk = cfg, sk
s = _dev.stream()
for _ in range(20):
    _lib.check(lib.agx_gemm_f32(s, g))
torch.cuda.synchronize()
st_buf.zero_(); torch.cuda.synchronize()
_lib.check(lib.agx_gemm_f32(s, g)); torch.cuda.synchronize()
report(st_buf, NB)
