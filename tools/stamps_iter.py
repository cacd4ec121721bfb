"""Per-workgroup phase timeline of ONE gemm_kernel launch class inside the captured training iteration (diagnostic
build tools/ab/libagx_stamps.so).  usage: stamps_iter.py B M K   (only launches with this M and K are stamped; the
last one of the iteration is reported, e.g. 32 18161 96 = the critic's fused dW_1 + RMSprop at 32 rows)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'tools'))
from stamps_common import report
import numpy as np
import torch
import bench
from adversarial_gene_expression_amd import _dev, _lib, tf_utils as T
_lib.LIB_PATH = os.path.join(R, 'tools', 'ab', 'libagx_stamps.so')
lib = _lib.load(); dev = _dev.device()
B, M, K = [int(v) for v in sys.argv[1:4]]
G, V, n_num, H, Z = bench.WORKLOADS['gtex']
gen = T.Generator(G, V, n_num, H, Z, seed=1)
disc = T.Critic(G, V, n_num, H, 1, seed=2)
gopt, dopt = T.RMSprop(5e-4), T.RMSprop(5e-4)
x, cc, nc = bench.synthetic_batch_pool(G, V, n_num, B * 8, seed=100)
T.set_seed(1000)
NB = 1 << 15
st_buf = torch.zeros(NB * 8, dtype=torch.int64, device=dev)
lib.agx_dbg_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]; lib.agx_dbg_stamps.restype = C.c_int
assert lib.agx_dbg_stamps(st_buf.data_ptr(), M, K) == 0
for i in range(6):
    s = slice((i % 8) * B, (i % 8 + 1) * B)
    T.train_iteration(x[s], cc[s], nc[s], gen, disc, gopt, dopt, 5)
torch.cuda.synchronize()
st_buf.zero_(); torch.cuda.synchronize()
s = slice(6 * B, 7 * B)
T.train_iteration(x[s], cc[s], nc[s], gen, disc, gopt, dopt, 5)
torch.cuda.synchronize()
report(st_buf, NB)
