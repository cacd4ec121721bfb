#!/usr/bin/env python3
"""Is the batch-32 iteration loop host- or GPU-bound?  Times the enqueue loop (no sync) and the
whole run, and the host cost of one train_iteration call with the GPU idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from adversarial_gene_expression_amd import tf_utils as T

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
G, V, n_num, H, Z = bench.WORKLOADS['gtex']
gen = T.Generator(G, V, n_num, H, Z, seed=1)
disc = T.Critic(G, V, n_num, H, 1, seed=2)
gopt, dopt = T.RMSprop(5e-4), T.RMSprop(5e-4)
x, cc, nc = bench.synthetic_batch_pool(G, V, n_num, B * 8, seed=100)
T.set_seed(1000)
EAGER = bool(os.environ.get('HB_EAGER'))   # the same launch sequence queued directly (no graph)
def it(i):
    s = slice((i % 8) * B, (i % 8 + 1) * B)
    if EAGER:
        pair = T._pair(gen, disc, B, 6)
        T._rng.offset += pair.iteration_eager(x[s], cc[s], nc[s], gen, disc, gopt, dopt, 5, 10.0, T._rng.seed, T._rng.offset)
    else:
        T.train_iteration(x[s], cc[s], nc[s], gen, disc, gopt, dopt, 5)
for i in range(30): it(i)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for i in range(n): it(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue %.1f us/iter, total %.1f us/iter' % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# host cost with an idle GPU: sync after every call
tt = 0.0
for i in range(100):
    torch.cuda.synchronize()
    a = time.perf_counter()
    it(i)
    tt += time.perf_counter() - a
print('host call %.1f us (GPU idle)' % (tt / 100 * 1e6))
