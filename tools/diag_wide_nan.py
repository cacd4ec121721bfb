"""Diagnostic: bench's batch order on the wide workload, GPU captured iteration vs fp64 oracle, step by step."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from oracle import cgan_oracle as O
from adversarial_gene_expression_amd import tf_utils as T, _dev, _lib
from adversarial_gene_expression_amd._dev import ptr
lib = _lib.load()
wl = 'wide'
G, V, n_num, H, Z = bench.WORKLOADS[wl]
B, nbc = 32, 5
gen = T.Generator(G, V, n_num, H, Z, seed=1)
disc = T.Critic(G, V, n_num, H, 1, seed=2)
def to_oracle(net):
    ws = net.get_weights(); nv = len(V)
    return {'emb': [w.astype(np.float64) for w in ws[:nv]], 'W': [w.astype(np.float64) for w in ws[nv::2]],
            'b': [w.astype(np.float64) for w in ws[nv + 1::2]]}
tg, td = O.to_torch(to_oracle(gen)), O.to_torch(to_oracle(disc))
ogo, odo = O.RMSprop(5e-4), O.RMSprop(5e-4)
go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
x, cc, nc = bench.synthetic_batch_pool(G, V, n_num, B * 8, seed=100)
torch.set_num_threads(16)
T.set_seed(1000)
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
for i in [0, 0, 1, 2]:
    s = slice((i % 8) * B, (i % 8 + 1) * B)
    off0 = T._rng.offset
    # regenerate this iteration's draws (eager RNG entry points advance the stream like the graph)
    zs, als = [], []
    for _ in range(nbc):
        zs.append(T.random_normal((B, Z)).cpu().numpy()); als.append(T.random_uniform((B,)).cpu().numpy().reshape(B, 1))
    zs.append(T.random_normal((B, Z)).cpu().numpy())
    T._rng.offset = off0
    pair = T._pair(gen, disc, B, nbc + 1)
    g = pair.iteration_graph(gen, disc, go, do, B, nbc, 10.0, T._rng.seed)
    losses = _dev.zeros((nbc + 1, 4))
    xb = x[s].contiguous()
    _lib.check(lib.agx_cgan_iter_launch(g, _dev.stream(), ptr(xb), xb.stride(0), ptr(cc[s].contiguous()), ptr(nc[s].contiguous()), T._rng.offset, ptr(losses)))
    T._rng.offset += int(lib.agx_cgan_iter_rng_draws(g))
    lg = losses.cpu().numpy()
    xo, co, no = t64(xb.cpu().numpy()), torch.tensor(cc[s].cpu().numpy().astype(np.int64)), t64(nc[s].cpu().numpy())
    for k in range(nbc):
        l, gp, _ = O.train_disc(xo, t64(zs[k]), co, no, tg, td, odo, t64(als[k]))
        print('batch %d step %d gpu wgan %.6g gp %.6g total %.6g | oracle gp %.6g total %.6g' % (i, k, lg[k, 0], lg[k, 1], lg[k, 2], float(gp), float(l)), flush=True)
    l, _ = O.train_gen(t64(zs[-1]), co, no, tg, td, ogo)
    print('batch %d gen gpu %.6g oracle %.6g' % (i, lg[nbc, 0], float(l)), flush=True)
