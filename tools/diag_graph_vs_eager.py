"""Diagnostic: captured iteration vs eager step functions vs fp64 oracle, nb_critic = 1..n (GPU box)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import keras_list, make_problem, t64, tl
from oracle import cgan_oracle as O
import test_gpu_fullsize as F
from adversarial_gene_expression_amd import tf_utils as T, _dev

name = sys.argv[1] if len(sys.argv) > 1 else 'gtex_b512'
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = F.CONFIGS[name]
B = c['B']
p = make_problem(G=c['G'], V=c['V'], H=c['H'], Z=c['Z'], B=B, seed=17)
p['x'] = F._f32(p['x']).astype(np.float64)
torch.set_num_threads(16)
x32, nc32 = F._f32(p['x']), F._f32(p['nc'])
flat = lambda net: np.concatenate([w.reshape(-1) for w in net.get_weights()]).astype(np.float64)
def stats(tag, a, b):
    e = np.abs(a - b)
    print('   %-22s q50 %.3g q99 %.3g q999 %.3g max %.3g frac>2e-6 %.3g' % (tag, np.quantile(e, .5), np.quantile(e, .99), np.quantile(e, .999), e.max(), (e > 2e-6).mean()))
for nbc in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1, nmax + 1):
    zs, als, end = F._graph_draws(T, 4242, B, c['Z'], nbc)
    # eager
    _, gen, disc = F._models(p)
    go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
    le = []
    for k in range(nbc):
        le.append(float(T.train_disc(x32, zs[k], p['cc'], nc32, gen, disc, do, alpha=als[k])))
    le.append(float(T.train_gen(zs[-1], p['cc'], nc32, gen, disc, go)))
    eg, ed = flat(gen), flat(disc)
    # graph
    _, gen2, disc2 = F._models(p)
    T.set_seed(4242)
    go2, do2 = T.RMSprop(5e-4), T.RMSprop(5e-4)
    dl, gl = T.train_iteration(x32, p['cc'], nc32, gen2, disc2, go2, do2, nbc)
    gg, gd = flat(gen2), flat(disc2)
    l64, g64, d64 = F._oracle_iteration(p, zs, als, torch.float64)
    l32, g32, d32 = F._oracle_iteration(p, zs, als, torch.float32)
    print('nb_critic', nbc, 'eager losses', le[-2:], 'graph', float(dl), float(gl), 'o64', l64[-2:], 'o32', l32[-2:])
    stats('disc graph-eager', gd, ed); stats('disc graph-o64', gd, d64); stats('disc eager-o64', ed, d64); stats('disc o32-o64', d32, d64)
    stats('gen graph-eager', gg, eg); stats('gen graph-o64', gg, g64); stats('gen eager-o64', eg, g64); stats('gen o32-o64', g32, g64)
    # where do graph and eager differ? (one ReLU mask flip moves one column of W_1)
    nv = len(c['V'])
    w1g, w1e = disc2.get_weights()[nv].astype(np.float64), disc.get_weights()[nv].astype(np.float64)
    big = np.abs(w1g - w1e) > 2e-6
    cols = big.sum(0)
    order = np.argsort(-cols)[:5]
    print('   W_1 entries differing > 2e-6 graph vs eager: %d of %d; per column top5: %s' % (big.sum(), big.size, [(int(j), int(cols[j])) for j in order]))
