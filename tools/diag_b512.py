"""Diagnostic: where does the 512-row iteration drift from the fp64 oracle? (GPU box)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import keras_list, make_problem, t64, tl
from oracle import cgan_oracle as O
import test_gpu_fullsize as F
from adversarial_gene_expression_amd import tf_utils as T, _dev

name = sys.argv[1] if len(sys.argv) > 1 else 'gtex_b512'
c = F.CONFIGS[name]
B = c['B']
p = make_problem(G=c['G'], V=c['V'], H=c['H'], Z=c['Z'], B=B, seed=17)
p['x'] = F._f32(p['x']).astype(np.float64)
torch.set_num_threads(16)
zs, als, end = F._graph_draws(T, 4242, B, c['Z'], 5)

def oracle_steps(dtype):
    tt = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    tg, td = O.to_torch(p['gp'], dtype=dtype), O.to_torch(p['dp'], dtype=dtype)
    do = O.RMSprop(5e-4)
    x, cc, nc = tt(p['x']), tl(p['cc']), tt(p['nc'])
    out = []
    for k in range(5):
        loss, gp, grads = O.disc_loss_and_grads(x, tt(zs[k]), cc, nc, tg, td, tt(als[k]))
        out.append((float(loss), float(gp), [g.numpy().astype(np.float64) for g in grads]))
        do.apply(O.param_list(td), grads)
    return out

o64 = oracle_steps(torch.float64)
o32 = oracle_steps(torch.float32)
# eager GPU steps with the same draws
_, gen, disc = F._models(p)
do = T.RMSprop(5e-4)
x32, nc32 = F._f32(p['x']), F._f32(p['nc'])
for k in range(5):
    loss = T.train_disc(x32, zs[k], p['cc'], nc32, gen, disc, do, alpha=als[k], apply=False)
    gg = [g.astype(np.float64) for g in disc.get_gradients()]
    l64, gp64, g64 = o64[k]
    l32, gp32, g32 = o32[k]
    print('step %d loss gpu %.7f o32 %.7f o64 %.7f' % (k, float(loss), l32, l64))
    for i, (a, b, c_) in enumerate(zip(gg, g32, g64)):
        c_ = c_.reshape(a.shape); b = b.reshape(a.shape)
        eg, eo = np.abs(a - c_), np.abs(b - c_)
        print('   var %d shape %s max|g| %.3g  rms err gpu %.3g o32 %.3g  max err gpu %.3g o32 %.3g'
              % (i, a.shape, np.abs(c_).max(), np.sqrt((eg ** 2).mean()), np.sqrt((eo ** 2).mean()), eg.max(), eo.max()))
    do.apply(disc)
