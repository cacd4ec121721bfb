"""Per-variable error of one critic step's and one generator step's gradients against the fp64 oracle at a shape that
takes the Gram / factored forms (AGX_NO_GRAM=1: the direct forms)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import keras_list, make_problem, t64, tl
from oracle import cgan_oracle as O
from adversarial_gene_expression_amd import tf_utils as T
G, H, B = int(sys.argv[1]), (int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[4])
V = (15, 3) if G > 10000 else (6, 8, 5, 11, 5, 3, 5)
p = make_problem(G=G, V=V, H=H, Z=64 if G > 10000 else 10, B=B, seed=17)
gen = T.Generator(p['G'], list(p['V']), p['n_num'], list(p['H']), p['Z'], weights=keras_list(p['gp']))
disc = T.Critic(p['G'], list(p['V']), p['n_num'], list(p['H']), 1, weights=keras_list(p['dp']))
f32 = lambda k: np.asarray(p[k], np.float32)
loss = T.train_disc(f32('x'), f32('z'), p['cc'], f32('nc'), gen, disc, T.RMSprop(5e-4), alpha=f32('alpha'), apply=False)
want_loss, want_gp, want = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']),
                                                 O.to_torch(p['dp']), t64(p['alpha']))
print('critic loss', float(loss), float(want_loss))
for i, (a, b) in enumerate(zip(disc.get_gradients(), want)):
    b = b.numpy().reshape(a.shape)
    print(' ', i, a.shape, 'err %.3e scale %.3e' % (np.abs(a - b).max(), np.abs(b).max()))
loss = T.train_gen(f32('z'), p['cc'], f32('nc'), gen, disc, T.RMSprop(5e-4), apply=False)
want_loss, want = O.gen_loss_and_grads(t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']), O.to_torch(p['dp']))
print('generator loss', float(loss), float(want_loss))
for i, (a, b) in enumerate(zip(gen.get_gradients(), want)):
    b = b.numpy().reshape(a.shape)
    print(' ', i, a.shape, 'err %.3e scale %.3e' % (np.abs(a - b).max(), np.abs(b).max()))
