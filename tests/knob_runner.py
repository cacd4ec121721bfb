"""Child process of tests/test_gpu_knobs.py: two captured iterations of an E. coli-sized cGAN (large enough for the
fused-optimizer / chained-input paths), without and with augmentation, under whatever AGX_* knobs the parent set in
the environment; losses and final weights go to the .npz named on the command line.  libagx reads its knobs once per
process, hence one process per knob.  Weights and inputs are helpers.make_problem's (non-zero biases): with the
library's default initialisation (zero biases) many hidden units sit within rounding of the ReLU boundary, and two
correct fp32 sequences that round differently then drift apart by 1e-5..1e-3 within two RMSprop steps (the torch
fp32 oracle does so against its own fp64 run: tests/diag_two_iterations.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(out_path, shape='ecoli'):
    from helpers import keras_list, make_problem
    from adversarial_gene_expression_amd import tf_utils as T
    G, V, H, Z, B = 4297, (6, 8, 5, 11, 5, 3, 5), (128, 128), 10, 32
    if shape == 'gram':    # batch >= the first hidden width: the Gram form of the penalty is the default (cgan.hip gram_ok)
        H, B = (160, 160), 256    # (widths the fused-optimizer tiling takes: the factored fake rows need it)
    p = make_problem(G=G, V=V, H=H, Z=Z, B=2 * B, seed=17)
    x, cc, nc = np.asarray(p['x'], np.float32), p['cc'], np.asarray(p['nc'], np.float32)
    out = {}
    for p_aug in (0.0, 0.25):
        gen = T.Generator(G, list(V), p['n_num'], list(H), Z, weights=keras_list(p['gp']))
        disc = T.Critic(G, list(V), p['n_num'], list(H), 1, weights=keras_list(p['dp']))
        gopt, dopt = T.RMSprop(5e-4), T.RMSprop(5e-4)
        T.set_seed(77)
        np.random.seed(5)      # augmentation trace constants
        losses = []
        for it in range(2):
            s = slice(it * B, (it + 1) * B)
            d, g = T.train_iteration(x[s], cc[s], nc[s], gen, disc, gopt, dopt, 5, p_aug=p_aug)
            losses += [float(d), float(g)]
        tag = 'aug' if p_aug else 'plain'
        out[tag + '_losses'] = np.array(losses)
        out[tag + '_w'] = np.concatenate([w.reshape(-1) for w in gen.get_weights() + disc.get_weights()])
        out[tag + '_sizes'] = np.array([w.size for w in gen.get_weights() + disc.get_weights()])
    np.savez(out_path, **out)


if __name__ == '__main__':
    main(sys.argv[1], *sys.argv[2:3])
