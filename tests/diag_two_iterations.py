"""N captured iterations against the fp64 oracle carried through the same N iterations (optimizer state included),
E. coli-sized problem without augmentation: a wrong gradient MAGNITUDE is invisible after one RMSprop step (the first
step moves every weight by lr / sqrt(1 - rho) whatever |g| is) and shows in the second.  Used by
test_gpu_fullsize.test_two_captured_iterations_match_the_oracle; by hand: python tests/diag_two_iterations.py [n]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import keras_list, make_problem, tl          # noqa: E402
from oracle import cgan_oracle as O                        # noqa: E402


def compare(n_it):
    from adversarial_gene_expression_amd import tf_utils as T
    G, V, H, Z, B, nbc, lr, seed = 4297, (6, 8, 5, 11, 5, 3, 5), (128, 128), 10, 32, 5, 5e-4, 4242
    p = make_problem(G=G, V=V, H=H, Z=Z, B=B, seed=17)
    p['x'] = np.asarray(p['x'], np.float32).astype(np.float64)
    gen = T.Generator(G, list(V), p['n_num'], list(H), Z, weights=keras_list(p['gp']))
    disc = T.Critic(G, list(V), p['n_num'], list(H), 1, weights=keras_list(p['dp']))
    # the draws of n_it iterations, regenerated through the eager RNG entry points
    T.set_seed(seed)
    draws = []
    for _ in range(n_it):
        zs, als = [], []
        for _ in range(nbc):
            zs.append(T.random_normal((B, Z)).cpu().numpy())
            als.append(T.random_uniform((B,)).cpu().numpy().reshape(B, 1))
        zs.append(T.random_normal((B, Z)).cpu().numpy())
        draws.append((zs, als))
    T.set_seed(seed)
    go, do = T.RMSprop(lr), T.RMSprop(lr)
    x32, nc32 = np.asarray(p['x'], np.float32), np.asarray(p['nc'], np.float32)
    got_l = []
    for _ in range(n_it):
        d, g = T.train_iteration(x32, p['cc'], nc32, gen, disc, go, do, nbc)
        got_l += [float(d), float(g)]
    got_g = np.concatenate([w.reshape(-1) for w in gen.get_weights()]).astype(np.float64)
    got_d = np.concatenate([w.reshape(-1) for w in disc.get_weights()]).astype(np.float64)
    torch.set_num_threads(16)
    res = {}
    for dtype in (torch.float64, torch.float32):
        tt = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        tg, td = O.to_torch(p['gp'], dtype=dtype), O.to_torch(p['dp'], dtype=dtype)
        ogo, odo = O.RMSprop(lr), O.RMSprop(lr)
        x, cc, nc = tt(p['x']), tl(p['cc']), tt(p['nc'])
        ls = []
        for zs, als in draws:
            for k in range(nbc):
                dl = float(O.train_disc(x, tt(zs[k]), cc, nc, tg, td, odo, tt(als[k]))[0])
            ls += [dl, float(O.train_gen(tt(zs[-1]), cc, nc, tg, td, ogo)[0])]
        flat = lambda tp: np.concatenate([t.detach().numpy().astype(np.float64).reshape(-1) for t in O.param_list(tp)])
        res[dtype] = (ls, flat(tg), flat(td))
    l64, g64, d64 = res[torch.float64]
    l32, g32, d32 = res[torch.float32]
    print('losses gpu', got_l)
    print('losses f64', l64)
    q = lambda e: (float(np.quantile(e, 0.99)), float(np.quantile(e, 0.999)), float(e.max()))
    print('gen  q99/q999/max  gpu-f64 %.2e %.2e %.2e | f32-f64 %.2e %.2e %.2e' % (q(np.abs(got_g - g64)) + q(np.abs(g32 - g64))))
    print('disc q99/q999/max  gpu-f64 %.2e %.2e %.2e | f32-f64 %.2e %.2e %.2e' % (q(np.abs(got_d - d64)) + q(np.abs(d32 - d64))))
    return dict(losses=np.array(got_l), losses64=np.array(l64), gen=q(np.abs(got_g - g64)), disc=q(np.abs(got_d - d64)),
                gen32=q(np.abs(g32 - g64)), disc32=q(np.abs(d32 - d64)))


if __name__ == '__main__':
    compare(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
