"""Child process of tests/test_gpu_dp_native.py: two loop iterations of a cGAN on ONE GPU through
  mode 'native': the library's overlapped data-parallel driver (agx_cgan_dp_iteration: split = 3 segments, the
                 largest weight-gradient product in row chunks, reduce-scatter / sharded optimizer / all-gather per
                 piece on a second stream) with a ONE-rank RCCL communicator (AGX_DP_FORCE=1; AGX_DP_CHUNKS pieces);
  mode 'emul':   the split = 2 segments with the whole-buffer optimizer step between them (what tf_utils._dp_update
                 does with one rank), driven from Python;
  mode 'single': the ordinary one-GPU captured iteration (fused optimizer epilogues).
Losses, weights and optimizer slots go to the .npz named on the command line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main(mode, size, opt_name, out_path):
    import torch
    from helpers import keras_list, make_problem
    from adversarial_gene_expression_amd import _dev, _lib, tf_utils as T
    from adversarial_gene_expression_amd._dev import ptr
    lib = _lib.load()
    if size == 'gtex512':
        G, V, H, Z, B, nbc = 18154, (15, 3), (256, 256), 64, 512, 2
    else:
        G, V, H, Z, B, nbc = 4297, (6, 8, 5), (64, 48), 10, 48, 3    # (generator output-layer gradient too large to ride the small-gradient launch)
    p = make_problem(G=G, V=V, H=H, Z=Z, B=2 * B, seed=29)
    x, cc, nc = np.asarray(p['x'], np.float32), p['cc'], np.asarray(p['nc'], np.float32)
    gen = T.Generator(G, list(V), p['n_num'], list(H), Z, weights=keras_list(p['gp']))
    disc = T.Critic(G, list(V), p['n_num'], list(H), 1, weights=keras_list(p['dp']))
    mk = (lambda: T.RMSprop(5e-4)) if opt_name == 'rmsprop' else (lambda: T.Adam(2e-4, beta_1=0.5))
    gopt, dopt = mk(), mk()
    T.set_seed(77)
    losses = []
    for it in range(int(os.environ.get('DP_RUNNER_ITERS', '2'))):
        s = slice(it * B, (it + 1) * B)
        if mode in ('native', 'single'):
            d, g = T.train_iteration(x[s], cc[s], nc[s], gen, disc, gopt, dopt, nbc)
            losses += [float(d), float(g)]
            continue
        xd, cd, nd = _dev.to_dev(x[s]), _dev.to_dev(cc[s], torch.int32), _dev.to_dev(nc[s])
        pair = T._pair(gen, disc, B, nbc + 1)
        gr = pair.iteration_graph(gen, disc, gopt, dopt, B, nbc, 10.0, T._rng.seed, batch_global=B, row_lo=0, split=2)
        _lib.check(lib.agx_cgan_iter_stage(gr, _dev.stream(), ptr(xd), xd.stride(0), ptr(cd), ptr(nd), T._rng.offset))
        for sg in range(nbc + 1):
            _lib.check(lib.agx_cgan_iter_launch_seg(gr, _dev.stream(), sg))
            net, opt = (disc, dopt) if sg < nbc else (gen, gopt)
            opt.apply_range(net, 0, net.n_flat, net.grads[:net.n_flat])
            if sg == nbc - 1:
                losses.append(float(disc.grads[disc.n_flat + 2]))
        losses.append(float(gen.grads[gen.n_flat]))
        T._rng.offset += int(lib.agx_cgan_iter_rng_draws(gr))
    torch.cuda.synchronize()
    sd = dopt.state_arrays(disc)
    slot = np.concatenate([a.reshape(-1) for a in sd['rms' if opt_name == 'rmsprop' else 'v']])
    np.savez(out_path, losses=np.array(losses), w=np.concatenate([w.reshape(-1) for w in gen.get_weights() + disc.get_weights()]),
             sizes=np.array([w.size for w in gen.get_weights() + disc.get_weights()]), slot=slot, iters=np.array([dopt.iterations, gopt.iterations]), offset=np.array([T._rng.offset]))


if __name__ == '__main__':
    main(*sys.argv[1:5])
