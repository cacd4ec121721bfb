"""WGAN-GP train step on the GPU (C-ABI agx_cgan_*) against the torch-CPU fp64 autograd oracle.

The reference pins nothing here (TF absent, no tests): "parity unpinned"; the oracle restates
tf_utils.py:45-258 line by line.  fp32 tolerances: losses 1e-5 relative; every gradient
max|err| <= 3e-5 * max|grad| + 1e-7; weights after n RMSprop steps: see test_weights_after_steps."""
import numpy as np
import pytest
import torch

from helpers import keras_list, make_problem, t64, tl
from oracle import cgan_oracle as O
from oracle import corr_oracle as CO

pytestmark = pytest.mark.gpu


def _models(p, slope=0.0):
    from adversarial_gene_expression_amd import tf_utils as T
    gen = T.Generator(p['G'], p['V'], p['n_num'], p['H'], p['Z'], weights=keras_list(p['gp']), slope=slope)
    disc = T.Critic(p['G'], p['V'], p['n_num'], p['H'], 1, weights=keras_list(p['dp']), slope=slope)
    return T, gen, disc


def _f32(p, k):
    return np.asarray(p[k], np.float32)


def _check_grads(got, want, rtol=3e-5):
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        b = np.asarray(b).reshape(a.shape)
        err = np.max(np.abs(a - b)) if a.size else 0.0
        assert err <= rtol * np.max(np.abs(b)) + 1e-7, ('variable %d' % i, err, np.max(np.abs(b)))


CFGS = [dict(), dict(B=1, seed=2), dict(B=12, H=(16,), seed=3), dict(B=33, H=(24, 16), G=130, seed=4),
        dict(B=7, H=(12, 10, 8), V=(6, 8, 5, 11, 5, 3, 5), Z=10, seed=5),
        dict(B=4, n_num=2, num_zero=False, seed=7), dict(B=32, G=300, H=(128, 128), V=(15, 3), Z=64, seed=8),
        # batches of >= the hidden widths: Gram form of the penalty, factored generator step (cgan.hip gram_ok, gen_fact_ok)
        dict(B=40, H=(16, 8), seed=9), dict(B=48, G=130, H=(24, 16), n_num=2, num_zero=False, seed=10),
        # direct form on the 64-wide tiling with G mod 64 in 1..32: the last tile's second 32-column block has no slab of
        # squared-norm partials (it used to land on the next row's slab 0: critic loss off by 2e-3 at this shape)
        dict(B=192, G=4297, H=(160, 96), V=(6, 8, 5), Z=10, seed=11),
        # batch == first hidden width: W_g^T fills the interpolated rows of the stacked input exactly
        dict(B=24, G=130, H=(24, 16), seed=12)]


@pytest.mark.parametrize('cfg', CFGS)
def test_forward_matches_oracle(gpu, cfg):
    p = make_problem(**cfg)
    T, gen, disc = _models(p)
    xg = gen([_f32(p, 'z'), p['cc'], _f32(p, 'nc')]).numpy()
    want = O.gen_forward(O.to_torch(p['gp']), t64(p['z']), tl(p['cc']), t64(p['nc'])).numpy()
    np.testing.assert_allclose(xg, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
    sc = disc([_f32(p, 'x'), p['cc'], _f32(p, 'nc')]).numpy()
    want = O.disc_forward(O.to_torch(p['dp']), t64(p['x']), tl(p['cc']), t64(p['nc'])).numpy()
    np.testing.assert_allclose(sc, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
    assert T.predict(p['cc'], _f32(p, 'nc'), gen, z=_f32(p, 'z')).shape == (p['B'], p['G'])


@pytest.mark.parametrize('cfg', CFGS)
def test_disc_step_loss_and_grads(gpu, cfg):
    p = make_problem(**cfg)
    T, gen, disc = _models(p)
    loss = T.train_disc(_f32(p, 'x'), _f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4),
                        alpha=_f32(p, 'alpha'), apply=False)
    want_loss, want_gp, want = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']),
                                                     O.to_torch(p['gp']), O.to_torch(p['dp']), t64(p['alpha']))
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * max(1.0, abs(float(want_loss)))
    _check_grads(disc.get_gradients(), [g.numpy() for g in want])


def test_disc_step_with_augmentation_and_leaky_slope(gpu):
    p = make_problem(B=6, seed=11)
    rs = np.random.RandomState(1)
    aug = dict(mask=rs.binomial(1, 0.5, 6).astype(np.float32), noise_fake=rs.normal(0, 0.5, p['G']).astype(np.float32),
               noise_real=rs.normal(0, 0.5, (6, p['G'])).astype(np.float32))
    T, gen, disc = _models(p)
    loss = T.train_disc(_f32(p, 'x'), _f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4),
                        alpha=_f32(p, 'alpha'), aug=aug, apply=False)
    wl, _, want = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']),
                                        O.to_torch(p['dp']), t64(p['alpha']), aug_mask=t64(aug['mask']),
                                        aug_noise_fake=t64(aug['noise_fake']), aug_noise_real=t64(aug['noise_real']))
    assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    _check_grads(disc.get_gradients(), [g.numpy() for g in want])
    # generator step with its [B, G] noise
    gl = T.train_gen(_f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4),
                     aug=dict(mask=aug['mask'], noise=aug['noise_real']), apply=False)
    wl, want = O.gen_loss_and_grads(t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']), O.to_torch(p['dp']),
                                    aug_mask=t64(aug['mask']), aug_noise=t64(aug['noise_real']))
    assert abs(float(gl) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    _check_grads(gen.get_gradients(), [g.numpy() for g in want])


def test_more_than_256_covariate_columns_take_the_generic_path(gpu):
    """The fused critic-tail kernels zero the covariate columns of a row with one 256-thread group;
    a critic with more covariate columns (n_num + sum of embedding widths) must fall back to the
    generic launch sequence instead of failing."""
    p = make_problem(B=6, G=40, H=(32, 16), n_num=300, num_zero=False, seed=13)
    T, gen, disc = _models(p)
    loss = T.train_disc(_f32(p, 'x'), _f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4),
                        alpha=_f32(p, 'alpha'), apply=False)
    wl, _, want = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']),
                                        O.to_torch(p['dp']), t64(p['alpha']))
    assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    _check_grads(disc.get_gradients(), [g.numpy() for g in want])
    T.set_seed(3)
    dl, gl = T.train_iteration(_f32(p, 'x'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4), T.RMSprop(5e-4), 2)
    assert np.isfinite(float(dl)) and np.isfinite(float(gl))


def test_leaky_slope_matches_closed_form(gpu):
    """slope > 0 (north_star's LeakyReLU option) against the NumPy restatement of the same backward."""
    from oracle import cgan_closed_form as CF
    p = make_problem(B=9, seed=21)
    T, gen, disc = _models(p, slope=0.2)
    loss = T.train_disc(_f32(p, 'x'), _f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4),
                        alpha=_f32(p, 'alpha'), apply=False)
    losses, want = CF.disc_grads(p['x'], p['z'], p['cc'], p['nc'], p['gp'], p['dp'], p['alpha'], slope=0.2)
    assert abs(float(loss) - losses[2]) <= 1e-5 * max(1.0, abs(losses[2]))
    _check_grads(disc.get_gradients(), CF.flat(want))


@pytest.mark.parametrize('cfg', CFGS)
def test_gen_step_loss_and_grads(gpu, cfg):
    p = make_problem(**cfg)
    T, gen, disc = _models(p)
    loss = T.train_gen(_f32(p, 'z'), p['cc'], _f32(p, 'nc'), gen, disc, T.RMSprop(5e-4), apply=False)
    want_loss, want = O.gen_loss_and_grads(t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']),
                                           O.to_torch(p['dp']))
    assert abs(float(loss) - float(want_loss)) <= 1e-5 * max(1.0, abs(float(want_loss)))
    _check_grads(gen.get_gradients(), [g.numpy() for g in want])
    # the critic is untouched by a generator step
    for a, b in zip(disc.get_weights(), keras_list(p['dp'])):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('opt', ['rmsprop', 'adam'])
def test_weights_after_steps(gpu, opt):
    """5 x (5 critic + 1 generator) steps with injected z / alpha.

    RMSprop/Adam divide by sqrt(EMA(g^2)) + 1e-7, so an entry whose gradient is ~1e-7 turns an fp32
    rounding difference into an O(lr) difference of its update ('RMSprop
    sensitivity').  The check is therefore a quantile: 99.9 % of all weights within 2e-6 of the fp64
    oracle, and (tie-breaker) that quantile not worse than 4x the fp32 oracle's own distance from
    the fp64 oracle + 1e-7.  The maximum is only printed: any optimizer trajectory satisfies an
    O(steps * lr) bound, so asserting one would check nothing."""
    p = make_problem(B=12, G=61, H=(24, 16), seed=31)
    T, gen, disc = _models(p)
    lr = 5e-4
    mk = (lambda: T.RMSprop(lr)) if opt == 'rmsprop' else (lambda: T.Adam(lr))
    mko = (lambda: O.RMSprop(lr)) if opt == 'rmsprop' else (lambda: O.Adam(lr))
    go, do = mk(), mk()
    p['x'] = _f32(p, 'x').astype(np.float64)        # both oracles see the fp32 inputs the GPU sees
    tg, td = O.to_torch(p['gp']), O.to_torch(p['dp'])
    sg, sd = O.to_torch(p['gp'], dtype=torch.float32), O.to_torch(p['dp'], dtype=torch.float32)
    ogo, odo, sgo, sdo = mko(), mko(), mko(), mko()
    t32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    rs = np.random.RandomState(5)
    losses, olosses = [], []
    n_it = 5
    for it in range(n_it):
        for _ in range(5):
            z, al = rs.randn(12, p['Z']).astype(np.float32), rs.rand(12, 1).astype(np.float32)
            losses.append(float(T.train_disc(_f32(p, 'x'), z, p['cc'], _f32(p, 'nc'), gen, disc, do, alpha=al)))
            olosses.append(float(O.train_disc(t64(p['x']), t64(z), tl(p['cc']), t64(p['nc']), tg, td, odo, t64(al))[0]))
            O.train_disc(t32(p['x']), t32(z), tl(p['cc']), t32(p['nc']), sg, sd, sdo, t32(al))
        z = rs.randn(12, p['Z']).astype(np.float32)
        losses.append(float(T.train_gen(z, p['cc'], _f32(p, 'nc'), gen, disc, go)))
        olosses.append(float(O.train_gen(t64(z), tl(p['cc']), t64(p['nc']), tg, td, ogo)[0]))
        O.train_gen(t32(z), tl(p['cc']), t32(p['nc']), sg, sd, sgo)
    np.testing.assert_allclose(losses, olosses, rtol=2e-4, atol=2e-4)
    flat = lambda tp: np.concatenate([t.detach().numpy().astype(np.float64).reshape(-1) for t in O.param_list(tp)])
    for net, tp, sp in ((gen, tg, sg), (disc, td, sd)):
        got = np.concatenate([w.reshape(-1) for w in net.get_weights()]).astype(np.float64)
        e64, o32 = np.abs(got - flat(tp)), np.abs(flat(sp) - flat(tp))
        q, qo = np.quantile(e64, 0.999), np.quantile(o32, 0.999)
        print('%s %s: q999 gpu-f64 %.3g, f32 oracle-f64 %.3g; max %.3g / %.3g' % (opt, net.kind, q, qo, e64.max(), o32.max()))
        assert q < 2e-6, q
        assert q <= 4 * qo + 1e-7, (q, qo)


def _noise_factory():
    rs = np.random.RandomState(77)
    return lambda kind, shape: rs.randn(*shape) if kind == 'z' else rs.rand(*shape)


def test_train_loop_trace_matches_oracle(gpu):
    from adversarial_gene_expression_amd import tf_utils as T, utils as U
    p = make_problem(G=37, B=70, seed=41)
    _, gen, disc = _models(p)
    x32 = p['x'].astype(np.float32)
    zt = np.random.RandomState(9).randn(70, p['Z'])
    saved = []
    n1 = _noise_factory()
    tr = T.train(x32, p['cc'], _f32(p, 'nc'), p['Z'], 6, 32, gen, disc,
                 lambda g: U.gamma_coef(x32, T.predict(p['cc'], _f32(p, 'nc'), g, z=zt.astype(np.float32))),
                 lambda: saved.append(1), verbose=False, noise_fn=lambda k, s: n1(k, s).astype(np.float32))
    tg, td = O.to_torch(p['gp']), O.to_torch(p['dp'])
    osaved = []
    n2 = _noise_factory()
    otr = O.train(p['x'], p['cc'], p['nc'], p['Z'], 6, 32, tg, td,
                  lambda gp_: CO.gamma_coef(p['x'], O.predict(tl(p['cc']), t64(p['nc']), gp_, t64(zt))),
                  lambda: osaved.append(1), O.RMSprop(5e-4), O.RMSprop(5e-4), n2)
    np.testing.assert_allclose(tr['gen_loss'], otr['gen_loss'], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(tr['disc_loss'], otr['disc_loss'], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(tr['score'], otr['score'], atol=2e-3)
    assert len(tr['score']) == 2 and len(saved) == len(osaved)      # evaluated at epochs 0 and 5
    assert tr['patience'] == otr['patience'] and tr['saved'] == otr['saved']


def test_reference_surface(gpu):
    """Names, defaults and call patterns a user of the reference relies on."""
    import inspect
    from adversarial_gene_expression_amd import tf_utils as T
    for name in ['limit_gpu', 'make_generator', 'make_discriminator', 'wasserstein_loss', 'generator_loss',
                 'gradient_penalty', 'discriminator_loss', 'train_disc', 'train_gen', 'train', 'predict']:
        assert callable(getattr(T, name))
    sig = inspect.signature(T.train)
    assert [k for k in sig.parameters][:10] == ['dataset', 'cat_covs', 'num_covs', 'z_dim', 'epochs', 'batch_size',
                                                'gen', 'disc', 'score_fn', 'save_fn']
    assert sig.parameters['nb_critic'].default == 5 and sig.parameters['patience'].default == 10
    assert sig.parameters['norm_scale'].default == 0.5 and sig.parameters['p_aug'].default == 0
    assert inspect.signature(T.train_disc).parameters['grad_penalty_weight'].default == 10
    assert inspect.signature(T.train_gen).parameters['norm_scale'].default == 1
    p = make_problem(B=5, seed=3)
    _, gen, disc = _models(p)
    assert gen.output.shape[-1] == p['G'] and gen.input[0].shape[-1] == p['Z']
    assert gen.count_params() == sum(a.size for a in keras_list(p['gp']))
    # loss helpers (tf_utils.py:140-181)
    rs = np.random.RandomState(0)
    r, f = rs.randn(9, 1).astype(np.float32), rs.randn(9, 1).astype(np.float32)
    np.testing.assert_allclose(float(T.generator_loss(f)), -f.mean(), atol=1e-6)
    np.testing.assert_allclose(float(T.discriminator_loss(r, f)), -r.mean() + f.mean(), atol=1e-6)
    np.testing.assert_allclose(float(T.wasserstein_loss(-np.ones_like(r), r)), -r.mean(), atol=1e-6)
    # gradient_penalty with the reference's closure (tf_utils.py:218)
    cc, nc = p['cc'], _f32(p, 'nc')
    xr, xf = _f32(p, 'x'), rs.randn(5, p['G']).astype(np.float32)
    gp = T.gradient_penalty(lambda x: disc([x, cc, nc], training=True), xr, xf, alpha=_f32(p, 'alpha'))
    dp_ = O.to_torch(p['dp'])
    want = O.gradient_penalty(lambda xi: O.disc_forward(dp_, xi, tl(cc), t64(nc)), t64(xr), t64(xf), t64(p['alpha']))
    np.testing.assert_allclose(float(gp), float(want), rtol=2e-5)
    # save / load round trip
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'models', 'gen_test.h5')
        gen.save(path)
        g2 = T.load_model(path)
        for a, b in zip(gen.get_weights(), g2.get_weights()):
            np.testing.assert_array_equal(a, b)


def test_ragged_and_max_batch_growth(gpu):
    """Step context grows when a larger batch arrives; a ragged tail (12) after 32 reuses it."""
    p32 = make_problem(B=40, seed=51)
    T, gen, disc = _models(p32)
    for B in (32, 12, 40):
        s = slice(0, B)
        loss = T.train_disc(_f32(p32, 'x')[s], _f32(p32, 'z')[s], p32['cc'][s], _f32(p32, 'nc')[s], gen, disc,
                            T.RMSprop(5e-4), alpha=_f32(p32, 'alpha')[s], apply=False)
        wl, _, want = O.disc_loss_and_grads(t64(p32['x'][s]), t64(p32['z'][s]), tl(p32['cc'][s]), t64(p32['nc'][s]),
                                            O.to_torch(p32['gp']), O.to_torch(p32['dp']), t64(p32['alpha'][s]))
        assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
        _check_grads(disc.get_gradients(), [g.numpy() for g in want])


def test_adam_graph_iteration_is_bit_identical_to_eager(gpu):
    """Adam on both networks (``train(..., gen_opt=Adam, disc_opt=Adam)``, tf_utils.py:261-262): the captured
    iteration reads each step's bias-corrected step size from device memory (staged per replay by the host,
    computed as for the eager call), so graph and eager runs agree bit for bit - weights, both moment slots,
    step counters, losses."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=61, B=70, H=(40, 16), seed=63)   # (hidden width > batch 32: the direct forms, as in the step functions)
    x32 = p['x'].astype(np.float32)
    out = []
    for use_graph in (False, True):
        _, gen, disc = _models(p)
        T.set_seed(4321)
        go, do = T.Adam(2e-4, beta_1=0.5), T.Adam(2e-4, beta_1=0.5)
        tr = T.train(x32, p['cc'], _f32(p, 'nc'), p['Z'], 3, 32, gen, disc, lambda g: 0.0, lambda: None, gen_opt=go,
                     disc_opt=do, verbose=False, use_graph=use_graph)
        sd, sg = do.state_arrays(disc), go.state_arrays(gen)
        out.append((gen.get_weights() + disc.get_weights() + sd['m'] + sd['v'] + sg['m'] + sg['v'], tr,
                    do.slot(disc)[2], go.slot(gen)[2], T._rng.offset))
    (w0, tr0, td0, tg0, o0), (w1, tr1, td1, tg1, o1) = out
    assert (td0, tg0, o0) == (td1, tg1, o1) and td0 == 3 * 3 * 5 and tg0 == 3 * 3
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)
    assert tr0['gen_loss'] == tr1['gen_loss'] and tr0['disc_loss'] == tr1['disc_loss']


@pytest.mark.parametrize('p_aug', [0, 0.25])
def test_graph_iteration_is_bit_identical_to_eager(gpu, p_aug):
    """train(use_graph=True) replays one hipGraph per iteration; with equal Philox seeds it must give
    exactly the weights, optimizer slots and losses of the eager step functions (ragged tail incl.)."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=61, B=70, H=(40, 16), seed=61)   # (hidden width > batch 32: the direct forms, as in the step functions)
    x32 = p['x'].astype(np.float32)
    out = []
    for use_graph in (False, True):
        _, gen, disc = _models(p)
        T._TRACE_CONSTS.clear()
        np.random.seed(5)
        T.set_seed(1234)
        go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
        tr = T.train(x32, p['cc'], _f32(p, 'nc'), p['Z'], 3, 32, gen, disc, lambda g: 0.0, lambda: None, gen_opt=go,
                     disc_opt=do, verbose=False, p_aug=p_aug, use_graph=use_graph)
        out.append((gen.get_weights(), disc.get_weights(), tr, do.state_arrays(disc)['rms'], T._rng.offset))
    (gw0, dw0, tr0, r0, o0), (gw1, dw1, tr1, r1, o1) = out
    assert o0 == o1
    for a, b in zip(gw0 + dw0 + r0, gw1 + dw1 + r1):
        np.testing.assert_array_equal(a, b)
    assert tr0['gen_loss'] == tr1['gen_loss'] and tr0['disc_loss'] == tr1['disc_loss']


def _iteration_by_segments(T, gens, discs, gopts, dopts, shards, Bg, nb_critic, seed, offset, p_aug=0):
    """Drive the split (data-parallel) form of the captured iteration for several 'ranks' that live
    on ONE GPU: every rank launches segment s, then the gradient buffers are summed across ranks
    (what the RCCL all-reduce does) before the next segment."""
    from adversarial_gene_expression_amd import _dev, _lib
    from adversarial_gene_expression_amd._dev import ptr
    lib = _lib.load()
    graphs, keep = [], []
    for r, (x, cc, nc, lo) in enumerate(shards):
        xd, cd, nd = _dev.to_dev(x), _dev.to_dev(cc, torch.int32), _dev.to_dev(nc)
        pair = T._pair(gens[r], discs[r], xd.shape[0], nb_critic + 1)
        g = pair.iteration_graph(gens[r], discs[r], gopts[r], dopts[r], xd.shape[0], nb_critic, 10.0, seed,
                                 batch_global=Bg, row_lo=lo, split=True)
        assert lib.agx_cgan_iter_num_segs(g) == nb_critic + 2
        _lib.check(lib.agx_cgan_iter_stage(g, _dev.stream(), ptr(xd), xd.stride(0), ptr(cd), ptr(nd), offset))
        graphs.append(g)
        keep.append((xd, cd, nd))
    d_loss = g_loss = None
    for s in range(nb_critic + 2):
        for g in graphs:
            _lib.check(lib.agx_cgan_iter_launch_seg(g, _dev.stream(), s))
        nets = discs if s < nb_critic else (gens if s == nb_critic else [])
        if nets:
            total = torch.stack([n.grads for n in nets]).sum(0)
            for n in nets:
                n.grads.copy_(total)
            if s == nb_critic - 1:
                d_loss = float(discs[0].grads[discs[0].n_flat + 2])
            if s == nb_critic:
                g_loss = float(gens[0].grads[gens[0].n_flat])
    torch.cuda.synchronize()
    return d_loss, g_loss


def test_split_iteration_equals_whole_graph(gpu):
    """One rank, split into nb_critic+2 segments == the single-graph iteration, bit for bit."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=61, B=24, H=(24, 16), seed=11)
    x, cc, nc = _f32(p, 'x'), p['cc'], _f32(p, 'nc')
    _, g0, d0 = _models(p)
    T.set_seed(77)
    go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
    dl0, gl0 = T.train_iteration(x, cc, nc, g0, d0, go, do, 3)
    _, g1, d1 = _models(p)
    go1, do1 = T.RMSprop(5e-4), T.RMSprop(5e-4)
    dl1, gl1 = _iteration_by_segments(T, [g1], [d1], [go1], [do1], [(x, cc, nc, 0)], 24, 3, 77, 0)
    assert float(dl0) == dl1 and float(gl0) == gl1
    for a, b in zip(g0.get_weights() + d0.get_weights(), g1.get_weights() + d1.get_weights()):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('cuts', [(0, 12, 24), (0, 7, 16, 24)])
def test_data_parallel_shards_reproduce_the_big_batch(gpu, cuts):
    """R ranks (emulated on one GPU: replicas + summed gradient buffers) holding row shards of a
    global batch give the weights of the one-GPU run on the whole batch (same draws: every rank
    keeps its rows of the GLOBAL z / alpha streams)."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=61, B=24, H=(24, 16), seed=12)
    x, cc, nc = _f32(p, 'x'), p['cc'], _f32(p, 'nc')
    _, g0, d0 = _models(p)
    T.set_seed(5)
    dl0, gl0 = T.train_iteration(x, cc, nc, g0, d0, T.RMSprop(5e-4), T.RMSprop(5e-4), 2)
    R = len(cuts) - 1
    gens, discs = [], []
    for _ in range(R):
        _, g, d = _models(p)
        gens.append(g)
        discs.append(d)
    shards = [(x[a:b], cc[a:b], nc[a:b], a) for a, b in zip(cuts[:-1], cuts[1:])]
    dl1, gl1 = _iteration_by_segments(T, gens, discs, [T.RMSprop(5e-4) for _ in range(R)],
                                      [T.RMSprop(5e-4) for _ in range(R)], shards, 24, 2, 5, 0)
    # RMSprop sensitivity (see test_weights_after_steps): a gradient entry ~1e-7 turns a summation-order
    # difference into an O(lr) difference of its update; stated tolerance as there.
    assert abs(float(dl0) - dl1) <= 2e-4 and abs(float(gl0) - gl1) <= 2e-4
    lr = 5e-4
    for r in range(R):
        for ref, net, steps in ((g0, gens[r], 1), (d0, discs[r], 2)):
            a = np.concatenate([w.reshape(-1) for w in ref.get_weights()])
            b = np.concatenate([w.reshape(-1) for w in net.get_weights()])
            err = np.abs(a - b)
            assert np.quantile(err, 0.99) < 2e-6, np.quantile(err, 0.99)   # ~2e3 weights: 99 %
            assert err.max() <= steps * lr * np.sqrt(10) + 1e-6, err.max()
        if r:   # replicas stay bit-identical to each other (same summed gradients, same update)
            for a, b in zip(discs[0].get_weights() + gens[0].get_weights(), discs[r].get_weights() + gens[r].get_weights()):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('p_aug', [0, 0.25])
def test_large_batch_graph_paths_match_eager(gpu, p_aug):
    """Batches >= 128 take the large-batch form of the captured iteration (RMSprop fused into the big
    weight-gradient products through the LDS-tiled kernels, K = 3B >= 384): same results as the eager
    step functions.  The ragged tail of 72 rows (< 128) takes the small-batch form."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=150, B=200, H=(32, 24), seed=71)
    x32 = p['x'].astype(np.float32)
    out = []
    for use_graph in (False, True):
        _, gen, disc = _models(p)
        T._TRACE_CONSTS.clear()
        np.random.seed(5)
        T.set_seed(4321)
        go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
        tr = T.train(x32, p['cc'], _f32(p, 'nc'), p['Z'], 2, 128, gen, disc, lambda g: 0.0, lambda: None, gen_opt=go,
                     disc_opt=do, verbose=False, p_aug=p_aug, use_graph=use_graph, nb_critic=3)
        out.append((gen.get_weights(), disc.get_weights(), tr))
    (gw0, dw0, tr0), (gw1, dw1, tr1) = out
    if p_aug == 0:
        # without augmentation the captured iteration at this batch (>= the hidden widths) never forms the generated
        # rows (cgan.hip iter_ff) while the step functions do: the same algebra, rounded differently
        for a, b in zip(gw0 + dw0, gw1 + dw1):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
        np.testing.assert_allclose(tr0['gen_loss'], tr1['gen_loss'], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(tr0['disc_loss'], tr1['disc_loss'], rtol=1e-5, atol=1e-5)
        assert any(np.abs(a - b).max() > 0 for a, b in zip(gw0 + dw0, gw1 + dw1))
        return
    for a, b in zip(gw0 + dw0, gw1 + dw1):
        np.testing.assert_array_equal(a, b)
    assert tr0['gen_loss'] == tr1['gen_loss'] and tr0['disc_loss'] == tr1['disc_loss']


def test_fused_optimizer_iteration_is_bit_identical_to_eager(gpu):
    """At batch < 128 the captured one-GPU iteration applies RMSprop in the epilogue of the two big
    weight-gradient products (critic W_1[:G], generator W_out) and never stores those gradients.  At
    these sizes (wide enough for the LDS-tiled kernels that carry the fused epilogue) the weights,
    optimizer slots and losses must still equal the eager gradient -> all-of-RMSprop sequence bit for bit."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=4297, B=44, H=(256, 192), V=(6, 8, 5), Z=10, seed=91)
    x32 = p['x'].astype(np.float32)
    out = []
    for use_graph in (False, True):
        _, gen, disc = _models(p)
        T._TRACE_CONSTS.clear()
        np.random.seed(5)
        T.set_seed(77)
        go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
        tr = T.train(x32, p['cc'], _f32(p, 'nc'), p['Z'], 2, 32, gen, disc, lambda g: 0.0, lambda: None, gen_opt=go,
                     disc_opt=do, verbose=False, use_graph=use_graph, nb_critic=2)
        out.append((gen.get_weights(), disc.get_weights(), tr, do.state_arrays(disc)['rms'], go.state_arrays(gen)['rms']))
        if use_graph:   # the fused path really ran: those gradients were consumed in registers, never stored
            nv = len(p['V'])
            assert not np.any(disc.get_gradients()[nv]) and np.any(disc.get_gradients()[nv + 2])
            assert not np.any(gen.get_gradients()[nv + 4]) and np.any(gen.get_gradients()[nv + 2])
    (gw0, dw0, tr0, rd0, rg0), (gw1, dw1, tr1, rd1, rg1) = out
    for a, b in zip(gw0 + dw0 + rd0 + rg0, gw1 + dw1 + rd1 + rg1):
        np.testing.assert_array_equal(a, b)
    assert tr0['gen_loss'] == tr1['gen_loss'] and tr0['disc_loss'] == tr1['disc_loss']


def test_eager_form_of_captured_iteration_is_bit_identical(gpu):
    """agx_cgan_iter_eager queues the captured iteration's launch sequence directly on the stream (what
    the event brackets of bench.py time): same weights, slots and losses as the replayed graph."""
    from adversarial_gene_expression_amd import _dev, tf_utils as T
    p = make_problem(G=4297, B=32, H=(128, 128), V=(15, 3), Z=64, seed=8)
    xd, cd, nd = _dev.to_dev(p['x'].astype(np.float32)), _dev.to_dev(p['cc'], torch.int32), _dev.to_dev(_f32(p, 'nc'))
    out = []
    for mode in ('graph', 'eager'):
        _, gen, disc = _models(p)
        T.set_seed(99)
        go, do = T.RMSprop(5e-4), T.RMSprop(5e-4)
        losses = _dev.zeros((3, 4))
        for _ in range(2):
            if mode == 'graph':
                dl, gl = T.train_iteration(xd, cd, nd, gen, disc, go, do, nb_critic=2)
                last = (float(dl), float(gl))
            else:
                pair = T._pair(gen, disc, 32, 3)
                T._rng.offset += pair.iteration_eager(xd, cd, nd, gen, disc, go, do, 2, 10.0, T._rng.seed, T._rng.offset,
                                                      losses)
                lc = losses.cpu().numpy()
                last = (float(lc[1, 2]), float(lc[2, 0]))
        out.append((gen.get_weights() + disc.get_weights() + do.state_arrays(disc)['rms'], last, T._rng.offset))
    (w0, l0, o0), (w1, l1, o1) = out
    assert o0 == o1 and l0 == l1
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)
