"""CPU checks of the training-path oracles.

The reference pins nothing on this path (no tests, TF absent) - "parity unpinned" - so the
autograd oracle is cross-checked here against (a) finite differences and (b) the NumPy
restatement of the hand-derived backward that the HIP step launches (cgan_closed_form)."""
import numpy as np
import pytest
import torch

from helpers import make_problem, t64, tl
from oracle import cgan_closed_form as CF
from oracle import cgan_oracle as O


def _oracle_disc(p, **aug):
    gp_, dp_ = O.to_torch(p['gp']), O.to_torch(p['dp'])
    loss, gp, grads = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']), gp_, dp_,
                                            t64(p['alpha']), 10.0, **aug)
    return float(loss), float(gp), [g.numpy() for g in grads]


@pytest.mark.parametrize('cfg', [dict(), dict(B=12, H=(16,), seed=3), dict(B=7, H=(12, 10, 8), V=(6, 8, 5, 11, 5, 3, 5), seed=5),
                                 dict(B=4, n_num=2, num_zero=False, seed=7)])
def test_closed_form_disc_matches_autograd(cfg):
    p = make_problem(**cfg)
    loss, gp, grads = _oracle_disc(p)
    losses, cf = CF.disc_grads(p['x'], p['z'], p['cc'], p['nc'], p['gp'], p['dp'], p['alpha'])
    assert abs(losses[2] - loss) < 1e-10 and abs(losses[1] - gp) < 1e-10
    for a, b in zip(CF.flat(cf), grads):
        np.testing.assert_allclose(a.reshape(b.shape), b, rtol=1e-9, atol=1e-11)


def test_closed_form_disc_with_augmentation():
    p = make_problem(B=6, seed=11)
    rs = np.random.RandomState(1)
    mask = rs.binomial(1, 0.5, 6).astype(np.float64)
    nf, nr = rs.normal(0, 0.5, p['G']), rs.normal(0, 0.5, (6, p['G']))
    loss, gp, grads = _oracle_disc(p, aug_mask=t64(mask), aug_noise_fake=t64(nf), aug_noise_real=t64(nr))
    losses, cf = CF.disc_grads(p['x'], p['z'], p['cc'], p['nc'], p['gp'], p['dp'], p['alpha'],
                               aug_mask=mask, aug_noise_fake=nf, aug_noise_real=nr)
    assert abs(losses[2] - loss) < 1e-10
    for a, b in zip(CF.flat(cf), grads):
        np.testing.assert_allclose(a.reshape(b.shape), b, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize('cfg', [dict(), dict(B=12, H=(16,), seed=3), dict(B=7, H=(12, 10, 8), V=(6, 8, 5, 11, 5, 3, 5), seed=5)])
def test_closed_form_gen_matches_autograd(cfg):
    p = make_problem(**cfg)
    gp_, dp_ = O.to_torch(p['gp']), O.to_torch(p['dp'])
    loss, grads = O.gen_loss_and_grads(t64(p['z']), tl(p['cc']), t64(p['nc']), gp_, dp_)
    l2, cf = CF.gen_grads(p['z'], p['cc'], p['nc'], p['gp'], p['dp'])
    assert abs(l2 - float(loss)) < 1e-12
    for a, b in zip(CF.flat(cf), grads):
        np.testing.assert_allclose(a.reshape(tuple(b.shape)), b.numpy(), rtol=1e-9, atol=1e-12)


def test_autograd_disc_gradient_finite_differences():
    """Central differences on a few entries of every critic variable (incl. the GP's double-backward)."""
    p = make_problem(B=4, H=(8, 6), G=11, seed=13)
    _, _, grads = _oracle_disc(p)
    names = [('emb', i) for i in range(len(p['dp']['emb']))]
    for l in range(len(p['dp']['W'])):
        names += [('W', l), ('b', l)]
    rs = np.random.RandomState(0)
    eps = 1e-6
    for (k, i), g in zip(names, grads):
        arr = p['dp'][k][i]
        for _ in range(3):
            idx = tuple(rs.randint(0, s) for s in arr.shape)
            old = arr[idx]
            arr[idx] = old + eps
            lp = _oracle_disc(p)[0]
            arr[idx] = old - eps
            lm = _oracle_disc(p)[0]
            arr[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[idx]) < 1e-5 * max(1.0, abs(fd)), (k, i, idx, fd, g[idx])


def test_gp_is_zero_for_unit_gradient_linear_critic():
    """A critic that is linear in x with ||w|| = 1 has slope 1 everywhere -> gp = 0."""
    G = 9
    w = np.zeros(G)
    w[2] = 1.0
    f = lambda xi: (xi @ torch.tensor(w)).reshape(-1, 1)
    rs = np.random.RandomState(0)
    gp = O.gradient_penalty(f, t64(rs.randn(5, G)), t64(rs.randn(5, G)), t64(rs.rand(5, 1)))
    assert abs(float(gp)) < 1e-24


def test_rmsprop_and_adam_formulas():
    w0 = np.array([1.0, -2.0, 0.5])
    g = np.array([0.1, -0.2, 0.0])
    w = [torch.tensor(w0.copy())]
    opt = O.RMSprop(5e-4)
    opt.apply(w, [torch.tensor(g)])
    rms = 0.1 * g * g
    np.testing.assert_allclose(w[0].numpy(), w0 - 5e-4 * g / (np.sqrt(rms) + 1e-7), rtol=1e-12)
    w = [torch.tensor(w0.copy())]
    O.Adam(1e-3).apply(w, [torch.tensor(g)])
    m, v = 0.1 * g, 0.001 * g * g
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w[0].numpy(), w0 - lr_t * m / (np.sqrt(v) + 1e-7), rtol=1e-12)


def test_dp_sharding_equals_full_batch():
    """Shard gradients computed with 1/B_global and summed == full-batch gradients (the DP contract)."""
    p = make_problem(B=9, seed=17)
    _, full = CF.disc_grads(p['x'], p['z'], p['cc'], p['nc'], p['gp'], p['dp'], p['alpha'])
    parts = []
    for lo, hi in ((0, 5), (5, 9)):
        s = slice(lo, hi)
        _, g = CF.disc_grads(p['x'][s], p['z'][s], p['cc'][s], p['nc'][s], p['gp'], p['dp'], p['alpha'][s], B_global=9)
        parts.append(CF.flat(g))
    for a, b0, b1 in zip(CF.flat(full), *parts):
        np.testing.assert_allclose(b0 + b1, a, rtol=1e-10, atol=1e-13)
