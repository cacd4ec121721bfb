"""Edge cases of the hot path against the oracles: degenerate shapes, constant / NaN columns, single rows,
batches larger than the data set.  The NumPy oracle (pinned by the reference's goldens) defines the expected
values, including where NumPy produces NaN / inf / empty results."""
import numpy as np
import pytest
import torch

from helpers import keras_list, lowrank, make_problem, t64, tl
from oracle import cgan_oracle as O
from oracle import corr_oracle as CO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,G', [(1, 5), (2, 1), (2, 2), (3, 129), (129, 3), (64, 128), (65, 127)])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_degenerate_correlation_shapes(gpu, N, G, dtype):
    from adversarial_gene_expression_amd import utils as U
    rs = np.random.RandomState(N * 131 + G)
    x = rs.randn(N, G).astype(dtype)
    tol = 1e-12 if dtype == np.float64 else 5e-6
    with np.errstate(all='ignore'):
        want = CO.pearson_correlation(x, x)          # N = 1: every std is 0 -> all NaN, like the reference
        got = U.pearson_correlation(x, x)
        assert got.shape == want.shape and got.dtype == want.dtype
        np.testing.assert_allclose(got, want, rtol=0, atol=tol, equal_nan=True)
        wl = CO.upper_diag_list(want)
        gl = U.upper_diag_list(want)
        assert gl.shape == wl.shape                   # G = 1: empty list; NaN entries dropped
        np.testing.assert_array_equal(gl, wl)
        np.testing.assert_allclose(U.standardize(x), CO.standardize(x), rtol=1e-5 if dtype == np.float32 else 1e-12,
                                   atol=tol, equal_nan=True)


def test_constant_and_nan_columns(gpu):
    from adversarial_gene_expression_amd import utils as U
    rs = np.random.RandomState(5)
    x = lowrank(rs, 40, 70, k=5)
    x[:, 3] = 2.5                  # zero variance -> NaN row and column (utils.py:819 divides by 0)
    x[7, 11] = np.nan              # a NaN sample poisons its gene
    with np.errstate(all='ignore'):
        want = CO.pearson_correlation(x, x)
        got = U.pearson_correlation(x, x)
        assert np.array_equal(np.isnan(got), np.isnan(want))
        np.testing.assert_allclose(got, want, rtol=0, atol=5e-6, equal_nan=True)
        np.testing.assert_array_equal(U.upper_diag_list(got), CO.upper_diag_list(got))
        y = x.copy()
        g1, g2 = U.gamma_coef(x, y), CO.gamma_coef(x, y)       # same NaN pattern on both sides: lists equally long
        assert abs(float(g1) - float(g2)) < 2e-5


@pytest.mark.parametrize('B', [1, 2, 3])
def test_tiny_batches(gpu, B):
    """Batch 1 (every mean over one sample), 2, 3: losses and gradients of both steps."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(B=B, G=45, H=(20, 12), seed=70 + B)
    f = lambda k: np.asarray(p[k], np.float32)
    gen = T.Generator(p['G'], p['V'], p['n_num'], p['H'], p['Z'], weights=keras_list(p['gp']))
    disc = T.Critic(p['G'], p['V'], p['n_num'], p['H'], 1, weights=keras_list(p['dp']))
    loss = T.train_disc(f('x'), f('z'), p['cc'], f('nc'), gen, disc, T.RMSprop(5e-4), alpha=f('alpha'), apply=False)
    wl, _, want = O.disc_loss_and_grads(t64(p['x']), t64(p['z']), tl(p['cc']), t64(p['nc']), O.to_torch(p['gp']),
                                        O.to_torch(p['dp']), t64(p['alpha']))
    assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    for a, b in zip(disc.get_gradients(), want):
        b = b.numpy().reshape(a.shape)
        assert np.abs(a - b).max() <= 3e-5 * np.abs(b).max() + 1e-7
    T.set_seed(B)
    dl, gl = T.train_iteration(f('x'), p['cc'], f('nc'), gen, disc, T.RMSprop(5e-4), T.RMSprop(5e-4), 5)
    assert np.isfinite(float(dl)) and np.isfinite(float(gl))


def test_batch_size_larger_than_the_data_set(gpu):
    """train() with batch_size > N runs ONE ragged batch per epoch (tf_utils.py:309-312 slices past the end)."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=37, B=21, seed=44)
    f = lambda k: np.asarray(p[k], np.float32)
    rs = np.random.RandomState(1)
    noise = lambda kind, shape: (rs.randn(*shape) if kind == 'z' else rs.rand(*shape)).astype(np.float32)
    gen = T.Generator(p['G'], p['V'], p['n_num'], p['H'], p['Z'], weights=keras_list(p['gp']))
    disc = T.Critic(p['G'], p['V'], p['n_num'], p['H'], 1, weights=keras_list(p['dp']))
    tr = T.train(f('x'), p['cc'], f('nc'), p['Z'], 2, 64, gen, disc, lambda g: 0.0, lambda: None, verbose=False,
                 noise_fn=noise)
    rs = np.random.RandomState(1)
    tg, td = O.to_torch(p['gp']), O.to_torch(p['dp'])
    otr = O.train(p['x'].astype(np.float32).astype(np.float64), p['cc'], p['nc'], p['Z'], 2, 64, tg, td,
                  lambda g: 0.0, lambda: None, O.RMSprop(5e-4), O.RMSprop(5e-4),
                  lambda kind, shape: (rs.randn(*shape) if kind == 'z' else rs.rand(*shape)).astype(np.float32))
    np.testing.assert_allclose(tr['gen_loss'], otr['gen_loss'], rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(tr['disc_loss'], otr['disc_loss'], rtol=5e-4, atol=5e-4)
    assert tr['patience'] == otr['patience'] and tr['saved'] == otr['saved']


def test_out_of_range_category_gives_a_zero_embedding_row(gpu):
    """An index outside [0, vocab) contributes a zero embedding (TF GPU Embedding semantics) and no gradient."""
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(B=4, seed=9)
    f = lambda k: np.asarray(p[k], np.float32)
    gen = T.Generator(p['G'], p['V'], p['n_num'], p['H'], p['Z'], weights=keras_list(p['gp']))
    cc = p['cc'].copy()
    cc[1, 0] = p['V'][0] + 3
    out = gen([f('z'), cc, f('nc')]).numpy()
    gp0 = {k: [a.copy() for a in v] for k, v in p['gp'].items()}
    gp0['emb'][0] = np.vstack([gp0['emb'][0], np.zeros((8, gp0['emb'][0].shape[1]))])   # padded table: the extra rows are zero
    want = O.gen_forward(O.to_torch(gp0), t64(p['z']), tl(cc), t64(p['nc'])).numpy()
    np.testing.assert_allclose(out, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))
