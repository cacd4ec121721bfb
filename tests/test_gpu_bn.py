"""Optional batch-norm / per-feature affine stage of the Dense epilogue (BASELINE.json north_star: "fused with
bias, LeakyReLU and batch-norm") against torch-CPU float64.  The reference has NO batch-norm
(tf_utils.py:85-88), so this stage is off on every parity path; its oracle is torch's own batch_norm.
Tolerance: 2e-5 * max|y| (fp32 GEMM over K <= 18161 + one rsqrt)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (M, N, K): tiny/ragged, batch-32 generator output layer (64x64 LDS tiles), critic layer 1 at 512 rows (split-K +
# reduce epilogue), skinny direct kernel
SHAPES = [(5, 7, 3), (33, 130, 71), (192, 4297, 128), (1024, 256, 18161), (32, 256, 4000)]


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('slope', [0.0, 0.2])
def test_dense_bn_inference_and_batch_statistics(gpu, M, N, K, slope):
    from adversarial_gene_expression_amd import _dev, ops
    rs = np.random.RandomState(M + N + K)
    x = rs.randn(M, K).astype(np.float32)
    W = (rs.randn(K, N) / np.sqrt(K)).astype(np.float32)
    b = rs.uniform(-0.5, 0.5, N).astype(np.float32)
    gamma, beta = rs.uniform(0.5, 1.5, N).astype(np.float32), rs.uniform(-0.3, 0.3, N).astype(np.float32)
    rmean, rvar = rs.uniform(-0.5, 0.5, N).astype(np.float32), rs.uniform(0.3, 2.0, N).astype(np.float32)
    eps = 1e-3                                   # Keras BatchNormalization default
    xd, Wd = _dev.padded_matrix(x)[0], _dev.padded_matrix(W)[0]
    dev = lambda a: _dev.to_dev(a)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    pre = t(x) @ t(W) + t(b)
    act = lambda y: F.leaky_relu(y, slope) if slope else torch.relu(y)
    # inference: running statistics folded into the GEMM epilogue (one launch)
    y, _, _ = ops.dense_bn(xd, Wd, dev(b), M, N, K, gamma=dev(gamma), beta=dev(beta), eps=eps, slope=slope,
                           running=(dev(rmean), dev(rvar)))
    want = act(F.batch_norm(pre, t(rmean), t(rvar), t(gamma), t(beta), False, 0.0, eps)).numpy()
    got = y[:, :N].cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    # training: statistics of this batch (biased variance)
    if M > 1:
        y, mean, var = ops.dense_bn(xd, Wd, dev(b), M, N, K, gamma=dev(gamma), beta=dev(beta), eps=eps, slope=slope)
        want = act(F.batch_norm(pre, None, None, t(gamma), t(beta), True, 0.0, eps)).numpy()
        got = y[:, :N].cpu().numpy()
        assert np.abs(got - want).max() <= 5e-5 * max(1.0, np.abs(want).max())
        np.testing.assert_allclose(mean.cpu().numpy(), pre.mean(0).numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(var.cpu().numpy(), pre.var(0, unbiased=False).numpy(), rtol=1e-4, atol=1e-6)


def test_affine_stage_is_off_by_default_and_validated(gpu):
    """A descriptor without col_scale gives bit-identical results to the round-1 epilogue; one pointer
    without the other is rejected."""
    from adversarial_gene_expression_amd import _dev, _lib, ops
    rs = np.random.RandomState(0)
    x, W = _dev.to_dev(rs.randn(40, 64).astype(np.float32)), _dev.to_dev(rs.randn(64, 96).astype(np.float32))
    b = _dev.to_dev(rs.randn(96).astype(np.float32))
    one, zero = _dev.to_dev(np.ones(96, np.float32)), _dev.to_dev(np.zeros(96, np.float32))
    y0 = ops.gemm(x, W, 40, 96, 64, bias=b, act=1)
    y1 = ops.gemm(x, W, 40, 96, 64, bias=b, act=1, col_scale=one, col_shift=zero)
    assert torch.equal(y0, y1)
    with pytest.raises(_lib.AgxError):
        ops.gemm(x, W, 40, 96, 64, bias=b, col_scale=one)


def _bn_models(G, V, H, Z, seed):
    """Generator with batch-norm + critic, the same weights in the product and in the torch-CPU oracle."""
    from helpers import keras_list, make_problem
    from oracle import cgan_bn_oracle as BO
    from adversarial_gene_expression_amd import tf_utils as T
    p = make_problem(G=G, V=V, H=H, Z=Z, B=48, seed=seed)
    bn = BO.init_bn(H, seed=seed + 5)
    ws = [np.asarray(a, np.float32) for a in p['gp']['emb']]
    for l, (w, b) in enumerate(zip(p['gp']['W'], p['gp']['b'])):
        ws += [np.asarray(w, np.float32), np.asarray(b, np.float32)]
        if l < len(H):
            ws += [np.asarray(bn[l][k], np.float32) for k in ('gamma', 'beta', 'mean', 'var')]
    gen = T.Generator(G, list(V), 1, list(H), Z, weights=ws, batch_norm=True)
    disc = T.Critic(G, list(V), 1, list(H), 1, weights=keras_list(p['dp']))
    return T, BO, p, bn, gen, disc


def _gen_lists(BO, tg, tbn, gen, which):
    """Oracle tensors and product arrays in the product's get_weights() order."""
    want, nv = [t for t in tg['emb']], len(tg['emb'])
    for l, (w, b) in enumerate(zip(tg['W'], tg['b'])):
        want += [w, b]
        if l < len(tbn):
            want += [tbn[l][k] for k in ('gamma', 'beta', 'mean', 'var')]
    return [t.detach().numpy() for t in want], which


@pytest.mark.parametrize('G,H,B', [(37, (16, 8), 5), (700, (64, 48), 33), (4297, (128, 128), 32)])
def test_generator_batch_norm_in_the_train_step(gpu, G, H, B):
    """`make_generator(..., batch_norm=True)`: Dense -> BatchNormalization -> ReLU in the generator's hidden layers,
    through the eager step functions.  train_disc runs the generator on its moving statistics (training=False,
    tf_utils.py:206), train_gen on batch statistics with the hand-derived backward and the moving-statistics update
    (training=True, :244), predict on the moving statistics again.  Against the torch-CPU float64 oracle
    (oracle/cgan_bn_oracle.py; no reference counterpart - the reference has no batch-norm, tf_utils.py:85-88):
    losses 1e-5 relative, every gradient 2e-4 * max|grad|, weights / moving statistics after three alternating
    RMSprop rounds: median error < 1e-6, 99 % < 1e-4, none beyond a few steps of lr * sqrt(10) - the stated bound of
    test_gpu_cgan.test_weights_after_steps: RMSprop divides by sqrt(EMA(g^2)) + 1e-7, so a gradient entry near 1e-7
    turns fp32 rounding into an O(lr) difference of its update, and batch statistics over 5-33 rows add their own."""
    from helpers import t64, tl
    from oracle import cgan_oracle as O
    V, Z = (15, 3), 6
    T, BO, p, bn, gen, disc = _bn_models(G, V, H, Z, seed=G)
    s = slice(0, B)
    f32 = lambda a: np.asarray(a, np.float32)
    x, z, cc, nc, al = f32(p['x'][s]), f32(p['z'][s]), p['cc'][s], f32(p['nc'][s]), f32(p['alpha'][s])
    tg, td, tbn = O.to_torch(p['gp']), O.to_torch(p['dp']), BO.bn_to_torch(bn)
    args64 = (t64(x), t64(z), tl(cc), t64(nc))
    assert len(gen.trainable_variables) == len(gen.get_weights()) - 2 * len(H)      # moving statistics are not trainable

    def check(got, want, what):
        for i, (a, b) in enumerate(zip(got, want)):
            b = np.asarray(b).reshape(a.shape)
            err = float(np.max(np.abs(a - b))) if a.size else 0.0
            assert err <= 2e-4 * float(np.max(np.abs(b))) + 1e-7, (what, i, err, float(np.max(np.abs(b))))

    # critic step: generator in inference mode
    loss = T.train_disc(x, z, cc, nc, gen, disc, T.RMSprop(5e-4), alpha=al, apply=False)
    wl, want = BO.disc_loss_and_grads(*args64, tg, tbn, td, t64(al))
    assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    check(disc.get_gradients(), [g.numpy() for g in want], 'critic')
    # generator step: training mode (gradients only; the moving statistics move in both)
    loss = T.train_gen(z, cc, nc, gen, disc, T.RMSprop(5e-4), apply=False)
    wl, want = BO.gen_loss_and_grads(t64(z), tl(cc), t64(nc), tg, tbn, td)
    assert abs(float(loss) - float(wl)) <= 1e-5 * max(1.0, abs(float(wl)))
    got = gen.get_gradients()
    nv, k, gi = len(V), len(V), 0
    for l in range(len(H) + 1):                       # product order per layer: W, b, gamma, beta, mean, var
        n = 6 if l < len(H) else 2
        check(got[k:k + min(n, 4)], [g.numpy() for g in want[nv + gi:nv + gi + min(n, 4)]], 'generator layer %d' % l)
        if n == 6:                                    # the moving statistics carry no gradient
            assert not np.any(got[k + 4]) and not np.any(got[k + 5])
        k += n
        gi += min(n, 4)
    check(got[:nv], [g.numpy() for g in want[:nv]], 'generator embeddings')
    w_now, _ = _gen_lists(BO, tg, tbn, gen, None)
    for a, b in zip(gen.get_weights(), w_now):        # the moving statistics took one update, nothing else moved
        np.testing.assert_allclose(a, np.asarray(b).reshape(a.shape), rtol=2e-6, atol=2e-7)
    # three alternating rounds with RMSprop (2 critic steps + 1 generator step each)
    go, do, ogo, odo = T.RMSprop(5e-4), T.RMSprop(5e-4), O.RMSprop(5e-4), O.RMSprop(5e-4)
    rs = np.random.RandomState(3)
    for it in range(3):
        for _ in range(2):
            zz, aa = f32(rs.randn(B, Z)), f32(rs.rand(B, 1))
            l0 = T.train_disc(x, zz, cc, nc, gen, disc, do, alpha=aa)
            l1 = BO.train_disc(t64(x), t64(zz), tl(cc), t64(nc), tg, tbn, td, odo, t64(aa))
            assert abs(float(l0) - float(l1)) <= 1e-4 * max(1.0, abs(float(l1)))
        zz = f32(rs.randn(B, Z))
        l0 = T.train_gen(zz, cc, nc, gen, disc, go)
        l1 = BO.train_gen(t64(zz), tl(cc), t64(nc), tg, tbn, td, ogo)
        assert abs(float(l0) - float(l1)) <= 1e-4 * max(1.0, abs(float(l1)))
    w_now, _ = _gen_lists(BO, tg, tbn, gen, None)
    a = np.concatenate([w.reshape(-1) for w in gen.get_weights()])
    b = np.concatenate([np.asarray(w).reshape(-1) for w in w_now])
    err = np.abs(a - b)
    print('batch-norm generator after 3 rounds: median %.3g q99 %.3g max %.3g' % (np.median(err), np.quantile(err, 0.99), err.max()))
    assert np.median(err) < 1e-6 and np.quantile(err, 0.99) < 1e-4 and err.max() < 3 * 5e-4 * np.sqrt(10) + 1e-6, \
        (np.median(err), np.quantile(err, 0.99), err.max())
    # predict: moving statistics (the oracle takes the product's current weights: the few O(lr) outliers above would
    # otherwise show in its output)
    ws = gen.get_weights()
    nv = len(V)
    pg = {'emb': ws[:nv], 'W': [], 'b': []}
    pbn, k = [], nv
    for l in range(len(H) + 1):
        pg['W'].append(ws[k])
        pg['b'].append(ws[k + 1])
        if l < len(H):
            pbn.append(dict(gamma=ws[k + 2], beta=ws[k + 3], mean=ws[k + 4], var=ws[k + 5]))
        k += 6 if l < len(H) else 2
    zz = f32(rs.randn(B, Z))
    got = T.predict(cc, nc, gen, z=zz)
    with torch.no_grad():
        want = BO.gen_forward(O.to_torch(pg), BO.bn_to_torch(pbn), t64(zz), tl(cc), t64(nc), training=False).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))


def test_batch_norm_generator_trains_through_train_and_round_trips_through_save(gpu, tmp_path):
    """train() with a batch-norm generator takes the eager step functions (no captured iteration), the critic refuses
    the option, save / load_model keep the four batch-norm vectors per layer."""
    from adversarial_gene_expression_amd import tf_utils as T
    rs = np.random.RandomState(0)
    G, V, Z = 61, [15, 3], 6
    x = rs.randn(70, G).astype(np.float32)
    cc = np.stack([rs.randint(0, v, 70) for v in V], 1).astype(np.int32)
    nc = np.zeros((70, 1), np.float32)
    gen = T.make_generator(G, V, 1, [24, 16], Z, batch_norm=True, seed=1)
    disc = T.make_discriminator(G, V, 1, [24, 16], seed=2)
    with pytest.raises(NotImplementedError):
        T.make_discriminator(G, V, 1, [24, 16], batch_norm=True)
    with pytest.raises(ValueError):
        T.train_iteration(x[:32], cc[:32], nc[:32], gen, disc, T.RMSprop(5e-4), T.RMSprop(5e-4))
    mv0 = gen.get_weights()[len(V) + 5].copy()
    tr = T.train(x, cc, nc, Z, 2, 32, gen, disc, lambda g: 0.0, lambda: None, verbose=False)
    assert np.all(np.isfinite(tr['gen_loss'])) and np.all(np.isfinite(tr['disc_loss']))
    assert np.abs(gen.get_weights()[len(V) + 5] - mv0).max() > 0          # the moving variance moved
    path = str(tmp_path / 'gen_bn')
    gen.save(path)
    g2 = T.load_model(path)
    assert g2.batch_norm and len(g2.get_weights()) == len(gen.get_weights())
    for a, b in zip(gen.get_weights(), g2.get_weights()):
        np.testing.assert_array_equal(a, b)
    z = rs.randn(9, Z).astype(np.float32)
    np.testing.assert_array_equal(T.predict(cc[:9], nc[:9], gen, z=z), T.predict(cc[:9], nc[:9], g2, z=z))
