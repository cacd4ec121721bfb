"""HBM-bound kernels of the train step (csrc/elementwise.hip) against NumPy, through the C-ABI."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _libs():
    from adversarial_gene_expression_amd import _dev, _lib
    return _dev, _lib, _lib.load()


def _embed_desc(_lib, vocab, edim, offs):
    e = _lib.EmbedDesc()
    e.n_cat = len(vocab)
    for i, (v, d, o) in enumerate(zip(vocab, edim, offs)):
        e.vocab[i], e.edim[i], e.table_off[i] = v, d, o
    return e


@pytest.mark.parametrize('B,row_mod', [(1, 0), (12, 0), (33, 0), (36, 12)])
def test_embed_concat_fwd_bwd(gpu, B, row_mod):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(B)
    vocab, edim = [6, 8, 5, 11], [3, 3, 3, 4]
    offs, off = [], 0
    for v, d in zip(vocab, edim):
        offs.append(off)
        off += (v * d + 63) & ~63
    tables = rs.randn(off).astype(np.float32)
    src = row_mod or B
    lead = rs.randn(src, 5).astype(np.float32)
    num = rs.randn(src, 2).astype(np.float32)
    cat = np.stack([rs.randint(0, v, src) for v in vocab], 1).astype(np.int32)
    width = 5 + 2 + sum(edim)
    e = _embed_desc(_lib, vocab, edim, offs)
    out = _dev.zeros((B, width + 3))
    td, ld_, nd, cd = _dev.to_dev(tables), _dev.to_dev(lead), _dev.to_dev(num), _dev.to_dev(cat, torch.int32)
    _lib.check(lib.agx_embed_concat_fwd(_dev.stream(), e, ptr(td), ptr(ld_), 5, 5, ptr(nd), 2, 2, ptr(cd), 4, B,
                                        row_mod, ptr(out), width + 3))
    rows = np.arange(B) % src
    ref = np.concatenate([lead[rows], num[rows]] + [tables[o:o + v * d].reshape(v, d)[cat[rows, i]]
                                                     for i, (v, d, o) in enumerate(zip(vocab, edim, offs))], 1)
    np.testing.assert_array_equal(out.cpu().numpy()[:, :width], ref)
    # backward: scatter-add (duplicates summed), untouched rows get exact zeros
    dout = rs.randn(B, width + 3).astype(np.float32)
    grad = _dev.to_dev(np.full(off, 7.0, np.float32))
    dd = _dev.to_dev(dout)
    _lib.check(lib.agx_embed_bwd(_dev.stream(), e, ptr(grad), ptr(dd), width + 3, 7, ptr(cd), 4, B, row_mod, 0))
    g = grad.cpu().numpy()
    c0 = 7
    for i, (v, d, o) in enumerate(zip(vocab, edim, offs)):
        want = np.zeros((v, d))
        np.add.at(want, cat[rows, i], dout[:, c0:c0 + d].astype(np.float64))
        np.testing.assert_allclose(g[o:o + v * d].reshape(v, d), want, atol=1e-5)
        c0 += d


def test_out_of_range_index_gives_zero_row(gpu):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    e = _embed_desc(_lib, [4], [3], [0])
    tables = _dev.to_dev(np.arange(12, dtype=np.float32) + 1)
    cat = _dev.to_dev(np.array([[1], [9], [-1]], np.int32), torch.int32)
    out = _dev.empty((3, 4))
    _lib.check(lib.agx_embed_concat_fwd(_dev.stream(), e, ptr(tables), None, 0, 0, None, 0, 0, ptr(cat), 1, 3, 0,
                                        ptr(out), 4))
    np.testing.assert_array_equal(out.cpu().numpy()[:, :3], [[4, 5, 6], [0, 0, 0], [0, 0, 0]])


@pytest.mark.parametrize('B,G', [(1, 1), (12, 1076), (33, 4297)])
def test_interp_aug_rownorm(gpu, B, G):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(G)
    real, fake = rs.randn(B, G).astype(np.float32), rs.randn(B, G).astype(np.float32)
    alpha = rs.rand(B).astype(np.float32)
    rd, fd = _dev.to_dev(real), _dev.to_dev(fake)
    inter = _dev.empty((B, G))
    ad = _dev.to_dev(alpha)   # keep every device buffer referenced until its kernel has been enqueued
    _lib.check(lib.agx_gp_interp(_dev.stream(), ptr(rd), G, ptr(fd), G, ptr(ad), ptr(inter), G, B, G))
    np.testing.assert_array_equal(inter.cpu().numpy(), real + alpha[:, None] * (fake - real))
    # augmentation: broadcast row and per-row noise
    mask = rs.binomial(1, 0.5, B).astype(np.float32)
    nf, nr = rs.randn(G).astype(np.float32), rs.randn(B, G).astype(np.float32)
    md, nfd, nrd = _dev.to_dev(mask), _dev.to_dev(nf), _dev.to_dev(nr)
    _lib.check(lib.agx_aug_add(_dev.stream(), ptr(fd), G, ptr(md), ptr(nfd), G, 1, B, G))
    _lib.check(lib.agx_aug_add(_dev.stream(), ptr(rd), G, ptr(md), ptr(nrd), G, 0, B, G))
    np.testing.assert_array_equal(fd.cpu().numpy(), fake + mask[:, None] * nf[None, :])
    np.testing.assert_array_equal(rd.cpu().numpy(), real + mask[:, None] * nr)
    # slopes and the scaled gradient
    g = rs.randn(B, G).astype(np.float32)
    gd = _dev.to_dev(g)
    slopes = _dev.empty(B)
    coef = 10.0 * 2.0 / B
    _lib.check(lib.agx_gp_rownorm_scale(_dev.stream(), ptr(gd), G, ptr(slopes), B, G, coef))
    s = np.sqrt(np.sum(g.astype(np.float64) ** 2, axis=1))
    np.testing.assert_allclose(slopes.cpu().numpy(), s, rtol=2e-6)
    np.testing.assert_allclose(gd.cpu().numpy(), g * (coef * (s - 1) / s)[:, None], rtol=2e-5, atol=1e-7)


def test_zero_slope_gives_nan_like_the_reference(gpu):
    """sqrt has no epsilon (tf_utils.py:169): a zero gradient row yields NaN, not a silent 0."""
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    gd = _dev.zeros((2, 8))
    gd[1] = 1.0
    slopes = _dev.empty(2)
    _lib.check(lib.agx_gp_rownorm_scale(_dev.stream(), ptr(gd), 8, ptr(slopes), 2, 8, 1.0))
    out = gd.cpu().numpy()
    assert np.all(np.isnan(out[0])) and np.all(np.isfinite(out[1]))


def test_rowdot_delta_init_colsums_loss(gpu):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(2)
    B, H = 11, 100
    R = 3 * B
    A = rs.randn(R, H).astype(np.float32)
    w = rs.randn(H).astype(np.float32)
    b = np.array([0.3], np.float32)
    Ad, wd = _dev.to_dev(A), _dev.to_dev(w)
    out = _dev.empty(R)
    bd = _dev.to_dev(b)
    _lib.check(lib.agx_rowdot(_dev.stream(), ptr(Ad), H, ptr(wd), ptr(bd), ptr(out), R, H))
    np.testing.assert_allclose(out.cpu().numpy(), A.astype(np.float64) @ w + 0.3, atol=2e-5)
    d = _dev.empty((R, H))
    inv = 1.0 / B
    _lib.check(lib.agx_delta_init(_dev.stream(), ptr(Ad), H, ptr(wd), ptr(d), H, R, H, B, B, -inv, inv, 1.0, 0.1))
    c = np.concatenate([np.full(B, -inv), np.full(B, inv), np.ones(B)]).astype(np.float32)
    ref = np.where(A > 0, 1.0, 0.1) * (c[:, None] * w[None, :])
    np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-6, atol=1e-9)
    cs = _dev.empty(H)
    _lib.check(lib.agx_colsum(_dev.stream(), ptr(Ad), H, 0, 2 * B, H, ptr(cs), 0))
    np.testing.assert_allclose(cs.cpu().numpy(), A[:2 * B].astype(np.float64).sum(0), atol=2e-5)
    cw, bo = _dev.empty(H), _dev.empty(1)
    _lib.check(lib.agx_colsum_weighted(_dev.stream(), ptr(Ad), H, R, H, B, B, -inv, inv, 1.0, ptr(cw), ptr(bo)))
    np.testing.assert_allclose(cw.cpu().numpy(), (c[:, None].astype(np.float64) * A).sum(0), atol=2e-5)
    assert float(bo.item()) == 0.0
    slopes = rs.rand(B).astype(np.float32) + 0.5
    losses = _dev.empty(4)
    sd = _dev.to_dev(slopes)
    _lib.check(lib.agx_wgan_loss(_dev.stream(), ptr(out), B, B, ptr(sd), B, inv, 10.0, 1.0, ptr(losses)))
    o = out.cpu().numpy().astype(np.float64)
    wg = -o[:B].mean() + o[B:2 * B].mean()
    gp = np.mean((slopes.astype(np.float64) - 1) ** 2)
    np.testing.assert_allclose(losses.cpu().numpy()[:3], [wg, gp, wg + 10 * gp], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('n', [1, 3, 4, 1027, 4715587])
def test_optimizers(gpu, n):
    from adversarial_gene_expression_amd import ops, _dev
    rs = np.random.RandomState(n % 97)
    w0 = rs.randn(n).astype(np.float32)
    g = (rs.randn(n) * 10.0 ** rs.uniform(-6, 0, n)).astype(np.float32)
    g[::7] = 0.0
    w, rms = _dev.to_dev(w0), _dev.zeros(n)
    gd = _dev.to_dev(g)
    ops.rmsprop(w, gd, rms, 5e-4)
    ops.rmsprop(w, gd, rms, 5e-4)
    g64, w64, r64 = g.astype(np.float64), w0.astype(np.float64), np.zeros(n)
    for _ in range(2):
        r64 = 0.9 * r64 + 0.1 * g64 * g64
        w64 = w64 - 5e-4 * g64 / (np.sqrt(r64) + 1e-7)
    np.testing.assert_allclose(rms.cpu().numpy(), r64, rtol=1e-5, atol=1e-30)
    # update magnitude is <= lr*sqrt(10) per step; fp32 rounding of w dominates the error
    np.testing.assert_allclose(w.cpu().numpy(), w64, rtol=0, atol=5e-7 + 1e-6 * np.abs(w64).max())
    np.testing.assert_array_equal(w.cpu().numpy()[::7], w0[::7])   # zero gradient => untouched weight
    w, m, v = _dev.to_dev(w0), _dev.zeros(n), _dev.zeros(n)
    w64, m64, v64 = w0.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in (1, 2):
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam(w, gd, m, v, lr_t)
        m64 = 0.9 * m64 + 0.1 * g64
        v64 = 0.999 * v64 + 0.001 * g64 * g64
        w64 = w64 - lr_t * m64 / (np.sqrt(v64) + 1e-7)
    np.testing.assert_allclose(w.cpu().numpy(), w64, rtol=0, atol=5e-7 + 1e-6 * np.abs(w64).max())


def test_rng_streams(gpu):
    from adversarial_gene_expression_amd import ops
    n = 1 << 20
    a = ops.rng(n, 123, 0, True).cpu().numpy()
    b = ops.rng(n, 123, 0, True).cpu().numpy()
    c = ops.rng(n, 124, 0, True).cpu().numpy()
    d = ops.rng(n, 123, n, True).cpu().numpy()
    np.testing.assert_array_equal(a, b)                 # deterministic in (seed, offset)
    assert not np.array_equal(a, c) and not np.array_equal(a, d)
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3 and np.all(np.isfinite(a))
    assert abs(np.mean(a ** 4) - 3.0) < 0.05            # kurtosis of N(0,1)
    u = ops.rng(n, 9, 0, False).cpu().numpy()
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 1e-3
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5e-3
    odd = ops.rng(5, 1, 0, True).cpu().numpy()           # ragged tail of a 4-wide counter block
    assert np.all(np.isfinite(odd))


def test_critic_head_and_gp_finalize(gpu):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(8)
    B, H = 7, 48
    R = 3 * B
    A = rs.randn(R, H).astype(np.float32)
    w, b = rs.randn(H).astype(np.float32), np.array([-0.2], np.float32)
    Ad, wd, bd = _dev.to_dev(A), _dev.to_dev(w), _dev.to_dev(b)
    out, d = _dev.empty(R), _dev.empty((R, H))
    inv = 1.0 / B
    _lib.check(lib.agx_critic_head(_dev.stream(), ptr(Ad), H, ptr(wd), ptr(bd), ptr(out), ptr(d), H, R, H, B, B,
                                   -inv, inv, 1.0, 0.0))
    np.testing.assert_allclose(out.cpu().numpy(), A.astype(np.float64) @ w - 0.2, atol=2e-5)
    c = np.concatenate([np.full(B, -inv), np.full(B, inv), np.ones(B)]).astype(np.float32)
    np.testing.assert_allclose(d.cpu().numpy(), np.where(A > 0, 1.0, 0.0) * (c[:, None] * w[None, :]), rtol=1e-6, atol=1e-9)
    # finalisation from per-slab partial sums
    nslab, ncov, ldx = 11, 7, 20
    part = rs.rand(B, nslab).astype(np.float32)
    d1 = rs.randn(B, H).astype(np.float32)
    xc = rs.randn(B, ldx).astype(np.float32)
    pd, d1d, xcd = _dev.to_dev(part), _dev.to_dev(d1), _dev.to_dev(xc)
    slopes, fs = _dev.empty(B), _dev.empty(B)
    coef = 10.0 * 2.0 / B
    _lib.check(lib.agx_gp_finalize(_dev.stream(), ptr(pd), nslab, nslab, ptr(slopes), ptr(fs), ptr(d1d), H, H,
                                   xcd.data_ptr() + 4 * 3, ldx, ncov, B, coef))
    s = np.sqrt(part.astype(np.float64).sum(1))
    f = coef * (s - 1) / s
    np.testing.assert_allclose(slopes.cpu().numpy(), s, rtol=2e-6)
    np.testing.assert_allclose(fs.cpu().numpy(), f, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(d1d.cpu().numpy(), d1 * f[:, None], rtol=2e-5, atol=1e-7)
    got = xcd.cpu().numpy()
    assert np.all(got[:, 3:3 + ncov] == 0) and np.array_equal(got[:, :3], xc[:, :3]) and np.array_equal(got[:, 10:], xc[:, 10:])


def test_rng_multi_matches_single_segment_streams(gpu):
    """All draws of an iteration in one launch == the per-call streams at the same positions."""
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd import ops
    from adversarial_gene_expression_amd._dev import ptr
    base = _dev.zeros(2, dtype=torch.int64)
    _lib.check(lib.agx_set_u64(_dev.stream(), ptr(base), 40))
    spec = [(37 * 6, 0, 1), (37, 224, 0), (5, 264, 1), (1027, 272, 0)]
    outs = [_dev.empty(n) for n, _, _ in spec]
    segs = (_lib.RngSeg * len(spec))()
    for s, o, (n, sub, normal) in zip(segs, outs, spec):
        s.out, s.n, s.sub_offset, s.normal = ptr(o), n, sub, normal
    _lib.check(lib.agx_rng_multi(_dev.stream(), 99, ptr(base), len(spec), segs))
    for o, (n, sub, normal) in zip(outs, spec):
        np.testing.assert_array_equal(o.cpu().numpy(), ops.rng(n, 99, 40 + sub, bool(normal)).cpu().numpy())


@pytest.mark.parametrize('aug', [False, True])
def test_stack_rows(gpu, aug):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(3)
    vocab, edim = [6, 3], [3, 2]
    offs = [0, 64]
    tables = rs.randn(128).astype(np.float32)
    B, G, n_num = 5, 37, 2
    ncov = n_num + sum(edim)
    ldx = 48
    src = rs.randn(B, 40).astype(np.float32)
    num = rs.randn(B, n_num).astype(np.float32)
    cat = np.stack([rs.randint(0, v, B) for v in vocab], 1).astype(np.int32)
    mask = rs.binomial(1, 0.5, B).astype(np.float32)
    noise = rs.randn(G).astype(np.float32)
    X0 = rs.randn(3 * B, ldx).astype(np.float32)
    X = _dev.to_dev(X0)
    e = _embed_desc(_lib, vocab, edim, offs)
    md, nz = (_dev.to_dev(mask), _dev.to_dev(noise)) if aug else (None, None)
    td, sd, nd, cd = _dev.to_dev(tables), _dev.to_dev(src), _dev.to_dev(num), _dev.to_dev(cat, torch.int32)
    _lib.check(lib.agx_stack_rows(_dev.stream(), e, ptr(td), ptr(sd), 40, B, B, ptr(nd), n_num, n_num, ptr(cd), 2, B,
                                  2 * B, G, ptr(X), ldx, ptr(md), ptr(nz), 0))
    want = X0.copy()
    want[B:2 * B, :G] = src[:, :G] + (mask[:, None] * noise[None, :] if aug else 0)
    rows = np.arange(2 * B) % B
    cov = np.concatenate([num[rows]] + [tables[o:o + v * d].reshape(v, d)[cat[rows, i]]
                                        for i, (v, d, o) in enumerate(zip(vocab, edim, offs))], 1)
    want[:2 * B, G:G + ncov] = cov
    np.testing.assert_array_equal(X.cpu().numpy(), want)


@pytest.mark.parametrize('B,H', [(5, 48), (70, 256), (33, 600)])
def test_small_grads_matches_separate_entry_points(gpu, B, H):
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(B)
    R = 3 * B
    vocab, edim, offs = [6, 3], [3, 2], [0, 64]
    e = _embed_desc(_lib, vocab, edim, offs)
    col0, nin = 11, 11 + sum(edim)
    D1 = rs.randn(R, H).astype(np.float32)
    D2 = rs.randn(R, 40).astype(np.float32)
    AL = rs.randn(R, 40).astype(np.float32)
    W = rs.randn(nin, H).astype(np.float32)
    cat = np.stack([rs.randint(0, v, B) for v in vocab], 1).astype(np.int32)
    out = rs.randn(R).astype(np.float32)
    slopes = (1 + 0.3 * rs.randn(B)).astype(np.float32)
    d = {k: _dev.to_dev(v) for k, v in dict(D1=D1, D2=D2, AL=AL, W=W, out=out, slopes=slopes).items()}
    catd = _dev.to_dev(cat, torch.int32)
    db1, db2, dw, dbo = _dev.empty(H), _dev.empty(40), _dev.empty(40), _dev.empty(1)
    grad = _dev.to_dev(np.full(128, 7.0, np.float32))
    losses = _dev.empty(4)
    inv = 1.0 / B
    j = _lib.SmallGrads()
    j.n_colsum = 2
    j.cs_A[0], j.cs_lda[0], j.cs_r0[0], j.cs_r1[0], j.cs_H[0], j.cs_out[0] = ptr(d['D1']), H, 0, 2 * B, H, ptr(db1)
    j.cs_A[1], j.cs_lda[1], j.cs_r0[1], j.cs_r1[1], j.cs_H[1], j.cs_out[1] = ptr(d['D2']), 40, 0, 2 * B, 40, ptr(db2)
    j.w_A, j.w_lda, j.w_R, j.w_H, j.w_n0, j.w_n1 = ptr(d['AL']), 40, R, 40, B, B
    j.w_c0, j.w_c1, j.w_c2 = -inv, inv, 1.0
    j.w_out, j.w_bias_out = ptr(dw), ptr(dbo)
    j.emb_grad, j.emb = ptr(grad), e
    j.emb_D, j.emb_W, j.emb_cat = ptr(d['D1']), ptr(d['W']), ptr(catd)
    j.emb_ldd, j.emb_H, j.emb_ldw, j.emb_col0, j.emb_ld_cat, j.emb_B, j.emb_rows = H, H, H, col0, 2, B, 2 * B
    j.l_losses, j.l_out, j.l_slopes = ptr(losses), ptr(d['out']), ptr(d['slopes'])
    j.l_n0, j.l_n1, j.l_ns = B, B, B
    j.l_inv_b, j.l_gp_weight, j.l_fake_sign = inv, 10.0, 1.0
    _lib.check(lib.agx_small_grads(_dev.stream(), j))
    np.testing.assert_allclose(db1.cpu().numpy(), D1[:2 * B].astype(np.float64).sum(0), atol=2e-5)
    np.testing.assert_allclose(db2.cpu().numpy(), D2[:2 * B].astype(np.float64).sum(0), atol=2e-5)
    c = np.concatenate([np.full(B, -inv), np.full(B, inv), np.ones(B)])
    np.testing.assert_allclose(dw.cpu().numpy(), (c[:, None] * AL.astype(np.float64)).sum(0), atol=2e-5)
    assert abs(float(dbo.cpu().numpy()[0])) < 1e-6
    g = grad.cpu().numpy()
    rows = np.arange(2 * B) % B
    c0 = col0
    for i, (v, dd, o) in enumerate(zip(vocab, edim, offs)):
        dcov = D1[:2 * B].astype(np.float64) @ W[c0:c0 + dd].astype(np.float64).T
        want = np.zeros((v, dd))
        np.add.at(want, cat[rows, i], dcov)
        np.testing.assert_allclose(g[o:o + v * dd].reshape(v, dd), want, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(want).max()))
        c0 += dd
    assert np.all(g[18:64] == 7.0)        # padding between tables untouched
    ls = losses.cpu().numpy()
    w = -out[:B].astype(np.float64).sum() * inv + out[B:2 * B].astype(np.float64).sum() * inv
    gp = ((slopes.astype(np.float64) - 1) ** 2).sum() * inv
    np.testing.assert_allclose(ls[:3], [w, gp, w + 10 * gp], rtol=1e-5, atol=1e-5)


def test_small_grads_with_the_maximum_number_of_jobs(gpu):
    """AGX_MAX_CAT embedding tables + AGX_MAX_LAYERS column sums + the weighted sum + the loss job in one launch:
    the host-computed block -> job segment table at its largest (elementwise.hip SmallPlan)."""
    _dev, _lib, lib = _libs()
    from adversarial_gene_expression_amd._dev import ptr
    rs = np.random.RandomState(77)
    B, H, NL, NC = 9, 72, 8, 16
    R = 3 * B
    vocab = [int(v) for v in rs.randint(1, 9, NC)]
    edim = [int(v) for v in rs.randint(1, 5, NC)]
    offs, o = [], 0
    for v, dd in zip(vocab, edim):
        offs.append(o)
        o += v * dd + 3                       # 3 floats of padding between tables
    e = _embed_desc(_lib, vocab, edim, offs)
    col0 = 5
    nin = col0 + sum(edim)
    D1 = rs.randn(R, H).astype(np.float32)
    W = rs.randn(nin, H).astype(np.float32)
    widths = [int(w) for w in rs.randint(1, 200, NL)]
    Ds = [rs.randn(R, w).astype(np.float32) for w in widths]
    AL = rs.randn(R, 33).astype(np.float32)
    cat = np.stack([rs.randint(0, v, B) for v in vocab], 1).astype(np.int32)
    out = rs.randn(R).astype(np.float32)
    slopes = (1 + 0.3 * rs.randn(B)).astype(np.float32)
    dD1, dW, dAL, dout, dsl = (_dev.to_dev(a) for a in (D1, W, AL, out, slopes))
    dDs = [_dev.to_dev(a) for a in Ds]
    catd = _dev.to_dev(cat, torch.int32)
    sums = [_dev.empty(w) for w in widths]
    dw, dbo = _dev.empty(33), _dev.empty(1)
    grad = _dev.to_dev(np.full(o, 7.0, np.float32))
    losses = _dev.empty(4)
    inv = 1.0 / B
    j = _lib.SmallGrads()
    j.n_colsum = NL
    for l in range(NL):
        j.cs_A[l], j.cs_lda[l], j.cs_r0[l], j.cs_r1[l], j.cs_H[l], j.cs_out[l] = ptr(dDs[l]), widths[l], l % 3, 2 * B, widths[l], ptr(sums[l])
    j.w_A, j.w_lda, j.w_R, j.w_H, j.w_n0, j.w_n1 = ptr(dAL), 33, R, 33, B, B
    j.w_c0, j.w_c1, j.w_c2 = -inv, inv, 1.0
    j.w_out, j.w_bias_out = ptr(dw), ptr(dbo)
    j.emb_grad, j.emb = ptr(grad), e
    j.emb_D, j.emb_W, j.emb_cat = ptr(dD1), ptr(dW), ptr(catd)
    j.emb_ldd, j.emb_H, j.emb_ldw, j.emb_col0, j.emb_ld_cat, j.emb_B, j.emb_rows = H, H, H, col0, NC, B, 2 * B
    j.l_losses, j.l_out, j.l_slopes = ptr(losses), ptr(dout), ptr(dsl)
    j.l_n0, j.l_n1, j.l_ns = B, B, B
    j.l_inv_b, j.l_gp_weight, j.l_fake_sign = inv, 10.0, 1.0
    _lib.check(lib.agx_small_grads(_dev.stream(), j))
    for l in range(NL):
        np.testing.assert_allclose(sums[l].cpu().numpy(), Ds[l][l % 3:2 * B].astype(np.float64).sum(0), atol=2e-5)
    c = np.concatenate([np.full(B, -inv), np.full(B, inv), np.ones(B)])
    np.testing.assert_allclose(dw.cpu().numpy(), (c[:, None] * AL.astype(np.float64)).sum(0), atol=2e-5)
    g = grad.cpu().numpy()
    rows = np.arange(2 * B) % B
    c0 = col0
    touched = np.zeros(o, bool)
    for i, (v, dd, of) in enumerate(zip(vocab, edim, offs)):
        dcov = D1[:2 * B].astype(np.float64) @ W[c0:c0 + dd].astype(np.float64).T
        want = np.zeros((v, dd))
        np.add.at(want, cat[rows, i], dcov)
        np.testing.assert_allclose(g[of:of + v * dd].reshape(v, dd), want, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(want).max()))
        touched[of:of + v * dd] = True
        c0 += dd
    assert np.all(g[~touched] == 7.0)
    ls = losses.cpu().numpy()
    w = -out[:B].astype(np.float64).sum() * inv + out[B:2 * B].astype(np.float64).sum() * inv
    gp = ((slopes.astype(np.float64) - 1) ** 2).sum() * inv
    np.testing.assert_allclose(ls[:3], [w, gp, w + 10 * gp], rtol=1e-5, atol=1e-5)


def test_empty_bracket_overhead_is_small_and_positive(gpu):
    """agx_prof_empty_bracket_us: what the two event records of a bracket add to the launch they
    enclose (bench.py subtracts it from the bracketed duration of the dominant kernel)."""
    import ctypes
    _dev, _lib, lib = _libs()
    _dev.device()
    us = ctypes.c_double(-1.0)
    _lib.check(lib.agx_prof_empty_bracket_us(_dev.stream(), 64, ctypes.byref(us)))
    assert 0.0 <= us.value < 50.0
    assert lib.agx_prof_empty_bracket_us(_dev.stream(), 0, ctypes.byref(us)) != 0   # n out of range


def test_occupancy_query_of_the_tile_configurations(gpu):
    """agx_prof_occupancy: resident workgroups per CU of the main GEMM instantiations as the HIP runtime computes
    them.  Pins what the tile heuristics assume: 64x64 tiles at least 3 per CU (K-tiles of 16:
    16 KB LDS each; 144 registers set the limit), the 4-wave 128x128 kernels 2 per CU (capped at 256 registers for
    that), the 8-wave one 1."""
    import ctypes as C
    from adversarial_gene_expression_amd import _dev, _lib
    lib = _lib.load()
    _dev.device()
    got = {}
    for which in range(6):
        nb, lds = C.c_int32(), C.c_int32()
        _lib.check(lib.agx_prof_occupancy(which, C.byref(nb), C.byref(lds)), 'agx_prof_occupancy')
        got[which] = (nb.value, lds.value)
    assert all(got[w][0] >= 3 and got[w][1] == 16384 for w in (0, 1, 4)), got
    assert got[2][0] >= 1 and got[3][0] == 2 and got[5][0] == 2, got
    assert all(got[w][1] == 65536 for w in (2, 3, 5)), got
    nb, lds = C.c_int32(), C.c_int32()
    assert lib.agx_prof_occupancy(99, C.byref(nb), C.byref(lds)) != 0


def test_rccl_communicator_of_one_rank(gpu):
    """agx_comm_* (RCCL bound with dlopen) on the one GPU of the box: a communicator of one rank, whose
    collectives are the identity, queued on the compute stream between kernels of the library.  More ranks
    need more GPUs (RCCL refuses two ranks on one device): the multi-rank algebra is covered by the gloo tests."""
    import ctypes as C
    import torch
    from adversarial_gene_expression_amd import _dev, _lib
    lib = _lib.load()
    ident = (C.c_char * 128)()
    _lib.check(lib.agx_comm_unique_id(ident), 'agx_comm_unique_id')
    comm = C.c_void_p()
    _lib.check(lib.agx_comm_init(ident, 1, 0, C.byref(comm)), 'agx_comm_init')
    try:
        assert lib.agx_comm_size(comm) == 1 and lib.agx_comm_rank(comm) == 0
        st = _dev.stream()
        x = torch.randn(4099, device='cuda')
        ref = x.clone()
        _lib.check(lib.agx_comm_allreduce_sum(comm, st, x.data_ptr(), x.numel(), 0), 'agx_comm_allreduce_sum')
        d = torch.randn(6, device='cuda', dtype=torch.float64)
        dref = d.clone()
        _lib.check(lib.agx_comm_allreduce_sum(comm, st, d.data_ptr(), 6, 1), 'agx_comm_allreduce_sum')
        out = torch.zeros(4096, device='cuda')
        _lib.check(lib.agx_comm_reduce_scatter_sum(comm, st, x.data_ptr(), out.data_ptr(), 4096), 'agx_comm_reduce_scatter_sum')
        gathered = torch.zeros(4096, device='cuda')
        _lib.check(lib.agx_comm_all_gather(comm, st, out.data_ptr(), gathered.data_ptr(), 4096), 'agx_comm_all_gather')
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(d, dref)
        assert torch.equal(out, ref[:4096]) and torch.equal(gathered, ref[:4096])
        assert lib.agx_comm_allreduce_sum(None, st, x.data_ptr(), 4, 0) == -1
    finally:
        lib.agx_comm_destroy(comm)
