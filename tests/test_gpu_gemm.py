"""fp32-MFMA GEMM family (csrc/gemm.hip) against NumPy fp64 through the C-ABI.
Tolerance: fp32 fma-chain accumulation => |err| <= 2e-6 * sum|a*b| (stated bound, K <= 4096)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev_mat(a, pad_ld=True):
    from adversarial_gene_expression_amd import _dev
    t, _ = _dev.padded_matrix(a) if pad_ld else (_dev.to_dev(a), None)
    return t


def _ref(A, B, ta, tb):
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    return (A64.T if ta else A64) @ (B64.T if tb else B64)


def _bound(A, B, ta, tb):
    return (np.abs(A.T if ta else A).astype(np.float64) @ np.abs(B.T if tb else B).astype(np.float64))


SHAPES = [(1, 1, 1), (5, 37, 16), (32, 256, 71), (33, 130, 65), (96, 64, 300), (12, 257, 129),
          (128, 128, 32), (129, 127, 33), (200, 70, 1030)]


@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize('cfg', [-1, 0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize('shape', SHAPES)
def test_gemm_layouts_configs(gpu, layout, cfg, shape):
    from adversarial_gene_expression_amd import ops
    ta, tb = layout
    M, N, K = shape
    rs = np.random.RandomState(M * 7 + N * 3 + K + ta * 2 + tb)
    A = rs.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    B = rs.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    out = ops.gemm(_dev_mat(A), _dev_mat(B), M, N, K, ta=ta, tb=tb, force_cfg=cfg)
    got = out.cpu().numpy()[:, :N]
    err = np.abs(got - _ref(A, B, ta, tb))
    assert np.all(err <= 2e-6 * _bound(A, B, ta, tb) + 1e-30)


@pytest.mark.parametrize('cfg', [-1, 1, 3])
@pytest.mark.parametrize('splitk', [2, 5, 64])
@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 0)])
def test_gemm_splitk(gpu, layout, splitk, cfg):
    from adversarial_gene_expression_amd import ops
    ta, tb = layout
    M, N, K = 45, 70, 1111
    rs = np.random.RandomState(splitk)
    A = rs.randn(*((K, M) if ta else (M, K))).astype(np.float32)
    B = rs.randn(*((N, K) if tb else (K, N))).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    from adversarial_gene_expression_amd import _dev
    out = ops.gemm(_dev_mat(A), _dev_mat(B), M, N, K, ta=ta, tb=tb, force_splitk=splitk, force_cfg=cfg, bias=_dev.to_dev(bias), act=1)
    ref = np.maximum(_ref(A, B, ta, tb) + bias, 0)
    err = np.abs(out.cpu().numpy()[:, :N] - ref)
    assert np.all(err <= 2e-6 * (_bound(A, B, ta, tb) + np.abs(bias)) + 1e-30)


def test_gemm_unaligned_operands_take_scalar_path(gpu):
    """ld not a multiple of 4 and a base pointer off 16 B: the non-vector loader must be used."""
    from adversarial_gene_expression_amd import ops, _dev
    rs = np.random.RandomState(1)
    M, N, K = 37, 18, 53
    A = rs.randn(M, K).astype(np.float32)
    B = rs.randn(K, N).astype(np.float32)
    Ad = _dev.to_dev(A)                         # ld = 53
    Bbuf = _dev.zeros(K * N + 1)
    Bd = Bbuf[1:].view(K, N)                    # base + 4 B
    Bd.copy_(torch.from_numpy(B))
    out = ops.gemm(Ad, Bd, M, N, K)
    np.testing.assert_allclose(out.cpu().numpy()[:, :N], A.astype(np.float64) @ B, rtol=0, atol=2e-5)


def test_gemm_epilogues(gpu):
    from adversarial_gene_expression_amd import ops, _dev
    rs = np.random.RandomState(5)
    M, N, K = 36, 50, 40
    A, B = rs.randn(M, K).astype(np.float32), rs.randn(K, N).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    add = rs.randn(12, N).astype(np.float32)
    mask = rs.randn(M, N).astype(np.float32)
    base = A.astype(np.float64) @ B
    Ad, Bd = _dev_mat(A), _dev_mat(B)
    # bias + leaky relu
    got = ops.gemm(Ad, Bd, M, N, K, bias=_dev.to_dev(bias), act=1, slope=0.2).cpu().numpy()[:, :N]
    v = base + bias
    np.testing.assert_allclose(got, np.where(v > 0, v, 0.2 * v), atol=3e-5)
    # alpha + addrow (row modulo)
    got = ops.gemm(Ad, Bd, M, N, K, alpha=0.5, addrow=_dev_mat(add)).cpu().numpy()[:, :N]
    np.testing.assert_allclose(got, 0.5 * base + add[np.arange(M) % 12], atol=3e-5)
    # separate mask (ReLU grad) and in-place mask
    got = ops.gemm(Ad, Bd, M, N, K, mask=_dev_mat(mask)).cpu().numpy()[:, :N]
    np.testing.assert_allclose(got, np.where(mask > 0, base, 0.0), atol=3e-5)
    C = _dev_mat(mask).clone()
    ops.gemm(Ad, Bd, M, N, K, mask=C, out=C, slope=0.1)
    np.testing.assert_allclose(C.cpu().numpy()[:, :N], np.where(mask > 0, base, 0.1 * base), atol=3e-5)
    # accumulate
    C = _dev_mat(mask).clone()
    ops.gemm(Ad, Bd, M, N, K, out=C, accumulate=True)
    np.testing.assert_allclose(C.cpu().numpy()[:, :N], mask + base, atol=3e-5)


def test_gemm_identity_with_asymmetric_b(gpu):
    """A = I with an asymmetric B catches a swapped row/col map in the C write."""
    from adversarial_gene_expression_amd import ops
    n = 96
    B = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)
    for cfg in (0, 1, 2, 3):
        out = ops.gemm(_dev_mat(np.eye(n, dtype=np.float32)), _dev_mat(B), n, n, n, force_cfg=cfg)
        np.testing.assert_array_equal(out.cpu().numpy()[:, :n], B)


def test_gemm_rejects_bad_arguments(gpu):
    from adversarial_gene_expression_amd import ops, _dev, AgxError
    a = _dev.zeros((8, 8))
    with pytest.raises(AgxError):
        ops.gemm(a, a, 8, 8, 8, ta=1, tb=1)
    with pytest.raises(AgxError):
        ops.gemm(a, a, 8, 16, 8)  # ldb < N


def test_gemm_gtex_shapes_bitwise_reproducible(gpu):
    """Full-size layer shapes (B=32 and 512): linearity in A and run-to-run bitwise equality."""
    from adversarial_gene_expression_amd import ops, _dev
    G, H = 18154, 256
    torch.manual_seed(0)
    W = torch.randn(G + 7, H, device=gpu)
    for B in (32, 512):
        X = torch.randn(3 * B, _dev.r4(G + 7), device=gpu)
        y1 = ops.gemm(X, W, 3 * B, H, G + 7)
        y2 = ops.gemm(X, W, 3 * B, H, G + 7)
        assert torch.equal(y1, y2)
        y3 = ops.gemm(2.0 * X, W, 3 * B, H, G + 7)
        assert torch.equal(y3, 2.0 * y1)   # scaling by 2 is exact in fp32
        ref = (X[:, :G + 7].double() @ W.double())
        assert float((y1[:, :H].double() - ref).abs().max()) < 5e-3  # K=18161, |sum| ~ 135


@pytest.mark.parametrize('cfg', [2, 3])
@pytest.mark.parametrize('M,N,K', [(5, 37, 16), (32, 1076, 128), (33, 4297, 70)])
def test_gemm_rowscale_and_row_sum_of_squares(gpu, M, N, K, cfg):
    """Side outputs used by the fused gradient-penalty path: per-row scaling and the deterministic
    per-32-column-slab partial sums of squares of the stored result."""
    from adversarial_gene_expression_amd import ops, _dev
    rs = np.random.RandomState(N)
    A, B = rs.randn(M, K).astype(np.float32), rs.randn(N, K).astype(np.float32)
    scale = rs.randn(M).astype(np.float32)
    nslab = (N + 31) // 32
    sq = _dev.zeros((M, nslab + 2))
    sd = _dev.to_dev(scale)
    out = ops.gemm(_dev_mat(A), _dev_mat(B), M, N, K, tb=1, rowscale=sd, rowsq=sq, force_cfg=cfg)
    ref = scale[:, None].astype(np.float64) * (A.astype(np.float64) @ B.T)
    got = out.cpu().numpy()[:, :N]
    np.testing.assert_allclose(got, ref, atol=3e-5)
    want_sq = np.stack([np.sum(got[:, 32 * s:32 * s + 32].astype(np.float64) ** 2, axis=1) for s in range(nslab)], 1)
    np.testing.assert_allclose(sq.cpu().numpy()[:, :nslab], want_sq, rtol=2e-6, atol=1e-12)
    sq2 = _dev.zeros((M, nslab + 2))
    ops.gemm(_dev_mat(A), _dev_mat(B), M, N, K, tb=1, rowscale=sd, rowsq=sq2, force_cfg=cfg)
    assert torch.equal(sq, sq2)   # fixed summation order


@pytest.mark.parametrize('cfg,sk', [(-1, 0), (3, 1), (3, 7), (1, 3), (0, 1)])
@pytest.mark.parametrize('R,N,K', [(5, 37, 16), (32, 256, 1100), (33, 130, 65)])
def test_gemm_interpolated_rows_by_linearity(gpu, R, N, K, cfg, sk):
    """A = [real; fake] (2R rows); C gets 3R rows: the third block is the layer applied to
    real + alpha*(fake - real) (tf_utils.py:162-167), computed as r + alpha*(f - r) of the products."""
    from adversarial_gene_expression_amd import ops, _dev
    rs = np.random.RandomState(R + K)
    A = rs.randn(2 * R, K).astype(np.float32)
    B = (rs.randn(K, N) / np.sqrt(K)).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    alpha = rs.rand(R).astype(np.float32)
    out = _dev.zeros((3 * R, _dev.r4(N)))
    ops.gemm(_dev_mat(A), _dev_mat(B), 2 * R, N, K, bias=_dev.to_dev(bias), act=1, slope=0.0, out=out,
             interp=(R, _dev.to_dev(alpha)), force_cfg=cfg, force_splitk=sk)
    inter = A[:R] + alpha[:, None] * (A[R:] - A[:R])
    full = np.concatenate([A, inter]).astype(np.float64) @ B.astype(np.float64) + bias
    np.testing.assert_allclose(out.cpu().numpy()[:, :N], np.maximum(full, 0), atol=2e-5)


@pytest.mark.parametrize('shape', [(18161, 256, 1536), (18161, 256, 1530), (18154, 256, 1030), (18161, 512, 1536)])
def test_stream_k_weight_gradient_product(gpu, shape):
    """Stream-K form of the TN weight-gradient product (gemm.hip SK: the (tile, K-tile) space dealt evenly over
    2 x CUs workgroups, partial tiles handed over inside the launch): the 512-row dW_1 (284 tiles of 128x128 on 256
    CUs), the same with a ragged K tail, the generator's output-layer gradient, a 512-wide first layer.  Against
    NumPy float64 with the family's fma-chain bound; bit-identical on a second launch (fixed summation order);
    every hand-off word back at zero afterwards and the spin-timeout word never set."""
    from adversarial_gene_expression_amd import _dev, _lib, ops
    lib = _lib.load()
    M, N, K = shape
    rs = np.random.RandomState(K)
    A = rs.randn(K, M).astype(np.float32)
    B = rs.randn(K, N).astype(np.float32)
    nflag = int(lib.agx_gemm_sk_flag_words())
    flags = torch.zeros(nflag, dtype=torch.int32, device='cuda')
    Ad, Bd = _dev_mat(A), _dev_mat(B)
    out1 = ops.gemm(Ad, Bd, M, N, K, ta=1, sk_flags=flags).clone()
    torch.cuda.synchronize()
    assert int(flags.abs().sum()) == 0
    plain = ops.gemm(Ad, Bd, M, N, K, ta=1)                 # the static tiling, one k chain per element
    out2 = ops.gemm(Ad, Bd, M, N, K, ta=1, sk_flags=flags)
    torch.cuda.synchronize()
    assert int(flags.abs().sum()) == 0
    assert torch.equal(out1[:, :N], out2[:, :N])
    assert not torch.equal(out1[:, :N], plain[:, :N])       # the stream-K kernel really ran (another summation order)
    got = out1.cpu().numpy()[:, :N]
    ref, bound = _ref(A, B, 1, 0), _bound(A, B, 1, 0)
    assert np.all(np.abs(got - ref) <= 2e-6 * bound + 1e-30)
    assert np.all(np.abs(plain.cpu().numpy()[:, :N] - ref) <= 2e-6 * bound + 1e-30)
