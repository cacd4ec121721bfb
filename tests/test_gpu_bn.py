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
