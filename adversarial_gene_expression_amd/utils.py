"""Drop-in mirror of the correlation / data helpers of the reference ``utils.py``.

Same names, argument meaning and error behaviour as the reference
(``/root/reference/utils.py:260-282, 808-873``); the arithmetic runs in HIP
kernels of libagx.so on an MI355X.  NumPy in -> NumPy out, like the reference.
Device tensors (torch, cuda) in -> ``DeviceArray`` out (no host round trip).

Precision: like NumPy, the arithmetic follows the input dtype.  float32 inputs take
the fp32 path (exact-fp32 MFMA, the throughput path); float64 inputs (the reference's
E. coli notebooks pass float64) take a float64 path end to end (fp64 matrix cores,
``agx_*_f64``): matrices agree with the reference to ~1e-12, the scalar Gamma to ~1e-10
(its five sums are raw fp64 moments of distances with mean ~1: three to four digits
of cancellation out of sixteen).  The float64 path is sized for the E. coli shapes
(N <= ~1000, G <= ~4300: a few ms; it computes the full G x G product from global
memory); a float64 caller at GTEx size gets correct results some 20x slower than the
fp32 path - cast to float32 for throughput.
"""
import numpy as np
import torch

from . import _dev, _lib
from ._dev import DeviceArray, ptr


def _is_dev(a):
    return isinstance(a, DeviceArray) or (isinstance(a, torch.Tensor) and a.is_cuda)


def _out(t, like_dev, dtype):
    if like_dev:
        return DeviceArray(t)
    a = t.cpu().numpy()
    return a.astype(dtype) if a.dtype != dtype else a


def _in_dtype(*arrs):
    for a in arrs:
        if isinstance(a, DeviceArray):
            a = a.tensor
        if (isinstance(a, np.ndarray) and a.dtype == np.float64) or (isinstance(a, torch.Tensor) and a.dtype == torch.float64):
            return np.float64
    return np.float32


# ---------------------------------------------------------------------------
# float64 building blocks (reference dtype semantics: float64 in -> float64 arithmetic)
# ---------------------------------------------------------------------------

def _mat64(a):
    t = _dev.to_dev(a, torch.float64)
    if t.dim() != 2:
        raise ValueError('expected a 2-D matrix, got shape %s' % (tuple(t.shape),))
    return t


def _colstats_dev64(xd):
    lib = _lib.load()
    n, g = xd.shape
    mean, std = _dev.empty(g, dtype=torch.float64), _dev.empty(g, dtype=torch.float64)
    nws = int(lib.agx_colstats_ws_doubles(n, g))
    ws = _dev.empty(nws, dtype=torch.float64)
    _lib.check(lib.agx_colstats_f64(_dev.stream(), ptr(xd), xd.stride(0), n, g, ptr(mean), ptr(std), ptr(ws), nws),
               'agx_colstats_f64')
    return mean, std


def _standardize_dev64(xd, mean, std):
    lib = _lib.load()
    n, g = xd.shape
    out = _dev.empty((n, g), dtype=torch.float64)
    _lib.check(lib.agx_standardize_f64(_dev.stream(), ptr(xd), xd.stride(0), n, g, ptr(mean), ptr(std), ptr(out), g),
               'agx_standardize_f64')
    return out


def _corr_dev64(x, y=None):
    """Pearson matrix of float64 columns (x with itself when y is None): device tensor [G1, G2] float64."""
    lib = _lib.load()
    xd = _mat64(x)
    xs = _standardize_dev64(xd, *_colstats_dev64(xd))
    if y is None:
        ys = xs
    else:
        yd = _mat64(y)
        ys = _standardize_dev64(yd, *_colstats_dev64(yd))
    n, g1, g2 = xs.shape[0], xs.shape[1], ys.shape[1]
    c = _dev.empty((g1, g2), dtype=torch.float64)
    _lib.check(lib.agx_corr_xty_f64(_dev.stream(), ptr(xs), g1, ptr(ys), g2, n, g1, g2, 1.0 / n, ptr(c), g2),
               'agx_corr_xty_f64')
    return c


def _r_from_sums64(s, n):
    """Pearson r from the five raw moment sums.  A (numerically) constant vector has no defined correlation: the
    reference divides 0 by 0 there and returns NaN (utils.py:816-824); raw moments would leave a rounding-level
    covariance over a zero variance (+-inf), so non-positive variances map to NaN explicitly."""
    with np.errstate(all='ignore'):
        ma, mb = s[0] / n, s[1] / n
        va, vb = s[2] / n - ma * ma, s[3] / n - mb * mb
        cov = s[4] / n - ma * mb
        if not (va > 0.0 and vb > 0.0):
            return np.float64(np.nan)
        return np.float64(cov / (np.sqrt(va) * np.sqrt(vb)))


def _gamma_coef64(x, y):
    """gamma_coef for float64 inputs (every rank computes the whole matrices: float64 callers are small)."""
    lib = _lib.load()
    cx, cy = _corr_dev64(x), _corr_dev64(y)
    g = cx.shape[0]
    sums = _dev.zeros(8, dtype=torch.float64)
    ws = _sums_ws()
    _lib.check(lib.agx_gamma_sums_f64(_dev.stream(), ptr(cx), g, ptr(cy), g, g, ptr(sums), ptr(ws)),
               'agx_gamma_sums_f64')
    s = sums.cpu().numpy()
    if s[5] == 0:
        return _r_from_sums64(s, g * (g - 1) // 2)
    lx, ly = _triu_pack_dev(cx, g, g), _triu_pack_dev(cy, g, g)
    assert lx.shape[0] == ly.shape[0]
    return np.float64(_pearson_1d_dev(lx, ly, one_minus=True))


# ---------------------------------------------------------------------------
# device-level building blocks
# ---------------------------------------------------------------------------

def _colstats_dev(xd, ld, n, g):
    lib = _lib.load()
    mean = _dev.empty(g)
    std = _dev.empty(g)
    nws = int(lib.agx_colstats_ws_doubles(n, g))
    ws = _dev.empty(nws, dtype=torch.float64)
    _lib.check(lib.agx_colstats(_dev.stream(), ptr(xd), ld, n, g, ptr(mean), ptr(std), ptr(ws), nws), 'agx_colstats')
    return mean, std


def _standardize_dev(xd, ld, n, g, mean, std):
    lib = _lib.load()
    out = _dev.empty((n, ld))
    if ld != g:
        out[:, g:].zero_()
    _lib.check(lib.agx_standardize(_dev.stream(), ptr(xd), ld, n, g, ptr(mean), ptr(std), ptr(out), ld),
               'agx_standardize')
    return out


def _std_columns_dev(x):
    xd, ld = _dev.padded_matrix(x)
    n, g = xd.shape[0], (x.shape[1])
    mean, std = _colstats_dev(xd, ld, n, g)
    return _standardize_dev(xd, ld, n, g, mean, std), ld, n, g


def _corr_self_dev(x, tile_range=None):
    """G x G Pearson matrix of the columns of x (device tensor [G, ldc])."""
    lib = _lib.load()
    xs, ld, n, g = _std_columns_dev(x)
    ldc = _dev.r4(g)
    c = _dev.empty((g, ldc))
    tb, te = tile_range if tile_range is not None else (0, -1)
    _lib.check(lib.agx_corr_xtx(_dev.stream(), ptr(xs), ld, n, g, 1.0 / n, ptr(c), ldc, tb, te), 'agx_corr_xtx')
    return c, ldc, g


def _corr_bipartite_dev(x, y):
    lib = _lib.load()
    xs, ldx, n, g1 = _std_columns_dev(x)
    ys, ldy, _, g2 = _std_columns_dev(y)
    ldc = _dev.r4(g2)
    c = _dev.empty((g1, ldc))
    ws_floats = int(lib.agx_gemm_ws_floats(g1, g2, n))
    ws = _dev.empty(max(ws_floats, 1))
    d = _lib.GemmDesc()
    d.A, d.B, d.C = ptr(xs), ptr(ys), ptr(c)
    d.M, d.N, d.K = g1, g2, n
    d.lda, d.ldb, d.ldc = ldx, ldy, ldc
    d.ta, d.tb = 1, 0
    d.alpha = 1.0 / n
    d.ws, d.ws_floats = ptr(ws), ws_floats
    d.force_cfg, d.force_splitk = -1, 0
    _lib.check(lib.agx_gemm_f32(_dev.stream(), d), 'agx_gemm_f32')
    return c, ldc, g1, g2


def _pearson_1d_dev(a, b, one_minus=False):
    """Scalar Pearson r of two equally long device vectors (five fp64 sums on
    device, closed form on host).  Follows utils.py:816-824 for 1-D inputs."""
    lib = _lib.load()
    n = a.numel()
    sums = _dev.zeros(8, dtype=torch.float64)
    ws = _sums_ws()      # bound to a local: the scratch must outlive the launches that use it on any stream
    if a.dtype == torch.float64:
        _lib.check(lib.agx_pearson_sums_f64(_dev.stream(), ptr(a), ptr(b), n, int(one_minus), ptr(sums), ptr(ws)),
                   'agx_pearson_sums_f64')
        return _r_from_sums64(sums.cpu().numpy(), n)
    _lib.check(lib.agx_pearson_sums(_dev.stream(), ptr(a), ptr(b), n, int(one_minus), ptr(sums), ptr(ws)),
               'agx_pearson_sums')
    return _r_from_sums(sums.cpu().numpy(), n)


_SUMS_WS = {}


def _sums_ws():
    """Scratch for the workgroups' partial sums of the deterministic five-sum reductions: ONE tensor per device,
    kept for the life of the process (a temporary would go back to the caching allocator before the kernels that
    write it have run, which is only safe while everything stays on one stream).  Calls that overlap on
    different streams of one device would share it: the eval helpers are single-stream."""
    dev = _dev.device()
    t = _SUMS_WS.get(dev.index)
    if t is None:
        t = _SUMS_WS[dev.index] = torch.empty(int(_lib.load().agx_sums_ws_doubles()), dtype=torch.float64, device=dev)
    return t


def _r_from_sums(s, n):
    with np.errstate(all='ignore'):
        ma, mb = s[0] / n, s[1] / n
        va, vb = s[2] / n - ma * ma, s[3] / n - mb * mb
        cov = s[4] / n - ma * mb
        if not (va > 0.0 and vb > 0.0):      # constant vector: the reference's 0 / 0 (utils.py:816-824)
            return np.float32(np.nan)
        return np.float32(cov / (np.sqrt(va) * np.sqrt(vb)))


def _triu_pack_dev(c, ldc, g):
    lib = _lib.load()
    f64 = c.dtype == torch.float64
    out = _dev.empty(max(g * (g - 1) // 2, 1), dtype=c.dtype)
    counts = _dev.empty(g + 1, dtype=torch.int64)
    offs = _dev.empty(g + 1, dtype=torch.int64)
    total = _dev.zeros(1, dtype=torch.int64)
    fn = lib.agx_triu_pack_f64 if f64 else lib.agx_triu_pack
    _lib.check(fn(_dev.stream(), ptr(c), ldc, g, ptr(out), ptr(counts), ptr(offs), ptr(total)), 'agx_triu_pack')
    return out[:int(total.item())]


# ---------------------------------------------------------------------------
# reference surface
# ---------------------------------------------------------------------------

def standardize(x, mean=None, std=None):
    """Shape x: (nb_samples, nb_vars).  Reference ``utils.py:260-268``."""
    dev_in = _is_dev(x)
    dt = _in_dtype(x)
    if dt == np.float64:
        xd = _mat64(x)
        g = xd.shape[1]
        if mean is None or std is None:
            m_, s_ = _colstats_dev64(xd)
        mean_d = m_ if mean is None else _dev.to_dev(np.broadcast_to(np.asarray(mean, np.float64), (g,)), torch.float64)
        std_d = s_ if std is None else _dev.to_dev(np.broadcast_to(np.asarray(std, np.float64), (g,)), torch.float64)
        return _out(_standardize_dev64(xd, mean_d, std_d), dev_in, dt)
    xd, ld = _dev.padded_matrix(x)
    n, g = xd.shape[0], (x.shape[1])
    if mean is None or std is None:
        m_, s_ = _colstats_dev(xd, ld, n, g)
    mean_d = m_ if mean is None else _dev.to_dev(np.broadcast_to(np.asarray(mean, np.float32), (g,)))
    std_d = s_ if std is None else _dev.to_dev(np.broadcast_to(np.asarray(std, np.float32), (g,)))
    out = _standardize_dev(xd, ld, n, g, mean_d, std_d)
    return _out(out[:, :g], dev_in, dt)


def split_train_test(x, train_rate=0.75):
    """Head/tail split (pure indexing, stays on the host).  Reference ``utils.py:271-282``."""
    nb_samples = x.shape[0]
    split_point = int(train_rate * nb_samples)
    return x[:split_point], x[split_point:]


def pearson_correlation(x, y):
    """Pearson matrix between the columns of x and of y (or scalar r for 1-D
    inputs).  Reference ``utils.py:808-824``; same AssertionError on a
    sample-count mismatch, same silent NaN/inf for zero-variance columns."""
    assert x.shape[0] == y.shape[0]
    dev_in = _is_dev(x)
    dt = _in_dtype(x, y)
    if len(x.shape) == 1:
        td = torch.float64 if dt == np.float64 else torch.float32
        return _pearson_1d_dev(_dev.to_dev(x, td).reshape(-1), _dev.to_dev(y, td).reshape(-1))
    if dt == np.float64:
        return _out(_corr_dev64(x, None if x is y else y), dev_in, dt)
    if x is y:
        c, ldc, g = _corr_self_dev(x)
        return _out(c[:, :g], dev_in, dt)
    c, ldc, g1, g2 = _corr_bipartite_dev(x, y)
    return _out(c[:, :g2], dev_in, dt)


def upper_diag_list(m_):
    """Condensed strict upper triangle, row-major, NaNs dropped.  Reference ``utils.py:837-849``."""
    dev_in = _is_dev(m_)
    if m_.shape[0] != m_.shape[1]:
        raise ValueError('upper_diag_list expects a square matrix')
    g = m_.shape[0]
    t = m_.tensor if isinstance(m_, DeviceArray) else m_
    if (isinstance(t, np.ndarray) and t.dtype == np.float64) or (isinstance(t, torch.Tensor) and t.dtype == torch.float64):
        md = _dev.to_dev(t, torch.float64)      # data movement only: stays float64, bit exact
        return _out(_triu_pack_dev(md, md.stride(0), g), dev_in, np.float64)
    md, ld = _dev.padded_matrix(m_)
    return _out(_triu_pack_dev(md, ld, g), dev_in, np.float32)


def correlations_list(x, y, corr_fn=pearson_correlation):
    """Reference ``utils.py:852-860``."""
    if corr_fn is pearson_correlation and x is y and _in_dtype(x) == np.float64:
        c = _corr_dev64(x)
        return _out(_triu_pack_dev(c, c.shape[1], c.shape[0]), _is_dev(x), np.float64)
    if corr_fn is pearson_correlation and x is y:
        dev_in = _is_dev(x)
        c, ldc, g = _corr_self_dev(x)
        return _out(_triu_pack_dev(c, ldc, g), dev_in, _in_dtype(x))
    return upper_diag_list(corr_fn(x, y))


def _rank_tile_range(g):
    """Contiguous share of the upper-triangular 128x128 tile list owned by this rank."""
    lib = _lib.load()
    world, rank = _dev.world()
    total = int(lib.agx_corr_num_tiles(g))
    per = (total + world - 1) // world
    return min(total, rank * per), min(total, (rank + 1) * per), world


def gamma_coef(x, y):
    """Gamma(D^X, D^Z) = Pearson r between the condensed distance lists
    ``1 - corr``.  Reference ``utils.py:863-873``.

    Both G x G matrices stay in HBM.  Without NaNs the five sums are taken
    straight over the two upper triangles (the 2 x G(G-1)/2 lists are never
    built); if any NaN is present the lists are packed with the reference's
    NaN-dropping and the reference's length assertion applies.

    Under torch.distributed (x, y replicated on every rank) the upper-triangular
    tile list of both matrices is dealt across the ranks: each GPU computes only
    its tiles of X^T X and Y^T Y and their partial sums; one all-reduce of six
    fp64 scalars combines them (no matrix is gathered)."""
    lib = _lib.load()
    dt = _in_dtype(x, y)
    if x.shape[1] != y.shape[1]:
        # lists of different length: reference fails at utils.py:821
        raise AssertionError
    if dt == np.float64:
        return _gamma_coef64(x, y)
    tb, te, world = _rank_tile_range(x.shape[1])
    sums = _dev.zeros(8, dtype=torch.float64)
    if world > 1:
        cx, ldx, g = _corr_self_dev(x, (tb, te))
        cy, ldy, _ = _corr_self_dev(y, (tb, te))
        _lib.check(lib.agx_gamma_sums_tiles(_dev.stream(), ptr(cx), ldx, ptr(cy), ldy, g, tb, te, ptr(sums),
                                            ptr(_sums_ws())), 'agx_gamma_sums_tiles')
        _dev.allreduce_sum(sums)
        s = sums.cpu().numpy()
        if s[5] == 0:
            r = _r_from_sums(s, g * (g - 1) // 2)
            return r if dt == np.float32 else np.float64(r)
        sums.zero_()        # NaNs present: every rank takes the packed (replicated) path below
    cx, ldx, g = _corr_self_dev(x)
    cy, ldy, g2 = _corr_self_dev(y)
    _lib.check(lib.agx_gamma_sums(_dev.stream(), ptr(cx), ldx, ptr(cy), ldy, g, ptr(sums), ptr(_sums_ws())),
               'agx_gamma_sums')
    s = sums.cpu().numpy()
    if s[5] == 0:
        r = _r_from_sums(s, g * (g - 1) // 2)
    else:
        lx = _triu_pack_dev(cx, ldx, g)
        ly = _triu_pack_dev(cy, ldy, g)
        assert lx.shape[0] == ly.shape[0]
        r = _pearson_1d_dev(lx, ly, one_minus=True)
    return r if dt == np.float32 else np.float64(r)


class GammaScorer:
    """``gamma_coef(x_test, .)`` with the fixed side cached in HBM.

    The reference's ``score_fn`` closure (gtex_tcga_gan.py:97-108, ecoli_gan.py:62-74) calls
    ``gamma_coef(x_test, x_gen)`` every 5 epochs and recomputes the correlation matrix of the
    never-changing ``x_test`` each time (utils.py:870).  Here that matrix (this rank's tiles of it
    under torch.distributed) is computed once; every call then costs one standardise + X^T X of the
    generated matrix and one pass of fused sums.  Accepts host arrays or device tensors
    (``predict(..., training=True)``), so the generated samples never leave the GPU.  Results are
    those of ``gamma_coef`` (same kernels, same summation order)."""

    def __init__(self, x_test):
        self.dtype = _in_dtype(x_test)
        self.g = x_test.shape[1]
        if self.dtype == np.float64:      # float64 test set: float64 arithmetic, matrix of the fixed side cached
            self.world, self._x, self.cx = 1, None, _corr_dev64(x_test)
            return
        self.tb, self.te, self.world = _rank_tile_range(self.g)
        self._x = x_test if self.world > 1 else None     # NaN fallback of the sharded path needs the input again
        rng = (self.tb, self.te) if self.world > 1 else None
        self.cx, self.ldx, _ = _corr_self_dev(x_test, rng)
        self._full = self.cx if self.world == 1 else None

    def __call__(self, x_gen):
        lib = _lib.load()
        if x_gen.shape[1] != self.g:
            raise AssertionError
        sums = _dev.zeros(8, dtype=torch.float64)
        g = self.g
        if self.dtype == np.float64:
            cy = _corr_dev64(_dev.to_dev(x_gen, torch.float64))
            _lib.check(lib.agx_gamma_sums_f64(_dev.stream(), ptr(self.cx), g, ptr(cy), g, g, ptr(sums), ptr(_sums_ws())),
                       'agx_gamma_sums_f64')
            s = sums.cpu().numpy()
            if s[5] == 0:
                return _r_from_sums64(s, g * (g - 1) // 2)
            lx, ly = _triu_pack_dev(self.cx, g, g), _triu_pack_dev(cy, g, g)
            assert lx.shape[0] == ly.shape[0]
            return np.float64(_pearson_1d_dev(lx, ly, one_minus=True))
        if self.world > 1:
            cy, ldy, _ = _corr_self_dev(x_gen, (self.tb, self.te))
            _lib.check(lib.agx_gamma_sums_tiles(_dev.stream(), ptr(self.cx), self.ldx, ptr(cy), ldy, g, self.tb, self.te,
                                                ptr(sums), ptr(_sums_ws())), 'agx_gamma_sums_tiles')
            _dev.allreduce_sum(sums)
            s = sums.cpu().numpy()
            if s[5] != 0:
                return gamma_coef(self._x, x_gen)
        else:
            cy, ldy, _ = _corr_self_dev(x_gen)
            _lib.check(lib.agx_gamma_sums(_dev.stream(), ptr(self.cx), self.ldx, ptr(cy), ldy, g, ptr(sums),
                                          ptr(_sums_ws())), 'agx_gamma_sums')
            s = sums.cpu().numpy()
            if s[5] != 0:
                lx = _triu_pack_dev(self.cx, self.ldx, g)
                ly = _triu_pack_dev(cy, ldy, g)
                assert lx.shape[0] == ly.shape[0]
                r = _pearson_1d_dev(lx, ly, one_minus=True)
                return r if self.dtype == np.float32 else np.float64(r)
        r = _r_from_sums(s, g * (g - 1) // 2)
        return r if self.dtype == np.float32 else np.float64(r)


# ---------------------------------------------------------------------------
# extended evaluation built on the correlation kernels (reference utils.py:827-834, 876-1105)
#
# The G x G correlation matrices and their condensed distance lists come from the MI355X; the
# hierarchical clustering itself stays SciPy's (`linkage('complete')`, `cophenet`), exactly as in the
# reference - it is sequential host work on a list the GPU has already produced.
# ---------------------------------------------------------------------------

def cosine_similarity(x, y):
    """Reference ``utils.py:827-834`` (two short host vectors)."""
    return np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y))


def _host(a):
    return a.numpy() if isinstance(a, DeviceArray) else (a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a))


def hierarchical_clustering(data, corr_fun=pearson_correlation):
    """SciPy complete-linkage matrix of the genes on ``1 - corr``.  Reference ``utils.py:1019-1032``."""
    from scipy.cluster.hierarchy import linkage
    y = 1 - _host(correlations_list(data, data, corr_fun))
    return linkage(y, 'complete')


def dendrogram_distance(l_matrix, condensed=True):
    """Dendrogrammatic distance between every pair of genes (height of the merge that first joins
    them).  The reference fills a G x G matrix with O(G^2) Python loops (``utils.py:1062-1092``); those
    values are exactly SciPy's cophenetic distances, taken here in one call (same numbers, same
    condensed order)."""
    from scipy.cluster.hierarchy import cophenet
    from scipy.spatial.distance import squareform
    d = cophenet(np.asarray(l_matrix, np.float64))
    return d if condensed else squareform(d)


def compare_cophenetic(l_matrix1, l_matrix2):
    """Pearson r between two dendrograms' distance lists.  Reference ``utils.py:1095-1105``."""
    return pearson_correlation(dendrogram_distance(l_matrix1), dendrogram_distance(l_matrix2))


def gamma_coefficients(expr_x, expr_z):
    """Gamma(D^X,D^Z), Gamma(D^X,T^X), Gamma(D^Z,T^Z), Gamma(T^X,T^Z).  Reference ``utils.py:876-903``.
    Each expression matrix is correlated ONCE on the GPU (the reference computes each Pearson matrix
    twice: for the distance list and again inside ``hierarchical_clustering``)."""
    from scipy.cluster.hierarchy import cophenet, linkage
    dists_x = 1 - _host(correlations_list(expr_x, expr_x))
    dists_z = 1 - _host(correlations_list(expr_z, expr_z))
    gamma_dx_dz = pearson_correlation(dists_x, dists_z)
    xl_matrix = linkage(dists_x, 'complete')
    gamma_dx_tx, _ = cophenet(xl_matrix, dists_x)
    zl_matrix = linkage(dists_z, 'complete')
    gamma_dz_tz, _ = cophenet(zl_matrix, dists_z)
    return gamma_dx_dz, gamma_dx_tx, gamma_dz_tz, compare_cophenetic(xl_matrix, zl_matrix)


def compute_tf_tg_corrs(expr, gene_symbols, tf_tg=None, flat=True):
    """Lists of TF-TG and TG-TG correlations per transcription factor.  Reference ``utils.py:906-941``.
    All of them are entries of ONE G x G Pearson matrix, computed once on the GPU and indexed on the
    host (the reference runs one small correlation per TF).  ``tf_tg``: dict TF symbol -> list of target
    symbols; the reference's RegulonDB loader (``tf_tg_interactions``) is outside this package."""
    if tf_tg is None:
        raise ValueError('pass tf_tg (dict TF -> target genes); the RegulonDB file loader is not part of this package')
    gene_symbols = np.array(gene_symbols)
    corr = _host(pearson_correlation(expr, expr))
    tf_tg_corr, tg_tg_corr = [], []
    for tf, tgs in tf_tg.items():
        tg_idxs = np.array([np.where(gene_symbols == tg)[0] for tg in tgs if tg in gene_symbols]).ravel()
        if tf in gene_symbols and len(tg_idxs) > 0:
            sub = corr[np.ix_(tg_idxs, tg_idxs)]
            v = sub[np.triu_indices(len(tg_idxs), k=1)]
            tg_tg_corr += [v[~np.isnan(v)].tolist()]          # upper_diag_list semantics (utils.py:844-849)
            tf_idx = np.argwhere(gene_symbols == tf)[0]
            tf_tg_corr += [corr[tf_idx[0], tg_idxs].ravel().tolist()]
    if flat:
        tf_tg_corr = [c for corr_l in tf_tg_corr for c in corr_l]
        tg_tg_corr = [c for corr_l in tg_tg_corr for c in corr_l]
    return tf_tg_corr, tg_tg_corr


def psi_coefficient(tf_tg_x, tf_tg_z, weights_type='nb_genes'):
    """Weighted mean cosine similarity of the TF-TG correlation vectors.  Reference ``utils.py:944-963``."""
    weights_sum = 0
    total_sum = 0
    for cx, cz in zip(tf_tg_x, tf_tg_z):
        weight = len(cx) if weights_type == 'nb_genes' else 1
        weights_sum += weight
        total_sum += weight * cosine_similarity(np.array(cx), np.array(cz))
    return total_sum / weights_sum


def phi_coefficient(tg_tg_x, tg_tg_z, weights_type='nb_genes'):
    """Weighted mean cosine similarity of the TG-TG correlation vectors.  Reference ``utils.py:966-988``."""
    weights_sum = 0
    total_sum = 0
    for cx, cz in zip(tg_tg_x, tg_tg_z):
        if len(cx) > 0:  # a TF that regulates one gene has an empty TG-TG list
            weight = 1
            if weights_type == 'nb_genes':
                weight = max(np.roots([1, 1, -2 * len(cx)]))  # number of regulated genes from the list length
            weights_sum += weight
            total_sum += weight * cosine_similarity(np.array(cx), np.array(cz))
    return total_sum / weights_sum


def compute_scores(expr_x, expr_z, gene_symbols, tf_tg=None):
    """[S_dist, S_dend, S_sdcc, S_tftg, S_tgtg].  Reference ``utils.py:991-1012`` (``tf_tg`` as in
    ``compute_tf_tg_corrs``)."""
    gamma_dx_dz, gamma_dx_tx, gamma_dz_tz, gamma_tx_tz = gamma_coefficients(expr_x, expr_z)
    r_tf_tg_corr, r_tg_tg_corr = compute_tf_tg_corrs(expr_x, gene_symbols, tf_tg, flat=False)
    s_tf_tg_corr, s_tg_tg_corr = compute_tf_tg_corrs(expr_z, gene_symbols, tf_tg, flat=False)
    return [gamma_dx_dz, gamma_tx_tz, (gamma_dx_tx - gamma_dz_tz) ** 2,
            psi_coefficient(r_tf_tg_corr, s_tf_tg_corr), phi_coefficient(r_tg_tg_corr, s_tg_tg_corr)]
