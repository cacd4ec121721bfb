"""NumPy oracle for the correlation / gamma-coefficient evaluation path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Each function is a
restatement of the reference algorithm with the reference file:line it
follows.  Parity PINNED by ``tests/golden/corr_*.npz`` (outputs of the
unmodified reference ``utils.py`` imported in the build container).
"""
import numpy as np


def standardize(x, mean=None, std=None):
    """Per-column z-score; population std (ddof=0); caller may inject the
    train-split statistics.  Follows reference ``utils.py:260-268``."""
    if mean is None:
        mean = np.mean(x, axis=0)
    if std is None:
        std = np.std(x, axis=0)
    return (x - mean) / std


def split_train_test(x, train_rate=0.75):
    """Head/tail split at ``int(train_rate * N)``.  Reference ``utils.py:271-282``."""
    split_point = int(train_rate * x.shape[0])
    return x[:split_point], x[split_point:]


def pearson_correlation(x, y):
    """Bipartite Pearson matrix ``x_.T @ y_ / N`` of column-standardised
    inputs; also valid for 1-D inputs (scalar r).  Zero-variance columns give
    NaN/inf silently.  Reference ``utils.py:808-824``."""
    assert x.shape[0] == y.shape[0]
    x_ = (x - np.mean(x, axis=0)) / np.std(x, axis=0)
    y_ = (y - np.mean(y, axis=0)) / np.std(y, axis=0)
    return np.dot(x_.T, y_) / x.shape[0]


def upper_diag_list(m_):
    """Strict upper triangle in row-major order with NaN entries dropped
    (the reference masks the lower triangle with NaN and then filters on
    ``~isnan``, so genuine NaNs above the diagonal vanish as well).
    Reference ``utils.py:837-849``."""
    m_ = np.asarray(m_)
    iu = np.triu_indices(m_.shape[0], k=1, m=m_.shape[1])
    v = m_[iu]
    return v[~np.isnan(v)]


def correlations_list(x, y, corr_fn=pearson_correlation):
    """``upper_diag_list(corr_fn(x, y))``.  Reference ``utils.py:852-860``."""
    return upper_diag_list(corr_fn(x, y))


def gamma_coef(x, y):
    """Gamma(D^X, D^Z): Pearson r between the two condensed distance lists
    ``1 - corr``.  Reference ``utils.py:863-873``."""
    dists_x = 1 - correlations_list(x, x)
    dists_y = 1 - correlations_list(y, y)
    return pearson_correlation(dists_x, dists_y)


def gamma_coef_f64(x, y):
    """fp64 tie-breaker for tolerance budgeting (not a reference function)."""
    return gamma_coef(np.asarray(x, np.float64), np.asarray(y, np.float64))
