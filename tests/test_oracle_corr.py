"""The NumPy oracle of the correlation path against golden vectors produced by the
UNMODIFIED reference utils.py (tools/gen_golden_corr.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from oracle import corr_oracle as CO

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'corr_n*.npz')))


def test_golden_files_present():
    assert len(CASES) >= 6


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_goldens(name):
    g = golden(name)
    x, y, y2 = g['x'], g['y'], g['y2']
    with np.errstate(all='ignore'):
        # same NumPy calls in the same order -> bit-exact (NaN positions included)
        np.testing.assert_array_equal(CO.pearson_correlation(x, x), g['pearson_xx'])
        np.testing.assert_array_equal(CO.pearson_correlation(x, y), g['pearson_xy'])
        np.testing.assert_array_equal(CO.upper_diag_list(g['pearson_xx']), g['upper_xx'])
        np.testing.assert_array_equal(CO.correlations_list(x, x), g['corr_list_xx'])
        np.testing.assert_array_equal(CO.standardize(x), g['standardize_x'])
        np.testing.assert_array_equal(CO.standardize(x, mean=np.mean(y, axis=0), std=np.std(y, axis=0)),
                                      g['standardize_x_with_y_stats'])
        tr, te = CO.split_train_test(x)
        np.testing.assert_array_equal(tr, g['split_train'])
        np.testing.assert_array_equal(te, g['split_test'])
        np.testing.assert_array_equal(CO.pearson_correlation(g['a1d'], g['b1d']), g['pearson_1d'])
        np.testing.assert_array_equal(CO.gamma_coef(x, x), g['gamma_x_x'])
        if 'gamma_x_y2' in g.files:
            np.testing.assert_array_equal(CO.gamma_coef(x, y2), g['gamma_x_y2'])
    assert CO.pearson_correlation(x, x).dtype == x.dtype


def test_upper_diag_drops_genuine_nans():
    g = golden('corr_upper_nan.npz')
    got = CO.upper_diag_list(g['m'])
    np.testing.assert_array_equal(got, g['upper'])
    assert got.shape[0] == 21 - 2


def test_sample_count_mismatch_asserts():
    with pytest.raises(AssertionError):
        CO.pearson_correlation(np.zeros((4, 3), np.float32), np.zeros((5, 3), np.float32))


def test_properties():
    rs = np.random.RandomState(3)
    x = rs.randn(40, 9).astype(np.float32)
    c = CO.pearson_correlation(x, x)
    np.testing.assert_allclose(c, c.T, atol=1e-6)
    np.testing.assert_allclose(np.diag(c), 1.0, atol=1e-5)
    assert np.all(np.abs(c) <= 1 + 1e-5)
    # invariant to per-gene positive affine maps
    c2 = CO.pearson_correlation(x * 3.0 + 2.0, x * 3.0 + 2.0)
    np.testing.assert_allclose(c, c2, atol=1e-5)
    np.testing.assert_allclose(c, np.corrcoef(x.T), atol=1e-5)
