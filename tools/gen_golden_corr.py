"""Generate golden vectors for the correlation path from the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  Imports the
reference ``utils.py`` with its two absent plotting-only dependencies
(``seaborn``, ``statsmodels.stats.multitest``; used at utils.py:5,10 by the
plotting helpers only) replaced by empty stub modules, calls the reference
functions on seeded inputs and stores inputs + outputs as small ``.npz``
files under ``tests/golden/``.  Only the data ships; no reference source.

    python tools/gen_golden_corr.py
"""
import os
import sys
import types
import warnings

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def import_reference_utils():
    for m in ['seaborn', 'statsmodels', 'statsmodels.stats', 'statsmodels.stats.multitest']:
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules['statsmodels.stats.multitest'].multipletests = None
    sys.path.insert(0, REF)
    import utils  # noqa
    return utils


def lowrank(rs, n, g, k=4, noise=0.5, dtype=np.float32):
    f = rs.randn(n, k)
    l = rs.randn(k, g)
    return (f @ l + noise * rs.randn(n, g)).astype(dtype)


def main():
    ru = import_reference_utils()
    os.makedirs(OUT, exist_ok=True)
    warnings.simplefilter('ignore')
    cases = [
        # name, N, G, dtype, constant column?
        ('n50_g7_f32', 50, 7, np.float32, False),
        ('n64_g48_f32', 64, 48, np.float32, False),
        ('n33_g17_const_f32', 33, 17, np.float32, True),
        ('n64_g48_f64', 64, 48, np.float64, False),
        ('n12_g130_f32', 12, 130, np.float32, False),   # N < G, ragged vs 32/64/128 tiles
        ('n257_g65_f32', 257, 65, np.float32, False),   # odd N, G one past a tile edge
    ]
    for i, (name, n, g, dt, const) in enumerate(cases):
        rs = np.random.RandomState(100 + i)
        x = lowrank(rs, n, g, dtype=dt)
        y = lowrank(rs, n, g, dtype=dt)          # same N (bipartite pearson)
        y2 = lowrank(rs, max(5, n // 2 + 3), g, dtype=dt)  # different N (gamma_coef)
        if const:
            x[:, 3] = 1.25                      # zero-variance gene -> NaN row/col
        out = {'x': x, 'y': y, 'y2': y2}
        out['pearson_xx'] = ru.pearson_correlation(x, x)
        out['pearson_xy'] = ru.pearson_correlation(x, y)
        out['upper_xx'] = ru.upper_diag_list(out['pearson_xx'])
        out['corr_list_xx'] = ru.correlations_list(x, x)
        out['standardize_x'] = ru.standardize(x)
        mu, sd = np.mean(y, axis=0), np.std(y, axis=0)
        out['standardize_x_with_y_stats'] = ru.standardize(x, mean=mu, std=sd)
        tr, te = ru.split_train_test(x)
        out['split_train'], out['split_test'] = tr, te
        if not const:
            out['gamma_x_y2'] = np.asarray(ru.gamma_coef(x, y2))
            out['gamma_x_x'] = np.asarray(ru.gamma_coef(x, x))
        else:
            # NaN-dropping keeps both lists the same length only for x vs x
            out['gamma_x_x'] = np.asarray(ru.gamma_coef(x, x))
        # 1-D pearson (the final step of gamma_coef)
        a = rs.randn(21).astype(dt)
        b = (0.7 * a + 0.3 * rs.randn(21)).astype(dt)
        out['a1d'], out['b1d'] = a, b
        out['pearson_1d'] = np.asarray(ru.pearson_correlation(a, b))
        np.savez_compressed(os.path.join(OUT, 'corr_%s.npz' % name), **out)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith(('pearson', 'gamma', 'upper'))})

    # upper_diag_list on a matrix with genuine NaNs above the diagonal (utils.py:844-849)
    rs = np.random.RandomState(7)
    m = rs.randn(7, 7).astype(np.float32)
    m[0, 3] = np.nan
    m[2, 5] = np.nan
    m[4, 1] = np.nan  # below the diagonal: irrelevant
    np.savez_compressed(os.path.join(OUT, 'corr_upper_nan.npz'), m=m, upper=ru.upper_diag_list(m))


if __name__ == '__main__':
    main()
