"""Oracle of the extended evaluation scores (oracle/eval_oracle.py) against goldens produced by the
unmodified reference utils.py (tools/gen_golden_eval.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import eval_oracle as EO

CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, 'tests', 'golden', 'eval_*.npz')))


def load(name):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name))
    return g, json.loads(str(g['tf_tg_json']))


def test_goldens_present():
    assert len(CASES) >= 2


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_goldens(name):
    g, tf_tg = load(name)
    x, z, sym = g['x'], g['z'], g['symbols']
    tol = dict(rtol=1e-4, atol=1e-5) if x.dtype == np.float32 else dict(rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(np.array(EO.gamma_coefficients(x, z), np.float64), g['gamma_coefficients'], **tol)
    lx = EO.hierarchical_clustering(x)
    np.testing.assert_allclose(lx, g['linkage_x'], **tol)
    np.testing.assert_array_equal(EO.dendrogram_distance(g['linkage_x']), g['dend_x'])            # copies of merge heights
    np.testing.assert_array_equal(EO.dendrogram_distance(g['linkage_x'], condensed=False), g['dend_x_square'])
    np.testing.assert_allclose(EO.compare_cophenetic(g['linkage_x'], g['linkage_z']), g['compare_cophenetic'], rtol=1e-9)
    tf_f, tg_f = EO.compute_tf_tg_corrs(x, sym, tf_tg)
    np.testing.assert_allclose(tf_f, g['tf_tg_flat'], **tol)
    np.testing.assert_allclose(tg_f, g['tg_tg_flat'], **tol)
    tf_x, tg_x = EO.compute_tf_tg_corrs(x, sym, tf_tg, flat=False)
    tf_z, tg_z = EO.compute_tf_tg_corrs(z, sym, tf_tg, flat=False)
    assert [len(l) for l in tf_x] == list(g['tf_tg_lens']) and [len(l) for l in tg_x] == list(g['tg_tg_lens'])
    for wt in ('nb_genes', 'ones'):
        np.testing.assert_allclose(EO.psi_coefficient(tf_x, tf_z, wt), g['psi_' + wt], **tol)
        np.testing.assert_allclose(EO.phi_coefficient(tg_x, tg_z, wt), g['phi_' + wt], **tol)
    np.testing.assert_allclose(np.array(EO.compute_scores(x, z, sym, tf_tg), np.float64), g['compute_scores'], **tol)
    np.testing.assert_allclose(EO.cosine_similarity(g['cos_a'], g['cos_b']), g['cosine'], rtol=1e-6)


def test_dendrogram_distance_equals_scipy_cophenet():
    """The shortcut the GPU-side mirror takes (utils.dendrogram_distance = scipy cophenet) gives the
    numbers of the reference's explicit cluster-pair loops."""
    from scipy.cluster.hierarchy import cophenet
    g, _ = load(CASES[0])
    np.testing.assert_array_equal(cophenet(g['linkage_x']), g['dend_x'])
