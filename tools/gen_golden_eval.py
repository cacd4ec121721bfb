"""Golden vectors for the extended evaluation scores from the UNMODIFIED reference ``utils.py``
(gamma_coefficients, hierarchical_clustering, dendrogram_distance, compare_cophenetic,
compute_tf_tg_corrs, psi_coefficient, phi_coefficient, compute_scores, cosine_similarity).

Runs only in the build container (needs /root/reference); only the .npz data ships.

    python tools/gen_golden_eval.py
"""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_corr import OUT, import_reference_utils  # noqa: E402


def main():
    ru = import_reference_utils()
    warnings.simplefilter('ignore')
    for i, (name, n1, n2, g, dt) in enumerate([('n60_g24_f32', 60, 41, 24, np.float32), ('n90_g40_f64', 90, 90, 40, np.float64)]):
        rs = np.random.RandomState(300 + i)
        load = rs.randn(5, g)                       # shared gene loadings: z resembles x, as a trained generator's output would
        x = (rs.randn(n1, 5) @ load + 0.5 * rs.randn(n1, g)).astype(dt)
        z = (rs.randn(n2, 5) @ load + 0.8 * rs.randn(n2, g)).astype(dt)
        symbols = np.array(['g%02d' % j for j in range(g)])
        # synthetic regulatory network: a few TFs, some with one target (empty TG-TG list), one target
        # that is not measured and one TF that is not measured (both skipped by the reference)
        tf_tg = {'g00': ['g03', 'g04', 'g05', 'g17'], 'g01': ['g06'], 'g02': ['g07', 'g08', 'zzz', 'g09'],
                 'absent_tf': ['g10', 'g11'], 'g12': ['g13', 'g14', 'g15', 'g16', 'g18', 'g19']}
        out = {'x': x, 'z': z, 'symbols': symbols, 'tf_tg_json': np.array(json.dumps(tf_tg))}
        out['gamma_coefficients'] = np.array(ru.gamma_coefficients(x, z), dtype=np.float64)
        lx = ru.hierarchical_clustering(x)
        lz = ru.hierarchical_clustering(z)
        out['linkage_x'], out['linkage_z'] = lx, lz
        out['dend_x'] = ru.dendrogram_distance(lx, condensed=True)
        out['dend_x_square'] = ru.dendrogram_distance(lx, condensed=False)
        out['compare_cophenetic'] = np.asarray(ru.compare_cophenetic(lx, lz))
        tf_f, tg_f = ru.compute_tf_tg_corrs(x, symbols, tf_tg=tf_tg, flat=True)
        out['tf_tg_flat'], out['tg_tg_flat'] = np.array(tf_f), np.array(tg_f)
        tf_x, tg_x = ru.compute_tf_tg_corrs(x, symbols, tf_tg=tf_tg, flat=False)
        tf_z, tg_z = ru.compute_tf_tg_corrs(z, symbols, tf_tg=tf_tg, flat=False)
        out['tf_tg_lens'] = np.array([len(l) for l in tf_x])
        out['tg_tg_lens'] = np.array([len(l) for l in tg_x])
        for wt in ('nb_genes', 'ones'):
            out['psi_' + wt] = np.asarray(ru.psi_coefficient(tf_x, tf_z, weights_type=wt))
            out['phi_' + wt] = np.asarray(ru.phi_coefficient(tg_x, tg_z, weights_type=wt))
        # compute_scores loads RegulonDB when no dictionary is given; same arithmetic with ours injected
        saved = ru.tf_tg_interactions
        ru.tf_tg_interactions = lambda *a, **k: tf_tg
        try:
            out['compute_scores'] = np.array(ru.compute_scores(x, z, symbols), dtype=np.float64)
        finally:
            ru.tf_tg_interactions = saved
        a, b = rs.randn(13).astype(dt), rs.randn(13).astype(dt)
        out['cos_a'], out['cos_b'], out['cosine'] = a, b, np.asarray(ru.cosine_similarity(a, b))
        np.savez_compressed(os.path.join(OUT, 'eval_%s.npz' % name), **out)
        print(name, out['gamma_coefficients'], out['compute_scores'])


if __name__ == '__main__':
    main()
