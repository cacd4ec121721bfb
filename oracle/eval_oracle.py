"""NumPy/SciPy oracle for the extended evaluation scores built on the correlation path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Restates reference ``utils.py:827-834,
876-1105`` (gamma coefficients with dendrogrammatic distances, TF-TG / TG-TG scores); clustering is
SciPy's ``linkage('complete')`` / ``cophenet`` exactly as in the reference.  Parity PINNED by
``tests/golden/eval_*.npz`` (outputs of the unmodified reference ``utils.py`` imported in the build
container, generator: ``tools/gen_golden_eval.py``).
"""
import numpy as np
from scipy.cluster.hierarchy import cophenet, linkage

from .corr_oracle import correlations_list, pearson_correlation, upper_diag_list


def cosine_similarity(x, y):
    """Reference ``utils.py:827-834``."""
    return np.dot(x, y) / (np.linalg.norm(x) * np.linalg.norm(y))


def hierarchical_clustering(data, corr_fun=pearson_correlation):
    """Complete-linkage clustering of the genes on the distance ``1 - corr``.  Reference ``utils.py:1019-1032``."""
    y = 1 - correlations_list(data, data, corr_fun)
    return linkage(y, 'complete')


def dendrogram_distance(l_matrix, condensed=True):
    """Distance between two genes = height of the merge that first joins them.  The reference fills
    a G x G matrix cluster pair by cluster pair in Python loops (``utils.py:1062-1092``); the values are
    exactly SciPy's cophenetic distances, restated here with the same loops for small inputs."""
    nb_genes = l_matrix.shape[0] + 1
    members = {i: [i] for i in range(nb_genes)}
    m = np.zeros((nb_genes, nb_genes))
    for i, (c1, c2, dist, _) in enumerate(l_matrix):
        a, b = members[int(c1)], members[int(c2)]
        for p in a:
            for q in b:
                m[p, q] = dist
                m[q, p] = dist
        members[nb_genes + i] = a + b
    return upper_diag_list(m) if condensed else m


def compare_cophenetic(l_matrix1, l_matrix2):
    """Reference ``utils.py:1095-1105``."""
    return pearson_correlation(dendrogram_distance(l_matrix1), dendrogram_distance(l_matrix2))


def gamma_coefficients(expr_x, expr_z):
    """Gamma(D^X,D^Z), Gamma(D^X,T^X), Gamma(D^Z,T^Z), Gamma(T^X,T^Z).  Reference ``utils.py:876-903``."""
    dists_x = 1 - correlations_list(expr_x, expr_x)
    dists_z = 1 - correlations_list(expr_z, expr_z)
    gamma_dx_dz = pearson_correlation(dists_x, dists_z)
    xl = hierarchical_clustering(expr_x)
    gamma_dx_tx, _ = cophenet(xl, dists_x)
    zl = hierarchical_clustering(expr_z)
    gamma_dz_tz, _ = cophenet(zl, dists_z)
    return gamma_dx_dz, gamma_dx_tx, gamma_dz_tz, compare_cophenetic(xl, zl)


def compute_tf_tg_corrs(expr, gene_symbols, tf_tg, flat=True):
    """Per transcription factor: TF-TG and TG-TG correlation lists.  Reference ``utils.py:906-941``
    (the interaction dictionary is an argument here; the reference can also load RegulonDB)."""
    gene_symbols = np.array(gene_symbols)
    tf_tg_corr, tg_tg_corr = [], []
    for tf, tgs in tf_tg.items():
        tg_idxs = np.array([np.where(gene_symbols == tg)[0] for tg in tgs if tg in gene_symbols]).ravel()
        if tf in gene_symbols and len(tg_idxs) > 0:
            expr_tgs = expr[:, tg_idxs]
            tg_tg_corr += [correlations_list(expr_tgs, expr_tgs).tolist()]
            tf_idx = np.argwhere(gene_symbols == tf)[0]
            expr_tf = expr[:, tf_idx]
            tf_tg_corr += [pearson_correlation(expr_tf[:, None], expr_tgs).ravel().tolist()]
    if flat:
        tf_tg_corr = [c for l in tf_tg_corr for c in l]
        tg_tg_corr = [c for l in tg_tg_corr for c in l]
    return tf_tg_corr, tg_tg_corr


def psi_coefficient(tf_tg_x, tf_tg_z, weights_type='nb_genes'):
    """Reference ``utils.py:944-963``."""
    weights_sum = total_sum = 0
    for cx, cz in zip(tf_tg_x, tf_tg_z):
        weight = len(cx) if weights_type == 'nb_genes' else 1
        weights_sum += weight
        total_sum += weight * cosine_similarity(np.array(cx), np.array(cz))
    return total_sum / weights_sum


def phi_coefficient(tg_tg_x, tg_tg_z, weights_type='nb_genes'):
    """Reference ``utils.py:966-988``."""
    weights_sum = total_sum = 0
    for cx, cz in zip(tg_tg_x, tg_tg_z):
        if len(cx) > 0:
            weight = 1
            if weights_type == 'nb_genes':
                weight = max(np.roots([1, 1, -2 * len(cx)]))
            weights_sum += weight
            total_sum += weight * cosine_similarity(np.array(cx), np.array(cz))
    return total_sum / weights_sum


def compute_scores(expr_x, expr_z, gene_symbols, tf_tg):
    """[S_dist, S_dend, S_sdcc, S_tftg, S_tgtg].  Reference ``utils.py:991-1012``."""
    g_dx_dz, g_dx_tx, g_dz_tz, g_tx_tz = gamma_coefficients(expr_x, expr_z)
    r_tf, r_tg = compute_tf_tg_corrs(expr_x, gene_symbols, tf_tg, flat=False)
    s_tf, s_tg = compute_tf_tg_corrs(expr_z, gene_symbols, tf_tg, flat=False)
    return [g_dx_dz, g_tx_tz, (g_dx_tx - g_dz_tz) ** 2, psi_coefficient(r_tf, s_tf), phi_coefficient(r_tg, s_tg)]
