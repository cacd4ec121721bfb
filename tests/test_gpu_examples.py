"""The counterpart entry scripts (examples/) run end to end on small synthetic inputs: preprocessing,
model construction, train() with its evaluation cadence, final Gamma(Dx, Dz)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'examples', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gtex_shaped_script(gpu, tmp_path, capsys):
    m = _load('train_gtex_shaped')
    counts, cat = m.synthetic_rnaseqdb(300, 150, seed=1)
    cfg = dict(m.CONFIG, hdim=32, latent_dim=8)
    gen, score = m.run(counts, cat, [15, 3], epochs=6, models_dir=str(tmp_path), config=cfg)
    out = capsys.readouterr().out
    assert 'Saving model ...' in out and 'Epoch 6.' in out and 'Gamma(Dx, Dz):' in out
    assert np.isfinite(score) and -1.0 <= score <= 1.0
    assert os.path.exists(os.path.join(str(tmp_path), 'gen_rnaseqdb.h5.npz')) or os.path.exists(os.path.join(str(tmp_path), 'gen_rnaseqdb.h5'))


def test_ecoli_shaped_script(gpu, capsys):
    m = _load('train_ecoli_shaped')
    expr, cond = m.synthetic_compendium(160, 90, seed=2)
    gen, score = m.run(expr, cond, epochs=3)
    out = capsys.readouterr().out
    assert 'Epoch 3.' in out and 'Gamma(Dx, Dz):' in out
    assert np.isfinite(score) and -1.0 <= score <= 1.0
