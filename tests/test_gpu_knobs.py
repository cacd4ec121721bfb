"""Every run-time knob of libagx selects between launch sequences that must compute the same
iteration.  The default sequence is compared with the oracle at every BASELINE size in test_gpu_fullsize.py; here each
knob's sequence is compared with the default's on an E. coli-sized problem (fused optimizer, chained input assembly,
tails, the folded step-0 input and the two-layer generator kernel are all active at that size), without and with
augmentation.  One child process per knob: the library reads its environment once."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (round 3: the knobs whose alternative only ever measured slower - AGX_OPT_BIG_TILES, AGX_DIRECT_SK_MULT,
#  AGX_TAIL_RPW, AGX_DIRECT_NW8_MIN, AGX_DIRECT8_NO_U4, AGX_TAIL_MAX_ROWS, AGX_FUSE_GEN_BIG, AGX_NO_FOLD0,
#  AGX_NO_CHAIN_AUG - are deleted; what is left selects the independent launch sequences other configurations
#  run anyway - wide shapes, data parallel, Adam - and doubles as a cross-check of the fused default)
KNOBS = [
    ('AGX_NO_TAIL', '1'), ('AGX_NO_FUSED_OPT', '1'), ('AGX_NO_FUSED_REST', '1'), ('AGX_NO_TN_FUSE', '1'),
    ('AGX_NO_CHAIN_INPUT', '1'), ('AGX_NO_DRAW_FUSE', '1'), ('AGX_NO_GEN_TAIL', '1'), ('AGX_TAIL_NSPLIT', '1'),
    ('AGX_TAIL_NSPLIT', '2'), ('AGX_NO_STAGE_DRAWS', '1'),
]


def _run(tmp_path, name, env_extra, shape='ecoli'):
    out = str(tmp_path / (name + '.npz'))
    env = {k: v for k, v in os.environ.items() if not k.startswith('AGX_')}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'knob_runner.py'), out, shape], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, (name, r.stderr[-2000:])
    return np.load(out)


@pytest.fixture(scope='module')
def default_run(tmp_path_factory):
    return _run(tmp_path_factory.mktemp('knobs'), 'default', {})


@pytest.mark.parametrize('knob,value', KNOBS, ids=['%s=%s' % kv for kv in KNOBS])
def test_knob_sequence_computes_the_default_iteration(gpu, tmp_path, default_run, knob, value):
    got = _run(tmp_path, 'knob', {knob: value})
    for tag in ('plain', 'aug'):
        ref_l, ref_w = default_run[tag + '_losses'], default_run[tag + '_w']
        assert np.all(np.isfinite(ref_l)) and np.all(np.isfinite(got[tag + '_losses']))
        # losses of the second iteration already depend on every update of the first
        np.testing.assert_allclose(got[tag + '_losses'], ref_l, rtol=2e-4, atol=2e-4)
        err = np.abs(got[tag + '_w'] - ref_w)
        # 12 RMSprop steps of 5e-4: a differently rounded gradient moves a weight by at most a few 1e-7
        # (see test_gpu_cgan.test_weights_after_steps for the sensitivity argument)
        assert np.quantile(err, 0.999) < 2e-6, (knob, tag, float(np.quantile(err, 0.999)), float(err.max()))


@pytest.mark.parametrize('knob', ['AGX_NO_GRAM', 'AGX_NO_FF', 'AGX_NO_FUSED_OPT', 'AGX_NO_TAIL'])
def test_gram_form_computes_the_direct_form(gpu, tmp_path, knob):
    """Batches of at least the first hidden width take the Gram form of the gradient penalty (cgan.hip gram_ok:
    |d_1 W_g^T|^2 = d_1 (W_g^T W_g) d_1^T, and so on): same algebra as the direct form (AGX_NO_GRAM), other rounding.  The
    full-size 512-row cases of test_gpu_fullsize.py and test_gpu_split8.py compare the Gram form with the oracle.
    On top of it the one-GPU iteration without augmentation never forms the generated rows (record_iteration `ff`:
    x_f = h W_out + b_out enters every product through W_out W_g); AGX_NO_FF keeps the Gram form and the factored generator
    step but materialises the rows, AGX_NO_GRAM selects the direct form of everything."""
    a = _run(tmp_path, 'gram', {}, 'gram')
    b = _run(tmp_path, 'direct', {knob: '1'}, 'gram')
    for tag in ('plain', 'aug'):
        assert np.all(np.isfinite(a[tag + '_losses']))
        np.testing.assert_allclose(a[tag + '_losses'], b[tag + '_losses'], rtol=2e-4, atol=2e-4)
        err = np.abs(a[tag + '_w'] - b[tag + '_w'])
        assert np.quantile(err, 0.999) < 2e-6, (tag, float(np.quantile(err, 0.999)), float(err.max()))
        # (the other two knobs: the large-batch forms without the fused optimizer / without the tail kernels, i.e. the
        #  direct forms on the generic launches - must land on the default too)
        if knob == 'AGX_NO_GRAM' or (knob == 'AGX_NO_FF' and tag == 'plain'):   # (with augmentation the generated rows are formed anyway)
            assert err.max() > 0, 'the two forms round differently: identical weights mean the knob selected nothing'
