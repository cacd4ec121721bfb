"""Host-side logic that needs no GPU: model-file config validation, batch sharding."""
import ast

import numpy as np
import pytest


def test_model_config_is_parsed_as_a_literal_and_validated():
    from adversarial_gene_expression_amd import tf_utils as T
    good = dict(kind='gen', x_dim=37, z_dim=6, vocab_sizes=[15, 3], nb_numeric=1, h_dims=[16, 8], slope=0.0)
    assert T._check_model_config(ast.literal_eval(repr(good))) == good
    # code is not a literal: ast.literal_eval refuses it before anything is constructed
    for bad in ("().__class__.__base__.__subclasses__()", "__import__('os').system('true')", "[x for x in (1,)]"):
        with pytest.raises((ValueError, SyntaxError)):
            ast.literal_eval(bad)
    for bad in (dict(good, kind='other'), dict(good, x_dim='37'), dict(good, h_dims=[]), dict(good, vocab_sizes=[0]),
                dict(good, x_dim=True), [1, 2], dict(good, slope='0'), {k: v for k, v in good.items() if k != 'z_dim'}):
        with pytest.raises(ValueError):
            T._check_model_config(bad)


def test_load_model_rejects_a_crafted_config(tmp_path):
    """A model file whose config string is an expression must fail in the parser (no GPU, no eval)."""
    from adversarial_gene_expression_amd import tf_utils as T
    path = str(tmp_path / 'evil.npz')
    np.savez(path, __config__=np.array("().__class__.__base__.__subclasses__()"), w000=np.zeros(3, np.float32))
    with pytest.raises((ValueError, SyntaxError)):
        T.load_model(path)


@pytest.mark.parametrize('n,world', [(32, 8), (12, 8), (4096, 8), (33, 2), (7, 3)])
def test_shard_partitions_the_batch(n, world):
    from adversarial_gene_expression_amd import tf_utils as T
    cuts = [T._shard(100, 100 + n, world, r) for r in range(world)]
    assert cuts[0][0] == 100 and cuts[-1][1] == 100 + n
    for (a, b), (c, d) in zip(cuts[:-1], cuts[1:]):
        assert b == c
    sizes = [b - a for a, b in cuts]
    assert max(sizes) - min(sizes) <= 1



def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N ranks itself (RANK / WORLD_SIZE / MASTER_* set),
    relays rank 0's JSON line and returns a failing rank's exit code (rehearsal mode: no GPU is touched)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env['AGX_BENCH_LAUNCH_ONLY'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0]) == {'launch_only': True, 'world': 4, 'master': '127.0.0.1'}
    env['AGX_BENCH_LAUNCH_RC_2'] = '7'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4'], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr)
