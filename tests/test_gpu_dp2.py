"""Two torch.distributed ranks (gloo; both processes on the one GPU of the box) through the PRODUCT's
distributed paths with the real HIP kernels:
  * train(): batch data parallel (row shards of the global batch, reduce-scatter / sharded RMSprop /
    all-gather per optimizer step) against the one-process run on the whole batch;
  * gamma_coef / GammaScorer: the upper-triangular tile list dealt over the ranks + one 6-scalar
    all-reduce, including the NaN fall-back (reference utils.py:863-873, :844-849);
  * make_score_fn: generator forward over row shards of the test set + all-gather.
A one-GPU box cannot run two RCCL ranks: the two-rank collectives here are gloo's; the last test drives the same
wrappers through RCCL (c10d's communicator and libagx's own) with a single rank."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _problem():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import lowrank, make_problem
    p = make_problem(G=130, B=58, H=(24, 16), seed=33)      # 58 rows: batches of 24, 24 and a ragged 10
    rs = np.random.RandomState(3)
    xt, yt = lowrank(rs, 60, 300, k=6), lowrank(rs, 45, 300, k=6)
    return p, xt, yt


def _run(world, rank, out_path):
    """The same script for the one-process reference and for each rank."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import keras_list
    from adversarial_gene_expression_amd import tf_utils as T, utils as U
    p, xt, yt = _problem()
    f32 = lambda a: np.asarray(a, np.float32)
    out = {}
    for opt in ('rmsprop', 'adam'):
        gen = T.Generator(p['G'], p['V'], p['n_num'], p['H'], p['Z'], weights=keras_list(p['gp']))
        disc = T.Critic(p['G'], p['V'], p['n_num'], p['H'], 1, weights=keras_list(p['dp']))
        T.set_seed(2024)
        mk = (lambda: T.RMSprop(5e-4)) if opt == 'rmsprop' else (lambda: T.Adam(2e-4, beta_1=0.5))
        go, do = mk(), mk()
        tr = T.train(f32(p['x']), p['cc'], f32(p['nc']), p['Z'], 2, 24, gen, disc, lambda g: 0.0, lambda: None,
                     gen_opt=go, disc_opt=do, verbose=False, nb_critic=3)
        out[opt + '_w'] = np.concatenate([w.reshape(-1) for w in gen.get_weights() + disc.get_weights()])
        out[opt + '_gl'], out[opt + '_dl'] = np.array(tr['gen_loss']), np.array(tr['disc_loss'])
        slot = do.state_arrays(disc)
        out[opt + '_slot'] = np.concatenate([a.reshape(-1) for a in slot['rms' if opt == 'rmsprop' else 'v']])
    # gamma paths
    out['gamma'] = np.array([float(U.gamma_coef(xt, yt))])
    sc = U.GammaScorer(xt)
    out['scorer'] = np.array([float(sc(yt))])
    xc, yc = xt.copy(), yt.copy()
    xc[:, 7] = 1.0
    yc[:, 7] = 2.0                                      # NaN row/column in both: lists stay equally long
    out['gamma_nan'] = np.array([float(U.gamma_coef(xc, yc))])
    out['scorer_nan'] = np.array([float(U.GammaScorer(xc)(yc))])
    try:
        U.gamma_coef(xt, yc)                            # NaNs on one side only: utils.py:821 fires
        out['assert'] = np.array([0.0])
    except AssertionError:
        out['assert'] = np.array([1.0])
    # score_fn with the generator forward over row shards
    gen = T.Generator(300, [15, 3], 1, [16, 8], 6, seed=3)
    rs = np.random.RandomState(5)
    cc = np.stack([rs.randint(0, v, 60) for v in (15, 3)], 1).astype(np.int32)
    nc = np.zeros((60, 1), np.float32)
    T.set_seed(9)
    out['score_fn'] = np.array([float(T.make_score_fn(xt, cc, nc)(gen))])
    T.set_seed(9)
    out['pred'] = T.predict_sharded(cc, nc, gen).cpu().numpy()
    np.savez(out_path, **out)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        _run(world, rank, os.path.join(out_dir, 'rank%d.npz' % rank))
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_the_one_process_run(gpu, tmp_path):
    import torch.multiprocessing as mp
    from oracle import corr_oracle as CO
    ref_path = str(tmp_path / 'ref.npz')
    _run(1, 0, ref_path)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(ref_path)
    ranks = [np.load(str(tmp_path / ('rank%d.npz' % r))) for r in range(world)]
    lr = 5e-4
    for opt in ('rmsprop', 'adam'):
        # ranks hold bit-identical replicas (all-gathered parameters)
        np.testing.assert_array_equal(ranks[0][opt + '_w'], ranks[1][opt + '_w'])
        np.testing.assert_array_equal(ranks[0][opt + '_slot'], ranks[1][opt + '_slot'])
        # and reproduce the one-process run on the whole batch: losses to 2e-4; weights: 99 % within 2e-6
        # (RMSprop sensitivity, see test_gpu_cgan.test_weights_after_steps; ~1e4 weights, 18 steps)
        for k in ('_gl', '_dl'):
            np.testing.assert_allclose(ranks[0][opt + k], ref[opt + k], rtol=2e-4, atol=2e-4)
        err = np.abs(ranks[0][opt + '_w'] - ref[opt + '_w'])
        assert np.quantile(err, 0.99) < 2e-6, (opt, np.quantile(err, 0.99))
        assert np.abs(ref[opt + '_w']).max() > 0 and err.max() < 18 * 3 * lr * np.sqrt(10)
        serr = np.abs(ranks[0][opt + '_slot'] - ref[opt + '_slot'])
        assert np.quantile(serr, 0.99) < 1e-6 * max(1.0, np.abs(ref[opt + '_slot']).max())
    p, xt, yt = _problem()
    want = float(CO.gamma_coef(xt, yt))
    for r in ranks:
        for k in ('gamma', 'scorer', 'gamma_nan', 'scorer_nan', 'score_fn'):
            assert abs(float(r[k][0]) - float(ref[k][0])) < 1e-6, (k, r[k], ref[k])
        assert abs(float(r['gamma'][0]) - want) < 2e-5
        assert r['assert'][0] == 1.0 and ref['assert'][0] == 1.0
        np.testing.assert_array_equal(r['pred'], ref['pred'])     # row shards of the global z draw: same matrix
    xc, yc = xt.copy(), yt.copy()
    xc[:, 7] = 1.0
    yc[:, 7] = 2.0
    with np.errstate(all='ignore'):
        assert abs(float(ref['gamma_nan'][0]) - float(CO.gamma_coef(xc, yc))) < 2e-5


def _rccl_worker(rank, port, out_path, native):
    """One rank over the 'nccl' (= RCCL) backend, initialised as bench.py does it: the collectives wrappers of the
    data-parallel step on device tensors.  With one rank every collective is the identity, but the call goes
    through c10d's RCCL communicator (or libagx's own with AGX_NATIVE_COMM=1) on the compute stream."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if native:
        os.environ['AGX_NATIVE_COMM'] = '1'
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from adversarial_gene_expression_amd import _dev
        src = torch.arange(4096, dtype=torch.float32, device='cuda') * 0.25 - 7.0
        a = src.clone()
        _dev.allreduce_sum(a)
        rs = torch.zeros(4096, dtype=torch.float32, device='cuda')
        _dev.reduce_scatter_sum(rs, src)
        ag = torch.zeros(4096, dtype=torch.float32, device='cuda')
        _dev.all_gather(ag, src)
        d = torch.tensor([1.5, -2.0, 3.25, 0.0, 1e300, -1e-300], dtype=torch.float64, device='cuda')
        d2 = d.clone()
        _dev.allreduce_sum(d2)
        torch.cuda.synchronize()
        np.savez(out_path, src=src.cpu().numpy(), a=a.cpu().numpy(), rs=rs.cpu().numpy(), ag=ag.cpu().numpy(),
                 d=d.cpu().numpy(), d2=d2.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('native', [False, True], ids=['c10d', 'agx_comm'])
def test_collectives_run_through_rccl_with_one_rank(gpu, tmp_path, native):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'rccl.npz')
    mp.spawn(_rccl_worker, args=(_free_port(), out, native), nprocs=1, join=True)
    r = np.load(out)
    for k in ('a', 'rs', 'ag'):
        np.testing.assert_array_equal(r[k], r['src'])
    np.testing.assert_array_equal(r['d2'], r['d'])
