"""Data-parallel contract of the train step on CPU: 2 processes, gloo backend.

The HIP kernels cannot run here, so each rank computes its shard's gradients with the oracle's
closed-form backward (scaled by 1/B_global like the kernels), and the PRODUCT's sharding / all-reduce
plumbing (tf_utils._world, _shard, _allreduce_grads with the loss scalars riding in the gradient
buffer's tail slots) combines them.  The sum must equal the full-batch gradients and loss."""
import os
import socket
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from adversarial_gene_expression_amd import tf_utils as T
        from helpers import make_problem
        from oracle import cgan_closed_form as CF
        assert T._world() == (world, rank)
        p = make_problem(B=9, seed=17)
        # every rank walks the same global batches as train(): full batch of 9 then a ragged tail of 3
        for lo_g, hi_g in ((0, 9), (6, 9)):
            lo, hi = T._shard(lo_g, hi_g, world, rank)
            bg = hi_g - lo_g
            s = slice(lo, hi)
            loss, g = CF.disc_grads(p['x'][s], p['z'][s], p['cc'][s], p['nc'][s], p['gp'], p['dp'], p['alpha'][s],
                                    B_global=bg)
            flat = np.concatenate([np.asarray(a, np.float64).reshape(-1) for a in CF.flat(g)])
            # the product keeps the loss scalars in 64 slots behind the gradients so they ride the all-reduce
            net = SimpleNamespace(grads=torch.tensor(np.concatenate([flat, [0.0, 0.0, float(loss[2])], np.zeros(61)])))
            T._allreduce_grads(net)
            sg = slice(lo_g, hi_g)
            floss, fg = CF.disc_grads(p['x'][sg], p['z'][sg], p['cc'][sg], p['nc'][sg], p['gp'], p['dp'], p['alpha'][sg])
            want = np.concatenate([np.asarray(a, np.float64).reshape(-1) for a in CF.flat(fg)])
            got = net.grads.numpy()
            np.testing.assert_allclose(got[:want.size], want, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(got[want.size + 2], float(floss[2]), rtol=1e-9)
        # shards tile the batch without gaps or overlap, sizes differ by at most one
        for n in (1, 2, 3, 12, 33):
            parts = [T._shard(5, 5 + n, world, r) for r in range(world)]
            assert parts[0][0] == 5 and parts[-1][1] == 5 + n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
        open(os.path.join(out_dir, 'ok%d' % rank), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), 'ok%d' % r)) for r in range(world))
