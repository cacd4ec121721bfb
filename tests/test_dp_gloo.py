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


class _SgdStub:
    """CPU stand-in for the HIP optimizers: same apply_range contract (parameters [lo, lo+n) from a shard)."""
    def __init__(self):
        self.iterations = 0

    def apply_range(self, net, lo, n, g, advance=True):
        net.params[lo:lo + n] -= 0.5 * g[:n]
        self.iterations += int(advance)


def _exchange_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from adversarial_gene_expression_amd import tf_utils as T
        # (n_flat, exchange pieces): first: the loss slots fall into the LAST rank's shard only; the chunked
        # layouts cut the 'kernel' (rows x ld at offset 64) into row blocks of 64 rows
        for n_flat, chunks in ((64, 0), (192, 0), (4096 + 64, 0), (64 + 300 * 8 + 64, 3), (64 + 1000 * 12 + 128, 4),
                               (64 + 50 * 8, 4)):
            os.environ['AGX_DP_CHUNKS'] = str(chunks)
            rs = np.random.RandomState(n_flat)
            w0 = rs.randn(n_flat).astype(np.float32)
            size = n_flat + T._LOSS_SLOTS + T._EXCH_SLACK
            pbuf, gbuf = torch.zeros(size), torch.zeros(size)
            pbuf[:n_flat] = torch.from_numpy(w0)
            ld = 12 if n_flat == 64 + 1000 * 12 + 128 else 8
            rows = max(1, (n_flat - 128) // ld)
            net = SimpleNamespace(n_flat=n_flat, _pbuf=pbuf, _gbuf=gbuf, params=pbuf[:n_flat],
                                  grads=gbuf[:n_flat + T._LOSS_SLOTS], _shards={},
                                  dims=[rows, ld, 1], ldw=[ld, 1], w_off=[64, 64 + rows * ld])
            grs = [np.random.RandomState(1000 + r).randn(n_flat).astype(np.float32) for r in range(world)]
            gbuf[:n_flat] = torch.from_numpy(grs[rank])
            gbuf[n_flat + 2] = 1.5 + rank                 # this rank's share of the loss scalar
            opt = _SgdStub()
            T._dp_update(net, opt)
            want = w0 - 0.5 * np.sum(grs, axis=0)
            np.testing.assert_allclose(net.params.numpy(), want, rtol=1e-6, atol=1e-6)
            assert float(T._reduced_loss(net, 2)) == sum(1.5 + r for r in range(world))
            assert opt.iterations == 1
            total, shard, lo = T._exchange_layout(net, world, rank)
            assert total % (4 * world) == 0 and total >= n_flat + T._LOSS_SLOTS and shard * world == total
            pieces = T._exchange_pieces(net, world, chunks)
            assert pieces[0][0] == 0 and pieces[-1][0] + pieces[-1][1] * world == total
            assert all(a[0] + a[1] * world == b[0] for a, b in zip(pieces, pieces[1:]))
            assert all(st % 4 == 0 and per % 4 == 0 and per > 0 for st, per in pieces)
            assert len(pieces) == (1 if chunks <= 1 else len(T._exchange_rows(net, chunks)) - 1)
            if chunks > 1 and rows >= 64 * chunks:
                assert len(pieces) == chunks
            # the slots come back assembled from every rank's shares (a collective)
            slot = torch.zeros(n_flat)
            for st, per in pieces:
                a, b = st + rank * per, min(st + (rank + 1) * per, n_flat)
                if b > a:
                    slot[a:b] = torch.arange(a, b, dtype=torch.float32)
            np.testing.assert_array_equal(T._gathered_slot(net, slot).numpy(), np.arange(n_flat, dtype=np.float32))
        open(os.path.join(out_dir, 'ex%d' % rank), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_reduce_scatter_sharded_update_all_gather(tmp_path):
    """tf_utils._dp_update on CPU tensors over gloo (2 and 3 ranks): reduce-scatter of the flat gradient
    buffer, optimizer on the own shard, all-gather of the parameters == full-gradient update everywhere;
    the loss scalars behind the gradients come back summed.  Flat (one piece) and chunked exchange layouts."""
    for world in (2, 3):
        d = tmp_path / ('w%d' % world)
        d.mkdir()
        mp.spawn(_exchange_worker, args=(world, _free_port(), str(d)), nprocs=world, join=True)
        assert all(os.path.exists(os.path.join(str(d), 'ex%d' % r)) for r in range(world))
