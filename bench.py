#!/usr/bin/env python3
"""Headline benchmark: cGAN train samples/s at GTEx shape (18154 genes) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu]

A "step" is one iteration of the reference train loop (tf_utils.py:316-323): nb_critic=5
critic steps (fresh z and alpha, same real batch) + 1 generator step on one batch, optimizer
updates included, synthetic data resident in HBM.  Default workload = BASELINE.json configs[1]
(GTEx-shaped cGAN, tissue+dataset covariates, batch 32 per GPU); for N > 1 the driver launches
one rank per GPU with torch.distributed.run and the same per-GPU batch (weak scaling) with an
RCCL all-reduce of the gradients per optimizer step.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     - the dominant kernel class of the timed region, timed live with HIP events on
                 the launch stream (agx_prof_*), algorithmic bytes/FLOPs per launch over its
                 mean duration, against the gfx950 peak (HBM 8 TB/s, fp32 MFMA 157.3 TFLOP/s)
  cpu_baseline - the torch-CPU restatement of the same iteration (oracle/, kind "port": the
                 reference's TF path cannot run offline) timed on this host's cores, N=1 only
  secondary    - the same measurement at 512 samples/GPU (BASELINE.json configs[2] per-GPU
                 shard, the MFMA-bound regime), N=1 only unless --secondary
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA dense peak

WORKLOADS = {
    # name: (G, vocab_sizes, n_num, h_dims, z_dim)
    'gtex': (18154, [15, 3], 1, [256, 256], 64),          # gtex_tcga_gan.py:11-18, notebook shapes
    'wide': (20000, [15, 3], 1, [1024, 1024], 64),        # BASELINE.json configs[4]
    'ecoli': (4297, [6, 8, 5, 11, 5, 3, 5], 1, [128, 128], 10),       # BASELINE.json configs[0] (all genes)
    'ecoli_crp': (1076, [6, 8, 5, 11, 5, 3, 5], 1, [128, 128], 10),   # as shipped: root_gene='CRP' (ecoli_gan.py:14)
}
P_AUG = {'ecoli': 0.25, 'ecoli_crp': 0.25}                # ecoli_gan.py:90; the GTEx script trains without augmentation


def algorithmic_flops_per_iter(G, V, n, H, Z, B, nb_critic=5):
    """SURVEY.md section 8(d) formulas (two hidden layers)."""
    E = sum(int(v ** 0.5) + 1 for v in V)
    Dg, Dd = Z + n + E, G + n + E
    H1, H2 = H
    gen_fwd = 2 * B * (Dg * H1 + H1 * H2 + H2 * G)
    disc_fwd = 2 * B * (Dd * H1 + H1 * H2 + H2)
    gp_inp = 2 * B * (H1 * H2 + H1 * G)
    bwd_w = 2 * B * (2 * H2 + 2 * H1 * H2 + Dd * H1)
    gp_2nd = 2 * B * (2 * G * H1 + 2 * H1 * H2) + B * H2
    critic = gen_fwd + 3 * disc_fwd + gp_inp + 2 * bwd_w + gp_2nd
    disc_bwd_x = 2 * B * (H2 + H1 * H2 + G * H1)
    gen_bwd = 2 * B * (2 * H2 * G + 2 * H1 * H2 + Dg * H1)
    gen_step = gen_fwd + disc_fwd + disc_bwd_x + gen_bwd
    return nb_critic * critic + gen_step


def synthetic_batch_pool(G, V, n_num, rows, seed):
    """X = F L + 0.5 eps, z-scored per gene, generated on the device; covariates uniform."""
    import torch
    g = torch.Generator(device='cuda').manual_seed(seed)
    f = torch.randn(rows, 32, device='cuda', generator=g)
    l = torch.randn(32, G, device='cuda', generator=g)
    x = f @ l + 0.5 * torch.randn(rows, G, device='cuda', generator=g)
    x = (x - x.mean(0)) / x.std(0, unbiased=False)
    cc = torch.stack([torch.randint(0, v, (rows,), device='cuda', generator=g) for v in V], 1).to(torch.int32)
    nc = torch.zeros(rows, n_num, device='cuda')
    return x.contiguous(), cc.contiguous(), nc


def prof_read_all(lib, _lib):
    out = {}
    for k in range(64):
        name = lib.agx_prof_name(k)
        if not name:
            break
        ms, cnt, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.agx_prof_read(k, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by)))
        if cnt.value:
            out[k] = dict(name=name.decode(), ms=ms.value, count=cnt.value, flops=fl.value, bytes=by.value)
    return out


def run_gpu(workload, B, steps, warmup, world, rank, nb_critic=5, roofline=True):
    import torch
    import torch.distributed as dist
    from adversarial_gene_expression_amd import _lib, tf_utils as T
    lib = _lib.load()
    G, V, n_num, H, Z = WORKLOADS[workload]
    gen = T.Generator(G, V, n_num, H, Z, seed=1)
    disc = T.Critic(G, V, n_num, H, 1, seed=2)
    gopt, dopt = T.RMSprop(5e-4), T.RMSprop(5e-4)     # tf_utils.py:285-288
    nbatches = 8
    x, cc, nc = synthetic_batch_pool(G, V, n_num, B * nbatches, seed=100 + rank)
    T.set_seed(1000)       # one global draw stream: every rank keeps its rows of it
    Bg = B * world

    use_graph = not os.environ.get('AGX_BENCH_EAGER')
    p_aug = P_AUG.get(workload, 0)
    np.random.seed(0)      # augmentation constants (NumPy draws at first use, like the reference's trace constants)
    last = [None]
    walk = [0]             # every iteration, probe or timed, takes the NEXT batch of the pool (see the note at the warm-up)

    def next_i():
        walk[0] += 1
        return walk[0] - 1

    def iteration(i):
        s = slice((i % nbatches) * B, (i % nbatches + 1) * B)
        xb, cb, nb = x[s], cc[s], nc[s]
        if use_graph:      # the path train() takes: one hipGraph per iteration (N > 1: one graph segment
            # per gradient exchange)
            last[0] = T.train_iteration(xb, cb, nb, gen, disc, gopt, dopt, nb_critic, p_aug=p_aug, batch_global=Bg,
                                        row_lo=rank * B)
            return
        for _ in range(nb_critic):
            z = T.random_normal((B, Z))
            al = T.random_uniform((B,))
            T.train_disc(xb, z, cb, nb, gen, disc, dopt, alpha=al, batch_global=Bg)
        z = T.random_normal((B, Z))
        T.train_gen(z, cb, nb, gen, disc, gopt, batch_global=Bg)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def eager_iteration(i):
        s = slice((i % nbatches) * B, (i % nbatches + 1) * B)
        xb, cb, nb = x[s], cc[s], nc[s]
        if world == 1 and use_graph:
            # the timed region's own launch sequence (fused epilogues, tail kernels), queued instead of replayed
            pair = T._pair(gen, disc, B, nb_critic + 1)
            T._rng.offset += pair.iteration_eager(xb, cb, nb, gen, disc, gopt, dopt, nb_critic, 10.0, T._rng.seed,
                                                  T._rng.offset)
            return
        for _ in range(nb_critic):
            T.train_disc(xb, T.random_normal((B, Z)), cb, nb, gen, disc, dopt, alpha=T.random_uniform((B,)),
                         batch_global=Bg)
        T.train_gen(T.random_normal((B, Z)), cb, nb, gen, disc, gopt, batch_global=Bg)

    def bracketed(filt, n):
        """n eager iterations with HIP-event brackets on the launch stream.  A queued busy-wait lets
        the host run ahead, so brackets time back-to-back GPU execution (no host launch gaps)."""
        torch.cuda.synchronize()
        lib.agx_prof_reset()
        lib.agx_prof_enable(filt)
        for i in range(n):
            if world == 1:
                _lib.check(lib.agx_prof_spin(torch.cuda.current_stream().cuda_stream, 4000 if B <= 64 else 1000))
            eager_iteration(next_i())
        torch.cuda.synchronize()
        lib.agx_prof_enable(-2)
        return prof_read_all(lib, _lib)

    dbg = (lambda m: print('[bench rank %d] %s' % (rank, m), file=sys.stderr, flush=True)) if os.environ.get('AGX_BENCH_DEBUG') else (lambda m: None)
    dbg('models built')
    if world > 1 and dist.get_backend() == 'nccl' and os.environ.get('AGX_DP_OVERLAP', '1') != '0':
        # The library-driven overlapped exchange (agx_cgan_dp_iteration over libagx's own RCCL communicator) is the
        # default under RCCL.  It has only ever run with ONE rank on the builder's one-GPU boxes: if any rank fails to
        # bring it up, ALL ranks fall back together to the c10d-driven exchange (tf_utils._dp_update).
        # Brought up in stages, each closed by a MIN all-reduce over c10d, so that no rank enters a collective of
        # the library's communicator that another rank never reaches: (1) the communicator itself, (2) one small
        # sum on it, checked, (3) the first overlapped iteration.
        from adversarial_gene_expression_amd import _dev

        def all_ok(stage, fn):
            ok = 1.0
            try:
                fn()
                torch.cuda.synchronize()
            except Exception as e:          # noqa: BLE001 - anything: the fall-back must be collective
                print('[bench rank %d] overlapped exchange: %s failed (%r)' % (rank, stage, e), file=sys.stderr, flush=True)
                ok = 0.0
            t = torch.tensor([ok], device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return float(t.item()) >= 1.0

        def probe_sum():
            one = torch.ones(256, device='cuda')
            _dev.allreduce_sum(one)
            torch.cuda.synchronize()
            if float(one[0].item()) != float(world) or float(one[-1].item()) != float(world):
                raise RuntimeError('sum of ones over %d ranks = %r' % (world, float(one[0].item())))

        def need_comm():
            if _dev.native_comm() is None:
                raise RuntimeError('no native communicator')

        up = all_ok('communicator', need_comm) and all_ok('probe all-reduce', probe_sum) \
            and all_ok('first iteration', lambda: iteration(next_i()))
        if not up:
            print('[bench rank %d] falling back to the c10d-driven exchange' % rank, file=sys.stderr, flush=True)
            os.environ['AGX_DP_OVERLAP'] = '0'
            os.environ['AGX_NATIVE_COMM'] = '0'
    iteration(next_i())                        # first call: workspace, graph capture
    sync()
    nprobe = 3
    # (the discovery probe - which kernel class dominates? - runs AFTER the timed region: its bracketed launches leave
    #  the GPU idle in between, and with the driver's --steps 20 --warmup 5 the five warm-up steps did not bring the
    #  clocks back: 64.1 k samples/s against 66.8 k with 200 timed steps)
    # the W untimed warm-up steps run right in front of the timed region
    # (batches are walked in sequence across probe, warm-up and timed region, as train() walks the data set:
    #  training twice in a row on one batch of this synthetic data drives the wide critic to a zero penalty slope,
    #  whose NaN - tf_utils.py:169 has no epsilon - the fp64 oracle reproduces at the same step:
    #  profiles/r02_wide_zero_slope_nan_matches_oracle.txt)
    for i in range(warmup):
        iteration(next_i())
        dbg('warmup %d queued' % i)
    sync()
    dbg('warmup done')
    t0 = time.perf_counter()
    for i in range(steps):
        iteration(next_i())
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_losses = None
    if last[0] is not None:   # the timed iterations computed something sane: losses of the last one, read after the clock stopped
        final_losses = [float(last[0][0]), float(last[0][1])]
        if not all(np.isfinite(v) for v in final_losses):
            raise RuntimeError('bench: non-finite losses after the timed region: %r' % (final_losses,))
    if not roofline:
        res = dict(value=Bg * steps / dt, ms_per_step=dt / steps * 1e3, final_losses=final_losses)
        del gen, disc, x
        torch.cuda.empty_cache()
        return res
    disc_prof = bracketed(-1, nprobe)          # discovery: which kernel class dominates?
    dominant = max(disc_prof, key=lambda k: disc_prof[k]['ms'])
    dbg('probe done')
    lib.agx_prof_reset()
    # dominant kernel, timed with HIP events on the launch stream right after the timed region
    # (a captured graph cannot host event brackets on ROCm 7.2, so the same iteration is replayed
    # through the eager entry points; profiles/ holds the rocprofv3 trace of the graph run itself)
    dom = bracketed(dominant, 5)[dominant]
    # what the two event records of a bracket add, measured with empty brackets on the same stream
    # (the rocprofv3 kernel trace under profiles/ has no such term): subtracted, both figures reported
    empty = ctypes.c_double(0.0)
    _lib.check(lib.agx_prof_empty_bracket_us(torch.cuda.current_stream().cuda_stream, 200, ctypes.byref(empty)))
    raw_us = dom['ms'] / dom['count'] * 1e3
    avg_s = max(raw_us - empty.value, 0.25 * raw_us) * 1e-6
    mfma_bound = B >= 256 and dom['name'] not in ('opt_rmsprop', 'opt_adam')
    if mfma_bound:
        ach = dom['flops'] / dom['count'] / avg_s / 1e12
        roof = dict(bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TF, unit='TFLOP/s', frac=ach / MFMA_F32_PEAK_TF)
    else:
        ach = dom['bytes'] / dom['count'] / avg_s / 1e9
        roof = dict(bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS)
    traffic, traffic_src = measured_traffic(dom['name'], B)
    roof.update(kernel=dom['name'], launches=dom['count'], avg_us=avg_s * 1e6, avg_us_bracketed=raw_us,
                bracket_overhead_us=empty.value, traffic=traffic, traffic_source=traffic_src,
                algorithmic_per_launch=(dom['flops'] if mfma_bound else dom['bytes']) / dom['count'])
    if dom['name'] == 'disc_dw_in' and world == 1 and B < 128:
        # what the launch carries besides the product SURVEY 8(d) prices (NOT in `achieved`): the next step's
        # generated rows copied into the other stacked-input buffer and the flat RMSprop of the other parameters
        roof['carried_bytes_not_counted'] = 8.0 * B * G + 20.0 * (disc.n_flat - disc.dims[0] * disc.ldw[0])
    tot = sum(v['ms'] for v in disc_prof.values())
    breakdown = {v['name']: round(v['ms'] / nprobe * 1e3, 1) for v in sorted(disc_prof.values(), key=lambda v: -v['ms'])}
    flops_iter = algorithmic_flops_per_iter(G, V, n_num, H, Z, B, nb_critic) if len(H) == 2 else None
    # what the launches of one iteration actually multiply (sum of their own 2MNK tags): less than the SURVEY 8(d) count
    # of the reference's algorithm, which whole_iter_* prices - the layer-1 forward of the interpolated rows comes from
    # the real and fake rows' by linearity, and from a batch of the hidden width on the penalty's input-gradient terms go
    # through the Gram matrix of W_1's gene rows (cgan.hip gram_ok)
    flops_exec = sum(v['flops'] for v in disc_prof.values()) / nprobe
    res = dict(value=Bg * steps / dt, ms_per_step=dt / steps * 1e3, roofline=roof, final_losses=final_losses,
               kernel_us_per_iter=breakdown, tagged_kernel_ms_per_iter=tot / nprobe,
               whole_iter_tflops=(flops_iter / (dt / steps) / 1e12) if flops_iter else None,
               whole_iter_mfma_frac=(flops_iter / (dt / steps) / 1e12 / MFMA_F32_PEAK_TF) if flops_iter else None,
               executed_tflops=flops_exec / (dt / steps) / 1e12, executed_mfma_frac=flops_exec / (dt / steps) / 1e12 / MFMA_F32_PEAK_TF,
               flops_note='whole_iter_* = SURVEY 8(d) flops of the reference algorithm / measured time; executed_* = the 2MNK the launches really multiply')
    lib.agx_prof_reset()
    del gen, disc, x
    torch.cuda.empty_cache()
    return res


def run_corr(world, rank, n_big, G=18154, n_eval=2287):
    """Correlation evaluation (BASELINE.json metric 'corr GFLOP/s'; configs[3] = 18154 x 100000).
    (1) gamma_coef at the GTEx evaluation shape (utils.py:863-873 as score_fn calls it every 5 epochs);
    (2) the X^T X contraction of the big matrix: this rank's contiguous share of the upper-triangular
        tile list (no collective in the contraction).  FLOP conventions: symmetric N*G*(G+1) (algorithmic)
        and full 2*N*G^2 (what the reference's np.dot executes)."""
    import torch
    from adversarial_gene_expression_amd import _dev, _lib, utils as U
    lib = _lib.load()
    out = {}
    g = torch.Generator(device='cuda').manual_seed(7)
    f = torch.randn(n_eval, 32, device='cuda', generator=g)
    xa = f @ torch.randn(32, G, device='cuda', generator=g) + 0.5 * torch.randn(n_eval, G, device='cuda', generator=g)
    xb = xa + 0.3 * torch.randn(n_eval, G, device='cuda', generator=g)
    U.gamma_coef(xa, xb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        gam = U.gamma_coef(xa, xb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out['gamma_coef_eval'] = dict(shape=[n_eval, G], ms=dt * 1e3, value=float(gam),
                                  tflops_sym=2 * n_eval * G * (G + 1.0) / dt / 1e12)
    del xa, xb, f
    torch.cuda.empty_cache()
    # big matrix, generated in row chunks to bound the temporary memory
    ld = _dev.r4(G)
    x = torch.empty((n_big, ld), device='cuda')
    for r0 in range(0, n_big, 8192):
        r1 = min(n_big, r0 + 8192)
        x[r0:r1] = torch.randn(r1 - r0, ld, device='cuda', generator=g)
    st = _dev.stream()
    mean, std = _dev.empty(G), _dev.empty(G)
    nws = int(lib.agx_colstats_ws_doubles(n_big, G))
    ws = _dev.empty(nws, dtype=torch.float64)
    xs = torch.empty_like(x)
    c = _dev.empty((G, ld))
    total = int(lib.agx_corr_num_tiles(G))
    per = (total + world - 1) // world
    tb, te = min(total, rank * per), min(total, (rank + 1) * per)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def once():
        ev[0].record()
        _lib.check(lib.agx_colstats(st, x.data_ptr(), ld, n_big, G, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), nws))
        _lib.check(lib.agx_standardize(st, x.data_ptr(), ld, n_big, G, mean.data_ptr(), std.data_ptr(), xs.data_ptr(), ld))
        ev[1].record()
        _lib.check(lib.agx_corr_xtx(st, xs.data_ptr(), ld, n_big, G, 1.0 / n_big, c.data_ptr(), ld, tb, te))
        ev[2].record()
    once()
    torch.cuda.synchronize()
    once()
    torch.cuda.synchronize()
    t_std, t_xtx = ev[0].elapsed_time(ev[1]) * 1e-3, ev[1].elapsed_time(ev[2]) * 1e-3
    spot = None
    if world == 1:   # accuracy of what was just timed: a 64 x 64 block of random gene pairs against float64 (utils.py:816-824)
        idx = torch.randperm(G, device='cuda', generator=g)[:64]
        sub = x[:, idx].double()
        sub = (sub - sub.mean(0)) / sub.std(0, unbiased=False)
        ref = sub.t() @ sub / n_big
        blk = c[:, :G][idx][:, idx]
        spot = dict(max_abs_err_64x64_vs_f64=float((blk.double() - ref).abs().max()),
                    max_abs_diag_minus_1=float((torch.diagonal(c[:, :G]) - 1).abs().max()),
                    symmetric=bool(torch.equal(blk, blk.t())))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([t_std, t_xtx], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_std, t_xtx = float(t[0]), float(t[1])
    sym = n_big * G * (G + 1.0)
    out['pearson_big'] = dict(shape=[n_big, G], n_gpus=world, xtx_ms=t_xtx * 1e3, standardize_ms=t_std * 1e3,
                              tflops_sym=sym / t_xtx / 1e12, tflops_full_convention=2.0 * n_big * G * G / t_xtx / 1e12,
                              mfma_frac=sym / t_xtx / 1e12 / MFMA_F32_PEAK_TF,
                              standardize_gbs=12.0 * n_big * G / t_std / 1e9,
                              tiles=[tb, te, total], accuracy=spot)
    return out


def corr_cpu_baseline(G=18154, n_eval=2287):
    """NumPy restatement of utils.py:808-824 (checked against the reference's goldens) at the GTEx
    evaluation shape on this host's cores."""
    from oracle import corr_oracle as CO
    rs = np.random.RandomState(0)
    x = rs.randn(n_eval, G).astype(np.float32)
    t0 = time.perf_counter()
    CO.pearson_correlation(x, x)
    dt = time.perf_counter() - t0
    return dict(value=2.0 * n_eval * G * G / dt / 1e9, unit='GFLOP/s (2NG^2 convention)', seconds=dt,
                cores=os.cpu_count(), kind='port', sample='one pearson_correlation(x, x) at N=%d, G=%d' % (n_eval, G))


def measured_traffic(kernel, B):
    """HBM bytes per launch of the dominant kernel.  NOT measured in this run: read from the rocprofv3 PMC
    passes (FETCH_SIZE x 2 + WRITE_SIZE per the guide's gfx950 correction, separate --pmc runs) that
    tools/collect_profiles.sh left in profiles/traffic.json for the committed tree.  (value, source)."""
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(p) as f:
            t = json.load(f)
        e = t.get('%s@B%d' % (kernel, B), {})
        v = e.get('hbm_bytes_per_launch')
        return v, ('profiles/traffic.json (static: rocprofv3 PMC of %s, not this run)' % e.get('profile', 'an earlier run')
                   if v is not None else None)
    except Exception:
        return None, None


def cpu_baseline(workload, B, budget_s=15.0, nb_critic=5):
    """torch-CPU fp32 restatement (oracle/cgan_oracle.py) of the same iteration on this host's cores."""
    import torch
    from oracle import cgan_oracle as O
    G, V, n_num, H, Z = WORKLOADS[workload]
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    gp = O.to_torch(O.init_params('gen', G, V, n_num, H, Z, seed=1), dtype=torch.float32)
    dp = O.to_torch(O.init_params('disc', G, V, n_num, H, seed=2), dtype=torch.float32)
    go, do = O.RMSprop(5e-4), O.RMSprop(5e-4)
    rs = np.random.RandomState(0)
    x = torch.tensor(rs.randn(B, G), dtype=torch.float32)
    cc = torch.tensor(np.stack([rs.randint(0, v, B) for v in V], 1))
    nc = torch.zeros(B, n_num)

    kd, kg = {}, {}
    if P_AUG.get(workload, 0):       # ecoli_gan.py:90: p_aug = 0.25, constants per trace (tf_utils.py:202-210, :240-247)
        pa = P_AUG[workload]
        kd = dict(aug_mask=torch.tensor(rs.binomial(1, pa, B), dtype=torch.float32),
                  aug_noise_fake=torch.tensor(rs.normal(0, 0.5, G), dtype=torch.float32),
                  aug_noise_real=torch.tensor(rs.normal(0, 0.5, (B, G)), dtype=torch.float32))
        kg = dict(aug_mask=torch.tensor(rs.binomial(1, pa, B), dtype=torch.float32),
                  aug_noise=torch.tensor(rs.normal(0, 0.5, (B, G)), dtype=torch.float32))

    def iteration():
        for _ in range(nb_critic):
            O.train_disc(x, torch.randn(B, Z), cc, nc, gp, dp, do, torch.rand(B, 1), **kd)
        O.train_gen(torch.randn(B, Z), cc, nc, gp, dp, go, **kg)

    iteration()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        iteration()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    return dict(value=B * n / el, unit='samples/s', cores=cores, kind='port',
                sample='%d iterations (5 critic + 1 generator steps, batch %d, %s shape) of the torch-CPU fp32 '
                       'restatement of tf_utils.py:186-258 in %.1f s; TensorFlow itself is unavailable offline'
                       % (n, B, workload, el))


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes (one rank per GPU) of this
    very command with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, BEFORE anything in this process has
    touched the GPU (children are spawned, nothing is exec'ed over an initialised process).  Rank 0's stdout
    (the one JSON line) passes through; the exit code is the first non-zero child's.  If one rank dies the
    others are ended by their exact PIDs (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, '-u', os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 128 - c
                print('bench.py: rank %d exited with %d; ending the other ranks' % (procs.index(p), c), file=sys.stderr)
                deadline = time.time() + 20
                while time.time() < deadline and any(q.poll() is None for q in live):
                    time.sleep(0.2)
                for q in live:
                    if q.poll() is None:
                        q.terminate()
                deadline = time.time() + 10
                while time.time() < deadline and any(q.poll() is None for q in live):
                    time.sleep(0.2)
                for q in live:
                    if q.poll() is None:
                        q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32, help='samples per GPU per iteration')
    ap.add_argument('--workload', default='gtex', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--secondary', dest='secondary', action='store_true', default=None)
    ap.add_argument('--no-secondary', dest='secondary', action='store_false')
    ap.add_argument('--corr-n', type=int, default=100000, help='samples of the big correlation matrix (0 = skip)')
    ap.add_argument('--no-extra', dest='extra', action='store_false', default=True,
                    help='skip the E. coli / wide batch-32 throughput entries')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (nothing has touched the GPU yet)
        sys.exit(self_launch(a.gpus))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('AGX_BENCH_LAUNCH_ONLY'):
        # launcher rehearsal without a GPU (tests/test_host_logic.py): each rank reports and leaves
        if rank == 0:
            print(json.dumps({'launch_only': True, 'world': world, 'master': os.environ.get('MASTER_ADDR')}))
        sys.exit(int(os.environ.get('AGX_BENCH_LAUNCH_RC_%d' % rank, '0')))
    import torch
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get('AGX_DIST_BACKEND', 'nccl')   # 'gloo' lets several ranks share one GPU (rehearsal only)
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    if a.gpus != world and rank == 0:
        print('bench.py: --gpus %d but WORLD_SIZE=%d: measuring %d rank(s)' % (a.gpus, world, world), file=sys.stderr)

    res = run_gpu(a.workload, a.batch, a.steps, a.warmup, world, rank)
    G, V, n_num, H, Z = WORKLOADS[a.workload]
    line = {
        'metric': 'cGAN train samples/sec at 18k genes', 'value': res['value'], 'unit': 'samples/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': res['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s cGAN train iteration (5 critic + 1 generator step, WGAN-GP, RMSprop), G=%d, '
                               'covariates V=%s, h=%s, z=%d, batch %d per GPU'
                               % ('GTEx-shaped' if a.workload == 'gtex' else a.workload, G, V, H, Z, a.batch),
                   'global_batch': a.batch * world, 'parallelism': 'dp%d' % world},
        'roofline': res['roofline'],
        'final_losses': res['final_losses'],
        'kernel_us_per_iter': res['kernel_us_per_iter'],
        'whole_iter_tflops': res['whole_iter_tflops'], 'whole_iter_mfma_frac': res['whole_iter_mfma_frac'],
        'executed_tflops': res['executed_tflops'], 'executed_mfma_frac': res['executed_mfma_frac'], 'flops_note': res['flops_note'],
    }
    # 512 samples per GPU at every N: at N = 8 this IS BASELINE.json configs[2] (batch 4096 data parallel)
    want_secondary = a.secondary if a.secondary is not None else (a.batch != 512)
    if want_secondary:
        try:
            s = run_gpu(a.workload, 512, max(10, a.steps // 8), max(3, a.warmup // 4), world, rank)
            line['secondary'] = {'workload': 'same network, 512 samples per GPU (BASELINE.json configs[2]: global batch %d '
                                             'over %d GPU(s))' % (512 * world, world),
                                 'value': s['value'], 'unit': 'samples/s', 'ms_per_step': s['ms_per_step'],
                                 'global_batch': 512 * world,
                                 'roofline': s['roofline'], 'kernel_us_per_iter': s['kernel_us_per_iter'],
                                 'whole_iter_tflops': s['whole_iter_tflops'],
                                 'whole_iter_mfma_frac': s['whole_iter_mfma_frac'], 'executed_tflops': s['executed_tflops'],
                                 'executed_mfma_frac': s['executed_mfma_frac'], 'final_losses': s['final_losses']}
        except Exception as e:                      # never lose the headline line to the secondary workload
            line['secondary'] = {'error': repr(e)[:200]}
        if world == 1 and 'error' not in line['secondary'] and not os.environ.get('AGX_NO_GRAM') and not os.environ.get('AGX_BENCH_NO_AB'):
            # the same 512-row workload with the DIRECT forms of the large-batch algebra (AGX_NO_GRAM=1: B x G input gradient
            # formed, generated rows materialised) in a child process - the library reads its knobs once -
            # so that the line carries its own A/B: what the algebra bought on this box
            try:
                import subprocess
                env = dict(os.environ, AGX_NO_GRAM='1')
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--gpus', '1', '--batch', '512', '--workload', a.workload,
                                    '--steps', str(max(10, a.steps // 8)), '--warmup', str(max(3, a.warmup // 4)), '--no-secondary',
                                    '--no-extra', '--no-cpu-baseline', '--corr-n', '0'], env=env, capture_output=True, text=True, timeout=240)
                dl = json.loads(r.stdout.strip().splitlines()[-1])
                line['secondary']['direct_forms'] = {'value': dl['value'], 'unit': 'samples/s', 'ms_per_step': dl['ms_per_step'],
                                                     'executed_mfma_frac': dl.get('executed_mfma_frac'),
                                                     'note': 'AGX_NO_GRAM=1, same workload, child process'}
            except Exception as e:
                line['secondary']['direct_forms'] = {'error': repr(e)[:200]}
        if world > 1:
            # ranks must stay in step: either every rank has the secondary line or none keeps it (a failure that is the
            # same on every rank - an argument check, a missing path - is agreed on here and skipped; one rank failing
            # INSIDE a collective cannot be helped from here)
            import torch
            import torch.distributed as dist
            bad = torch.tensor([1.0 if 'error' in line['secondary'] else 0.0], device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if float(bad.item()) > 0 and 'error' not in line['secondary']:
                line['secondary'] = {'error': 'another rank failed in the secondary workload'}
    if world == 1 and a.extra and a.workload == 'gtex' and a.batch == 32:
        # the other BASELINE.json configurations at batch 32 from the same tree (throughput only; their parity
        # cases are tests/test_gpu_fullsize.py): configs[0] E. coli (all genes / as shipped) and configs[4] wide
        extra = {}
        for wl in ('ecoli', 'ecoli_crp', 'wide'):
            try:
                r = run_gpu(wl, 32, max(20, a.steps // 4), max(5, a.warmup // 2), 1, 0, roofline=False)
                Gx, Vx, nx, Hx, Zx = WORKLOADS[wl]
                extra[wl] = dict(value=r['value'], unit='samples/s', ms_per_step=r['ms_per_step'], G=Gx, h=Hx, batch=32,
                                 p_aug=P_AUG.get(wl, 0), final_losses=r['final_losses'],
                                 whole_iter_tflops=algorithmic_flops_per_iter(Gx, Vx, nx, Hx, Zx, 32) / (r['ms_per_step'] * 1e-3) / 1e12)
            except Exception as e:
                extra[wl] = {'error': repr(e)[:200]}
        try:   # configs[4] in the MFMA-bound regime as well
            r = run_gpu('wide', 512, max(5, a.steps // 20), 2, 1, 0, roofline=False)
            Gx, Vx, nx, Hx, Zx = WORKLOADS['wide']
            fl = algorithmic_flops_per_iter(Gx, Vx, nx, Hx, Zx, 512)
            extra['wide_b512'] = dict(value=r['value'], unit='samples/s', ms_per_step=r['ms_per_step'], G=Gx, h=Hx, batch=512,
                                      final_losses=r['final_losses'], whole_iter_tflops=fl / (r['ms_per_step'] * 1e-3) / 1e12,
                                      whole_iter_mfma_frac=fl / (r['ms_per_step'] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF)
        except Exception as e:
            extra['wide_b512'] = {'error': repr(e)[:200]}
        line['extra'] = extra
    if a.corr_n > 0:
        try:
            corr = run_corr(world, rank, a.corr_n)
        except Exception as e:                      # never lose the train line to the secondary metric
            corr = {'error': repr(e)[:200]}
        line['corr'] = corr
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(a.workload, a.batch)
        if 'extra' in line:    # BASELINE.json configs[0]: the E. coli cGAN on the CPU path (as shipped, and all genes)
            for wl in ('ecoli_crp', 'ecoli'):
                if isinstance(line['extra'].get(wl), dict) and 'value' in line['extra'][wl]:
                    line['extra'][wl]['cpu_baseline'] = cpu_baseline(wl, 32, budget_s=5.0)
        line['corr_cpu_baseline'] = corr_cpu_baseline()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
