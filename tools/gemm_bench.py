"""Micro-benchmark of agx_gemm_f32 on the layer shapes of the cGAN (and square probes).

    python tools/gemm_bench.py [--shapes gtex512|gtex32|square] [--iters N]

Prints per shape: tile config, split-K, mean time over N launches (torch events on the launch
stream), TFLOP/s and the fraction of the 157.3 TFLOP/s fp32-MFMA peak.  Measurement tool only."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarial_gene_expression_amd import _dev, ops  # noqa: E402


def shapes_for(name):
    G, H, Dd = 18154, 256, 18161
    if name == 'square':
        return [('sq4096', 4096, 4096, 4096, 0, 0), ('sq2048', 2048, 2048, 2048, 0, 0), ('sq8192', 8192, 8192, 8192, 0, 0)]
    if name == 'gram':   # the Gram form of the penalty's input-gradient terms
        return [('gram W^T W    TN', H, H, G, 1, 0), ('gram WT WT^T  NT', H, H, G, 0, 1), ('T = d1 M      NN', 512, H, H, 0, 0),
                ('S = d1^T f d1 TN', H, H, 512, 1, 0), ('dw_in K=1280  TN', Dd, H, 1280, 1, 0), ('dw_in K=1536  TN', Dd, H, 1536, 1, 0),
                ('disc_dx       NT', 512, G, H, 0, 1), ('gp2_in        NN', 512, H, G, 0, 0)]
    if name == 'ff':     # factored fake rows: headroom + real rows through the critic's first layer
        return [('hd+real 1032  NN', 1032, H, Dd, 0, 0), ('hd 520        NN', 520, H, Dd, 0, 0), ('real 512      NN', 512, H, Dd, 0, 0),
                ('old fwd 1024  NN', 1024, H, Dd, 0, 0), ('dw_in K=1032  TN', Dd, H, 1032, 1, 0)]
    B = 512 if name == 'gtex512' else 32
    return [
        ('gen_fwd_out   NN', B, G, H, 0, 0),
        ('disc_fwd_in   NN', 3 * B, H, Dd, 0, 0),
        ('disc_dx       NT', B, G, H, 0, 1),
        ('gp2_in        NN', B, H, G, 0, 0),
        ('disc_dw_in    TN', Dd, H, 3 * B, 1, 0),
        ('gen_dw_out    TN', H, G, B, 1, 0),
        ('gen_bwd_out   NT', B, H, G, 0, 1),
        ('hidden        NN', 3 * B, H, H, 0, 0),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='gtex512')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--cfg', type=int, default=-1)
    ap.add_argument('--splitk', type=int, default=0)
    a = ap.parse_args()
    dev = _dev.device()
    for name, M, N, K, ta, tb in shapes_for(a.shapes):
        A = torch.randn((K, _dev.r4(M)) if ta else (M, _dev.r4(K)), device=dev)
        Bm = torch.randn((N, _dev.r4(K)) if tb else (K, _dev.r4(N)), device=dev)
        out = _dev.empty((M, _dev.r4(N)))
        for _ in range(3):
            ops.gemm(A, Bm, M, N, K, ta=ta, tb=tb, out=out, force_cfg=a.cfg, force_splitk=a.splitk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            ops.gemm(A, Bm, M, N, K, ta=ta, tb=tb, out=out, force_cfg=a.cfg, force_splitk=a.splitk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
        gbs = 4.0 * (M * K + K * N + M * N) / (us * 1e-6) / 1e9
        print('%-18s M=%6d N=%6d K=%6d  %9.1f us  %7.2f TFLOP/s (%.1f%% of 157.3)  %8.1f GB/s algorithmic'
              % (name, M, N, K, us, tf, 100 * tf / 157.3, gbs), flush=True)


if __name__ == '__main__':
    main()
