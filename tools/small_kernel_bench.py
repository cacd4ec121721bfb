"""Back-to-back GPU time of small launches (host launch gaps hidden behind a queued busy-wait).
Measurement tool only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarial_gene_expression_amd import _dev, _lib, ops  # noqa: E402

lib = _lib.load()
dev = _dev.device()


def timed(fn, n=200, spin_us=30000):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.agx_prof_spin(_dev.stream(), spin_us)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def gemm_case(M, N, K, ta=0, tb=0, cfg=-1, sk=0):
    A = torch.randn((K, _dev.r4(M)) if ta else (M, _dev.r4(K)), device=dev)
    B = torch.randn((N, _dev.r4(K)) if tb else (K, _dev.r4(N)), device=dev)
    out = _dev.empty((M, _dev.r4(N)))
    ws = _dev.empty(max(1, 128 * M * N))
    g = _lib.GemmDesc()
    g.A, g.B, g.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
    g.ta, g.tb, g.alpha = ta, tb, 1.0
    g.ws, g.ws_floats = ws.data_ptr(), ws.numel()
    g.force_cfg, g.force_splitk = cfg, sk
    st = _dev.stream()
    return timed(lambda: lib.agx_gemm_f32(st, g)), (A, B, out, ws)


def main():
    x = _dev.empty(4096)
    st = _dev.stream()
    print('rng 4096 floats           %7.2f us' % timed(lambda: lib.agx_rng_normal(st, x.data_ptr(), 4096, 1, 0)))
    for (M, N, K, ta, tb) in [(32, 256, 71, 0, 0), (32, 256, 256, 0, 0), (96, 256, 256, 0, 0), (96, 256, 256, 0, 1),
                              (256, 256, 96, 1, 0), (96, 256, 18161, 0, 0), (32, 18154, 256, 0, 0), (32, 18154, 256, 0, 1),
                              (32, 256, 18154, 0, 0), (18161, 256, 96, 1, 0)]:
        for cfg in (-1, 1, 3):
            for sk in ((0,) if K < 1000 else (0, 32, 64, 128)):
                t, keep = gemm_case(M, N, K, ta, tb, cfg, sk)
                print('gemm M=%5d N=%5d K=%5d ta=%d tb=%d cfg=%2d sk=%3d  %8.2f us' % (M, N, K, ta, tb, cfg, sk, t), flush=True)
                del keep

if __name__ == '__main__':
    main()
