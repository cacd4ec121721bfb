"""Thin op-level wrappers over the C-ABI (one call = one libagx entry point).

Used by the model ``__call__`` path and by the parity tests; the training step
itself is sequenced in C++ (csrc/cgan.hip) to keep Python out of the hot loop.
All arguments and results are device tensors (torch used as allocations only).
"""
import torch

from . import _dev, _lib
from ._dev import ptr


def gemm(A, B, M, N, K, *, ta=0, tb=0, alpha=1.0, bias=None, addrow=None, act=0, slope=0.0, mask=None,
         out=None, accumulate=False, rowscale=None, rowsq=None, force_cfg=-1, force_splitk=0, use_ws=True,
         interp=None, col_scale=None, col_shift=None, sk_flags=None):
    """out[M, N] = epilogue(alpha * op(A) @ op(B)); leading dimensions are the row strides.
    sk_flags: zero-initialised int32 device tensor of agx_gemm_sk_flag_words() words - allows the stream-K form."""
    lib = _lib.load()
    if out is None:
        out = _dev.empty((M, _dev.r4(N)))
    g = _lib.GemmDesc()
    g.A, g.B, g.C = ptr(A), ptr(B), ptr(out)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = A.stride(0), B.stride(0), out.stride(0)
    g.ta, g.tb = ta, tb
    g.alpha = alpha
    g.bias = ptr(bias)
    if addrow is not None:
        g.addrow, g.addmod, g.ldadd = ptr(addrow), addrow.shape[0], addrow.stride(0)
    g.act, g.slope = act, slope
    if mask is not None:
        g.mask, g.ldmask = ptr(mask), mask.stride(0)
    g.accumulate = int(accumulate)
    g.rowscale = ptr(rowscale)
    g.col_scale, g.col_shift = ptr(col_scale), ptr(col_shift)
    if rowsq is not None:
        g.rowsq, g.rowsq_ld = ptr(rowsq), rowsq.stride(0)
    ws = None
    if use_ws:
        wsf = max(int(lib.agx_gemm_ws_floats(M, N, K)), (force_splitk if force_splitk > 0 else 0) * M * N,
                  M * N if interp is not None else 0)
        ws = _dev.empty(max(wsf, 1))
        g.ws, g.ws_floats = ptr(ws), wsf
    g.force_cfg, g.force_splitk = force_cfg, force_splitk
    if sk_flags is not None:
        g.sk_flags, g.sk_flag_words = ptr(sk_flags), sk_flags.numel()
    if interp is not None:   # (rows, alpha): A = [real; fake], out gets the interpolated block too
        g.interp_rows, g.interp_alpha = interp[0], ptr(interp[1])
    _lib.check(lib.agx_gemm_f32(_dev.stream(), g), 'agx_gemm_f32')
    return out


def dense_bn(x, W, b, M, N, K, *, gamma=None, beta=None, eps=1e-3, slope=0.0, act=1, running=None):
    """Dense + batch-norm + (leaky-)ReLU, the optional stage BASELINE.json's north_star names (the reference has
    no batch-norm, tf_utils.py:85-88: nothing on the parity path calls this).
    running=(mean, var): inference form - statistics folded into a per-feature scale / shift applied in the
    GEMM epilogue, ONE launch.  running=None: batch statistics (biased variance over the M rows, as Keras
    BatchNormalization in training mode): GEMM with bias -> column statistics -> folded affine + activation in
    place.  Returns (y, mean, var) with the statistics used."""
    lib = _lib.load()
    scale, shift = _dev.empty(N), _dev.empty(N)
    if running is not None:
        mean, var = running
        _lib.check(lib.agx_bn_fold(_dev.stream(), ptr(mean), ptr(var), 0, ptr(gamma), ptr(beta), eps, ptr(scale),
                                   ptr(shift), N), 'agx_bn_fold')
        return gemm(x, W, M, N, K, bias=b, act=act, slope=slope, col_scale=scale, col_shift=shift), mean, var
    y = gemm(x, W, M, N, K, bias=b)
    mean, std = _dev.empty(N), _dev.empty(N)
    nws = int(lib.agx_colstats_ws_doubles(M, N))
    ws = _dev.empty(nws, dtype=torch.float64)
    _lib.check(lib.agx_colstats(_dev.stream(), ptr(y), y.stride(0), M, N, ptr(mean), ptr(std), ptr(ws), nws), 'agx_colstats')
    _lib.check(lib.agx_bn_fold(_dev.stream(), ptr(mean), ptr(std), 1, ptr(gamma), ptr(beta), eps, ptr(scale), ptr(shift), N),
               'agx_bn_fold')
    _lib.check(lib.agx_affine_act(_dev.stream(), ptr(y), y.stride(0), M, N, ptr(scale), ptr(shift), act, slope),
               'agx_affine_act')
    return y, mean, std * std


def colsum(A, r0, r1, H, out=None, accumulate=False):
    lib = _lib.load()
    if out is None:
        out = _dev.zeros(H)
    _lib.check(lib.agx_colsum(_dev.stream(), ptr(A), A.stride(0), r0, r1, H, ptr(out), int(accumulate)), 'agx_colsum')
    return out


def rmsprop(w, g, rms, lr, rho=0.9, eps=1e-7, grad_scale=1.0):
    lib = _lib.load()
    _lib.check(lib.agx_opt_rmsprop(_dev.stream(), ptr(w), ptr(g), ptr(rms), w.numel(), lr, rho, eps, grad_scale),
               'agx_opt_rmsprop')


def adam(w, g, m, v, lr_t, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0):
    lib = _lib.load()
    _lib.check(lib.agx_opt_adam(_dev.stream(), ptr(w), ptr(g), ptr(m), ptr(v), w.numel(), lr_t, b1, b2, eps,
                                grad_scale), 'agx_opt_adam')


def rng(n, seed, offset=0, normal=True):
    lib = _lib.load()
    out = _dev.empty(n)
    fn = lib.agx_rng_normal if normal else lib.agx_rng_uniform
    _lib.check(fn(_dev.stream(), ptr(out), n, seed, offset), 'agx_rng')
    return out


def synchronize():
    torch.cuda.synchronize()
