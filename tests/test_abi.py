"""CPU checks of the C-ABI: the library loads, exports every symbol include/agx.h declares, the
Python binding table covers exactly those symbols, and argument validation runs without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'agx.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(agx_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from adversarial_gene_expression_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.agx_version() >= 100


def test_struct_sizes_match_header_layout():
    """ctypes mirrors of the by-value/by-pointer structs (natural alignment, same field order)."""
    from adversarial_gene_expression_amd import _lib
    assert ctypes.sizeof(_lib.EmbedDesc) == 8 + 64 + 64 + 128 - 4 + 4  # int32 + pad, 2 x int32[16], int64[16]
    assert ctypes.sizeof(_lib.GemmDesc) % 8 == 0 and ctypes.sizeof(_lib.StepArgs) % 8 == 0
    # every mirror against the size the library itself was compiled with
    lib = _lib.load()
    mirrors = [_lib.GemmDesc, _lib.EmbedDesc, _lib.SmallGrads, _lib.RngSeg, _lib.MlpDesc, _lib.CganConfig,
               _lib.StepArgs, _lib.IterArgs, _lib.DpArgs]
    for i, m in enumerate(mirrors):
        assert lib.agx_sizeof(i) == ctypes.sizeof(m), (m.__name__, lib.agx_sizeof(i), ctypes.sizeof(m))
    assert lib.agx_sizeof(len(mirrors)) == -1


def test_host_side_validation_and_queries_without_gpu():
    from adversarial_gene_expression_amd import _lib
    lib = _lib.load()
    assert lib.agx_corr_num_tiles(18154) == 142 * 143 // 2
    assert lib.agx_corr_num_tiles(128) == 1 and lib.agx_corr_num_tiles(129) == 3
    assert lib.agx_gemm_ws_floats(4096, 18154, 256) == 0          # enough tiles: no split-K
    assert lib.agx_gemm_ws_floats(96, 256, 18161) > 0             # few tiles, deep K: split-K scratch
    g = _lib.GemmDesc()
    assert lib.agx_gemm_f32(None, g) == 0                          # M = N = 0 is a no-op
    g.M = g.N = g.K = 4
    assert lib.agx_gemm_f32(None, g) == -1                         # null operands rejected before any launch
    assert b'null operand' in lib.agx_last_error()
    cfg = _lib.CganConfig()
    h = ctypes.c_void_p()
    assert lib.agx_cgan_create(cfg, ctypes.byref(h)) == -1         # empty config rejected
    assert lib.agx_cgan_destroy(None) == 0


def test_cgan_descriptor_roundtrip_without_gpu():
    """A GTEx-shaped descriptor validates and reports a sane workspace (host memory only)."""
    from adversarial_gene_expression_amd import _lib
    lib = _lib.load()

    def mlp(lead, out, H=(256, 256), V=(15, 3), n_num=1):
        e = [int(v ** 0.5) + 1 for v in V]
        dims = [lead + n_num + sum(e)] + list(H) + [out]
        d = _lib.MlpDesc()
        off = 0
        a64 = lambda v: (v + 63) & ~63
        d.emb.n_cat = len(V)
        for i, (v, k) in enumerate(zip(V, e)):
            d.emb.vocab[i], d.emb.edim[i], d.emb.table_off[i] = v, k, off
            off = a64(off + v * k)
        d.n_layers = len(dims) - 1
        for i, v in enumerate(dims):
            d.dims[i] = v
        for l in range(d.n_layers):
            n = dims[l + 1]
            d.ldw[l] = 1 if n == 1 else (n + 3) & ~3
            d.w_off[l] = off
            off = a64(off + dims[l] * d.ldw[l])
            d.b_off[l] = off
            off = a64(off + n)
        d.n_num, d.n_flat = n_num, off
        return d

    cfg = _lib.CganConfig()
    cfg.x_dim, cfg.z_dim, cfg.max_batch = 18154, 64, 512
    cfg.gen, cfg.disc = mlp(64, 18154), mlp(18154, 1)
    h = ctypes.c_void_p()
    assert lib.agx_cgan_create(cfg, ctypes.byref(h)) == 0, lib.agx_last_error()
    nbytes = lib.agx_cgan_workspace_bytes(h)
    assert 3 * 512 * 18164 * 4 < nbytes < 2 << 30
    assert lib.agx_cgan_gen_forward(h, None, 1, None, None, 8, 1, 18154) == -4   # not bound yet
    assert lib.agx_cgan_destroy(h) == 0


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from adversarial_gene_expression_amd import AgxError, utils
    import numpy as np
    with pytest.raises(AgxError):
        utils.pearson_correlation(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32))
