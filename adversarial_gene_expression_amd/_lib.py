"""ctypes binding of libagx.so (declared in include/agx.h).

The HIP library is the product: if it is missing or fails to load this module
raises - there is no CPU fallback anywhere in the package.  PyTorch-ROCm is
used only to own device memory and streams (and torch.distributed for RCCL).
"""
import ctypes as C
import os

AGX_MAX_LAYERS = 8
AGX_MAX_CAT = 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libagx.so')


class AgxError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p),
        ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
        ('lda', C.c_int32), ('ldb', C.c_int32), ('ldc', C.c_int32),
        ('ta', C.c_int32), ('tb', C.c_int32),
        ('alpha', C.c_float),
        ('bias', C.c_void_p),
        ('addrow', C.c_void_p),
        ('addmod', C.c_int32), ('ldadd', C.c_int32),
        ('act', C.c_int32),
        ('slope', C.c_float),
        ('mask', C.c_void_p),
        ('ldmask', C.c_int32),
        ('accumulate', C.c_int32),
        ('rowscale', C.c_void_p),
        ('rowsq', C.c_void_p),
        ('rowsq_ld', C.c_int32),
        ('ws', C.c_void_p),
        ('ws_floats', C.c_int64),
        ('force_cfg', C.c_int32),
        ('force_splitk', C.c_int32),
        ('prof_id', C.c_int32),
        ('prof_tagged', C.c_int32),
        ('interp_rows', C.c_int32),
        ('interp_alpha', C.c_void_p),
        ('col_scale', C.c_void_p),
        ('col_shift', C.c_void_p),
        ('sk_flags', C.c_void_p),
        ('sk_flag_words', C.c_int32),
    ]


class EmbedDesc(C.Structure):
    _fields_ = [
        ('n_cat', C.c_int32),
        ('vocab', C.c_int32 * AGX_MAX_CAT),
        ('edim', C.c_int32 * AGX_MAX_CAT),
        ('table_off', C.c_int64 * AGX_MAX_CAT),
    ]


class MlpDesc(C.Structure):
    _fields_ = [
        ('n_layers', C.c_int32),
        ('dims', C.c_int32 * (AGX_MAX_LAYERS + 1)),
        ('ldw', C.c_int32 * AGX_MAX_LAYERS),
        ('w_off', C.c_int64 * AGX_MAX_LAYERS),
        ('b_off', C.c_int64 * AGX_MAX_LAYERS),
        ('emb', EmbedDesc),
        ('n_num', C.c_int32),
        ('n_flat', C.c_int64),
        ('bn', C.c_int32), ('bn_eps', C.c_float), ('bn_momentum', C.c_float),
        ('bn_gamma_off', C.c_int64 * AGX_MAX_LAYERS), ('bn_beta_off', C.c_int64 * AGX_MAX_LAYERS),
        ('bn_mean_off', C.c_int64 * AGX_MAX_LAYERS), ('bn_var_off', C.c_int64 * AGX_MAX_LAYERS),
    ]


class CganConfig(C.Structure):
    _fields_ = [
        ('x_dim', C.c_int32), ('z_dim', C.c_int32),
        ('gen', MlpDesc), ('disc', MlpDesc),
        ('slope', C.c_float),
        ('max_batch', C.c_int32),
        ('max_steps', C.c_int32),
    ]


class StepArgs(C.Structure):
    _fields_ = [
        ('x', C.c_void_p), ('ldx', C.c_int32),
        ('z', C.c_void_p), ('cat', C.c_void_p), ('num', C.c_void_p), ('alpha', C.c_void_p),
        ('B', C.c_int32), ('B_global', C.c_int32),
        ('gp_weight', C.c_float),
        ('aug_mask', C.c_void_p), ('aug_noise_fake', C.c_void_p), ('aug_noise_real', C.c_void_p),
        ('losses', C.c_void_p),
    ]


class IterArgs(C.Structure):
    _fields_ = [
        ('B', C.c_int32), ('nb_critic', C.c_int32),
        ('gp_weight', C.c_float),
        ('lr_disc', C.c_float), ('lr_gen', C.c_float), ('rho', C.c_float), ('eps', C.c_float),
        ('disc_rms', C.c_void_p), ('gen_rms', C.c_void_p),
        ('seed', C.c_uint64),
        ('aug_mask_disc', C.c_void_p), ('aug_noise_fake', C.c_void_p), ('aug_noise_real', C.c_void_p),
        ('aug_mask_gen', C.c_void_p), ('aug_noise_gen', C.c_void_p),
        ('B_global', C.c_int32), ('row_lo', C.c_int32), ('split', C.c_int32),
        ('disc_losses', C.c_void_p), ('gen_losses', C.c_void_p),
        ('opt_kind', C.c_int32), ('beta1', C.c_float), ('beta2', C.c_float),
        ('disc_m', C.c_void_p), ('disc_v', C.c_void_p), ('gen_m', C.c_void_p), ('gen_v', C.c_void_p),
        ('adam_lr_t_host', C.c_void_p),
    ]


AGX_MAX_DP_PIECES = 16


class DpNet(C.Structure):
    _fields_ = [
        ('n_pieces', C.c_int32),
        ('rows', C.c_int32 * (AGX_MAX_DP_PIECES + 1)),
        ('start', C.c_int64 * AGX_MAX_DP_PIECES),
        ('per', C.c_int64 * AGX_MAX_DP_PIECES),
        ('slot0', C.c_void_p), ('slot1', C.c_void_p),
        ('lr', C.c_float),
    ]


class DpArgs(C.Structure):
    _fields_ = [
        ('disc', DpNet), ('gen', DpNet),
        ('opt_kind', C.c_int32),
        ('rho', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
        ('lr_t_host', C.c_void_p),
    ]


class RngSeg(C.Structure):
    _fields_ = [('out', C.c_void_p), ('n', C.c_int64), ('sub_offset', C.c_uint64), ('normal', C.c_int32),
                ('skip', C.c_int64)]


class SmallGrads(C.Structure):
    _fields_ = [
        ('n_colsum', C.c_int32),
        ('cs_A', C.c_void_p * AGX_MAX_LAYERS),
        ('cs_lda', C.c_int32 * AGX_MAX_LAYERS), ('cs_r0', C.c_int32 * AGX_MAX_LAYERS),
        ('cs_r1', C.c_int32 * AGX_MAX_LAYERS), ('cs_H', C.c_int32 * AGX_MAX_LAYERS),
        ('cs_out', C.c_void_p * AGX_MAX_LAYERS),
        ('w_A', C.c_void_p),
        ('w_lda', C.c_int32), ('w_R', C.c_int32), ('w_H', C.c_int32), ('w_n0', C.c_int32), ('w_n1', C.c_int32),
        ('w_c0', C.c_float), ('w_c1', C.c_float), ('w_c2', C.c_float),
        ('w_out', C.c_void_p), ('w_bias_out', C.c_void_p),
        ('emb_grad', C.c_void_p),
        ('emb', EmbedDesc),
        ('emb_D', C.c_void_p), ('emb_W', C.c_void_p),
        ('emb_cat', C.c_void_p),
        ('emb_ldd', C.c_int32), ('emb_H', C.c_int32), ('emb_ldw', C.c_int32), ('emb_col0', C.c_int32),
        ('emb_ld_cat', C.c_int32), ('emb_B', C.c_int32), ('emb_rows', C.c_int32),
        ('l_losses', C.c_void_p),
        ('l_out', C.c_void_p), ('l_slopes', C.c_void_p),
        ('l_n0', C.c_int32), ('l_n1', C.c_int32), ('l_ns', C.c_int32),
        ('l_inv_b', C.c_float), ('l_gp_weight', C.c_float), ('l_fake_sign', C.c_float),
        ('l_out_parts', C.c_int32), ('l_out_ld', C.c_int32),
    ]


_vp, _i32, _i64, _f32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> (restype, argtypes); every symbol include/agx.h declares
SIGNATURES = {
    'agx_version': (C.c_int, []),
    'agx_last_error': (C.c_char_p, []),
    'agx_init': (C.c_int, [C.c_int]),
    'agx_prof_enable': (C.c_int, [_i32]),
    'agx_prof_reset': (C.c_int, []),
    'agx_prof_read': (C.c_int, [_i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'agx_prof_name': (C.c_char_p, [_i32]),
    'agx_prof_spin': (C.c_int, [_vp, _i32]),
    'agx_prof_empty_bracket_us': (C.c_int, [_vp, _i32, C.POINTER(C.c_double)]),
    'agx_prof_enabled': (C.c_int, []),
    'agx_prof_clock': (C.c_int, [_vp, _i32, _i32, _vp]),
    'agx_prof_occupancy': (C.c_int, [_i32, C.POINTER(_i32), C.POINTER(_i32)]),
    'agx_sizeof': (_i64, [_i32]),
    'agx_gemm_f32': (C.c_int, [_vp, C.POINTER(GemmDesc)]),
    'agx_gemm_ws_floats': (_i64, [_i32, _i32, _i32]),
    'agx_gemm_sk_flag_words': (_i64, []),
    'agx_bn_fold': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _f32, _vp, _vp, _i32]),
    'agx_affine_act': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _f32]),
    'agx_bn_backward': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp]),
    'agx_bn_moving_update': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i32]),
    'agx_embed_concat_fwd': (C.c_int, [_vp, C.POINTER(EmbedDesc), _vp, _vp, _i32, _i32, _vp, _i32, _i32,
                                       _vp, _i32, _i32, _i32, _vp, _i32]),
    'agx_embed_bwd': (C.c_int, [_vp, C.POINTER(EmbedDesc), _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32]),
    'agx_gp_interp': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32]),
    'agx_aug_add': (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32]),
    'agx_rowdot': (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32]),
    'agx_delta_init': (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32]),
    'agx_critic_head': (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32]),
    'agx_gp_finalize': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _f32]),
    'agx_gp_rownorm_scale': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _f32]),
    'agx_wgan_loss': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _f32, _f32, _f32, _vp]),
    'agx_colsum': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32]),
    'agx_colsum_weighted': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp, _vp]),
    'agx_opt_rmsprop': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32]),
    'agx_opt_adam': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32]),
    'agx_rng_normal': (C.c_int, [_vp, _vp, _i64, _u64, _u64]),
    'agx_rng_uniform': (C.c_int, [_vp, _vp, _i64, _u64, _u64]),
    'agx_rng_at': (C.c_int, [_vp, _vp, _i64, _u64, _vp, _u64, _i32]),
    'agx_set_u64': (C.c_int, [_vp, _vp, _u64]),
    'agx_rng_multi': (C.c_int, [_vp, _u64, _vp, _i32, C.POINTER(RngSeg)]),
    'agx_stack_rows': (C.c_int, [_vp, C.POINTER(EmbedDesc), _vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32,
                                 _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32]),
    'agx_small_grads': (C.c_int, [_vp, C.POINTER(SmallGrads)]),
    'agx_colstats_ws_doubles': (_i64, [_i32, _i32]),
    'agx_colstats': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64]),
    'agx_colstats_noalloc': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'agx_standardize': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32]),
    'agx_corr_xtx': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _i32, _i64, _i64]),
    'agx_corr_num_tiles': (_i64, [_i32]),
    'agx_triu_pack': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'agx_triu_pack_f64': (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'agx_sums_ws_doubles': (_i64, []),
    'agx_pearson_sums': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    'agx_gamma_sums': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    'agx_gamma_sums_tiles': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i64, _i64, _vp, _vp]),
    'agx_colstats_f64': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64]),
    'agx_standardize_f64': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32]),
    'agx_corr_xty_f64': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, C.c_double, _vp, _i32]),
    'agx_pearson_sums_f64': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    'agx_gamma_sums_f64': (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    'agx_cgan_create': (C.c_int, [C.POINTER(CganConfig), C.POINTER(_vp)]),
    'agx_cgan_destroy': (C.c_int, [_vp]),
    'agx_cgan_workspace_bytes': (_i64, [_vp]),
    'agx_cgan_bind': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    'agx_cgan_gen_forward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32]),
    'agx_cgan_disc_forward': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    'agx_cgan_disc_grads': (C.c_int, [_vp, _vp, C.POINTER(StepArgs)]),
    'agx_cgan_gen_grads': (C.c_int, [_vp, _vp, C.POINTER(StepArgs)]),
    'agx_cgan_iter_capture': (C.c_int, [_vp, _vp, C.POINTER(IterArgs), C.POINTER(_vp)]),
    'agx_cgan_iter_launch': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _u64, _vp]),
    'agx_cgan_iter_stage': (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _u64]),
    'agx_cgan_iter_num_segs': (_i32, [_vp]),
    'agx_cgan_iter_launch_seg': (C.c_int, [_vp, _vp, _i32]),
    'agx_cgan_iter_rng_draws': (_i64, [_vp]),
    'agx_cgan_iter_set_lr': (C.c_int, [_vp, C.POINTER(C.c_float), _i32]),
    'agx_cgan_iter_eager': (C.c_int, [_vp, _vp, C.POINTER(IterArgs), _vp, _i32, _vp, _vp, _u64, _vp, C.POINTER(_i64)]),
    'agx_cgan_iter_destroy': (C.c_int, [_vp]),
    'agx_cgan_dp_iteration': (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(DpArgs), _vp, _i32, _vp, _vp, _u64]),
    'agx_comm_unique_id': (C.c_int, [_vp]),
    'agx_comm_init': (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    'agx_comm_destroy': (C.c_int, [_vp]),
    'agx_comm_size': (_i32, [_vp]),
    'agx_comm_rank': (_i32, [_vp]),
    'agx_comm_allreduce_sum': (C.c_int, [_vp, _vp, _vp, _i64, _i32]),
    'agx_comm_reduce_scatter_sum': (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    'agx_comm_all_gather': (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
}

_lib = None


def load():
    """Load libagx.so once; raise AgxError (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AgxError('libagx.so not found at %s - build it with `python -c "import __graft_entry__ as g; '
                       'g.build()"` or `make -C adversarial_gene_expression_amd/csrc`; there is no CPU '
                       'fallback' % LIB_PATH)
    # PyTorch ships its own HIP runtime (torch/lib/libamdhip64.so).  Loading libagx.so FIRST would pull in the
    # system one (/opt/rocm) beside it, and whichever of the two runtimes opens the device second then reports
    # "no ROCm-capable device" (measured on the MI355X box).  With torch imported first libagx binds to the
    # runtime already in the process: one runtime, shared streams and allocations.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().agx_last_error()
        raise AgxError('%s failed (code %d): %s' % (what or 'libagx call', rc, (msg or b'').decode()))


_initialised = set()


def init_device(device_index):
    """agx_init on first use of a device (verifies gfx950, caches the CU count)."""
    if device_index not in _initialised:
        check(load().agx_init(int(device_index)), 'agx_init')
        _initialised.add(device_index)
