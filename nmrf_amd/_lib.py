"""ctypes binding of libnmrf_hip.so (the C ABI declared in include/nmrf_hip.h).

The library is loaded lazily and EXPLICITLY: if it is missing the product raises
-- there is no CPU or eager-PyTorch fallback for the hot path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnmrf_hip.so")      # (tools may point this at another build BEFORE the first load(): bench.py --lib)
ABI_VERSION = 27

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float

# name -> argtypes (return type is always int unless listed in _RESTYPE)
PROTOTYPES = {
    "nmrf_abi_version": [],
    "nmrf_cost_volume_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_dpn_filter_softmax_f32": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P],
    "nmrf_nms_topk_f32": [_P, _L, _I, _I, _F, _I, _P, _P],
    "nmrf_seed_features_f32": [_P, _P, _L, _I, _I, _I, _F, _P, _P, _I, _P],
    "nmrf_seed_select_f32": [_P, _P, _L, _I, _I, _I, _F, _I, _F, _P, _P, _P, _P, _I, _P],
    "nmrf_fourier_embed_f32": [_P, _L, _F, _P, _I, _P, _P],
    "nmrf_mlp_chain_f32": [_I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _L, _P, _I, _I, _P, _P, _I, _I, _P, _P],
    "nmrf_nmp_block16_pair_f32": [_P, _P, _P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _I, _I, _P, _P, _P, _P, _F, _P, _I, _I, _P, _I, _L,
                                  _P, _P, _P, _P, _P, _I, _P, _P],
    "nmrf_heads_wta_f32": [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nmrf_refine_head_epilogue_f32": [_P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "nmrf_ln_concat_f32": [_P, _P, _P, _F, _P, _I, _I, _L, _I, _P, _I, _P],
    "nmrf_add_ln_concat_f32": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _L, _I, _P, _I, _P],
    "nmrf_stripe_attn_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "nmrf_warp_corr_concat_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    "nmrf_warp_corr_concat_fourier_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _F, _P, _I, _P, _P],
    "nmrf_self_attn_f32": [_P, _L, _I, _I, _I, _P, _P],
    "nmrf_window_attn_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "nmrf_window_table_pack_f32": [_P, _I, _I, _P, _P],
    "nmrf_window_attn6_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_linear_smalln_f32": [_P, _P, _P, _L, _I, _I, _I, _P, _P],
    "nmrf_superpixel_downsample_f32": [_P, _P, _I, _I, _I, _I, _P, _P],
    "nmrf_wta_median_f32": [_P, _P, _P, _I, _I, _I, _I, _P, _P],
    "nmrf_refine_epilogue_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "nmrf_instance_norm_f32": [_P, _P, _L, _L, _F, _I, _I, _P, _P, _P],
    "nmrf_msda_forward_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_msda_forward_f64": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_msda_backward_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nmrf_msda_backward_f64": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nmrf_pack_split_weight_f32": [_P, _I, _I, _I, _F, _P, _P],
    "nmrf_nmp_block16_f32": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _I, _I, _P, _I, _I, _I, _L, _P, _P, _P, _P, _P, _I, _P, _P],
    "nmrf_pack_split_weight16_f32": [_P, _I, _I, _I, _F, _P, _P],
    "nmrf_selftest_mfma16x16_f16split": [_P, _P, _I, _P, _P],
    "nmrf_instance_stats_f32": [_P, _L, _L, _P, _P],
    "nmrf_instance_apply_f32": [_P, _P, _P, _P, _I, _L, _L, _F, _I, _I, _P, _P],
    "nmrf_conv_split_f32": [_P, _I, _I, _I, _I, _P, _I, _F, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P],
    "nmrf_prep_images_s2d_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_prep_images_s2d_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_prep_images_u8": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_range_scan_f32": [_P, _L, _P, _P],
    "nmrf_nmp_block16_clock_records": [_P, _I],
    "nmrf_build_stamp": [],
    "nmrf_host_copy_nt": [_P, _P, ctypes.c_size_t],
    "nmrf_host_read_evict": [_P, _P, ctypes.c_size_t],
    "nmrf_conv3x3_split_f32": [_P, _I, _I, _I, _I, _P, _I, _F, _P, _I, _I, _F, _I, _P, _P, _P],
    "nmrf_conv1x1_in_relu_f32": [_P, _I, _I, _L, _I, _I, _P, _I, _F, _P, _I, _F, _P, _I, _P, _I, _P, _P],
    "nmrf_conv1x1_f32": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _F, _P, _I, _F, _P, _I, _P, _I, _P, _P],
    "nmrf_prep_images_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_bias_avgpool2_f32": [_P, _P, _L, _I, _I, _I, _P, _P, _P],
    "nmrf_selftest_mfma_f32": [_P, _P, _I, _P, _P],
    "nmrf_selftest_mfma_f16split": [_P, _P, _I, _I, _P, _P],
    "nmrf_selftest_lds_dma": [_P, _P, _I, _P],
    "nmrf_gemm_split_f32": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _P, _I, _I, _L, _P, _P, _P],
    "nmrf_sum_partials_f32": [_P, _I, _L, _L, _P, _P],
    "nmrf_sum_partials_grouped_f32": [_P, _I, _L, _L, _I, _P, _P],
    "nmrf_sum_partials_tree_f32": [_P, _I, _L, _L, _P, _P],
    "nmrf_absmax_f32": [_P, _L, _P, _P],
    "nmrf_from_kv16_f32": [_P, _L, _P, _P],
    "nmrf_colsum_partials_f32": [_P, _L, _I, _I, _P, _P],
    "nmrf_bias_act_f32": [_P, _P, _L, _I, _I, _P, _P, _P],
    "nmrf_act_bwd_f32": [_P, _P, _L, _I, _P, _P],
    "nmrf_layernorm_f32": [_P, _P, _P, _L, _I, _F, _P, _P],
    "nmrf_layernorm_bwd_f32": [_P, _P, _P, _L, _I, _F, _I, _P, _P, _P, _P],
    "nmrf_window_attn_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "nmrf_self_attn_bwd_f32": [_P, _P, _L, _I, _I, _I, _P, _P],
    "nmrf_stripe_attn_bwd_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "nmrf_unfold5_f32": [_P, _L, _I, _I, _I, _P, _P],
    "nmrf_fold5_f32": [_P, _L, _I, _I, _P, _P],
    "nmrf_softmax_bwd_f32": [_P, _P, _L, _I, _P, _P],
    "nmrf_cost_volume_bwd_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "nmrf_seed_taps_bwd_f32": [_P, _P, _L, _I, _I, _I, _I, _P, _P],
    "nmrf_warp_corr_concat_bwd_f32": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
}

# exported only by libnmrf_hip_debug.so (include/nmrf_hip_debug.h): reference kernels for A/B runs, never launched by the product
DEBUG_PROTOTYPES = {
    "nmrf_token_linear_f32": [_P, _P, _P, _P, _P, _F, _P, _I, _I, _P, _P, _P, _I, _L, _I, _I, _I, _P, _P],
    "nmrf_pack_linear_weight_f32": [_P, _I, _I, _P, _P],
    "nmrf_conv3x3_wino_f32": [_P, _P, _I, _I, _I, _I, _I, _P, _P],
    "nmrf_wino_pack_filter_f32": [_P, _I, _I, _P, _P],
    "nmrf_nmp_block_f32": [_P, _P, _P, _I, _P, _P, _P, _F, _P, _P, _P, _P, _F, _P, _I, _I, _P, _I, _I, _I, _L, _P, _P, _P, _P, _P, _P, _P],
}
DEBUG_LIB_PATH = os.path.join(_HERE, "lib", "libnmrf_hip_debug.so")

_lib = None
_dbg = None


class NmrfHipError(RuntimeError):
    pass


def load():
    """Return the loaded CDLL; raise loudly if libnmrf_hip.so is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NmrfHipError(
            "libnmrf_hip.so not found at %s.  The NMRF hot path has no CPU/PyTorch fallback: build the HIP "
            "library first (python -m nmrf_amd.build, or __graft_entry__.build())." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _I
    lib.nmrf_strerror.argtypes = [_I]
    lib.nmrf_strerror.restype = ctypes.c_char_p
    lib.nmrf_build_stamp.restype = ctypes.c_char_p
    ver = lib.nmrf_abi_version()
    if ver != ABI_VERSION:
        raise NmrfHipError("libnmrf_hip.so ABI %d != binding ABI %d: rebuild" % (ver, ABI_VERSION))
    _lib = lib
    return lib


def load_debug():
    """The tools / test build of the library (python -m nmrf_amd.build --debug): everything the product library exports plus the
    reference kernels of include/nmrf_hip_debug.h.  Only NMRF_LINEAR=fp32 (A/B parity runs), tests and tools ask for it."""
    global _dbg
    if _dbg is not None:
        return _dbg
    if not os.path.exists(DEBUG_LIB_PATH):
        raise NmrfHipError("libnmrf_hip_debug.so not found at %s: the fp32-MFMA reference kernels live in the tools build only "
                           "(python -m nmrf_amd.build --debug)" % DEBUG_LIB_PATH)
    lib = ctypes.CDLL(DEBUG_LIB_PATH)
    for name, argtypes in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _I
    if lib.nmrf_abi_version() != ABI_VERSION:
        raise NmrfHipError("libnmrf_hip_debug.so ABI %d != binding ABI %d: rebuild" % (lib.nmrf_abi_version(), ABI_VERSION))
    lib.nmrf_build_stamp.restype = ctypes.c_char_p
    # the tools library must come from the same sources as the product library it is compared with (an A/B library named through
    # LIB_PATH is exempt: different sources are its point)
    if os.path.abspath(LIB_PATH) == os.path.join(_HERE, "lib", "libnmrf_hip.so"):
        want, got = load().nmrf_build_stamp().decode(), lib.nmrf_build_stamp().decode()
        if want != got:
            raise NmrfHipError("libnmrf_hip_debug.so was built from other sources than libnmrf_hip.so (%s vs %s): "
                               "python -m nmrf_amd.build" % (got, want))
    _dbg = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().nmrf_strerror(code).decode()
        raise NmrfHipError("%s failed: %s (code %d)" % (what, msg, code))
