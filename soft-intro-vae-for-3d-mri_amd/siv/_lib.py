"""ctypes binding of libsiv.so (the C ABI declared in include/siv.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIV_LIB", os.path.join(_HERE, "libsiv.so"))   # SIV_LIB: A/B builds of the library

_f = C.c_void_p      # device pointers travel as integers
_i, _l, _fl, _sz, _u64 = C.c_int, C.c_long, C.c_float, C.c_size_t, C.c_uint64

# name -> argument types; every function returns int (0 = ok)
SIGNATURES = {
    "siv_set_conv_path": [_i],
    "siv_pack_cache_configure": [_f, _sz],
    "siv_pack_cache_add_weights": [_f, _sz],
    "siv_pack_cache_remove_weights": [_f],
    "siv_pack_cache_invalidate": [],
    "siv_pack_cache_invalidate_range": [_f, _sz],
    "siv_conv3d_k3_fwd": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv3d_k3_fwd_bnstats": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _fl, _f, _sz, _f],
    "siv_conv3d_k3_dgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv3d_k3_dgrad_up": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv3d_k3_dgrad_up_supported": [_i, _i],
    "siv_conv3d_k3_wgrad": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv_transpose3d_k3_fwd": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv_transpose3d_k3_dgrad": [_f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv_transpose3d_k3_wgrad": [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_supported": [_i],
    "siv_conv1_bnact_fwd": [_f, _f, _f, _f, _f, _f, _u64, _u64, _f, _fl, _fl, _fl, _i, _f, _f, _f, _f, _f, _fl, _f,
                            _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _u64, _u64, _f, _fl, _fl, _fl, _f, _f, _f, _i,
                            _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_fwd_bf16": [_f, _f, _f, _f, _f, _f, _u64, _u64, _f, _fl, _fl, _fl, _i, _f, _f, _f, _f, _f, _fl, _f,
                                 _f, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_fwd_bits": [_f, _f, _f, _f, _f, _f, _u64, _u64, _f, _fl, _fl, _fl, _i, _f, _f, _f, _f, _f, _fl, _f,
                                 _f, _f, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_bwd_bits": [_f, _f, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _fl, _f, _f, _f, _i,
                                 _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_conv1_bnact_wgrad_bits": [_f, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _fl, _f, _f, _f, _f, _i,
                                   _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_bnact_conv3d_k3_to1_fwd_bf16": [_f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _f, _i, _i, _i, _i, _i, _f],
    "siv_bnact_conv3d_k3_to1_dgrad_bf16": [_f, _f, _f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f,
                                           _sz, _f],
    "siv_bnact_conv3d_k3_to1_wgrad_bf16": [_f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_bnact_conv3d_k3_to1_supported": [_i, _i, _i, _i, _i],
    "siv_bnact_conv3d_k3_to1_fwd": [_f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _f],
    "siv_bnact_conv3d_k3_to1_dgrad": [_f, _f, _f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_bnact_conv3d_k3_to1_wgrad": [_f, _f, _f, _f, _f, _fl, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_upsample2_bwd": [_f, _f, _i, _i, _i, _i, _i, _f],
    "siv_conv1x1_fwd": [_f, _f, _f, _f, _l, _i, _i, _f],
    "siv_conv1x1_dgrad": [_f, _f, _f, _l, _i, _i, _f],
    "siv_conv1x1_wgrad": [_f, _f, _f, _f, _l, _i, _i, _i, _f, _sz, _f],
    "siv_linear_fwd": [_f, _f, _f, _f, _i, _i, _i, _f, _sz, _f],
    "siv_linear_dgrad": [_f, _f, _f, _i, _i, _i, _f, _sz, _f],
    "siv_linear_wgrad": [_f, _f, _f, _f, _i, _i, _i, _i, _f],
    "siv_add_act_fwd": [_f, _f, _fl, _f, _l, _f],
    "siv_add_act_bwd": [_f, _f, _f, _fl, _f, _l, _f],
    "siv_preprocess": [_f, _f, _i, _l, _fl, _f, _sz, _f],
    "siv_flip_shift": [_f, _f, _f, _i, _i, _i, _i, _f],
    "siv_affine_resample": [_f, _f, _f, _i, _i, _i, _i, _i, _fl, _f, _sz, _f],
    "siv_bn_stats": [_f, _l, _i, _f, _f, _f, _f, _f, _fl, _f, _sz, _f],
    "siv_bn_act_fwd": [_f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _i, _i, _i, _i, _i, _f],
    "siv_bn_act_bwd": [_f, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _f, _f, _f, _i,
                       _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_bn_stats_grouped": [_f, _l, _i, _i, _f, _f, _f, _f, _f, _fl, _f, _sz, _f],
    "siv_bn_act_fwd_grouped": [_f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _i, _i, _i, _i, _i, _i, _f],
    "siv_bn_act_bwd_grouped": [_f, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _f, _f, _f, _i,
                               _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_bn_act_bf16_supported": [_i],
    "siv_bn_act_fwd_bf16": [_f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _f, _f, _i, _i, _i, _i, _i, _i, _f],
    "siv_bn_act_bwd_bf16": [_f, _f, _f, _f, _f, _f, _f, _f, _fl, _fl, _i, _fl, _f, _f, _f, _f, _i,
                            _i, _i, _i, _i, _i, _i, _f, _sz, _f],
    "siv_act_dropout_fwd": [_f, _f, _fl, _fl, _f, _l, _f],
    "siv_act_dropout_bwd": [_f, _f, _f, _fl, _fl, _f, _l, _f],
    "siv_reparam_fwd": [_f, _f, _f, _fl, _f, _l, _f],
    "siv_reparam_bwd": [_f, _f, _f, _fl, _f, _f, _l, _f],
    "siv_kl_fwd": [_f, _f, _f, _i, _l, _f],
    "siv_kl_bwd": [_f, _f, _f, _f, _f, _i, _l, _f],
    "siv_sse_fwd": [_f, _f, _f, _i, _l, _f, _sz, _f],
    "siv_sse_fwd_noalloc": [_f, _f, _f, _i, _l, _f],
    "siv_sse_bwd": [_f, _f, _f, _f, _f, _i, _l, _f],
    "siv_sintro_loss_e_fwd": [_f, _f, _f, _f, _f, _f, _i, _fl, _fl, _fl, _fl, _f, _f],
    "siv_sintro_loss_e_bwd": [_f, _f, _f, _f, _f, _i, _fl, _fl, _fl, _fl, _f, _f],
    "siv_sintro_loss_d_fwd": [_f, _f, _f, _f, _f, _i, _fl, _fl, _fl, _fl, _f, _f],
    "siv_sintro_loss_d_bwd": [_f, _i, _fl, _fl, _fl, _fl, _f, _f],
    "siv_adam_step": [_f, _f, _f, _f, _l, _fl, _fl, _fl, _fl, _i, _f],
    "siv_adam_step_at": [_f, _f, _f, _f, _l, _f, _fl, _fl, _fl, _f, _f],
    "siv_counter_add": [_f, C.c_int64, _f],
    "siv_randn_at": [_f, _l, _u64, _u64, _f, _f],
    "siv_dropout_keep_at": [_f, _l, _fl, _u64, _u64, _f, _f],
    "siv_randn": [_f, _l, _u64, _u64, _f],
    "siv_dropout_keep": [_f, _l, _fl, _u64, _u64, _f],
    "siv_axpby": [_fl, _f, _fl, _f, _l, _f],
    "siv_scale": [_fl, _f, _l, _f],
    "siv_f32_to_bf16": [_f, _f, _l, _f],
    "siv_bf16_to_f32": [_f, _f, _l, _f],
}

_lib = None


class SivError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libsiv.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SivError(
            f"{LIB_PATH} not found: build it with `make -C soft-intro-vae-for-3d-mri_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.siv_last_error.restype = C.c_char_p
    lib.siv_last_error.argtypes = []
    lib.siv_pack_cache_hits.restype = C.c_long
    lib.siv_pack_cache_hits.argtypes = []
    lib.siv_last_conv_kernel.restype = C.c_char_p
    lib.siv_last_conv_kernel.argtypes = []
    lib.siv_version.restype = C.c_int
    lib.siv_version.argtypes = []
    lib.siv_workspace_bytes.restype = C.c_size_t
    lib.siv_conv1_bnact_bits_bytes.restype = C.c_size_t
    lib.siv_conv1_bnact_bits_bytes.argtypes = [_i] * 5
    lib.siv_workspace_bytes.argtypes = []
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().siv_last_error().decode("utf-8", "replace")
        raise SivError(f"{what} failed (code {rc}): {msg}")
