"""ctypes binding of libunipose_hip.so (the C ABI declared in include/unipose_hip.h).

The product path has exactly one backend: the HIP library built in-tree by
``unipose_amd.build.build_library()`` (``__graft_entry__.build()``).  If it is missing the import
of any op fails loudly — there is no PyTorch / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunipose_hip.so")

# device pointers and the hipStream_t travel as integers (tensor.data_ptr(), stream.cuda_stream) -> c_void_p


class ConvDesc(C.Structure):
    """up_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "C", "Cp", "ldx", "K", "R", "S", "stride", "pad", "dil", "P", "Q", "ldy", "Kp")]


class BnFold(C.Structure):
    """up_bn_fold: BatchNorm finalize folded into the launch that writes the statistics partials (ABI 10)"""
    _fields_ = [("eps", C.c_float), ("momentum", C.c_float), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("scale", C.c_void_p),
                ("shift", C.c_void_p), ("folded", C.c_int32)]


class ConvEpilogue(C.Structure):
    """up_conv_epilogue"""
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("ldr", C.c_int32), ("relu", C.c_int32), ("stats", C.c_void_p), ("fold", C.POINTER(BnFold))]


_i, _i64, _f, _u64, _sz, _p = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t, C.c_void_p


class BnReduceSlot(C.Structure):
    """up_bn_reduce_slot"""
    _fields_ = [("y", C.c_void_p), ("relu_bits", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("partial", C.c_void_p), ("ld", C.c_int32), ("C", C.c_int32), ("group_stride", C.c_int32),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("gsum", C.c_void_p), ("folded", C.c_int32)]


class DgradEpilogue(C.Structure):
    """up_dgrad_epilogue"""
    _fields_ = [("add", C.c_void_p), ("add_relu_bits", C.c_void_p), ("bn", C.POINTER(BnReduceSlot)), ("ld_add", C.c_int32),
                ("groups", C.c_int32)]


_D, _E = C.POINTER(ConvDesc), C.POINTER(ConvEpilogue)

# name -> (restype, argtypes); mirrors include/unipose_hip.h one to one
SIGNATURES = {
    "up_last_error": (C.c_char_p, []),
    "up_abi_version": (_i, []),
    "up_pack_weights": (_i, [_D, _p, _p, _p, _p]),
    "up_conv2d_fwd": (_i, [_D, _p, _p, _p, _E, _p]),
    "up_pack_weights_batched": (_i, [_p, _i, _p]),
    "up_conv_stats_tiles": (_i, [_D]),
    "up_conv_stats_tiles_math": (_i, [_D, _i]),
    "up_conv_split_parts": (_i, [_D]),
    "up_stream_release": (_i, [_p]),
    "up_stream_create_cu_mask": (_i, [C.POINTER(C.c_uint32), _i, C.POINTER(_p)]),
    "up_stream_destroy": (_i, [_p]),
    "up_probe_placement": (_i, [_i, _p, _p]),
    "up_conv_tune": (_i, [C.c_char_p, _i]),
    "up_conv_counter": (C.c_longlong, [C.c_char_p]),
    "up_conv_wgrad_visits": (_i, [_D, C.POINTER(C.c_double)]),
    "up_conv_tap_visits": (_i, [_D, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "up_conv2d_bwd_data": (_i, [_D, _p, _p, _p, _p, _i, _p]),
    "up_conv2d_bwd_data_tiles": (_i, [_D]),
    "up_conv2d_bwd_data_tiles_math": (_i, [_D, _i]),
    "up_conv2d_bwd_data_ex": (_i, [_D, _p, _p, _p, C.POINTER(DgradEpilogue), _i, _p]),
    "up_conv_stats_tiles_grouped": (_i, [_D, _i]),
    "up_conv2d_bwd_data_tiles_grouped": (_i, [_D, _i]),
    "up_bn_bwd_groups_prereduced_ok": (_i, [_i64, _i, _i, _i]),
    "up_bn_stats_groups_t": (_i, [_p, _i, _i64, _i, _i, _i, _p, _f, _f, _p, _p, _p, _p, _p, _p]),
    "up_bn_bwd_groups_finalized_t": (_i, [_p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i64, _i, _i, _i, _p]),
    "up_bn_bwd_groups_prereduced_t": (_i, [_p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _p, _i, _i64, _i, _i, _i, _p]),
    "up_conv2d_fwd_grouped": (_i, [_D, _p, _p, _p, _p, _i, _p]),
    "up_bn_bwd_prereduced_t": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _i64, _i,
                                    _i, _p]),
    "up_bn_bwd_finalized_t": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _i64, _i, _i, _p]),
    "up_pack_weights_bf16": (_i, [_D, _p, _p, _p, _p, _p, _p]),
    "up_pack_weights_bf16_batched": (_i, [_p, _i, _p]),
    "up_conv2d_fwd_bf16": (_i, [_D, _p, _p, _p, _p, _E, _i, _p]),
    "up_conv2d_bwd_data_bf16": (_i, [_D, _p, _p, _p, _p, _p, _i, _i, _p]),
    "up_conv2d_bwd_weight_workspace": (_sz, [_D]),
    "up_conv2d_bwd_weight": (_i, [_D, _p, _p, _p, _p, _p, _sz, _p]),
    "up_conv2d_bwd_weight_bf16": (_i, [_D, _p, _p, _p, _p, _p, _sz, _p]),
    "up_conv2d_bwd_weight_bf16s": (_i, [_D, _p, _p, _p, _p, _p, _sz, _p]),
    "up_conv2d_bwd_weight_acc": (_i, [_D, _p, _p, _p, _p, _p, _sz, _i, _i, _p]),
    "up_bn_eval_coeffs": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "up_bn_finalize": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "up_bn_apply": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i64, _i, _p]),
    "up_bn_bwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _i64, _i, _p]),
    "up_bn_bwd_workspace": (_sz, [_i64, _i]),
    "up_bn_apply_t": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i64, _i, _i, _p]),
    "up_bn_bwd_t": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _i64, _i, _i, _p]),
    "up_bn_bwd_acc_t": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _sz, _i64, _i, _i, _p]),
    "up_bn_exact_stats_t": (_i, [_p, _i, _i64, _i, _i, _i, _p, _p]),
    "up_bn_batch_stats_tiles": (_i, [_i64]),
    "up_bn_batch_stats_t": (_i, [_p, _i, _i64, _i, _i, _i, _p, _p]),
    "up_bn_finalize_groups": (_i, [_p, _i, _i, _i, _i64, _f, _f, _p, _p, _p, _p, _p, _p]),
    "up_bn_apply_groups_t": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i64, _i, _i, _i, _p]),
    "up_bn_apply_centered_t": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i64, _i, _i, _p]),
    "up_bn_bwd_groups_workspace": (_sz, [_i64, _i, _i]),
    "up_bn_bwd_groups_t": (_i, [_p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _sz, _i64, _i, _i, _i, _p]),
    "up_relu_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "up_copy2d": (_i, [_p, _i, _p, _i, _i64, _i, _p]),
    "up_add2d": (_i, [_p, _i, _p, _i, _p, _i, _i64, _i, _p]),
    "up_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "up_nhwc_to_nchw": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "up_maxpool3s2_fwd": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "up_maxpool3s2_bwd": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_bilinear_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_bilinear_bwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_gap_fwd": (_i, [_p, _i, _p, _i, _i, _i, _p]),
    "up_gap_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "up_avgpool9s8_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_dropout_fwd": (_i, [_p, _p, _p, _p, _i64, _f, _u64, _p]),
    "up_dropout_bwd": (_i, [_p, _p, _p, _i64, _f, _p]),
    "up_nchw_to_nhwc_t": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "up_nhwc_to_nchw_t": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "up_maxpool3s2_fwd_t": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_maxpool3s2_bwd_t": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_bilinear_fwd_t": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_bilinear_bwd_t": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "up_gap_fwd_t": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "up_gap_bwd_t": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "up_dropout_fwd_t": (_i, [_p, _p, _p, _p, _i64, _f, _u64, _i, _p]),
    "up_dropout_fwd_step_t": (_i, [_p, _p, _p, _p, _i64, _f, _u64, _p, _i, _p]),
    "up_dropout_bwd_t": (_i, [_p, _p, _p, _i64, _f, _i, _p]),
    "up_mse_fwd": (_i, [_p, _p, _p, _p, _i64, _p]),
    "up_mse_bwd": (_i, [_p, _p, _p, _p, _i64, _p]),
    "up_mse_workspace": (_sz, [_i64]),
    "up_lstm0_fwd": (_i, [_p, _i, _p, _p, _i, _i64, _i, _p]),
    "up_lstm0_bwd": (_i, [_p, _i, _p, _p, _i, _p, _i64, _i, _p]),
    "up_lstm_fwd": (_i, [_p, _i, _p, _i, _p, _p, _i, _i64, _i, _p]),
    "up_lstm_bwd": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _i64, _i, _p]),
    "up_heatmap_argmax": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "up_make_heatmaps": (_i, [_p, _i, _i, _i, _i, C.c_double, C.c_double, _p, _p]),
    "up_make_gaussian_maps": (_i, [_p, _i, _i, _i, C.c_double, _p, _p]),
    "up_normalize_image": (_i, [_p, _i, _i, _i, _i, _f, _f, _p, _p]),
    "up_pck_accuracy": (_i, [_p, _p, _i, _i, _i, _i, _i, C.c_double, C.c_double, _p, _p, _p, _p, _p, _p]),
    "up_peak_mask": (_i, [_p, _i, _i, _i, _p, _p]),
    "up_box_argmax": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p]),
    "up_unipose_plan_create": (_i, [_p, C.POINTER(_p)]),
    "up_unipose_plan_destroy": (None, [_p]),
    "up_unipose_plan_num_convs": (_i, [_p]),
    "up_unipose_plan_conv_name": (C.c_char_p, [_p, _i]),
    "up_unipose_plan_conv_shape": (_i, [_p, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "up_unipose_plan_set_conv": (_i, [_p, _i, _p, _p, _p]),
    "up_unipose_plan_workspace": (_sz, [_p]),
    "up_unipose_forward": (_i, [_p, _p, _p, _p, _sz, _p]),
    "up_profile_variants": (_i, []),
    "up_profile_variant_name": (C.c_char_p, [_i]),
    "up_profile_begin": (_i, []),
    "up_profile_enable": (_i, [_i]),
    "up_profile_live_flops": (_i, [C.POINTER(C.c_double), _i]),
    "up_profile_end": (_i, [C.POINTER(C.c_double), _i]),
}

_lib = None
# Set ONLY by tests/conftest.py when it points the binding at the CPU emulation build of the same
# kernel sources (tests/emu).  The product never sets it: ops then refuse non-CUDA tensors.
_ALLOW_HOST_POINTERS = False


class UniPoseHipError(RuntimeError):
    pass


def load(path: str | None = None):
    """dlopen the library and attach the prototypes.  Raises if it is missing (no fallback)."""
    global _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise UniPoseHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  unipose_amd has no non-HIP fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib():
    return _lib if _lib is not None else load()


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib().up_last_error().decode(errors="replace")
        if status == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise UniPoseHipError(f"{what}: status {status}: {msg}")
