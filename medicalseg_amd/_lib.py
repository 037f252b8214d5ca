"""ctypes binding of libmsegk.so (include/msegk.h).

The library is the ONLY compute path of this package: there is no CPU or
PyTorch fallback.  Importing this module without the built library raises; calling
into it without a GPU raises from msk_ctx_create.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSEGK_LIB") or os.path.join(_HERE, "lib", "libmsegk.so")   # MSEGK_LIB: A/B builds of tools/ab_build.sh


class MskError(RuntimeError):
    pass


class MskTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_int32), ("d", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("ld", C.c_int32)]


class MskConvDesc(C.Structure):
    _fields_ = [("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
                ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32)]


class MskBnFin(C.Structure):
    """msk_bn_fin (include/msegk.h): the arguments of msk_bn_finalize(world = 1) as a struct"""
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("momentum", C.c_float),
                ("count", C.c_double), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("save_mean", C.c_void_p), ("save_invstd", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


NULL_TENSOR = MskTensor(None, 0, 0, 0, 0, 0, 0)

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_u64, _u32 = C.c_uint64, C.c_uint32
_T, _CD = MskTensor, MskConvDesc

# name -> (restype, argtypes); mirrors include/msegk.h one to one
SIGNATURES = {
    "msk_version": (_i, []),
    "msk_device_count": (_i, [C.POINTER(_i)]),
    "msk_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "msk_ctx_destroy": (_i, [_vp]),
    "msk_last_error": (C.c_char_p, [_vp]),
    "msk_sync": (_i, [_vp]),
    "msk_join_side": (_i, [_vp]),
    "msk_device_name": (_i, [_vp, C.c_char_p, _i]),
    "msk_device_pci_bus_id": (_i, [_vp, C.c_char_p, _i]),
    "msk_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "msk_free": (_i, [_vp, _vp]),
    "msk_memset": (_i, [_vp, _vp, _i, _sz]),
    "msk_weights_changed": (_i, [_vp, _vp, _sz]),
    "msk_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "msk_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "msk_h2d_async": (_i, [_vp, _vp, _vp, _sz]),
    "msk_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "msk_pinned_alloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "msk_pinned_free": (_i, [_vp, _vp]),
    "msk_mem_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "msk_timer_start": (_i, [_vp]),
    "msk_timer_stop": (_i, [_vp, C.POINTER(_f)]),
    "msk_mark": (_i, [_vp, _i]),
    "msk_ctx_wait": (_i, [_vp, _vp]),
    "msk_mark_elapsed": (_i, [_vp, _i, _i, C.POINTER(_f)]),
    "msk_prof_enable": (_i, [_vp, _i]),
    "msk_prof_reset": (_i, [_vp]),
    "msk_prof_report": (_i, [_vp, C.c_char_p, _i, C.POINTER(_i)]),
    "msk_set_option": (_i, [_vp, C.c_char_p, _i]),
    "msk_get_option": (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    "msk_ncdhw_to_ndhwc": (_i, [_vp, _vp, _T]),
    "msk_ndhwc_to_ncdhw": (_i, [_vp, _T, _vp]),
    "msk_conv3d_fwd": (_i, [_vp, _CD, _T, _vp, _vp, _T]),
    "msk_conv_fold_bn": (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_long, _vp, _vp]),
    "msk_conv3d_fwd_act": (_i, [_vp, _CD, _T, _vp, _vp, _vp, _T]),
    "msk_conv3d_xform_bytes": (_sz, [_vp, _CD, _T, _i]),
    "msk_conv3d_fwd_ex": (_i, [_vp, _CD, _T, _vp, _vp, _T, _vp, _vp]),
    "msk_conv3d_wgrad_ex": (_i, [_vp, _CD, _T, _T, _vp, _vp, _i, _vp]),
    "msk_amax_new": (_vp, [_vp, _i]),
    "msk_conv3d_fwd_ex3": (_i, [_vp, _CD, _T, _vp, _vp, _T, _vp, _vp, _vp, _vp]),
    "msk_conv3d_fwd_in": (_i, [_vp, _CD, _T, _vp, _vp, _T, _vp, _vp, _vp, _vp, _i]),
    "msk_bn_stats_fin": (_i, [_vp, _T, _vp, _vp]),
    "msk_affine_act_bwd_reduce_pg": (_i, [_vp, _T, _vp, _vp, _T, _vp, _vp, _vp, _T, _vp, _vp, _i, _vp, _vp, _vp]),
    "msk_add_act_join_bwd_pg": (_i, [_vp, _T, _vp, _vp, _vp, _T, _vp, _vp, _vp, _T, _T, _T, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "msk_conv3d_fwd_ex2": (_i, [_vp, _CD, _T, _vp, _vp, _T, _vp, _vp, _vp]),
    "msk_affine_act_fwd_amax": (_i, [_vp, _T, _vp, _vp, _T, _vp, _T, _vp]),
    "msk_affine_act_fwd_amax2": (_i, [_vp, _T, _vp, _vp, _T, _vp, _T, _vp, _T]),
    "msk_affine_act_join_fwd_amax": (_i, [_vp, _T, _vp, _vp, _vp, _T, _vp, _T, _vp]),
    "msk_copy_scale_amax": (_i, [_vp, _T, _vp, _T, _i, _vp]),
    "msk_conv3d_bwd_bnact_bytes": (_sz, [_vp, _CD, _T, _T]),
    "msk_conv3d_bwd_bnact": (_i, [_vp, _CD, _T, _vp, _T, _vp, _vp, _vp, _vp, _vp, _vp, _T, _vp, _d, _T, _T, _i, _vp, _i, _vp, _vp, _vp]),
    "msk_convT3d_bwd_bnact": (_i, [_vp, _CD, _T, _vp, _T, _vp, _vp, _vp, _vp, _vp, _T, _vp, _d, _T, _T, _i, _vp, _i]),
    "msk_conv3d_bwd_bnact_split": (_i, [_vp, _CD, _T, _vp, _T, _vp, _vp, _vp, _vp, _vp, _vp, _T, _vp, _d, _T, _T, _i, _vp, _i, _vp, _vp, _vp, _T, _T, C.POINTER(_i)]),
    "msk_conv3d_bwd_bnact_acc": (_i, [_vp, _CD, _T, _vp, _T, _vp, _vp, _vp, _vp, _vp, _vp, _T, _vp, _d, _T, _T, _i, _vp, _i, _vp, _vp, _vp, _T, _T, _T, C.POINTER(_i)]),
    "msk_conv3d_bwd_inact": (_i, [_vp, _CD, _T, _vp, _T, _vp, _vp, _vp, _vp, _vp, _i, _T, _vp, _i, _d, _T, _i, _vp, _i, _vp, _vp, _vp]),
    "msk_conv3d_bwd_bnact_c1": (_i, [_vp, _CD, _T, _T, _vp, _vp, _vp, _vp, _vp, _T, _T, _vp, _d, _vp, _i]),
    "msk_conv3d_dgrad": (_i, [_vp, _CD, _T, _vp, _T, _i]),
    "msk_conv3d_dgrad_ex": (_i, [_vp, _CD, _T, _vp, _T, _i, _vp]),
    "msk_conv3d_wgrad_ex2": (_i, [_vp, _CD, _T, _T, _vp, _vp, _i, _vp, _vp]),
    "msk_conv3d_wgrad_ex3": (_i, [_vp, _CD, _T, _T, _vp, _vp, _i, _vp, _vp, _vp]),
    "msk_conv3d_wgrad": (_i, [_vp, _CD, _T, _T, _vp, _vp, _i]),
    "msk_convT3d_fwd": (_i, [_vp, _CD, _T, _vp, _vp, _T]),
    "msk_convT3d_fwd_ex": (_i, [_vp, _CD, _T, _vp, _vp, _T, _vp, _vp]),
    "msk_convT3d_dgrad": (_i, [_vp, _CD, _T, _vp, _T, _i]),
    "msk_convT3d_wgrad": (_i, [_vp, _CD, _T, _T, _vp, _vp, _i]),
    "msk_bn_stats": (_i, [_vp, _T, _vp]),
    "msk_bn_finalize": (_i, [_vp, _vp, _i, _d, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msk_bn_eval_coeffs": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "msk_affine_act_fwd": (_i, [_vp, _T, _vp, _vp, _T, _vp, _T]),
    "msk_affine_act_bwd_reduce": (_i, [_vp, _T, _vp, _vp, _T, _vp, _vp, _vp, _T, _vp]),
    "msk_affine_act_bwd_reduce_ex": (_i, [_vp, _T, _vp, _vp, _T, _vp, _vp, _vp, _T, _vp, _vp]),
    "msk_affine_act_bwd_apply": (_i, [_vp, _T, _vp, _vp, _T, _vp, _vp, _vp, _vp, _T, _vp, _d, _i, _T, _T, _i]),
    "msk_affine_act_bwd_apply_amax": (_i, [_vp, _T, _vp, _vp, _T, _vp, _vp, _vp, _vp, _T, _vp, _d, _i, _T, _T, _i, _vp]),
    "msk_affine_act_param_grads": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i]),
    "msk_add_act_bwd": (_i, [_vp, _T, _T, _vp, _T, _T, _T, _i, _vp]),
    "msk_affine_act_join_fwd": (_i, [_vp, _T, _vp, _vp, _vp, _T, _vp, _T]),
    "msk_add_act_join_bwd": (_i, [_vp, _T, _vp, _vp, _vp, _T, _vp, _T, _T, _T, _i, _vp]),
    "msk_add_act_join_bwd_ex": (_i, [_vp, _T, _vp, _vp, _vp, _T, _vp, _vp, _vp, _T, _T, _T, _i, _vp, _vp, _vp]),
    "msk_bn_bias_grad": (_i, [_vp, _i, _vp, _vp, _vp, _i]),
    "msk_copy_scale": (_i, [_vp, _T, _vp, _T, _i]),
    "msk_dropout_mask": (_i, [_vp, _u64, _u64, _u32, _i, _f, _vp]),
    "msk_channel_sum": (_i, [_vp, _T, _vp, _i]),
    "msk_argmax_c": (_i, [_vp, _T, _vp]),
    "msk_softmax_c": (_i, [_vp, _T, _T]),
    "msk_class_weights": (_i, [_vp, _T, _vp]),
    "msk_loss_fwd": (_i, [_vp, _T, _vp, _vp, _i, _vp, _vp]),
    "msk_loss_bwd": (_i, [_vp, _T, _vp, _vp, _i, _vp, _f, _f, _T]),
    "msk_elu_fwd": (_i, [_vp, _T, _f, _T]),
    "msk_elu_bwd": (_i, [_vp, _T, _T, _f, _T, _i]),
    "msk_loss_fwd_ex": (_i, [_vp, _T, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "msk_loss_bwd_ex": (_i, [_vp, _T, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _T]),
    "msk_sgd_momentum": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f]),
    "msk_sgd_momentum_eager": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f]),
    "msk_sgd_momentum_finish": (_i, [_vp]),
    "msk_adam": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _d, _d, _f, _f]),
    "msk_resample3d": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i]),
    "msk_crop_resample3d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i]),
    "msk_flip3d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "msk_rotate3d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _d, _i]),
    "msk_hu_norm": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f]),
    "msk_minmax_norm": (_i, [_vp, _vp, _vp, _sz, _i, _f, _f]),
    "msk_max_norm": (_i, [_vp, _vp, _vp, _sz]),
    "msk_label_remap": (_i, [_vp, _vp, _sz, _vp, _vp, _i]),
    "msk_interp_trilinear_fwd": (_i, [_vp, _T, _T]),
    "msk_interp_scratch_bytes": (_i, [_vp, _T, _T, C.POINTER(_sz)]),
    "msk_interp_trilinear_bwd": (_i, [_vp, _T, _T, _i, _vp, _sz]),
    "msk_dp_unique_id": (_i, [C.c_char_p]),
    "msk_dp_rccl_version": (_i, [C.POINTER(_i)]),
    "msk_dp_init": (_i, [_vp, C.c_char_p, _i, _i]),
    "msk_dp_allreduce_sum": (_i, [_vp, _vp, _sz]),
    "msk_dp_allreduce_stats": (_i, [_vp, _vp, _sz]),
    "msk_dp_allreduce_async": (_i, [_vp, _vp, _sz]),
    "msk_dp_wait": (_i, [_vp]),
    "msk_dp_allgather": (_i, [_vp, _vp, _vp, _sz]),
    "msk_dp_broadcast": (_i, [_vp, _vp, _sz, _i]),
    "msk_dp_barrier": (_i, [_vp]),
    "msk_dp_destroy": (_i, [_vp]),
}

UNIQUE_ID_BYTES = 128

_lib = None


def load():
    """Load libmsegk.so; raises MskError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # RCCL shares device buffers between the ranks of a node through dmabuf IPC; the legacy IPC mode fails on hosts whose
    # driver only supports dmabuf ("hipIpcGetMemHandle: invalid argument").  Must be in the environment before the HIP
    # runtime initialises, i.e. before the library is loaded; an explicit setting wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not os.path.exists(LIB_PATH):
        raise MskError(
            f"{LIB_PATH} is missing: build it with ./build.sh (or __graft_entry__.build()). "
            "medicalseg_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error(ctx=None) -> str:
    msg = load().msk_last_error(ctx)
    return msg.decode() if msg else ""
