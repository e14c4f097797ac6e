"""ctypes binding of libmvs_hip.so (the C ABI declared in include/mvs_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C mvsformer_amd/csrc`` and is the ONLY
compute path of this package: there is no PyTorch or CPU fallback, so a missing library is a hard error.

``import torch`` must precede the ``CDLL`` call: torch's wheel bundles its own HIP runtime under the same
SONAME (libamdhip64.so.7) and the dynamic loader then binds our library to that already-loaded runtime, so the
``hipStream_t`` handles torch hands us belong to the runtime that launches our kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (load order, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVS_HIP_LIB: diagnostics only - another BUILD of the same library (tests/test_hip_multistream.py's variants); never a fallback
LIB_PATH = os.environ.get("MVS_HIP_LIB") or os.path.join(_HERE, "libmvs_hip.so")
ABI_VERSION = 40

from ctypes import c_double  # noqa: E402

P, I, F, L, Dbl = c_void_p, c_int, c_float, c_int64, c_double

# name -> (restype, argtypes); mirrors include/mvs_hip.h one to one (tests/test_abi.py cross-checks the header)
class AdamTensor(ctypes.Structure):
    """``MvsAdamTensor`` of include/mvs_hip.h."""
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("n", ctypes.c_int64)]


class WgradJob(ctypes.Structure):
    """``MvsWgradJob`` of include/mvs_hip.h."""
    _fields_ = [("A", ctypes.c_void_p), ("Bt", ctypes.c_void_p), ("dW", ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ("nbatch", "CA", "CB", "CBout", "Dp", "Hp", "Wp", "Db", "Hb", "Wb", "sd", "shw", "taps", "reserved")]


SIGNATURES = {
    "mvs_version": (I, []),
    "mvs_last_error": (c_char_p, []),
    "mvs_proj_prepare": (I, [P, I, I, P, P]),
    "mvs_proj_relative": (I, [P, P, I, P, P]),
    "mvs_warp_fwd": (I, [P, P, P, I, I, I, I, I, I, P, P, P]),
    "mvs_nchw_to_nhwc": (I, [P, P, I, I, L, P]),
    "mvs_nchw_to_nhwc_multi": (I, [P, P, P, P, P, I, P]),
    "mvs_cv_entropy_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, I, P]),
    "mvs_vis_fwd": (I, [P, P, I, I, I, P, P]),
    "mvs_cv_aggregate_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, I, P]),
    "mvs_cv_aggregate_fwd_bf16": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, P, I, P]),
    "mvs_cv_corr_store_bytes": (L, [I, I, I, I, I, I, I]),
    "mvs_cv_corr_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, P]),
    "mvs_cv_merge_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, P, P]),
    "mvs_cv_corr_rows_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, P, P, I, P]),
    "mvs_cv_merge_rows_fwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, P]),
    "mvs_cv_tiled_workspace_bytes": (L, [I, I, I, I, I, I]),
    "mvs_conv3d_x3_supported": (I, [I, I, I, I]),
    "mvs_conv3d_x3_packed_bytes": (L, [I, I, I, I]),
    "mvs_conv3d_x3_pack_weights": (I, [P, I, I, I, I, P, P]),
    "mvs_conv3d_x3_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "mvs_deconv3d_x3_supported": (I, [I, I, I]),
    "mvs_deconv3d_x3_packed_bytes": (L, [I, I, I]),
    "mvs_deconv3d_x3_pack_weights": (I, [P, I, I, I, P, P]),
    "mvs_deconv3d_x3_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "mvs_conv3d_small_supported": (I, [I, I, I, I]),
    "mvs_conv3d_small_packed_bytes": (L, [I, I, I, I]),
    "mvs_conv3d_small_pack_weights": (I, [P, I, I, I, I, P, P]),
    "mvs_conv3d_small_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "mvs_tail_x3_packed_bytes": (L, []),
    "mvs_tail_x3_pack_weights": (I, [P, P, P]),
    "mvs_tail_x3_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "mvs_cv_tiled_entropy_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, I, P, P]),
    "mvs_cv_tiled_aggregate_fwd": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, P, I, P, P]),
    "mvs_conv3d_packed_floats": (L, [I, I, I]),
    "mvs_conv3d_pack_weights": (I, [P, I, I, I, P, P]),
    "mvs_conv3d_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P]),
    "mvs_deconv3d_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "mvs_prob3_fwd": (I, [P, P, I, I, I, I, I, P, P]),
    "mvs_head_fwd": (I, [P, P, P, P, I, P, F, I, I, I, I, I, P, P, P, P, P]),
    "mvs_bn_reduce_workspace_bytes": (L, [I, I, L]),
    "mvs_bn_stats": (I, [P, I, I, L, P, P, P]),
    "mvs_bn_finalize": (I, [P, P, P, P, P, F, F, Dbl, P, I, P, P, P, P, P]),
    "mvs_bn_finalize_grouped": (I, [P, P, P, P, P, F, F, Dbl, P, I, I, P, P, P, P, P]),
    "mvs_affine_act": (I, [P, P, P, P, I, I, I, L, P, P]),
    "mvs_bn_bwd_reduce": (I, [P, P, P, P, P, P, I, I, I, L, P, P, P]),
    "mvs_bn_bwd_apply": (I, [P, P, P, P, P, P, P, P, Dbl, P, I, I, I, L, P, P]),
    "mvs_conv3d_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_bf16_packed_elems": (L, [I, I]),
    "mvs_bf16_pack_weights": (I, [P, I, I, I, I, I, P, P]),
    "mvs_bf16_pack_weights2": (I, [P, I, I, I, I, I, P, I, I, I, P, P]),
    "mvs_bf16_conv3d": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_bf16_conv3d_stats_workspace_bytes": (L, [I, I, I, I, I]),
    "mvs_bf16_conv3d_stats": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, P, P, P]),
    "mvs_bf16_conv3d_wgrad_workspace_bytes": (L, [I, I, I, I, I, I]),
    "mvs_bf16_conv3d_wgrad": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_bf16_from_f32_ncdhw": (I, [P, P, I, I, L, P]),
    "mvs_bf16_to_f32_ncdhw": (I, [P, P, I, I, L, P]),
    "mvs_bf16_bn_reduce_workspace_bytes": (L, [I, L, I, L]),
    "mvs_bf16_bn_stats": (I, [P, I, L, I, L, P, P, P]),
    "mvs_bf16_affine_act": (I, [P, P, P, P, I, I, L, I, L, P, P]),
    "mvs_bf16_bn_train_fwd": (I, [P, P, I, I, L, I, L, P, P, P, P, F, F, P, P, P, P, P]),
    "mvs_bf16_bn_bwd_reduce": (I, [P, P, P, P, P, P, I, I, L, I, L, P, P, P]),
    "mvs_bf16_bn_bwd_apply": (I, [P, P, P, P, P, P, P, P, Dbl, P, I, I, L, I, L, P, P]),
    "mvs_bf16_bn_bwd_apply_dgb": (I, [P, P, P, P, P, P, P, P, Dbl, P, I, I, L, I, L, P, P, P]),
    "mvs_bf16_conv3d_bn_fwd_workspace_bytes": (L, [I, I, I, I, I]),
    "mvs_bf16_conv3d_bn_fwd": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P, F, F, P, P, P, P]),
    "mvs_bf16_conv3d_bnbwd": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, P, P, I, I, P, P, P, P]),
    "mvs_bf16_packed_elems_taps": (L, [I, I, I]),
    "mvs_bf16_pack_table_bytes": (L, [I]),
    "mvs_bf16_pack_table_fill": (I, [P, I, I, P, I, I, I, I, I, I, P]),
    "mvs_bf16_pack_table_run": (I, [P, I, I, P]),
    "mvs_bf16_conv3d_taps": (I, [P, P, P, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_adamw_step": (I, [P, I, F, F, F, F, F, I, P, P]),
    "mvs_bf16_wgrad_group_workspace_bytes": (L, [P, I]),
    "mvs_bf16_wgrad_group": (I, [P, I, P, L, P]),
    "mvs_bf16_conv3d_wgrad_taps": (I, [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_bf16_head_fwd": (I, [P, P, P, I, L, P, P]),
    "mvs_bf16_head_bwd_workspace_bytes": (L, [L]),
    "mvs_bf16_head_bwd": (I, [P, P, P, P, L, P, P, P, P]),
    "mvs_cv_aggregate_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, P]),
    "mvs_cv_aggregate_bwd_lds": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, I, I, P, P]),
    "mvs_cv_aggregate_bwd_own": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, P, P, I, I, P, P]),
    "mvs_softmax_bwd": (I, [P, P, I, I, L, P, P]),
    "mvs_prob1_bwd": (I, [P, P, P, I, I, L, P, P, P]),
    "mvs_sigmoid_fwd": (I, [P, L, P, P]),
    "mvs_sigmoid_bwd": (I, [P, P, L, P, P]),
    "mvs_ewise_mul": (I, [P, P, L, P, P]),
    "mvs_nhwc_to_nchw": (I, [P, P, I, I, L, P]),
    "mvs_depth_regression": (I, [P, P, I, I, I, I, I, P, P]),
    "mvs_conf_regression": (I, [P, I, I, I, I, I, P, P]),
    "mvs_mixup_head": (I, [P, P, I, I, I, I, P, P, P]),
    "mvs_prob1_fwd": (I, [P, P, P, I, I, L, P, P]),
    "mvs_geo_filter_workspace_bytes": (L, [I, I]),
    "mvs_geo_filter_fwd": (I, [P, P, P, P, I, I, I, I, F, F, F, P, P, P, P, P, P, P, P]),
    "mvs_vis_filter_fwd": (I, [P, P, P, P, I, I, I, I, F, F, F, P, P, P, P]),
    "mvs_geo_filter_dynamic_fwd": (I, [P, P, P, P, I, I, I, I, F, F, P, P, P, P, P, P, P, P]),
    "mvs_vis_filter_dynamic_fwd": (I, [P, P, I, I, I, I, F, F, P, P, P, P, P]),
    "mvs_ce_loss_fwd": (I, [P, P, P, P, I, I, L, I, F, P, P, P, P, P, P]),
    "mvs_mixup_ce_loss_fwd": (I, [P, P, P, P, I, I, L, I, F, P, P, P, P]),
    "mvs_reg_loss_fwd": (I, [P, P, P, P, P, I, I, L, I, F, P, P, P, P]),
    "mvs_was_loss_acc_floats": (L, [I, L]),
    "mvs_was_loss_fwd": (I, [P, P, P, P, I, I, L, I, F, F, P, P, P, P]),
    "mvs_ce_loss_bwd_scale": (I, [P, P, L, P, P, F, P]),
    "mvs_ce_loss_acc_floats": (L, [I, L]),
    "mvs_bf16_embed_ch0": (I, [P, P, L, P]),
    "mvs_gemm_x3": (I, [P, P, P, I, I, I, I, I, I, I, I, L, L, L, L, L, L, I, I, I, I, I, F, P, P, I, P, P, P]),
    "mvs_attention_x3": (I, [P, P, P, I, I, I, I, I, F, P]),
    "mvs_layernorm": (I, [P, P, P, P, L, I, F, P]),
    "mvs_conv2d_gemm_x3": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "mvs_partials_reduce": (I, [P, I, I, P, P]),
    "mvs_upsample2x_add": (I, [P, P, P, I, I, I, P]),
    "mvs_upsample2x_bwd": (I, [P, P, I, I, I, P]),
    "mvs_softmax_rows": (I, [P, P, L, I, F, P]),
    "mvs_bicubic_resize": (I, [P, P, I, I, I, I, I, F, F, P]),
    "mvs_x3p_bytes": (L, [L, I]),
    "mvs_x3p_pack": (I, [P, P, L, I, I, L, P]),
    "mvs_x3p_unpack": (I, [P, P, L, I, I, L, P]),
    "mvs_layernorm_x3p": (I, [P, P, P, P, L, I, I, I, F, P]),
    "mvs_gemm_x3p": (I, [P, P, I, I, I, L, L, P, I, P, P, I, P, P, P]),
    "mvs_conv_x3p": (I, [P, L, I, P, L, I, I, I, I, I, I, P, I, P, P, I, P, P, P]),
    "mvs_gemm_x3p_qkv": (I, [P, P, I, I, I, I, L, L, P, F, P, P, P, P]),
    "mvs_attention_x3p": (I, [P, P, P, P, I, I, I, I, P]),
    "mvs_cls_attention_x3p": (I, [P, P, P, I, I, I, I, P]),
    "mvs_conv3d_wino_supported": (I, [I, I, I, I, I]),
    "mvs_conv3d_wino_packed_floats": (L, [I, I]),
    "mvs_conv3d_wino_pack_weights": (I, [P, I, I, P, P]),
    "mvs_conv3d_wino_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "mvs_vis_wino_prepare": (I, [P, P, P]),
    "mvs_vis_wino_fwd": (I, [P, P, P, I, I, I, P, P]),
    "mvs_vis_x3_prepare": (I, [P, P, P]),
    "mvs_vis_x3_fwd": (I, [P, P, P, I, I, I, P, P]),
    "mvs_deconv3d_prob1_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "mvs_prob_filter": (I, [P, I, I, L, P, P, P, P]),
    "mvs_init_inverse_range": (I, [P, I, I, I, I, I, P, P]),
    "mvs_schedule_inverse_range": (I, [P, P, I, F, I, I, I, I, P, P]),
    "mvs_conf_accumulate": (I, [P, I, I, I, P, I, I, F, P]),
    "mvs_conv2d_packed_floats": (L, [I, I, I]),
    "mvs_conv2d_pack_weights": (I, [P, I, I, I, P, P]),
    "mvs_conv2d_bn_lrelu": (I, [P, P, P, P, I, I, I, I, I, I, I, F, P, P]),
    "mvs_fpn_level_cp_prepared_bytes": (L, [I]),
    "mvs_fpn_level_cp_prepare": (I, [P, P, P, I, P, P]),
    "mvs_fpn_level_cp": (I, [P, P, P, P, P, I, I, I, I, P, P]),
    "mvs_conv2d_x3s_supported": (I, [I, I, I, I]),
    "mvs_conv2d_x3s_prepared_bytes": (L, [I, I, I, I]),
    "mvs_conv2d_x3s_prepare": (I, [P, P, I, I, I, I, P, P]),
    "mvs_conv2d_x3s_bn_lrelu": (I, [P, I, P, P, I, I, I, I, I, I, I, F, P, P]),
    "mvs_conv2d_x3_supported": (I, [I, I, I, I]),
    "mvs_conv2d_x3_prepared_bytes": (L, [I, I, I]),
    "mvs_conv2d_x3_prepare": (I, [P, P, I, I, I, P, P]),
    "mvs_conv2d_x3_bn_lrelu": (I, [P, P, P, I, I, I, I, I, I, I, F, P, P]),
    "mvs_fpn_packed_floats": (L, [I]),
    "mvs_fpn_pack_weights": (I, [P, I, P, P]),
    "mvs_fpn_out0": (I, [P, P, P, P, I, I, I, P, P]),
    "mvs_fpn_level": (I, [P, P, P, P, P, P, P, I, I, I, I, P, P, P]),
    "mvs_fpn_level_layout": (I, [P, P, P, P, P, P, P, I, I, I, I, P, I, P, P]),
    "mvs_conv2d_x3_bn_lrelu_layout": (I, [P, I, P, P, I, I, I, I, I, I, I, F, P, P, P]),
    "mvs_fpn_level_x3s_prepared_bytes": (L, [I]),
    "mvs_fpn_level_x3s_prepare": (I, [P, P, I, P, P]),
    "mvs_fpn_level_x3s": (I, [P, P, P, P, P, P, I, I, I, I, P, I, P, P]),
    "mvs_fpn_level_x3_prepared_bytes": (L, [I]),
    "mvs_fpn_level_x3_prepare": (I, [P, P, P, I, P, P]),
    "mvs_fpn_level_x3": (I, [P, P, P, P, P, I, I, I, I, P, P]),
}

_lib = None


class MvsHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and return the library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvsHipError(
            "%s is missing: the HIP path has not been built (run `python -c \"import __graft_entry__ as g; g.build()\"` "
            "or `make -C mvsformer_amd/csrc`).  mvsformer_amd has no CPU/PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    got = lib.mvs_version()
    if got != ABI_VERSION:
        raise MvsHipError("libmvs_hip.so ABI version %d, binding expects %d — rebuild" % (got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().mvs_last_error()
        raise MvsHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
