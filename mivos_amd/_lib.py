"""ctypes binding of libmivos_hip.so (C ABI: include/mivos_hip.h).

The product path has NO fallback: if the shared library is missing, cannot be loaded or the
device is not a gfx950, every compute entry point raises.  (Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C mivos_amd/csrc``.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIVOS_HIP_LIB") or os.path.join(_HERE, "libmivos_hip.so")      # (MIVOS_HIP_LIB: A/B of two builds of the library, tuning only)

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvDesc(C.Structure):
    """mivos_conv_desc"""
    _fields_ = [("x", vp), ("w", vp), ("scale", vp), ("bias", vp), ("res", vp), ("y", vp), ("y2", vp),
                ("N", i32), ("H", i32), ("W", i32), ("Cin", i32), ("Cout", i32), ("KH", i32), ("KW", i32),
                ("stride", i32), ("pad", i32), ("Ho", i32), ("Wo", i32), ("split", i32),
                ("relu_in", i32), ("relu_out", i32), ("precision", i32),
                ("x_nstride", i64), ("x_pstride", i64), ("y_nstride", i64), ("y_pstride", i64),
                ("y2_nstride", i64), ("y2_pstride", i64), ("res_nstride", i64), ("res_pstride", i64),
                ("workspace", vp), ("workspace_bytes", i64),
                ("x_rstride", i64), ("y_rstride", i64), ("res_rstride", i64),
                ("x_border", i32), ("x_format", i32), ("y_format", i32), ("res_format", i32), ("dilation", i32), ("chip_share", i32), ("status", vp)]


class InterleaveDesc(C.Structure):
    """mivos_interleave_desc"""
    _fields_ = [("plane", vp * 16), ("nstride", i64 * 16), ("cval", f32 * 16), ("C", i32)]


class FusionLayer(C.Structure):
    """mivos_fusion_layer"""
    _fields_ = [("w16", vp), ("scale16", vp), ("bias", vp)]


class FusionNetDesc(C.Structure):
    """mivos_fusion_net_desc"""
    _fields_ = [("layer", FusionLayer * 5), ("final_w", vp), ("final_bias", vp), ("x16", vp), ("planes", vp), ("logits", vp), ("scratch", vp),
                ("scratch_floats", i64), ("batch", i32), ("height", i32), ("width", i32), ("workspace", vp),
                ("workspace_bytes", i64)]


# name -> (restype, argtypes); every symbol include/mivos_hip.h declares
PROTOTYPES = {
    "mivos_version": (C.c_int, []),
    "mivos_last_error": (C.c_char_p, []),
    "mivos_device_check": (C.c_int, [C.c_int]),
    "mivos_conv2d_fused": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "mivos_fusion_net_scratch_floats": (i64, [C.c_int, C.c_int, C.c_int]),
    "mivos_fusion_net_forward": (C.c_int, [C.POINTER(FusionNetDesc), vp]),
    "mivos_fusion_resblock": (C.c_int, [vp, vp, C.POINTER(FusionLayer), C.POINTER(FusionLayer), C.c_int, C.c_int, C.c_int, vp]),
    "mivos_fusion_head": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_stem7x7s2_planes": (C.c_int, [C.POINTER(InterleaveDesc), C.c_int, vp, C.c_int, C.c_float, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_fusion_conv1_planes": (C.c_int, [C.POINTER(InterleaveDesc), C.POINTER(FusionLayer), vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_fusion_wgrad_scratch_floats": (i64, []),
    "mivos_fusion_wgrad3x3": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_fusion_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, i64, vp]),
    "mivos_fusion_kth_loss": (C.c_int, [vp, vp, vp, C.c_int, i64, vp]),
    "mivos_fusion_loss_grad": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, i64, vp]),
    "mivos_mul_positive": (C.c_int, [vp, vp, i64, vp]),
    "mivos_adam_step": (C.c_int, [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, vp]),
    "mivos_pack_weights_f16x3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "mivos_pack_activation_sh32": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_unpack_activation_sh32": (C.c_int, [vp, i64, i64, i64, vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_pack_weights_f16x3_dma_bytes": (i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mivos_pack_weights_f16x3_dma": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "mivos_conv2d_variant": (C.c_int, [C.c_int, C.c_int]),
    "mivos_conv2d_variant_f16x3": (C.c_int, [C.c_int, C.c_int]),
    "mivos_conv2d_variant_pp": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mivos_conv2d_set_fold_mode": (C.c_int, [C.c_int]),
    "mivos_maxpool3x3s2": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_upsample2x_add": (C.c_int, [vp, i64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_upsample2x_add_multi": (C.c_int, [vp, i64, vp, vp, vp, vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_maxpool3x3s2_sh32": (C.c_int, [vp, vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_tap_sum9": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_memory_read_workspace_bytes": (i64, [C.c_int, i64, C.c_int, C.c_int]),
    "mivos_memory_read_topk": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_set_q128_min": (i64, [i64]),
    "mivos_memory_read_set_q256_min": (i64, [i64]),
    "mivos_memory_read_set_hifirst": (C.c_int, [C.c_int]),
    "mivos_memory_read_set_workgroups": (C.c_int, [C.c_int]),
    "mivos_memory_read_dense_workspace_bytes": (i64, [C.c_int, i64, C.c_int]),
    "mivos_memory_read_dense": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, i64, i64, i64, C.c_int, C.c_int, i64, C.c_int, vp, i64, vp]),
    "mivos_memory_read_topk_any_workspace_bytes": (i64, [C.c_int, i64, C.c_int]),
    "mivos_memory_read_topk_any": (C.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, vp, vp, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_plan": (C.c_int, [C.c_int, i64, C.c_int, C.c_int, C.c_int, C.POINTER(i32)]),
    "mivos_memory_read_select": (C.c_int, [vp, i64, vp, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_finalize": (C.c_int, [vp, i64, vp, i64, i64, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_finalize_sh32": (C.c_int, [vp, i64, vp, vp, i64, i64, i64, C.c_int, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_topk_indices": (C.c_int, [vp, i64, vp, vp, vp, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_read_finalize_indices": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_memory_split_keys": (C.c_int, [vp, i64, vp, i64, C.c_int, i64, vp]),
    "mivos_memory_read_select_f16x3": (C.c_int, [vp, i64, vp, C.c_int, i64, C.c_int, C.c_int, vp, i64, vp]),
    "mivos_attention_align": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "mivos_area_pool16": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_resize_bilinear": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_aggregate_wbg": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, C.c_int, vp]),
    "mivos_aggregate_sbg": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, C.c_int, vp]),
    "mivos_aggregate_wbg_channel": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, i64, C.c_int, C.c_int, vp]),
    "mivos_attention_weights": (C.c_int, [vp, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_argmax_u8": (C.c_int, [vp, i64, vp, C.c_int, i64, vp]),
    "mivos_mask_diff": (C.c_int, [vp, vp, vp, vp, i64, vp]),
    "mivos_sigmoid": (C.c_int, [vp, vp, i64, vp]),
    "mivos_mask_others": (C.c_int, [vp, vp, C.c_int, i64, vp]),
    "mivos_resize_bilinear_nhwc": (C.c_int, [vp, vp, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_global_avgpool": (C.c_int, [vp, vp, C.c_int, i64, C.c_int, vp]),
    "mivos_dilate3x3": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "mivos_ingest_u8": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, i64, i64, i64, C.c_int, C.c_int, C.POINTER(f32), C.POINTER(f32), vp]),
    "mivos_resize_bicubic": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64, i64, C.c_int, C.c_int, vp]),
    "mivos_onehot_nearest": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, i64, i64, C.c_int, C.c_int, vp]),
    "mivos_interleave_planes": (C.c_int, [C.POINTER(InterleaveDesc), vp, C.c_int, i64, vp]),
}

_lib = None


class MivosHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MivosHipError(f"{LIB_PATH} not found: the HIP library is the only compute path of mivos_amd "
                                "(build it with __graft_entry__.build()); there is no CPU/PyTorch fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().mivos_last_error().decode(errors="replace")
        if rc == -4:      # MIVOS_ERR_TOPK_RANGE: same exception type + wording as torch.topk in the reference
            raise RuntimeError(msg)
        raise MivosHipError(f"libmivos_hip error {rc}: {msg}")
