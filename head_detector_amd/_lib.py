"""ctypes binding of libvgh.so (include/vgh.h).  There is NO fallback: if the HIP library is missing
or fails to load, everything that needs it raises -- loudly -- instead of computing on the CPU."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# VGH_LIB_PATH: load another build of the library (tools/ use it for the -DVGH_EXPERIMENTS build, which is never shipped)
LIB_PATH = os.environ.get("VGH_LIB_PATH") or os.path.join(HERE, "libvgh.so")
ABI_VERSION = 7  # = VGH_ABI_VERSION of include/vgh.h

VGH_OP_STEM, VGH_OP_CONV, VGH_OP_SPP_POOL, VGH_OP_FORK = 0, 1, 2, 3
VGH_ACT_NONE, VGH_ACT_RELU, VGH_ACT_SILU = 0, 1, 2
VGH_IMG_F32_NCHW, VGH_IMG_U8_NHWC = 0, 1
VGH_FMT_BF16, VGH_FMT_F32, VGH_FMT_BF16X2, VGH_FMT_F16X2, VGH_FMT_FP8, VGH_FMT_F16, VGH_FMT_I8 = 0, 1, 2, 3, 4, 5, 6
NUM_FLAME_PARAMS = 413


class BufDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("pitch", C.c_int32), ("is_f32", C.c_int32), ("scale", C.c_float)]


class OpDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("in_buf", C.c_int32), ("in_coff", C.c_int32), ("cin", C.c_int32),
        ("out_buf", C.c_int32), ("out_coff", C.c_int32), ("cout_pad", C.c_int32),
        ("cout_store", C.c_int32),
        ("out_split", C.c_int32), ("out_coff2", C.c_int32),
        ("res_buf", C.c_int32), ("res_coff", C.c_int32),
        ("alpha", C.c_float),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32),
        ("shuffle", C.c_int32),
        ("w_off", C.c_int64),
        ("b_off", C.c_int64),
        ("force_cfg", C.c_int32),
        ("lane", C.c_int32),
        ("grp_cout", C.c_int32), ("grp_in_stride", C.c_int32),
    ]


class ConvCall(C.Structure):
    _fields_ = [
        ("in_dev", C.c_void_p), ("in_pitch", C.c_int64), ("in_coff", C.c_int32), ("cin", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("wpack_dev", C.c_void_p), ("bias_dev", C.c_void_p),
        ("out_dev", C.c_void_p), ("out_pitch", C.c_int64),
        ("out_coff", C.c_int32), ("cout_pad", C.c_int32), ("cout_store", C.c_int32), ("out_split", C.c_int32),
        ("out_coff2", C.c_int32), ("out_f32", C.c_int32),
        ("res_dev", C.c_void_p), ("res_pitch", C.c_int64), ("res_coff", C.c_int32),
        ("alpha", C.c_float),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("act", C.c_int32), ("shuffle", C.c_int32),
        ("force_cfg", C.c_int32),
        ("grp_cout", C.c_int32), ("grp_in_stride", C.c_int32),
        ("fmt", C.c_int32), ("out_scale", C.c_float),
        ("out_fp8", C.c_int32), ("gscale_dev", C.c_void_p), ("diag_dev", C.c_void_p),
    ]


class HeadLevel(C.Structure):
    _fields_ = [("pred_dev", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32), ("pitch", C.c_int32), ("stride", C.c_int32)]


VGH_MAX_LEVELS = 4


class DetectCfg(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("level_buf", C.c_int32 * VGH_MAX_LEVELS), ("level_h", C.c_int32 * VGH_MAX_LEVELS), ("level_w", C.c_int32 * VGH_MAX_LEVELS),
                ("level_pitch", C.c_int32 * VGH_MAX_LEVELS), ("level_stride", C.c_int32 * VGH_MAX_LEVELS), ("shape_live", C.c_int32), ("expr_live", C.c_int32),
                ("pre_k", C.c_int32), ("keep_k", C.c_int32), ("max_batch", C.c_int32)]


class DetectOut(C.Structure):
    _fields_ = [("boxes_dev", C.c_void_p), ("scores_dev", C.c_void_p), ("flame_dev", C.c_void_p), ("counts_dev", C.c_void_p),
                ("n_heads_dev", C.c_void_p), ("head_image_dev", C.c_void_p), ("head_capacity", C.c_int32), ("unpad_dev", C.c_void_p),
                ("verts_dev", C.c_void_p), ("rot_dev", C.c_void_p), ("rpy_dev", C.c_void_p), ("proj_dev", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("pack_path", C.c_char_p), ("max_batch", C.c_int32), ("pre_nms_top_k", C.c_int32), ("keep_top_k", C.c_int32),
                ("max_heads", C.c_int32), ("batch_split", C.c_int32), ("overlap", C.c_int32)]


class CtxInfo(C.Structure):
    _fields_ = [("variant", C.c_char * 32), ("image_size", C.c_int32), ("max_batch", C.c_int32), ("arena_batch", C.c_int32), ("num_anchors", C.c_int32),
                ("pre_nms_top_k", C.c_int32), ("keep_top_k", C.c_int32), ("num_vertices", C.c_int32), ("shape_live", C.c_int32), ("expr_live", C.c_int32),
                ("precision", C.c_int32), ("flops_per_image", C.c_double)]


SCRATCH_BOXES_ALL, SCRATCH_SCORES_ALL, SCRATCH_TOPK_IDX, SCRATCH_KEEP_IDX, SCRATCH_HEAD_ROW = range(5)

# every symbol include/vgh.h declares: (restype, argtypes)
_P, _I, _I64, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SYMBOLS = {
    "vgh_version": (C.c_char_p, []),
    "vgh_last_error": (C.c_char_p, []),
    "vgh_abi_version": (_I, []),
    "vgh_net_create": (_I, [_I, _I, _I, C.POINTER(BufDesc), _I, C.POINTER(OpDesc), _I, _P, _I64, _P, _I64, C.POINTER(_P)]),
    "vgh_net_destroy": (None, [_P]),
    "vgh_net_forward": (_I, [_P, _P, _I, _I, _P]),
    "vgh_net_profile": (_I, [_P, _P, _I, _I, _P, _P]),
    "vgh_net_capture": (_I, [_P, _P, _I, _I, _P]),
    "vgh_net_forward_graph": (_I, [_P, _P]),
    "vgh_net_buffer": (_P, [_P, _I]),
    "vgh_net_buffer_bytes": (_I64, [_P, _I]),
    "vgh_net_set_cfg": (_I, [_P, _I, _I]),
    "vgh_net_set_b2b": (_I, [_P, _I]),
    "vgh_net_stem_fused": (_I, [_P]),
    "vgh_net_b2b_pairs": (_I, [_P]),
    "vgh_net_set_split": (_I, [_P, _I]),
    "vgh_net_max_batch": (_I, [_P]),
    "vgh_net_image_size": (_I, [_P]),
    "vgh_conv2d": (_I, [C.POINTER(ConvCall), _P]),
    "vgh_pack_conv_weights": (_I, [_P, _I, _I, _I, _P]),
    "vgh_pack_conv_weights_split": (_I, [_P, _I, _I, _I, _I, _P, C.POINTER(_F)]),
    "vgh_pack_conv_weights_fp8": (_I, [_P, _I, _I, _I, _P, _P]),
    "vgh_pack_conv_weights_i8": (_I, [_P, _I, _I, _I, _P, _P]),
    "vgh_conv_num_cfgs": (_I, []),
    "vgh_conv_cfg_name": (C.c_char_p, [_I]),
    "vgh_conv_cfg_cout_tile": (_I, [_I]),
    "vgh_conv_split_num_cfgs": (_I, []),
    "vgh_conv_split_cfg_name": (C.c_char_p, [_I]),
    "vgh_conv_split_cfg_ok": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "vgh_conv_cfg_ok": (_I, [_I, _I, _I, _I, _I, _I]),
    "vgh_conv_set_max_blocks_per_xcd": (_I, [_I]),
    "vgh_net_set_i8_diag": (_I, [_I]),
    "vgh_net_op_has_diag": (_I, [_P, _I]),
    "vgh_head_decode": (_I, [C.POINTER(HeadLevel), _I, _I, _P, _P, _P]),
    "vgh_topk": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "vgh_gather_candidates": (_I, [C.POINTER(HeadLevel), _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "vgh_nms": (_I, [_P, _P, _I, _I, _F, _F, _I, _P, _P, _P]),
    "vgh_compact": (_I, [_P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "vgh_topk_nms_workspace_bytes": (_I64, [_I, _I, _I, _I]),
    "vgh_topk_nms": (_I, [_P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vgh_flame_create": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, C.POINTER(_P)]),
    "vgh_flame_destroy": (None, [_P]),
    "vgh_flame_decode": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "vgh_flame_decode_indirect": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vgh_detector_create": (_I, [_P, _P, C.POINTER(DetectCfg), C.POINTER(_P)]),
    "vgh_detector_destroy": (None, [_P]),
    "vgh_detector_candidates": (_I, [_P, _P, _I, _I, _P]),
    "vgh_detector_decode_candidates": (_I, [_P, _I, _I, _P]),
    "vgh_detector_candidate_buffers": (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "vgh_detector_set_flame": (_I, [_P, _P]),
    "vgh_detector_scratch": (_P, [_P, _I]),
    "vgh_detector_select": (_I, [_P, _I, _F, _F, C.POINTER(DetectOut), _P]),
    "vgh_detector_set_overlap": (_I, [_P, _I]),
    "vgh_detector_set_lazy_flame": (_I, [_P, _I]),
    "vgh_detector_join": (_I, [_P, _P]),
    "vgh_detector_record": (_I, [_P, _P, _P]),
    "vgh_detect": (_I, [_P, _P, _I, _I, _F, _F, C.POINTER(DetectOut), _P]),
    "vgh_flame_lbs": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "vgh_flame_set_matrix_path": (_I, [_I]),
    "vgh_create": (_I, [C.POINTER(Config), C.POINTER(_P)]),
    "vgh_destroy": (None, [_P]),
    "vgh_ctx_last_error": (C.c_char_p, [_P]),
    "vgh_ctx_get_info": (_I, [_P, C.POINTER(CtxInfo)]),
    "vgh_ctx_detect": (_I, [_P, _P, _I, _I, _F, _F, C.POINTER(DetectOut), _P]),
    "vgh_ctx_join": (_I, [_P, _P]),
    "vgh_ctx_net": (_P, [_P]),
    "vgh_ctx_flame": (_P, [_P]),
    "vgh_ctx_detector": (_P, [_P]),
    "vgh_rasterize": (_I, [_P, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
    "vgh_pncc_render": (_I, [_P, _I, _I, _P, _I, _P, _P, _I, _I, _P, _P]),
    "vgh_refined_head_bbox": (_I, [_P, _I, _I, _P, _I, _P, _P]),
    "vgh_letterbox": (_I, [_P, _I, _I, _I, _I64, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "vgh_stream_create": (_I, [_I, C.POINTER(_P)]),
    "vgh_stream_acquire": (_I, [_I, C.POINTER(_P), _I, C.POINTER(_P)]),
    "vgh_stream_release": (_I, [_I, _P]),
    "vgh_streams_overlap": (_I, [_P, _P]),
    "vgh_stream_blocked_behind": (_I, [_P, _P]),
    "vgh_stream_spin": (_I, [_P, _I]),
    "vgh_detector_streams": (_I, [_P, _P, C.POINTER(_P)]),
    "vgh_stream_destroy": (_I, [_P]),
    "vgh_stream_sync": (_I, [_P]),
    "vgh_event_create": (_I, [C.POINTER(_P)]),
    "vgh_event_destroy": (_I, [_P]),
    "vgh_event_record": (_I, [_P, _P]),
    "vgh_event_elapsed_ms": (_I, [_P, _P, C.POINTER(_F)]),
}

# A/B knobs of the -DVGH_EXPERIMENTS build (libvgh_exp.so, include/vgh.h's last block): bound when the loaded library has them, absent from the product library
EXPERIMENT_SYMBOLS = {
    "vgh_net_set_lane_lag": (_I, [_I]),
    "vgh_net_set_fuse_stem": (_I, [_P, _I]),
    "vgh_stem_set_mfma": (_I, [_I]),
    "vgh_conv_set_nt_store": (_I, [_I]),
    "vgh_detector_set_side_priority": (_I, [_P, _I]),
    "vgh_streams_interleave_permille": (_I, [_P, _P]),
}

_lib: Optional[C.CDLL] = None


class VghError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libvgh.so and bind every declared symbol. Raises VghError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VghError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m head_detector_amd.build` "
            "(needs hipcc). There is no CPU fallback in this package."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise VghError(f"failed to load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VghError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    if lib.vgh_abi_version() != ABI_VERSION:  # the ctypes structs below mirror include/vgh.h at this revision: another one means other struct sizes
        raise VghError(f"{LIB_PATH} has ABI revision {lib.vgh_abi_version()}, this binding is written for {ABI_VERSION} (stale build?)")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().vgh_last_error().decode("utf-8", "replace")
        raise VghError(f"libvgh error {rc}: {msg}")


def ptr(t) -> int:
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
