"""ctypes binding of libyolopoint_hip.so (the C ABI declared in include/yolopoint_hip.h).

The library is the product: there is no CPU fallback.  Importing this module never touches a
GPU (so the symbol table can be checked on a CPU-only host); calling a compute entry point
without a HIP device raises.
"""
import ctypes as C
import os
from .switches import sw

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libyolopoint_hip.so")

YP_F16, YP_BF16, YP_F32 = 0, 1, 2
YP_FP8, YP_FP8_BF8 = 3, 4          # 8-bit convolution INPUTS (e4m3 x e4m3 | filter e4m3 x activation e5m2); results are bf16
YP_ACT_NONE, YP_ACT_SILU = 0, 1


class YpError(RuntimeError):
    pass


class YpView(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("cstride", C.c_int32),
                ("coff", C.c_int32), ("C", C.c_int32), ("ups", C.c_int32)]


class YpConvDesc(C.Structure):
    _fields_ = [("in0", YpView), ("in1", YpView), ("out", YpView), ("res", YpView), ("out2", YpView),
                ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("dtype", C.c_int32), ("out_f32", C.c_int32), ("B", C.c_int32),
                ("Hi", C.c_int32), ("Wi", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32),
                ("stride_h", C.c_int32), ("stride_w", C.c_int32), ("pad_h", C.c_int32), ("pad_w", C.c_int32),
                ("Kpad", C.c_int32), ("Npad", C.c_int32), ("act", C.c_int32), ("tile", C.c_int32),
                ("dil_h", C.c_int32), ("dil_w", C.c_int32), ("in0_zero_stuffed", C.c_int32), ("ksplit", C.c_int32),
                ("atomic_accumulate", C.c_int32), ("tail_zero", C.c_int32),
                ("pre_weight", C.c_void_p), ("pre_bias", C.c_void_p), ("pre_Kpad", C.c_int32), ("pre_Npad", C.c_int32),
                ("pre_act", C.c_int32), ("post_act", C.c_int32),
                ("post_weight", C.c_void_p), ("post_bias", C.c_void_p), ("post_Kpad", C.c_int32), ("post_Npad", C.c_int32),
                ("bn_partial", C.c_void_p), ("split_slabs", C.c_void_p), ("split_stride", C.c_int64),
                ("scale_in", C.c_void_p), ("scale_w", C.c_void_p), ("out_phase", C.c_int32), ("reserved_", C.c_int32),
                ("stem_x", C.c_void_p), ("stem_weight", C.c_void_p), ("stem_bias", C.c_void_p),
                ("stem_Kpad", C.c_int32), ("stem_act", C.c_int32), ("stem_C", C.c_int32), ("stem_reserved_", C.c_int32)]


class YpDetectDesc(C.Structure):
    _fields_ = [("na", C.c_int32), ("no", C.c_int32), ("stride", C.c_float), ("anchors_px", C.c_float * 16),
                ("x_out", C.c_void_p), ("z_out", C.c_void_p), ("rows_total", C.c_int32), ("row_offset", C.c_int32)]


class YpOpArgs(C.Structure):
    _fields_ = [("op", C.c_int32), ("pad_", C.c_int32), ("v", YpView * 4), ("f", C.c_void_p * 4), ("g", C.c_void_p * 4),
                ("p", C.c_void_p * 2), ("n", C.c_size_t * 2), ("i", C.c_int32 * 8), ("s", C.c_float * 4)]


(OP_BN_STATS, OP_BN_APPLY, OP_BN_BWD, OP_UPS2_BWD, OP_ADD_VIEWS, OP_MAXPOOL5_BWD, OP_L2NORM_BWD, OP_DETECT_BWD_PACK, OP_TO_CHWB,
 OP_COL_SUM, OP_MEMSET0, OP_PACK_NCHW, OP_L2NORM, OP_SPPF_POOL, OP_CAST_F32, OP_MAXPOOL2, OP_PACK_WEIGHT, OP_WGRAD, OP_WGRAD_UNPACK, OP_MAXPOOL2_BWD, OP_WGRAD_UNPACK_BATCH, OP_WGRAD_GROUP,
 OP_SUM_SLABS, OP_STEM_WGRAD, OP_QUANT_FP8) = range(10, 35)
LANE_MAIN, LANE_SIDE, LANE_JOIN = 0, 1, 2

_i, _f, _p, _sz, _i64 = C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_int64
# name -> (restype, argtypes); must list every symbol of include/yolopoint_hip.h
SIGNATURES = {
    "yp_last_error": (C.c_char_p, []),
    "yp_version": (_i, []),
    "yp_device_count": (_i, []),
    "yp_conv2d": (_i, [C.POINTER(YpConvDesc), _p]),
    "yp_conv2d_detect": (_i, [C.POINTER(YpConvDesc), C.POINTER(YpDetectDesc), _p]),
    "yp_conv_kpad": (_i, [_i, _i]),
    "yp_sum_slabs": (_i, [_p, _p, _sz, _i, _p]),
    "yp_bn_act_apply_grouped_q8": (_i, [YpView, YpView, YpView, _i, _i, _i, _p, _p, _p, _p, _i, YpView, _p, _p, _p]),
    "yp_bn_act_bwd_grouped_q8": (_i, [YpView, YpView, YpView, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _p, _sz, YpView, _p, _p, _p]),
    "yp_quantize_fp8": (_i, [YpView, YpView, _i, _i, _i, _p, _p, _p]),
    "yp_fp8_update_scales": (_i, [_p, _p, _p, _i, _f, _p]),
    "yp_pack_weight_fp8_batch": (_i, [_p, _i, _i, _p]),
    "yp_adam_step": (_i, [_p, _p, _p, _p, _sz, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _p]),
    "yp_sum_slabs_tree": (_i, [_p, _p, _sz, _i, _i, _p]),
    "yp_stem_wgrad": (_i, [YpView, YpView, _i, _i, _p, _p, _p]),
    "yp_stem_wgrad_slabs": (_i, [_i, _i, _i]),
    "yp_stem_conv": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, YpView, _i, _p]),
    "yp_pack_input": (_i, [_p, _i, _i, _i, _i, YpView, _i, _p]),
    "yp_unpack_nchw": (_i, [YpView, _i, _i, _i, _p, _p]),
    "yp_sppf_pool": (_i, [YpView, YpView, YpView, YpView, _i, _i, _p]),
    "yp_maxpool2": (_i, [YpView, YpView, _i, _i, _p]),
    "yp_l2norm_f32": (_i, [YpView, YpView, _i, _i, _p]),
    "yp_detect_decode": (_i, [YpView, _i, _i, _i, _f, C.POINTER(_f), _p, _p, _i, _i, _p]),
    "yp_bn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "yp_bn_stats": (_i, [YpView, _i, _i, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "yp_bn_act_apply": (_i, [YpView, YpView, YpView, _i, _i, _p, _p, _p, _p, _i, _p]),
    "yp_bn_act_bwd": (_i, [YpView, YpView, YpView, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _p, _sz, _p]),
    "yp_ups2_bwd": (_i, [YpView, YpView, _i, _i, _i, _p]),
    "yp_add_views": (_i, [YpView, YpView, _i, _i, _i, _p]),
    "yp_maxpool5_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "yp_maxpool5_bwd": (_i, [YpView, YpView, YpView, _i, _i, _i, _p, _sz, _p]),
    "yp_l2norm_bwd_f32": (_i, [YpView, YpView, YpView, _i, _i, _p]),
    "yp_detect_bwd_pack": (_i, [_p, _i, _i, _i, YpView, _i, _p, _p]),
    "yp_to_chwb": (_i, [YpView, _i, _i, _i, _p, _i, _p]),
    "yp_cast_from_f32": (_i, [YpView, YpView, _i, _i, _p]),
    "yp_conv_wgrad": (_i, [YpView, YpView, _i, _i, _i, _i, _p, _p]),
    "yp_plan_set_lane": (_i, [_p, _i, _i]),
    "yp_plan_add_callback": (_i, [_p, _p, _p]),
    "yp_stream_pick": (_i, [_p, _i, _p]),
    "yp_stream_forget": (_i, [_p]),
    "yp_bn_finalize": (_i, [_p, _i, _i, C.c_double, _f, _f, _p, _p, _p, _p, _p]),
    "yp_bn_stats_grouped": (_i, [YpView, _i, _i, _i, _f, _f, _p, _p, _p, _p, _p, _sz, _p]),
    "yp_bn_finalize_grouped": (_i, [_p, _i, _i, _i, C.c_double, _f, _f, _p, _p, _p, _p, _p]),
    "yp_bn_act_apply_grouped": (_i, [YpView, YpView, YpView, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "yp_bn_act_bwd_grouped": (_i, [YpView, YpView, YpView, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _p, _sz, _p]),
    "yp_conv_bn_partial_rows": (_i, [_p, _p]),
    "yp_wgrad_group_entry_bytes": (_sz, []),
    "yp_wgrad_group_pack": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "yp_wgrad_block": (_i, [YpView, YpView, _i, _i]),
    "yp_wgrad_group_run": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "yp_wgrad_partial_elems": (_sz, [YpView, YpView, _i, _i, _i, _i, _i]),
    "yp_wgrad_group_pack_det": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "yp_conv_wgrad_q8": (_i, [YpView, YpView, _p, _p, _i, _i, _i, _p, _p]),
    "yp_wgrad_group_pack_q8": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "yp_wgrad_group_run_det": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "yp_pack_weight_batch": (_i, [_p, _i, _i, _i, _p]),
    "yp_wgrad_unpack_batch": (_i, [_p, _i, _i, _p]),
    "yp_infonce_fwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "yp_sampling_set_max_workgroups": (_i, [_i]),
    "yp_infonce_fwd_grad": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _i, _p]),
    "yp_infonce_bwd_db": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    "yp_infonce_rows16": (_i, [_p, C.c_size_t, _p, _p]),
    "yp_infonce_fwd_grad_h": (_i, [_p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _i, _p]),
    "yp_infonce_bwd_db_h": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    "yp_infonce_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "yp_homo_combine": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "yp_points_sample_taps": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "yp_points_sample_bwd_sorted": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p]),
    "yp_points_sample_fwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _p]),
    "yp_points_sample_bwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "yp_detloss_workspace_bytes": (_sz, [_i, _i, _i]),
    "yp_detloss": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "yp_cell_mask_workspace_bytes": (_sz, [_i, _i, _i]),
    "yp_cell_mask": (_i, [_p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "yp_detloss2d": (_i, [_p, _p, _p, _p, _p, _f, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "yp_nce_cells": (_i, [_p, _p, _i, _i, _i, _p, _p, _p]),
    "yp_nce_select": (_i, [_p, _p, _i, _i, _i, _i, C.c_uint64, _p, _p, _p]),
    "yp_nce_negatives": (_i, [_i, _i, C.c_uint64, _p, _p, _i, _p]),
    "yp_csr_build": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "yp_csr_workspace_ints": (_sz, [_i, _i]),
    "yp_fill_zero": (_i, [_p, _sz, _p]),
    "yp_multi_add": (_i, [_p, _i, _i, _p]),
    "yp_counters_add": (_i, [_p, _i, _i64, _p]),
    "yp_loss_combine5": (_i, [_p, _i, _p, _i, _p, _f, _f, _f, _f, _p, _p, _p, _f, C.c_double, _p]),
    "yp_objloss_level": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "yp_objloss_level_dev": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    "yp_build_targets": (_i, [_p, _i, _p, _i, _i, _p, _f, _i, _p, _p, _p, _p, _p, _p]),
    "yp_box_nms_classes": (_i, [_p, _i, _i, _i, _f, _f, _i, _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "yp_maxpool2_bwd": (_i, [YpView, YpView, YpView, _i, _i, _i, _p]),
    "yp_wgrad_unpack": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "yp_pack_weight": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p]),
    "yp_col_sum": (_i, [YpView, _i, _i, _p, _i, _p, _sz, _p]),
    "yp_run_op": (_i, [C.POINTER(YpOpArgs), _p]),
    "yp_plan_add_op": (_i, [_p, C.POINTER(YpOpArgs)]),
    "yp_kp_decode": (_i, [_p, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _p, _p]),
    "yp_kp_nms_workspace_bytes": (_sz, [_i, _i, _i]),
    "yp_kp_nms_candidate_count_offset": (_sz, [_i, _i, _i]),
    "yp_kp_nms": (_i, [_p, _i, _i, _i, _f, _i, _i, _p, _p, _i, _p, _sz, _p]),
    "yp_kp_nms_async": (_i, [_p, _i, _i, _i, _f, _i, _i, _p, _p, _i, _p, _sz, _i, _p, _p]),
    "yp_box_nms_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "yp_box_nms": (_i, [_p, _i, _i, _i, _f, _f, _i, _i, _i, _i, _f, _p, _p, _p, _sz, _p]),
    "yp_pts_box_filter": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "yp_desc_sample": (_i, [_p, _i, _i, _i, _i64, _i64, _i64, _p, _i, _i, _p, _p]),
    "yp_mnn_workspace_bytes": (_sz, [_i, _i]),
    "yp_mnn_match": (_i, [_p, _i, _p, _i, _i, _f, _p, _p, _i, _p, _sz, _p]),
    "yp_plan_create": (_i, [C.POINTER(_p)]),
    "yp_plan_destroy": (_i, [_p]),
    "yp_plan_add_conv": (_i, [_p, C.POINTER(YpConvDesc)]),
    "yp_plan_add_conv_detect": (_i, [_p, C.POINTER(YpConvDesc), C.POINTER(YpDetectDesc)]),
    "yp_plan_add_sppf_pool": (_i, [_p, YpView, YpView, YpView, YpView, _i, _i]),
    "yp_plan_add_l2norm": (_i, [_p, YpView, YpView, _i, _i]),
    "yp_plan_add_detect_decode": (_i, [_p, YpView, _i, _i, _i, _f, C.POINTER(_f), _p, _p, _i, _i]),
    "yp_plan_num_ops": (_i, [_p]),
    "yp_plan_patch_op_view": (_i, [_p, _i, _i, YpView]),
    "yp_plan_set_deps": (_i, [_p, _i, C.POINTER(_i), _i]),
    "yp_plan_graph_is_parallel": (_i, [_p]),
    "yp_plan_instantiate_graph": (_i, [_p, _p]),
    "yp_plan_run": (_i, [_p, _p]),
    "yp_plan_profile": (_i, [_p, _p, C.POINTER(_f)]),
    "yp_plan_time": (_i, [_p, _p, _i, C.POINTER(_f)]),
}

_lib = None


def lib():
    """Load the shared library (built by __graft_entry__.build()); fail loudly when absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm bundles its own libamdhip64: it must be in the process first, so that this library binds to the SAME
        # HIP runtime (loading ours first pulls /opt/rocm's copy in, and torch then finds no usable device)
        import torch  # noqa: F401
        path = sw("YP_HIP_LIB") or LIB_PATH      # override: A/B-compare two builds of the library in one session
        if not os.path.exists(path):
            raise YpError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
        l = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise YpError(f"libyolopoint_hip error {rc}: {lib().yp_last_error().decode()}")


def require_gpu():
    if lib().yp_device_count() < 1:
        raise YpError("no HIP device: the YOLOPoint hot path runs on MI355X only (no CPU fallback)")


def stream_ptr(stream=None):
    """The HIP stream a native launch goes to: torch's current stream of the CURRENT device.  Kernels run on the current device,
    so a launch whose stream belongs to another device is refused (the public entry points are wrapped in `guarded`, which makes the
    device of their tensors current for the duration of the call)."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    if s.device.index != torch.cuda.current_device():
        raise YpError(f"native launch on stream of cuda:{s.device.index} while cuda:{torch.cuda.current_device()} is current")
    return C.c_void_p(s.cuda_stream)


def _cuda_device_of(obj, depth=0):
    import torch
    if isinstance(obj, torch.Tensor):
        return obj.device if obj.is_cuda else None
    if isinstance(obj, torch.device):
        return obj if obj.type == "cuda" and obj.index is not None else None
    if depth < 2 and isinstance(obj, (list, tuple)):
        for o in obj:
            d = _cuda_device_of(o, depth + 1)
            if d is not None:
                return d
    if depth < 2 and isinstance(obj, dict):
        for o in obj.values():
            d = _cuda_device_of(o, depth + 1)
            if d is not None:
                return d
    if depth == 0 and not isinstance(obj, (str, bytes, int, float)) and hasattr(obj, "device"):
        try:
            return _cuda_device_of(torch.device(obj.device) if isinstance(obj.device, str) else obj.device, 2)
        except Exception:
            return None
    return None


def guarded(fn):
    """Decorator of the public entry points: run `fn` with the device of its first cuda tensor (or of `self.device`) current, as
    PyTorch's own operators do per call -- a model on cuda:1 works while cuda:0 is the current device."""
    import functools

    @functools.wraps(fn)
    def wrap(*a, **k):
        import torch
        dev = None
        for o in list(a) + list(k.values()):
            dev = _cuda_device_of(o)
            if dev is not None:
                break
        if dev is None or not torch.cuda.is_available() or dev.index == torch.cuda.current_device():
            return fn(*a, **k)
        with torch.cuda.device(dev):
            return fn(*a, **k)
    return wrap


def dtype_code(dt):
    import torch
    if dt in (YP_F16, YP_BF16, YP_F32):
        return dt
    table = {torch.float16: YP_F16, torch.bfloat16: YP_BF16, torch.float32: YP_F32,
             "f16": YP_F16, "fp16": YP_F16, "bf16": YP_BF16, "f32": YP_F32, "fp32": YP_F32}
    if dt not in table:
        raise YpError(f"unsupported compute dtype {dt!r}")
    return table[dt]


def torch_dtype(code):
    import torch
    return {YP_F16: torch.float16, YP_BF16: torch.bfloat16, YP_F32: torch.float32}[code]
