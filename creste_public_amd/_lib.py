"""ctypes binding of libcreste_hip.so (include/creste_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, every op
raises `HipLibraryError` -- it never silently routes to a PyTorch/CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libcreste_hip.so")

ABI_VERSION = 12         # creste_abi_version() of the library this binding was written against
ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2
PREC_F32, PREC_BF16, PREC_BF16X3, PREC_BF16X6, PREC_F16X3 = 0, 1, 2, 3, 4


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("wpk", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("a_scale", C.c_void_p), ("row_mask", C.c_void_p), ("out", C.c_void_p),
                ("work", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "in_cs", "Ho", "Wo", "Cout", "out_cs",
                                         "out_co", "res_cs", "KH", "KW", "stride", "pad_t", "pad_l",
                                         "act", "prec", "algo", "flags")] + \
               [("a_amax", C.c_void_p), ("out_amax", C.c_void_p), ("w_unscale", C.c_void_p), ("up_src", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("up_H", "up_W", "up_C", "up_cs")] + [("out_stats", C.c_void_p)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "creste_last_error": (C.c_char_p, []),
    "creste_abi_version": (_i, []),
    "creste_conv2d_nhwc": (_i, [C.POINTER(ConvDesc), _vp]),
    "creste_conv_stat_rows": (_i, [C.POINTER(ConvDesc)]),
    "creste_conv_supported": (_i, [_i, _i, _i, _i]),
    "creste_conv_wino_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "creste_conv_wino_weight_bytes": (_i64, [_i, _i, _i]),
    "creste_conv_wino_pack_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "creste_conv_wino_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "creste_conv_wino4_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "creste_conv_wino4_weight_bytes": (_i64, [_i, _i, _i]),
    "creste_conv_wino4_pack_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "creste_conv_wino4_workspace_bytes": (_i64, [_i, _i, _i, _i, _i, _i]),
    "creste_conv_wino4_gemm_probe": (_i, [_i]),
    "creste_conv_wino4_gemm_last_ms": (_i, [_vp]),
    "creste_conv_packed_weight_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "creste_conv_pack_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "creste_conv_pack_weight_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "creste_absmax_nhwc_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "creste_dwconv2d_nhwc_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 11 + [_vp]),
    "creste_se_partial_count": (_i, [_i, _i]),
    "creste_dwconv_se_nhwc_f32": (_i, [_vp] * 6 + [_i] * 11 + [_vp]),
    "creste_dwconv_se_tile_partial_count": (_i, [_i] * 5),
    "creste_dwconv_se_tile_f32": (_i, [_vp] * 6 + [_i] * 10 + [_vp]),
    "creste_dwconv_tile_f32": (_i, [_vp] * 4 + [_i] * 11 + [_vp]),
    "creste_se_gate_f32": (_i, [_vp] * 7 + [_i] * 4 + [_vp]),
    "creste_se_gate_partial_f32": (_i, [_vp, _i] + [_vp] * 5 + [_i] * 4 + [_vp]),
    "creste_mbconv_partial_count": (_i, [_i] * 7),
    "creste_stem_dw_partial_count": (_i, [_i] * 4),
    "creste_stem_dw_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "creste_mbconv_expand_dw_f32": (_i, [_vp, _i, _i, _i, _i, _i] + [_vp] * 7 + [_i] * 7 + [_vp]),
    "creste_upsample_concat_nhwc_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, _i,
                                              _f, _f, _vp, _vp]),
    "creste_maxpool2_nhwc_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "creste_maxpool_nhwc_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "creste_fill_u32": (_i, [_vp, C.c_uint32, _i64, _vp]),
    "creste_spin_us": (_i, [C.c_int, C.c_int, _vp]),
    "creste_max2_f32": (_i, [_vp, _vp, _vp, _vp]),
    "creste_affine_act_nhwc_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "creste_resize_plane_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _f, _f, _vp]),
    "creste_nchw_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "creste_nhwc_to_nchw_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "creste_lidar_depth_image_f32": (_i, [_vp, _i, _vp, _i, _i, _i64, _i, _i, _i, C.c_double, _vp, _i64, _vp]),
    "creste_lidar_pixels_to_depth_f64": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "creste_depth_expectation_f32": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "creste_pixel_geometry_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp,
                                        _vp, _i, _i, _vp]),
    "creste_bev_splat_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "creste_bev_splat_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _i, _i, _f, _vp, _vp, _vp,
                                   _vp, _vp]),
    "creste_bev_splat_mode_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _f, _i, _i, _f, _i, _vp, _vp, _vp,
                                        _vp, _vp]),
    "creste_bev_splat_plan_f32": (_i, [_vp, _i, _i, _f, _f, _f, _f, _i, _i, _vp, _vp, _vp]),
    "creste_bev_splat_plan_keyed_f32": (_i, [_i, _i, _i, _i, _vp, _vp, _vp]),
    "creste_pixel_geometry_keyed_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i,
                                             _f, _f, _f, _f, _i, _i, _vp, _vp, _vp]),
    "creste_bev_splat_gather_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "creste_conv1x1_chain3_weight_bytes": (_i64, [_i, _i]),
    "creste_conv1x1_chain3_f32": (_i, [_vp, _i, _i64, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "creste_upconv2x_ring_fix_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp]),
    "creste_value_iteration_workspace_bytes": (_i64, [_i, _i, _i]),
    "creste_value_iteration_f32": (_i, [_vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "creste_value_iteration_chunked_f32": (_i, [_vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "creste_conv_wgrad_workspace_bytes": (_i64, [_i] * 6),
    "creste_conv_wgrad_f32": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 8 + [_vp, _vp]),
    "creste_conv_flip_weight_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "creste_bn_workspace_bytes": (_i64, [_i]),
    "creste_bn_train_forward_f32": (_i, [_vp, _i, _i64, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                         _vp, _vp, _vp]),
    "creste_bn_train_forward_stats_f32": (_i, [_vp, _i, _i64, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                               _vp, _vp, _i, _vp]),
    "creste_bn_train_tangent_f32": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "creste_bn_train_backward_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "creste_bn_relu_train_backward_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "creste_bn_act_train_backward_f32": (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i,
                                              _vp, _vp, _vp]),
    "creste_pointwise2_f32": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "creste_maxpool2_idx_f32": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "creste_maxpool2_route_f32": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "creste_upsample_bwd_nhwc_f32": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "creste_conv_wgrad_strided_workspace_bytes": (_i64, [_i] * 6),
    "creste_conv_wgrad_strided_f32": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 12 + [_vp, _vp]),
    "creste_conv_wgrad_bf16x6": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 12 + [_vp, _vp]),
    "creste_conv_wgrad_wino4_supported": (_i, [_i] * 8),
    "creste_conv_wgrad_wino4_workspace_bytes": (_i64, [_i] * 5),
    "creste_conv_wgrad_wino4": (_i, [_vp, _i, _vp, _i, _vp] + [_i] * 8 + [_vp, _vp]),
    "creste_conv_wgrad_f16x3": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp] + [_i] * 12 + [_vp, _vp]),
    "creste_dwconv_dgrad_f32": (_i, [_vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "creste_dwconv_wgrad_workspace_bytes": (_i64, [_i, _i]),
    "creste_dwconv_wgrad_f32": (_i, [_vp, _vp, _vp] + [_i] * 11 + [_vp, _vp]),
    "creste_train_pointwise_f32": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _i64, _i, _vp, _vp]),
    "creste_sample_reduce_workspace_bytes": (_i64, [_i, _i]),
    "creste_sample_reduce_f32": (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _i, _f, _vp, _vp]),
    "creste_se_fc_forward_f32": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "creste_se_fc_backward_f32": (_i, [_vp] * 8 + [_i, _i, _i, _vp]),
    "creste_fc_wgrad_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "creste_loss_workspace_bytes": (_i64, []),
    "creste_depth_ce_loss_f32": (_i, [_vp, _i, _vp, _i64, _i, _f, _f, _f, _vp, _i, _vp, _vp, _vp]),
    "creste_mse_loss_f32": (_i, [_vp, _i, _vp, _i, _i64, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "creste_bev_ce_loss_f32": (_i, [_vp, _i, _i, _vp, _i, _i64, _i64, _vp, _vp, _i, _i, _f, _f, _vp, _i, _vp, _vp, _vp]),
    "creste_smooth_l1_loss_f32": (_i, [_i, _vp, _i, _vp, _i64, _i64, _i, _f, _f, _f, _i, _f, _vp, _i, _vp, _vp, _vp]),
    "creste_label_minmax_i64": (_i, [_vp, _i64, _vp, _vp]),
    "creste_remap_labels_i64": (_i, [_vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "creste_group_by_class_workspace_bytes": (_i64, [_i64, _i]),
    "creste_group_by_class_i64": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "creste_pick_cells_i32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "creste_gather_rows_f32": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp]),
    "creste_scatter_rows_f32": (_i, [_vp, _vp, _i64, _i, _vp, _i, _vp]),
    "creste_bev_splat_bwd_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _i, _vp,
                                      _vp, _vp]),
    "creste_bev_splat_mode_bwd_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _i, _vp, _i, _vp,
                                           _vp, _vp]),
    "creste_depth_expectation_bwd_f32": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "creste_zero_insert_nhwc_f32": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "creste_pixel_geometry_bwd_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp,
                                           _vp, _vp, _vp]),
    "creste_multipos_con_workspace_bytes": (_i64, [_i, _i, _i]),
    "creste_multipos_con_forward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "creste_multipos_con_backward_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "creste_expected_svf_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp,
                                      _vp]),
    "creste_trajectory_scores_f32": (_i, [_vp, _i, _i, _f, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "creste_trajectory_scores_grouped_f32": (_i, [_vp, _i, _i, _f, _i, _i, _vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "creste_irl_visitation_mix_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i64, _vp, _vp, _vp, _vp, _vp]),
    "creste_hip_model_load": (_i, [C.c_char_p, _i, C.POINTER(_vp)]),
    "creste_hip_model_free": (_i, [_vp]),
    "creste_hip_model_info": (C.c_char_p, [_vp]),
    "creste_hip_model_num_inputs": (_i, [_vp]),
    "creste_hip_model_num_outputs": (_i, [_vp]),
    "creste_hip_model_num_streams": (_i, [_vp]),
    "creste_hip_model_input": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                                    C.POINTER(_i64), C.POINTER(_i64)]),
    "creste_hip_model_output": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i),
                                     C.POINTER(_i64), C.POINTER(_i64)]),
    "creste_hip_model_infer": (_i, [_vp, C.POINTER(_vp), _i, _vp]),
    "creste_hip_memcpy_d2h": (_i, [_vp, _vp, _i64]),
}

_lib = None
_recorder = None          # a PlanRecorder while deploy.export_plan traces a forward: every launching call and stream-order edge


class PlanRecorder(list):
    """(name, args, stream handle) of every launching call of a traced forward, in host issue order; stream-order edges as
    the pseudo calls ("__record__" | "__wait__", [("i", event id)], stream handle).  `pipelined`: the traced forward may run
    in parts on several streams (ops.forward_in_parts); otherwise it is kept on one stream."""

    def __init__(self, pipelined: bool = False):
        super().__init__()
        self.pipelined = bool(pipelined)
        self._events = {}
        self._keep = []          # every event seen stays alive for the trace: id() of a freed event is handed out again

    MAX_STREAMS, MAX_EVENTS = 16, 4096      # what csrc/plan_runtime.cpp's loader accepts (deploy.export_plan checks before writing)

    def event_id(self, ev) -> int:
        k = id(ev)
        if k not in self._events:
            self._events[k] = len(self._events)
            self._keep.append(ev)
        return self._events[k]

    num_events = property(lambda self: len(self._events))


def event_record(ev, stream):
    """ev.record(stream) -- and, while a plan is traced, the edge's first half (the C runtime replays it with an event of its own)"""
    ev.record(stream)
    if _recorder is not None:
        _recorder.append(("__record__", [("i", _recorder.event_id(ev))], int(stream.cuda_stream)))


def event_wait(stream, ev):
    """stream.wait_event(ev) -- recorded like event_record"""
    stream.wait_event(ev)
    if _recorder is not None:
        _recorder.append(("__wait__", [("i", _recorder.event_id(ev))], int(stream.cuda_stream)))


def stream_wait_stream(dst, src):
    """dst.wait_stream(src): everything issued on `src` so far is ordered before what `dst` does next"""
    import torch
    ev = torch.cuda.Event()
    event_record(ev, src)
    event_wait(dst, ev)


class _RecordingLib:
    """The ctypes handle with every LAUNCHING entry point (int return, trailing stream argument) wrapped: the call is
    appended to the active recorder -- scalar arguments by value, a creste_conv_desc by a copy of its bytes -- and then
    executed.  Host-side queries (workspace sizes, `*_supported`) pass through unrecorded."""

    def __init__(self, lib, rec):
        self._lib_, self._rec = lib, rec

    def __getattr__(self, name):
        fn = getattr(self._lib_, name)
        sig = SIGNATURES.get(name)
        if sig is None or not is_launch(name):
            return fn
        rec = self._rec

        def call(*args):
            saved = []
            for a, ty in zip(args[:-1], sig[1][:-1]):
                if ty in (_vp,):
                    saved.append(("p", int(a) if a is not None else 0))
                elif ty in (_i, C.c_uint32):
                    saved.append(("i", int(a)))
                elif ty is _i64:
                    saved.append(("l", int(a)))
                elif ty is _f:
                    saved.append(("f", float(a)))
                elif ty is C.c_double:
                    saved.append(("d", float(a)))
                else:                                   # byref(ConvDesc)
                    saved.append(("desc", bytes(a._obj)))
            rec.append((name, saved, int(args[-1] or 0)))
            return fn(*args)
        return call


# int-returning entry points whose trailing void* is NOT a stream (queries / probes): never recorded into a plan
NOT_LAUNCHES = ("creste_conv_supported", "creste_se_partial_count", "creste_conv_wino4_gemm_last_ms")


def is_launch(name: str) -> bool:
    """Entry points that launch work on a stream: int return and a trailing void* stream (csrc/plan_dispatch.inc)."""
    res, args = SIGNATURES[name]
    return (res is _i and bool(args) and args[-1] is _vp and not name.startswith("creste_hip_model")
            and name not in NOT_LAUNCHES)


def load(path: str | None = None):
    """Load (once) and return the ctypes handle; raises HipLibraryError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib if _recorder is None else _RecordingLib(_lib, _recorder)
    # torch ships its own libamdhip64; it must be in the process BEFORE this library is opened so that
    # both share ONE HIP runtime (otherwise the kernels register with a second runtime that owns no
    # device: "no ROCm-capable device is detected").
    import torch  # noqa: F401
    p = path or os.environ.get("CRESTE_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise HipLibraryError(
            f"{p} not found: build it with `python -m creste_public_amd.build` "
            "(the HIP path has no PyTorch/CPU fallback)")
    try:
        lib = C.CDLL(p)
    except OSError as e:
        raise HipLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{p} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    if lib.creste_abi_version() != ABI_VERSION:
        raise HipLibraryError(f"{p} has ABI version {lib.creste_abi_version()}, this package binds version "
                              f"{ABI_VERSION}: rebuild with `python -m creste_public_amd.build`")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().creste_last_error()
        raise HipLibraryError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
