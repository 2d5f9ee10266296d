"""ctypes binding of libpmf_amd.so (include/pmf_amd.h).

The product path has NO fallback: if the shared library is missing or its struct
layout disagrees with this binding, importing a compute entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpmf_amd.so")

MAX_SRC, MAX_TAPS = 5, 49
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3
SRC_RELU, SRC_BCAST = 1, 2
EP_STAT_X_ONLY = 1
WGRAD_S3 = 1
PMF_E_ARG, PMF_E_UNSUPPORTED = -1, -2
CFG_DIRECT_TAPS = 1 << 24      # pmf_conv_desc_t.cfg: direct multi-tap variant (PMF_CFG_DIRECT_TAPS)
CFG_WS = 1 << 25               # ... the wave-scheduled N-split kernel (PMF_CFG_WS, csrc/conv_ws.hip)

(OP_CONV, OP_WGRAD, OP_PACK, OP_BN_FINALIZE, OP_BN_EVAL, OP_BN_BWD_REDUCE, OP_BN_BWD_APPLY, OP_ADD_ACT,
 OP_ADD_ACT_BWD, OP_ACT_BWD, OP_AVGPOOL, OP_AVGPOOL_BWD, OP_MAXPOOL, OP_MAXPOOL_BWD, OP_BILINEAR,
 OP_BILINEAR_BWD, OP_PSHUFFLE, OP_PSHUFFLE_BWD, OP_GATE, OP_GATE_BWD, OP_GMEAN, OP_GMEAN_BWD, OP_COLSUM,
 OP_SOFTMAX, OP_SOFTMAX_BWD, OP_NCHW2NHWC, OP_FILL, OP_PMASK_FROM, OP_PMASK_POOL, OP_PMASK_MUL, OP_PMASK_MUL_BWD,
 OP_VEC_ADD, OP_WGRAD_PART, OP_WGRAD_RED, OP_WGRAD_RED_MULTI, OP_BN_BWD_FOLD, OP_BN_BWD_SMALL, OP_BCAST) = range(1, 39)

OP_NAMES = {v: k for k, v in list(globals().items()) if k.startswith("OP_")}


class Src(C.Structure):
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("cmul", C.c_void_p),
                ("C", C.c_int32), ("ldc", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("flags", C.c_int32), ("cmul_ld", C.c_int32)]


class ConvDst(C.Structure):
    _fields_ = [("out", C.c_void_p), ("out_ldc", C.c_int32), ("C", C.c_int32), ("accumulate", C.c_int32),
                ("ep_cmul_ld", C.c_int32), ("ep_cmul", C.c_void_p), ("ep_relu_x", C.c_void_p),
                ("ep_relu_scale", C.c_void_p), ("ep_relu_shift", C.c_void_p), ("ep_relu_ldc", C.c_int32),
                ("ep_flags", C.c_int32), ("stats", C.c_void_p), ("ep_stat_mean", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("nsrc", C.c_int32),
                ("src", Src * MAX_SRC), ("ntaps", C.c_int32),
                ("tdy", C.c_int8 * (MAX_TAPS + 3)), ("tdx", C.c_int8 * (MAX_TAPS + 3)),
                ("in_stride", C.c_int32), ("gather", C.c_int32), ("w", C.c_void_p), ("ldw", C.c_int32),
                ("bias", C.c_void_p), ("act", C.c_int32), ("out", C.c_void_p),
                ("out_ldc", C.c_int32), ("out_H", C.c_int32), ("out_W", C.c_int32), ("out_sy", C.c_int32),
                ("out_sx", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32), ("accumulate", C.c_int32),
                ("ep_cmul", C.c_void_p), ("ep_cmul_ld", C.c_int32), ("ep_relu_x", C.c_void_p),
                ("ep_relu_scale", C.c_void_p), ("ep_relu_shift", C.c_void_p), ("ep_relu_ldc", C.c_int32),
                ("stats", C.c_void_p), ("ep_pmask", C.c_void_p), ("splitk_ws", C.c_void_p),
                ("splitk_ws_bytes", C.c_int64), ("cfg", C.c_int32), ("ep_flags", C.c_int32),
                ("ep_stat_mean", C.c_void_p), ("w_s3", C.c_void_p), ("ndst", C.c_int32), ("dst", ConvDst * MAX_SRC),
                ("splitk_tickets", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("nsrc", C.c_int32),
                ("src", Src * MAX_SRC), ("ntaps", C.c_int32),
                ("tdy", C.c_int8 * (MAX_TAPS + 3)), ("tdx", C.c_int8 * (MAX_TAPS + 3)),
                ("tap_widx", C.c_int8 * (MAX_TAPS + 3)),
                ("in_stride", C.c_int32), ("gather", C.c_int32), ("dz", C.c_void_p), ("dz_ldc", C.c_int32),
                ("partial", C.c_void_p), ("nsplit", C.c_int32), ("dw_oihw", C.c_void_p),
                ("Cin_real", C.c_int32), ("KHW", C.c_int32), ("accumulate", C.c_int32),
                ("dbias_rows", C.c_void_p), ("dbias_nrows", C.c_int32), ("dbias_ld", C.c_int32),
                ("dbias_out", C.c_void_p), ("cfg", C.c_int32), ("flags", C.c_int32)]


class View(C.Structure):
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("cmul", C.c_void_p),
                ("ldc", C.c_int32), ("cmul_ld", C.c_int32), ("flags", C.c_int32)]


class SmallArgs(C.Structure):
    _fields_ = [("p", C.c_void_p * 12), ("l", C.c_int64 * 4), ("i", C.c_int32 * 16), ("f", C.c_float * 4),
                ("v", View * 3)]


class _OpU(C.Union):
    _fields_ = [("conv", ConvDesc), ("wgrad", WgradDesc), ("sm", SmallArgs)]


class Op(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("u", _OpU)]


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32), ("KHW", C.c_int32),
                ("ntaps", C.c_int32), ("transpose", C.c_int32), ("K_pad", C.c_int32), ("ldw", C.c_int32),
                ("CT", C.c_int32), ("tiles_ci", C.c_int32), ("block_start", C.c_int32),
                ("tap_idx", C.c_int8 * (MAX_TAPS + 3)), ("format", C.c_int32), ("w_ld", C.c_int32)]


_lib = None


class PMFLibraryError(RuntimeError):
    pass


def lib():
    """Load libpmf_amd.so once; raise loudly (no fallback) if it is absent or ABI-incompatible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PMFLibraryError(
            "pmf_amd: %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pmf_amd/csrc`).  There is no CPU / PyTorch fallback for the HIP hot path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.pmf_sizeof.restype = C.c_int
    L.pmf_sizeof.argtypes = [C.c_int]
    for which, st in enumerate((Src, ConvDesc, WgradDesc, View, SmallArgs, Op, PackJob)):
        if L.pmf_sizeof(which) != C.sizeof(st):
            raise PMFLibraryError("pmf_amd ABI mismatch for %s: lib %d vs binding %d"
                                  % (st.__name__, L.pmf_sizeof(which), C.sizeof(st)))
    L.pmf_version.restype = C.c_char_p
    L.pmf_plan_run.restype = C.c_int
    L.pmf_plan_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    L.pmf_plan_run_range.restype = C.c_int
    L.pmf_plan_run_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
    L.pmf_conv_fwd.restype = C.c_int
    L.pmf_conv_fwd.argtypes = [C.POINTER(ConvDesc), C.c_void_p]
    L.pmf_conv_wgrad.restype = C.c_int
    L.pmf_conv_wgrad.argtypes = [C.POINTER(WgradDesc), C.c_void_p]
    L.pmf_conv_wgrad_nsplit.restype = C.c_int
    L.pmf_conv_wgrad_nsplit.argtypes = [C.POINTER(WgradDesc)]
    L.pmf_conv_wgrad_workspace.restype = C.c_int64
    L.pmf_conv_wgrad_workspace.argtypes = [C.POINTER(WgradDesc)]
    L.pmf_conv_multi_ok.restype = C.c_int
    L.pmf_conv_multi_ok.argtypes = [C.POINTER(ConvDesc)]
    L.pmf_conv_fwd_stat_rows.restype = C.c_int
    L.pmf_conv_fwd_stat_rows.argtypes = [C.POINTER(ConvDesc)]
    L.pmf_col_rows.restype = C.c_int
    L.pmf_col_rows.argtypes = [C.c_int64, C.c_int32]
    L.pmf_pack_tile_ci.restype = C.c_int
    L.pmf_pack_tile_ci.argtypes = [C.c_int32, C.c_int32]
    L.pmf_pack_weights_batched.restype = C.c_int
    L.pmf_pack_weights_batched.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pmf_bn_bwd_small_ok.restype = C.c_int
    L.pmf_bn_bwd_small_ok.argtypes = [C.c_int64, C.c_int32]
    L.pmf_bn_bwd_small.restype = C.c_int
    L.pmf_bn_bwd_small.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    L.pmf_knn_vote.restype = C.c_int
    L.pmf_knn_vote.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
    L.pmf_knn_vote_batch.restype = C.c_int
    L.pmf_knn_vote_batch.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                        C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]
    L.pmf_knn_vote_batch_prob.restype = C.c_int
    L.pmf_knn_vote_batch_prob.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                                             C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmf_project_scatter.restype = C.c_int
    L.pmf_project_scatter.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_void_p]
    L.pmf_project_scatter2.restype = C.c_int
    L.pmf_project_scatter2.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]
    L.pmf_crop_pad.restype = C.c_int
    L.pmf_crop_pad.argtypes = [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    L.pmf_lovasz_grad.restype = C.c_int
    L.pmf_lovasz_grad.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmf_project_v2_index.restype = C.c_int
    L.pmf_project_v2_index.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_float] + [C.c_void_p] * 10
    L.pmf_project_v2_index_scaled.restype = C.c_int
    L.pmf_project_v2_index_scaled.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_double] + \
        [C.c_void_p] * 10
    L.pmf_project_v2_scatter.restype = C.c_int
    L.pmf_project_v2_scatter.argtypes = [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                             C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmf_conv_wgrad_reduce_plan.restype = C.c_int
    L.pmf_conv_wgrad_reduce_plan.argtypes = [C.c_void_p, C.c_void_p]
    L.pmf_conv_wgrad_reduce_multi.restype = C.c_int
    L.pmf_conv_wgrad_reduce_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pmf_flip_rotate_crop.restype = C.c_int
    L.pmf_flip_rotate_crop.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)] + \
        [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pmf_color_jitter.restype = C.c_int
    L.pmf_color_jitter.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    L.pmf_merge_pred.restype = C.c_int
    L.pmf_merge_pred.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
    L.pmf_merge_pred_fallback.restype = C.c_int
    L.pmf_merge_pred_fallback.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
    L.pmf_points_transform.restype = C.c_int
    L.pmf_points_transform.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p, C.c_void_p]
    L.pmf_range_project_index.restype = C.c_int
    L.pmf_range_project_index.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_int32, C.c_int32] + [C.c_void_p] * 5
    L.pmf_range_project_gather.restype = C.c_int
    L.pmf_range_project_gather.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + \
        [C.c_void_p] * 10
    L.pmf_plan_capture.restype = C.c_int
    L.pmf_plan_capture.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    L.pmf_graph_launch.restype = C.c_int
    L.pmf_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
    L.pmf_graph_pieces.restype = C.c_int
    L.pmf_graph_pieces.argtypes = [C.c_void_p]
    L.pmf_plan_event_wait.restype = C.c_int
    L.pmf_plan_event_wait.argtypes = [C.c_int32, C.c_void_p]
    L.pmf_adamw_range.restype = C.c_int
    L.pmf_adamw_range.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 5 + [C.c_void_p, C.c_void_p]
    L.pmf_normalise_inplace.restype = C.c_int
    L.pmf_normalise_inplace.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int64, C.c_void_p]
    L.pmf_sgd_range.restype = C.c_int
    L.pmf_sgd_range.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_double] * 4 + [C.c_int32, C.c_int32, C.c_void_p]
    L.pmf_plan_issue_order.restype = C.c_int
    L.pmf_plan_issue_order.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.pmf_plan_lanes.restype = C.c_int
    L.pmf_plan_lanes.argtypes = [C.c_int]
    L.pmf_graph_destroy.restype = C.c_int
    L.pmf_graph_destroy.argtypes = [C.c_void_p]
    L.pmf_loss_rows.restype = C.c_int
    L.pmf_loss_rows.argtypes = [C.c_int64]
    L.pmf_loss_chunks.restype = C.c_int
    L.pmf_loss_chunks.argtypes = [C.c_int64]
    L.pmf_loss_pixel.restype = C.c_int
    L.pmf_loss_pixel.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_float] + \
        [C.c_void_p] * 8
    L.pmf_loss_lovasz.restype = C.c_int
    L.pmf_loss_lovasz.argtypes = [C.c_void_p] * 3 + [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_float, C.c_float] + \
        [C.c_void_p] * 7
    L.pmf_loss_pixel_w.restype = C.c_int
    L.pmf_loss_pixel_w.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p] + \
        [C.c_void_p] * 8
    L.pmf_loss_lovasz_w.restype = C.c_int
    L.pmf_loss_lovasz_w.argtypes = [C.c_void_p] * 3 + [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p] + \
        [C.c_void_p] * 7
    L.pmf_loss_sort_workspace.restype = C.c_int64
    L.pmf_loss_sort_workspace.argtypes = [C.c_int32, C.c_int64]
    L.pmf_loss_lovasz_sort.restype = C.c_int
    L.pmf_loss_lovasz_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_float,
                                       C.c_float] + [C.c_void_p] * 8
    L.pmf_loss_lovasz_sort_w.restype = C.c_int
    L.pmf_loss_lovasz_sort_w.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p] + \
        [C.c_void_p] * 8
    L.pmf_fill.restype = C.c_int
    L.pmf_fill.argtypes = [C.c_void_p, C.c_float, C.c_int64, C.c_void_p]
    _lib = L
    return L


EXPORTS = [
    "pmf_conv_fwd", "pmf_conv_wgrad", "pmf_conv_wgrad_partial", "pmf_conv_wgrad_reduce", "pmf_conv_wgrad_reduce_plan", "pmf_conv_wgrad_reduce_multi", "pmf_conv_wgrad_workspace", "pmf_conv_wgrad_nsplit", "pmf_pack_tile_ci",
    "pmf_pack_weights_batched", "pmf_conv_fwd_stat_rows", "pmf_conv_s3_eligible", "pmf_conv_ws_ok", "pmf_conv_fwd_stat_rows_max", "pmf_conv_fwd_kstages", "pmf_col_rows", "pmf_bn_finalize", "pmf_bn_eval_affine", "pmf_bn_bwd_reduce", "pmf_bn_bwd_fold", "pmf_bn_bwd_apply",
    "pmf_add_act", "pmf_add_act_bwd", "pmf_act_bwd", "pmf_avgpool3s2", "pmf_avgpool3s2_bwd", "pmf_maxpool3s2",
    "pmf_maxpool3s2_bwd", "pmf_bilinear2x", "pmf_bilinear2x_bwd", "pmf_pixel_shuffle2", "pmf_pixel_shuffle2_bwd",
    "pmf_fusion_gate", "pmf_fusion_gate_bwd", "pmf_global_mean", "pmf_global_mean_bwd", "pmf_broadcast_rows", "pmf_colsum", "pmf_colsum_rows",
    "pmf_pmask_from", "pmf_pmask_pool", "pmf_pmask_mul", "pmf_pmask_mul_bwd", "pmf_vec_add", "pmf_softmax_nhwc_to_nchw", "pmf_softmax_bwd_nchw_to_nhwc", "pmf_logits_nhwc_to_nchw", "pmf_logits_bwd_nchw_to_nhwc", "pmf_nchw_to_nhwc", "pmf_fill", "pmf_debug_col", "pmf_bn_bwd_small_ok", "pmf_bn_bwd_small", "pmf_knn_vote", "pmf_knn_vote_batch", "pmf_knn_vote_batch_prob", "pmf_merge_pred", "pmf_merge_pred_fallback",
    "pmf_project_scatter", "pmf_project_scatter2", "pmf_project_v2_index", "pmf_project_v2_index_scaled", "pmf_project_v2_scatter", "pmf_points_transform", "pmf_range_project_index", "pmf_range_project_gather", "pmf_crop_pad", "pmf_flip_rotate_crop", "pmf_color_jitter", "pmf_lovasz_grad", "pmf_loss_rows", "pmf_loss_chunks", "pmf_loss_pixel", "pmf_loss_lovasz", "pmf_loss_pixel_w", "pmf_loss_lovasz_w", "pmf_loss_sort_workspace", "pmf_loss_lovasz_sort", "pmf_loss_lovasz_sort_w", "pmf_plan_run", "pmf_plan_run_range", "pmf_plan_capture", "pmf_graph_launch", "pmf_graph_destroy", "pmf_plan_lanes", "pmf_plan_issue_order", "pmf_graph_pieces", "pmf_plan_event_wait", "pmf_adamw_range", "pmf_sgd_range", "pmf_normalise_inplace", "pmf_sizeof", "pmf_version", "pmf_conv_multi_ok",
]


def check(rc, what="pmf call", failed_at=None):
    if rc != 0:
        extra = "" if failed_at is None else " at op #%d" % failed_at
        raise RuntimeError("%s failed with code %d%s (negative = argument error, positive = hipError_t)"
                           % (what, rc, extra))
