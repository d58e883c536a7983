"""ctypes binding of the C-ABI HIP library (include/soswsod_hip.h).

The product path has NO fallback: if libsoswsod_hip.so is missing or a symbol cannot be
resolved, importing this module raises.  (Build it with sos-wsod_amd/csrc/build.sh or
__graft_entry__.build().)
"""
import ctypes
import os

import torch  # noqa: F401  — FIRST: torch bundles its own HIP runtime (torch/lib/libamdhip64.so); loading our library before
#                it would bind it to /opt/rocm's copy instead, and the process then holds two runtimes of which only torch's
#                has an initialised device (every launch from this library then fails with hipErrorNoDevice = 100)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SW_LIB_PATH") or os.path.join(_HERE, "libsoswsod_hip.so")     # SW_LIB_PATH: development builds (kernel A/B runs)

SW_F32, SW_BF16 = 0, 1

c_int, c_long, c_float, c_void_p, c_u64 = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint64


class Epilogue(ctypes.Structure):
    """struct sw_epilogue"""
    _fields_ = [
        ("bias", c_void_p), ("relu", c_int), ("drop_mask", c_void_p), ("ld_drop", c_long), ("drop_scale", c_float),
        ("relu_ref", c_void_p), ("ld_ref", c_long), ("ref_scale", c_float), ("ref_dtype", c_int),
        ("out_dtype", c_int), ("accumulate_atomic", c_int), ("absmax_out", c_void_p),
        ("drop_seed", c_u64), ("drop_offset", c_u64), ("drop_hash_p", c_float), ("drop_offset_dev", c_void_p),
        ("splitk_workspace", c_void_p),
        ("residual", c_void_p), ("ld_res", c_long), ("res_dtype", c_int),
        ("fold_row_scale", c_void_p),
        ("sgd_fused", c_void_p), ("sgd_momentum", c_float), ("sgd_grad_scale", c_float),       # const sw_sgd_tensor* (round 6)
    ]


class WgradProblem(ctypes.Structure):
    """sw_wgrad_problem"""
    _fields_ = [("nimg", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int),
                ("dilation", ctypes.c_int), ("nsplit", ctypes.c_int), ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p),
                ("slabs", ctypes.c_void_p)]


class WgradFold(ctypes.Structure):
    """sw_wgrad_fold"""
    _fields_ = [("Cin", ctypes.c_int), ("Cout", ctypes.c_int), ("nslab", ctypes.c_int), ("workspace", ctypes.c_void_p),
                ("dw_oihw", ctypes.c_void_p)]


class GemmKKProblem(ctypes.Structure):
    """sw_gemm_kk_problem"""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("slabs", ctypes.c_void_p), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
                ("K", ctypes.c_int32), ("nsplit", ctypes.c_int32), ("lda", ctypes.c_long), ("ldb", ctypes.c_long)]


class SplitkFold(ctypes.Structure):
    """sw_splitk_fold"""
    _fields_ = [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("nslab", ctypes.c_int32), ("accumulate", ctypes.c_int32),
                ("workspace", ctypes.c_void_p), ("C", ctypes.c_void_p), ("ldc", ctypes.c_long), ("row_scale", ctypes.c_void_p)]


class ConvProblem(ctypes.Structure):
    """sw_conv_problem"""
    _fields_ = [("nimg", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
                ("in_", ctypes.c_void_p), ("wk", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ep", ctypes.POINTER(Epilogue))]


class WinogradPrep(ctypes.Structure):
    """sw_winograd_prep"""
    _fields_ = [("w", ctypes.c_void_p), ("U", ctypes.c_void_p), ("Cout", ctypes.c_int32), ("Cin", ctypes.c_int32), ("mode", ctypes.c_int32)]


class CopyDesc(ctypes.Structure):
    """sw_copy_desc"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_long)]


class ColsumPart(ctypes.Structure):
    """struct sw_colsum_part_desc"""
    _fields_ = [("M", c_int), ("N", c_int), ("X", c_void_p), ("ld", c_long), ("workspace", c_void_p)]


class StageDesc(ctypes.Structure):
    """struct sw_stage_desc"""
    _fields_ = [("w", c_void_p), ("bn_weight", c_void_p), ("bn_bias", c_void_p), ("bn_mean", c_void_p), ("bn_var", c_void_p),
                ("scale", c_void_p), ("shift", c_void_p), ("dst", c_void_p),
                ("kind", ctypes.c_int32), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("block_start", ctypes.c_int32)]


class ColsumFold(ctypes.Structure):
    """sw_colsum_fold_desc"""
    _fields_ = [("N", ctypes.c_int), ("n_partial_rows", ctypes.c_int), ("workspace", ctypes.c_void_p), ("out", ctypes.c_void_p)]


class SgdTensor(ctypes.Structure):
    """struct sw_sgd_tensor"""
    _fields_ = [
        ("param", c_void_p), ("grad", c_void_p), ("momentum_buf", c_void_p), ("n", c_long), ("lr", c_float),
        ("weight_decay", c_float), ("first_step", c_int), ("stage_kind", c_int), ("stage_dtype", c_int),
        ("stage0", c_void_p), ("stage1", c_void_p), ("d0", c_int), ("d1", c_int), ("d2", c_int), ("ld0", c_long), ("ld1", c_long),
        ("hyper_dev", c_void_p),
    ]


SGD_MAX_TENSORS = 24          # SW_SGD_MAX_TENSORS
_EP = ctypes.POINTER(Epilogue)
_F4 = ctypes.POINTER(c_float)

# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES = {
    "sw_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long,
                        _EP, c_int, c_void_p]),
    "sw_gemm_splitk_workspace_floats": (c_long, [c_int, c_int, c_int, c_int]),
    "sw_gemm_sgd_fused_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "sw_conv3x3_igemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, _EP,
                                 c_void_p]),
    "sw_conv3x3_wgrad": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_void_p]),
    "sw_conv3x3_wgrad_workspace_floats": (c_long, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "sw_conv_weight_prep": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_maxpool2x2_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_maxpool2x2_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p]),
    "sw_preprocess": (c_int, [c_int, c_int, c_int, c_int, c_void_p, _F4, _F4, c_void_p, c_void_p]),
    "sw_preprocess_multi": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p), _F4, _F4, c_void_p, c_void_p]),
    "sw_roi_pool_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                                c_void_p, c_float, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "sw_roi_pool_fwd_workspace_bytes": (c_long, [c_int, c_int, c_int, c_int]),
    "sw_roi_pool_fwd_ws": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int,
                                   c_void_p, c_float, c_void_p, c_void_p, c_int, c_long, c_void_p, c_long, c_void_p]),
    "sw_roi_pool_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_long, c_void_p, c_int,
                                c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "sw_absmax": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_wsddn_workspace_floats": (c_long, [c_int, c_int, c_int]),
    "sw_wsddn_mil": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "sw_mean_views": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_oicr_mean_probs": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p]),
    "sw_mine_workspace_bytes": (c_long, [c_int, c_int, c_int]),
    "sw_oicr_mine_label": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_float,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p]),
    "sw_oicr_refine_loss": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, _F4, c_void_p, c_void_p, c_long, c_void_p,
                                    c_void_p, c_void_p]),
    "sw_oicr_predict": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_int, c_int, c_void_p, _F4, c_float, c_void_p,
                                c_void_p, c_void_p]),
    "sw_detect_workspace_bytes": (c_long, [c_int, c_int]),
    "sw_detect_postprocess": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_detect_workspace_bytes2": (c_long, [c_int, c_int, c_int]),
    "sw_detect_postprocess2": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sw_colsum": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_colsum_acc": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_int, c_void_p]),
    "sw_colsum_workspace_floats": (c_long, [c_int, c_int, c_int]),
    "sw_colsum_partial": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p]),
    "sw_colsum_fold": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_conv3x3_wgrad_scaled": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_void_p, c_void_p]),
    "sw_conv3x3_wgrad_acc": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p]),
    "sw_conv3x3_wgrad_small": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_conv3x3_wgrad_small_acc": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sw_conv3x3_wgrad_slabs": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                       c_void_p]),
    "sw_conv3x3_wgrad_fold": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_conv3x3_wgrad_fold_acc": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sw_conv3x3_wgrad_fold_multi": (c_int, [c_int, ctypes.POINTER(WgradFold), c_void_p]),
    "sw_conv3x3_multi": (c_int, [c_int, c_int, ctypes.POINTER(ConvProblem), c_void_p]),
    "sw_conv3x3_winograd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, _EP, c_void_p]),
    "sw_winograd_weight_prep": (c_int, [c_int, ctypes.POINTER(WinogradPrep), c_void_p]),
    "sw_gemm_kk_grouped": (c_int, [c_int, c_int, ctypes.POINTER(GemmKKProblem), c_void_p]),
    "sw_gemm_kk_grouped_slabs": (c_long, [c_int, c_int, c_int]),
    "sw_splitk_fold_multi": (c_int, [c_int, ctypes.POINTER(SplitkFold), c_void_p]),
    "sw_colsum_fold_multi": (c_int, [c_int, ctypes.POINTER(ColsumFold), c_void_p]),
    "sw_colsum_partial_multi": (c_int, [c_int, c_int, ctypes.POINTER(ColsumPart), c_void_p]),
    "sw_conv3x3_wgrad_grouped": (c_int, [c_int, c_int, ctypes.POINTER(WgradProblem), c_void_p]),
    "sw_convert_2d": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "sw_split_bf16x3": (c_int, [c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_void_p]),
    "sw_nchw_to_nhwc": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_relu_bwd": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_relu_bwd_out": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_scale_cols": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p]),
    "sw_to_f32": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_fill_zero": (c_int, [c_void_p, c_long, c_void_p]),
    "sw_dropout_mask": (c_int, [c_void_p, c_long, c_u64, c_u64, c_float, c_void_p]),
    "sw_sgd_momentum_step": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_int, c_float,
                                     c_void_p]),
    "sw_convert_2d_t": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "sw_sgd_multi": (c_int, [c_int, ctypes.POINTER(SgdTensor), c_float, c_float, c_void_p]),
    "sw_loss_finalize": (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_scale_cols_loss": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p, c_long, c_void_p]),
    "sw_resize_pass_u8": (c_int, [c_int, c_int, c_int, c_long, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p]),
    "sw_color_jitter_u8": (c_int, [c_int, c_int, c_int, c_void_p, c_float, ctypes.c_double, c_float, c_void_p, c_void_p, c_void_p]),
    "sw_transpose_2d": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p]),
    "sw_stage_blocks": (c_int, [c_int, c_int, c_int]),
    "sw_stage_weights_multi": (c_int, [c_int, c_int, c_void_p, c_int, c_float, c_void_p]),
    "sw_ema_multi": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_long), ctypes.c_double, c_void_p]),
    "sw_threshold_select": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "sw_conv3x3_relu_pool2": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_weighted_sum": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_scale_scalars": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_counter_add": (c_int, [c_void_p, c_u64, c_void_p]),
    "sw_focal_loss": (c_int, [c_int, c_int, c_void_p, c_long, c_void_p, c_float, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    "sw_copy_multi": (c_int, [c_int, ctypes.POINTER(CopyDesc), c_void_p]),
    "sw_pack_views": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    # ---- Stage-3 detector (csrc/detector.hip)
    "sw_preprocess_pad": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, _F4, _F4, c_void_p, c_void_p]),
    "sw_stem_conv7x7": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_maxpool3x3s2": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_subsample2x": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_scatter2x": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_add_relu": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "sw_upsample2x_add": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_downsample2x_sum": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sw_roi_align_fwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_long, c_void_p]),
    "sw_roi_align_bwd": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_void_p, c_void_p,
                                 c_int, c_void_p, c_void_p, c_void_p]),
    "sw_roi_align_bwd_fx": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_void_p, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_fx_to_float": (c_int, [c_int, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_wsddn_scores_bwd_workspace_floats": (c_long, [c_int, c_int]),
    "sw_wsddn_scores_bwd": (c_int, [c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p]),
    "sw_scale_col_blocks": (c_int, [c_long, c_int, c_int, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sw_rpn_unpack": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    "sw_rpn_unpack_bwd": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_int), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long,
                                  c_void_p]),
    "sw_roi_assign_levels": (c_int, [c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_long), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "sw_decode_boxes": (c_int, [c_long, c_long, c_void_p, c_long, c_void_p, _F4, c_float, c_void_p, c_void_p]),
    "sw_rpn_select_workspace_bytes": (c_long, [c_int, c_int, ctypes.POINTER(c_int)]),
    "sw_rpn_select_pack": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                   ctypes.POINTER(c_int), c_long, c_int, _F4, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_long, c_void_p]),
    "sw_rpn_label_workspace_bytes": (c_long, [c_int, c_long, c_int]),
    "sw_rpn_label_anchors": (c_int, [c_int, c_long, c_void_p, c_void_p, ctypes.POINTER(c_int), c_float, c_float, c_int, c_int,
                                     ctypes.POINTER(c_u64), c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "sw_roi_label_sample": (c_int, [c_int, c_void_p, c_int, c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p, c_void_p, c_int,
                                    c_float, c_int, c_int, c_int, ctypes.POINTER(c_u64), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "sw_rpn_loss_workspace_floats": (c_long, []),
    "sw_rpn_loss": (c_int, [c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _F4, c_float, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p]),
    "sw_version": (ctypes.c_char_p, []),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU / eager fallback exists). "
            "Build it with `bash sos-wsod_amd/csrc/build.sh` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing -> loud failure
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


class HipKernelError(RuntimeError):
    pass


def check(rc, name):
    if rc != 0:
        raise HipKernelError(f"{name} failed with code {rc}")
