"""ctypes binding of libkagnn_hip.so (C ABI: include/kagnn_hip.h).

The library is the product; there is no fallback.  If it is missing or a call fails, the
caller gets a RuntimeError -- nothing in ``kagnn_amd`` silently computes on the CPU or through
stock torch ops instead.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# KAGNN_LIB: load another build of the same ABI (A/B timing of kernel variants inside one process launch)
LIB_PATH = os.environ.get("KAGNN_LIB") or os.path.join(_HERE, "lib", "libkagnn_hip.so")

PREC_FP32 = 0
PREC_SPLIT = 1
PREC_FP32_GRID = 2      # exact fp32 on per-feature, non-uniform knot rows (after update_grid)
PREC_HALF = 3           # build-defined reduced precision: the split kernels with ONE fp16 product per fp32 product (kagnn_hip.h)
DTYPE_F32, DTYPE_BF16 = 0, 1

_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "kagnn_version": (c_int32, []),
    "kagnn_last_error": (ctypes.c_char_p, []),
    "kagnn_stage_timer_enable": (c_int32, [ctypes.c_char_p]),
    "kagnn_stage_timer_disable": (c_int32, []),
    "kagnn_stage_timer_collect": (c_int32, [_P, _P, _P, c_int32, POINTER(c_int32)]),
    "kagnn_csr_workspace_bytes": (c_int32, [c_int64, c_int64, POINTER(c_size_t)]),
    "kagnn_csr_build": (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P, c_int32, _P, c_int64,
                                  POINTER(c_int64), _P, c_size_t, _P]),
    "kagnn_csr_small_ok": (c_int32, [c_int64, c_int64]),
    "kagnn_csr_small_workspace_bytes": (c_int32, [c_int64, POINTER(c_size_t)]),
    "kagnn_csr_build_small": (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_gcn_deg_inv_sqrt": (c_int32, [_P, _P, c_int64, _P, _P]),
    "kagnn_aggregate_workspace_bytes": (c_int32, [c_int64, c_int32, POINTER(c_size_t)]),
    "kagnn_aggregate_sum": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int32, c_float,
                                      _P, _P, _P, c_int32, _P, c_int64, c_int32, _P, c_size_t, _P]),
    "kagnn_aggregate_sum_add": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int64, c_int32, c_float,
                                          _P, _P, _P, c_int32, _P, c_int64, c_int32, _P, c_int64, _P, c_size_t, _P]),
    "kagnn_aggregate_sum_bf16": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P, c_int64, c_int32, c_float,
                                           _P, _P, _P, c_int32, _P, c_int64, c_int32, _P, c_size_t, _P]),
    "kagnn_rows_to_bf16": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P]),
    "kagnn_aggregate_gine": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, c_int64,
                                       c_int32, c_float, _P]),
    "kagnn_aggregate_gine_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P,
                                           c_int64, _P, _P, _P, c_int64, c_int32, c_float, _P]),
    "kagnn_segment_pool": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, _P]),
    "kagnn_segment_broadcast": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int32, c_int32, _P]),
    "kagnn_p2p_reduce_scatter": (c_int32, [_P, c_int32, c_int32, c_int64, c_int32, c_int64, _P, c_int64, _P]),
    "kagnn_p2p_all_gather": (c_int32, [_P, c_int32, c_int64, c_int32, c_int64, _P, c_int64, _P]),
    "kagnn_embedding_fwd": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int32, _P, c_int64, c_int32, _P]),
    "kagnn_embedding_bwd_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, POINTER(c_size_t)]),
    "kagnn_embedding_bwd": (c_int32, [_P, c_int64, c_int64, _P, c_int64, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "kagnn_kan_pack_bytes": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32,
                                       POINTER(c_size_t), POINTER(c_size_t)]),
    "kagnn_kan_pack": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P]),
    "kagnn_kan_pack_batch": (c_int32, [c_int32, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P]),
    "kagnn_kan_fwd_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                POINTER(c_size_t)]),
    "kagnn_kan_linear_fwd": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, _P, _P, c_int64, _P, c_size_t, _P]),
    "kagnn_kan_fwd_parts_ok": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "kagnn_kan_linear_fwd_parts": (c_int32, [_P, _P, _P, c_int32, c_int64, _P, c_int32, c_int32, c_int32, c_int32,
                                             c_int32, _P, _P, c_int64, _P, c_size_t, _P]),
    "kagnn_kan_fwd_moments_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                        POINTER(c_size_t)]),
    "kagnn_kan_linear_fwd_moments": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int32, c_int32, c_int32,
                                               c_int32, _P, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "kagnn_kan_linear_bwd_input": (c_int32, [_P, c_int64, _P, c_int64, c_int64, _P, c_int32, c_int32,
                                             c_int32, c_int32, c_int32, _P, _P, c_int64, c_int32, _P]),
    "kagnn_kan_bwd_weight_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32,
                                                       c_int32, POINTER(c_size_t)]),
    "kagnn_kan_linear_bwd_weight": (c_int32, [_P, c_int64, _P, c_int64, c_int64, _P, c_int32, c_int32,
                                              c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P,
                                              c_size_t, _P]),
    "kagnn_gin_kan_layer_workspace_bytes": (c_int32, [c_int64, c_int32, _P, c_int32, c_int32, c_int32, c_int64, c_int64,
                                                      POINTER(c_size_t), POINTER(c_size_t)]),
    "kagnn_gin_kan_layer_fwd": (c_int32, [_P, c_int32, c_int64, c_int64, _P, _P, _P, c_int64, c_int32, c_float, c_int32, _P,
                                          _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_gin_kan_layer_bwd": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, c_int64, c_int32, c_float, c_int32, _P, _P, _P,
                                          _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int32, c_int64, c_int32, _P, _P, _P,
                                          _P, c_size_t, _P]),
    "kagnn_gin_kan_layer_bwd_add": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, c_int64, c_int32, c_float, c_int32, _P, _P, _P,
                                              _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int32, c_int64, c_int32, _P, c_int64,
                                              _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_gin_kan_layer_bwd_bn_workspace_bytes": (c_int32, [c_int64, c_int32, POINTER(c_size_t)]),
    "kagnn_gin_kan_layer_bwd_bn": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P,
                                             c_int64, _P, _P, _P, c_int64, c_int32, c_float, c_int32, _P, _P, _P,
                                             _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int32, c_int64, c_int32, _P, c_int64,
                                             _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_gin_kan_layer_bwd_bn_sums_workspace_bytes": (c_int32, [c_int64, c_int32, c_int64, POINTER(c_size_t)]),
    "kagnn_gin_kan_layer_bwd_bn_sums": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P,
                                                  _P, _P, c_int64, _P, _P, _P,
                                                  c_int64, _P, _P, _P, c_int64, c_int32, c_float, c_int32, _P, _P, _P,
                                                  _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int32, c_int64, c_int32, _P, c_int64,
                                                  _P, _P, _P, _P, c_size_t, _P]),
    # x ldx ea lde N rowptr col perm self L widths bw sw sc knots G K mode acts pf pd cmean cm2 ws bytes stream
    "kagnn_gine_kan_layer_fwd": (c_int32, [_P, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_float, c_int32, _P, _P, _P, _P, _P,
                                           c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    # g ldg bn_y ld bn_w mean rstd g_w g_b x ldx ea lde N rowptr_t col_t perm_t self L widths sw sc knots G K mode acts pd gx ldgx gea ldge gbw gsw gsc ws bytes stream
    "kagnn_gine_kan_layer_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, _P, _P, _P,
                                           c_float, c_int32, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P, c_int64,
                                           _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_gine_kan_stack_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, _P, c_int32, c_int32, c_int32, POINTER(c_size_t), POINTER(c_size_t)]),
    # x ldx ea lde N rowptr col perm self nconv L widths bw sw sc knots G K mode acts pf pd bnw bnb rm rv mom eps h mean rstd ws bytes stream
    "kagnn_gine_kan_stack_fwd": (c_int32, [_P, c_int64, _P, c_int64, c_int64, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P,
                                           c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    # g ldg x ldx ea lde N rowptr_t col_t perm_t self nconv L widths sw sc knots G K mode acts pd h bnw mean rstd gx ldgx gea ldge gbnw gbnb gbw gsw gsc ws bytes stream
    "kagnn_gine_kan_stack_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P,
                                           c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P,
                                           _P, c_size_t, _P]),
    "kagnn_kan_bwd_input_sums_ok": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "kagnn_kan_bwd_input_sums_workspace_bytes": (c_int32, [c_int64, c_int32, POINTER(c_size_t)]),
    "kagnn_kan_linear_bwd_input_affine_sums": (c_int32, [_P, c_int64, _P, _P, _P, _P, c_int64, c_int64, _P, c_int32, c_int32,
                                                         c_int32, c_int32, c_int32, _P, _P, c_int64, _P, _P, c_size_t, _P]),
    "kagnn_kan_bsplines": (c_int32, [_P, c_int64, c_int64, _P, c_int32, c_int32, c_int32, _P, _P]),
    "kagnn_kan_grid_refit_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "kagnn_kan_grid_refit": (c_int32, [_P, c_int64, c_int64, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P,
                                       _P, c_size_t, _P]),
    "kagnn_fastkan_fwd_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "kagnn_fastkan_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, c_int32, _P, c_float, _P, _P,
                                    c_float, _P, _P, _P, _P, c_int64, _P, c_int32, _P, c_size_t, _P]),
    "kagnn_fastkan_bwd_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, c_int32, c_int32, POINTER(c_size_t)]),
    "kagnn_fastkan_bwd": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32, _P,
                                    c_float, _P, _P, c_float, _P, _P, _P, _P, c_int64, _P, _P, _P, _P,
                                    _P, c_int32, _P, c_size_t, _P]),
    "kagnn_fastkan_row_moments": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P]),
    "kagnn_fastkan_merge_moments": (c_int32, [_P, c_int32, c_int64, c_int32, c_float, _P, _P]),
    "kagnn_fastkan_shard_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, c_int32, _P, c_float, _P, _P,
                                          _P, _P, _P, _P, _P, c_int64, c_int32, _P, c_size_t, _P]),
    "kagnn_fastkan_shard_bwd": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, c_int32, _P,
                                          c_float, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P,
                                          _P, c_int32, c_int32, _P, c_size_t, _P]),
    "kagnn_fastkan_shard_bwd_finish": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, _P, _P, _P,
                                                 _P, _P, c_int64, _P, _P, c_int32, _P, c_size_t, _P]),
    "kagnn_kagin_model_struct_bytes": (c_int32, []),
    "kagnn_kagin_model_sizes": (c_int32, [_P, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t)]),
    "kagnn_kagin_model_fwd": (c_int32, [_P, _P]),
    "kagnn_kagin_model_bwd": (c_int32, [_P, _P]),
    "kagnn_gat_att_grad_workspace_bytes": (c_int32, [c_int64, c_int32, c_int32, POINTER(c_size_t)]),
    "kagnn_gat_att_grad": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "kagnn_softmax_xent_workspace_bytes": (c_int32, [c_int64, POINTER(c_size_t)]),
    "kagnn_softmax_xent_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_softmax_xent_bwd": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, c_int32, _P, _P, _P, _P, c_int64, _P]),
    "kagnn_l1_loss_fwd": (c_int32, [_P, _P, c_int64, _P, _P]),
    "kagnn_l1_loss_bwd": (c_int32, [_P, _P, c_int64, _P, _P, _P]),
    "kagnn_adam_step": (c_int32, [c_int32, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_int64, _P]),
    "kagnn_gat_logits": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "kagnn_gat_fwd": (c_int32, [_P, c_int64, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, _P, c_int64, _P, _P, _P, c_int64,
                                c_int32, _P]),
    "kagnn_gat_bwd": (c_int32, [_P, c_int64, _P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                _P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int32, _P]),
    "kagnn_batchnorm_stats_affine": (c_int32, [_P, _P, c_int64, c_int32, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P]),
    "kagnn_aggregate_sum_affine": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int32, c_float, _P, _P, _P, c_int64, c_int32,
                                             _P, c_int64, _P, c_size_t, _P]),
    "kagnn_gin_kan_layer_fwd_affine": (c_int32, [_P, c_int64, c_int64, _P, _P, _P, c_int64, c_int32, c_float, _P, _P, c_int32, _P,
                                                 _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_kan_linear_fwd_parts_affine": (c_int32, [_P, _P, _P, _P, c_int32, c_int64, _P, c_int32, c_int32, c_int32, c_int32,
                                                    c_int32, _P, _P, c_int64, _P, c_size_t, _P]),
    "kagnn_kan_linear_bwd_input_affine": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int64, _P, c_int32, c_int32,
                                                    c_int32, c_int32, c_int32, _P, _P, c_int64, c_int32, _P]),
    "kagnn_kan_linear_bwd_weight_affine": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int64, _P, c_int32, c_int32,
                                                     c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "kagnn_batchnorm_workspace_bytes": (c_int32, [c_int64, c_int32, POINTER(c_size_t)]),
    "kagnn_batchnorm_fwd": (c_int32, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_float, c_float, c_int32,
                                      _P, _P, c_float, c_uint64, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "kagnn_batchnorm_bwd": (c_int32, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, c_int32, c_float, c_uint64,
                                      _P, c_int64, _P, _P, _P, c_size_t, _P]),
}

MODEL_MAX_LAYERS, MODEL_MAX_CONVS, MODEL_MAX_TABLES, MODEL_MAX_READOUT = 16, 16, 16, 8


class KaginModel(ctypes.Structure):
    """mirror of ``kagnn_kagin_model_t`` (include/kagnn_hip.h); ``load()`` checks its size against the library's"""
    _fields_ = [
        ("num_nodes", c_int64), ("num_edges", c_int64), ("num_graphs", c_int64), ("hidden", c_int64),
        ("num_atom_tables", c_int64), ("num_bond_tables", c_int64), ("x_stride", c_int64), ("e_stride", c_int64),
        ("num_convs", c_int64), ("num_layers", c_int64), ("grid_size", c_int64), ("spline_order", c_int64), ("mode", c_int64),
        ("num_readout", c_int64), ("readout_grid_size", c_int64), ("readout_spline_order", c_int64),
        ("readout_widths", c_int64 * (MODEL_MAX_READOUT + 1)), ("readout_modes", c_int64 * MODEL_MAX_READOUT),
        ("atom_rows", c_int64 * MODEL_MAX_TABLES), ("bond_rows", c_int64 * MODEL_MAX_TABLES),
        ("x_index", c_void_p), ("e_index", c_void_p),
        ("atom_table", c_void_p * MODEL_MAX_TABLES), ("bond_table", c_void_p * MODEL_MAX_TABLES),
        ("rowptr", c_void_p), ("col", c_void_p), ("perm", c_void_p), ("rowptr_t", c_void_p), ("col_t", c_void_p), ("perm_t", c_void_p),
        ("seg_ptr", c_void_p), ("edge_src", c_void_p), ("edge_dst", c_void_p), ("csr_flags", c_void_p), ("knots", c_void_p),
        ("base_weight", c_void_p * MODEL_MAX_LAYERS), ("spline_weight", c_void_p * MODEL_MAX_LAYERS), ("spline_scaler", c_void_p * MODEL_MAX_LAYERS),
        ("bn_weight", c_void_p * MODEL_MAX_CONVS), ("bn_bias", c_void_p * MODEL_MAX_CONVS),
        ("running_mean", c_void_p * MODEL_MAX_CONVS), ("running_var", c_void_p * MODEL_MAX_CONVS),
        ("readout_knots", c_void_p * MODEL_MAX_READOUT), ("readout_base_weight", c_void_p * MODEL_MAX_READOUT),
        ("readout_spline_weight", c_void_p * MODEL_MAX_READOUT), ("readout_spline_scaler", c_void_p * MODEL_MAX_READOUT),
        ("saved", c_void_p), ("saved_bytes", c_int64), ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("out", c_void_p), ("g_out", c_void_p), ("ld_g_out", c_int64), ("grads", c_void_p),
        ("self_scale", c_float * MODEL_MAX_CONVS), ("momentum", c_float * MODEL_MAX_CONVS), ("eps", c_float * MODEL_MAX_CONVS),
    ]


EXPORTED = tuple(_SIGNATURES)
_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m kagnn_amd._build` (needs hipcc). "
                "kagnn_amd has no CPU or eager-torch fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.kagnn_kagin_model_struct_bytes() != ctypes.sizeof(KaginModel):
            raise RuntimeError(f"kagnn_kagin_model_t is {lib.kagnn_kagin_model_struct_bytes()} bytes in {LIB_PATH}, its ctypes mirror "
                               f"{ctypes.sizeof(KaginModel)}: header and kagnn_amd/_lib.py disagree")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().kagnn_last_error()
        raise RuntimeError(f"libkagnn_hip {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def call(name: str, *args) -> None:
    check(getattr(load(), name)(*args), name)
