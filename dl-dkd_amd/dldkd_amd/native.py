"""ctypes binding of libdldkd_hip.so (C ABI: include/dldkd_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  torch is used only
for device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdldkd_hip.so")
ABI_VERSION = 7

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_void_p = ctypes.c_void_p
_c_size_t = ctypes.c_size_t
_c_long = ctypes.c_long

# name -> (restype, argtypes); must list every symbol include/dldkd_hip.h declares
SIGNATURES = {
    "dldkd_abi_version": (_c_int, []),
    "dldkd_last_error": (ctypes.c_char_p, []),
    "dldkd_gemm_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "dldkd_packed_queries_bytes": (_c_size_t, [_c_int]),
    "dldkd_packed_gallery_bytes": (_c_size_t, [_c_int, _c_int]),
    "dldkd_simpool_eval_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "dldkd_pack_queries_bf16": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_pack_gallery_bf16": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_eval_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_eval_pairs_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                                _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_eval_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "dldkd_simpool_finish_range": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_int, _c_int,
                                             _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_stream_wait_counter": (_c_int, [_c_void_p, _c_void_p, ctypes.c_int32]),
    "dldkd_simpool_finish": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_gemm_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "dldkd_layernorm_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                      _c_float, _c_void_p]),
    "dldkd_attention_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    "dldkd_modpool_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    "dldkd_simpool_rank_partials": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p, _c_void_p,
                                              _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_rank_partials_thr": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p, _c_void_p,
                                                  _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_rank_partials_count": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_rank_gt": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_layernorm_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                          _c_long, _c_int, _c_float, _c_void_p, _c_float, _c_void_p]),
    "dldkd_layernorm_bwd_groups_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                          _c_long, _c_int, _c_float, _c_void_p, _c_float, _c_void_p, _c_void_p]),
    "dldkd_layernorm_dropout_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                              _c_float, _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_layernorm_dropout_rows_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_float,
                                                   _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_layernorm_dropout_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_float,
                                               _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_gemm_f32x2": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                                  _c_void_p]),
    "dldkd_split2_bf16_jobs": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "dldkd_gemm_bf16_nt16_planes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                             _c_void_p, _c_long, _c_long, _c_void_p]),
    "dldkd_row_invnorm2_planes_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "dldkd_simpool_train_fwd_planes": (_c_int, [_c_void_p] * 6 + [_c_int, _c_int, _c_int, _c_int] + [_c_void_p] * 6),
    "dldkd_layernorm_ex_f32": (_c_int, [_c_void_p] * 10 + [_c_long, _c_int, _c_float, _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p,
                                         _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_train_emit": (_c_int, [_c_void_p, _c_void_p, _c_int] + [_c_void_p] * 8 + [_c_long] + [_c_void_p] * 9),
    "dldkd_layernorm_dropout_bf16_dual": (_c_int, [_c_void_p] * 8 + [_c_long, _c_int, _c_float, _c_float, ctypes.c_uint64, ctypes.c_uint64,
                                                    ctypes.c_uint64, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "dldkd_attention_train_fwd_bf16io": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float, ctypes.c_uint64,
                                                   ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_attention_train_bwd_bf16io": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float,
                                                   ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_tower_train_pack_bytes": (_c_size_t, [_c_int]),
    "dldkd_tower_train_pack": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "dldkd_tower_train_prepare": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "dldkd_tower_train_f1": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_float, _c_float, ctypes.c_uint64, ctypes.c_uint64,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_train_f3": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p,
                                       _c_void_p, _c_void_p, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_train_b3": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, ctypes.c_uint64, ctypes.c_uint64,
                                       _c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_train_b1": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float,
                                       ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_train_dw_workspace_bytes": (_c_size_t, [_c_int, _c_long]),
    "dldkd_tower_train_dw": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_void_p, _c_void_p, _c_void_p,
                                       _c_size_t, _c_void_p, _c_void_p]),
    "dldkd_tower_train_dw_pos": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_void_p, _c_void_p, _c_void_p,
                                           _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_long, _c_void_p]),
    "dldkd_tower_train_dw_ln": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_void_p, _c_void_p, _c_void_p,
                                          _c_size_t, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                          _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_mask_lens_f32": (_c_int, [_c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "dldkd_colsum_bf16": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_long, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_branch_losses_f32": (_c_int, [_c_void_p] * 11 + [_c_int] * 7 + [_c_float] * 6 + [_c_void_p] * 5 + [_c_int, _c_void_p, _c_void_p]),
    "dldkd_branch_losses_scale_f32": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_sum_scalars_f32": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "dldkd_zero_scratch_f32": (_c_int, [_c_void_p, _c_int, _c_void_p]),
    "dldkd_set_zero_by_memset": (_c_int, [_c_int]),
    "dldkd_inproj_bwd_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_long]),
    "dldkd_inproj_bwd_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_void_p, _c_void_p,
                                       _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p,
                                       ctypes.c_size_t, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_gemm_bf16_dw_bias": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_size_t,
                                          _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_gemm_bf16_mixed": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        _c_int, _c_void_p, _c_size_t, _c_void_p, _c_void_p]),
    "dldkd_gemm_bf16_nt": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                     _c_void_p, _c_void_p]),
    "dldkd_layernorm_groups_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                             _c_float, _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_cast_bf16": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p]),
    "dldkd_gemm_bf16_nt16_ok": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "dldkd_gemm_bf16_nt16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                       _c_void_p, _c_void_p]),
    "dldkd_gemm_bf16_nt_ok": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "dldkd_colsum_f32": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_long, _c_void_p]),
    "dldkd_relu_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p]),
    "dldkd_axpy_f32": (_c_int, [_c_void_p, _c_void_p, _c_float, _c_long, _c_void_p]),
    "dldkd_mul_f32": (_c_int, [_c_void_p, _c_void_p, _c_float, _c_void_p, _c_long, _c_void_p]),
    "dldkd_normalize_rows_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "dldkd_normalize_rows_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "dldkd_clip_pool_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "dldkd_clip_pool_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "dldkd_modpool_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int,
                                        _c_void_p]),
    "dldkd_kl_frame_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_int, _c_int, _c_int, _c_void_p,
                                     _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_nce_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float, _c_float, _c_int,
                                _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_triplet_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_float, _c_int, _c_int, _c_void_p,
                                    _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_sum_f32": (_c_int, [_c_void_p, _c_long, _c_void_p, _c_void_p]),
    "dldkd_attention_train_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float, ctypes.c_uint64,
                                                ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_attention_train_bwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float,
                                                ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_attention_train_fwd_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float, ctypes.c_uint64,
                                                 ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_attention_train_bwd_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_float,
                                                 ctypes.c_uint64, ctypes.c_uint64, _c_void_p, _c_void_p]),
    "dldkd_row_invnorm_f32": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "dldkd_row_invnorm2_f32": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "dldkd_row_invnorm2_cast_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                             _c_void_p]),
    "dldkd_simpool_train_fwd_bf16in": (_c_int, [_c_void_p] * 6 + [_c_int] * 4 + [_c_void_p] * 6),
    "dldkd_simpool_train_fwd_f32": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int,
                                              _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_simpool_train_bwd_f32": (_c_int, [_c_void_p] * 13 + [_c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_bert_adam_step_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                           _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_float, _c_float,
                                           _c_float, _c_void_p]),
    "dldkd_bert_adam_update_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                             _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_float, _c_float,
                                             _c_float, _c_void_p]),
    "dldkd_gather_sumsq_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_count_above_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p]),
    "dldkd_fold_ln_linear_h16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p,
                                            _c_void_p, _c_void_p]),
    "dldkd_in_proj_h16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int,
                                     _c_float, _c_int, _c_void_p]),
    "dldkd_segment_mean_l2norm_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_float, _c_void_p]),
    "dldkd_upload_words": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p]),
    "dldkd_rows_to_h16_stats": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_void_p,
                                           _c_void_p, _c_void_p]),
    "dldkd_in_proj_h16_rows128b": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                              _c_long, _c_int, _c_int, _c_void_p, _c_long, _c_void_p]),
    "dldkd_in_proj_h16_rows128b_out16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                              _c_long, _c_int, _c_int, _c_void_p, _c_long, _c_void_p]),
    "dldkd_in_proj_h16_rows128b_ok": (_c_int, [_c_int]),
    "dldkd_gather_pad_rows_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                            _c_void_p]),
    "dldkd_fold_ln_linear_h16_frag": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                                                 _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_in_proj_h16_full": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                          _c_float, _c_int, _c_void_p]),
    "dldkd_in_proj_h16_rows128": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                             _c_float, _c_int, _c_void_p]),
    "dldkd_in_proj_h16_rows128_ok": (_c_int, [_c_int]),
    "dldkd_in_proj_h16_rows128_groups": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                                    _c_float, _c_int, _c_void_p, _c_long, _c_void_p]),
    "dldkd_row_meanrstd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_float, _c_void_p]),
    "dldkd_linear_lngrad": (_c_int, [_c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_void_p, _c_void_p, _c_void_p,
                                           ctypes.c_size_t, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p]),
    "dldkd_fold_ln_linear_planes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                             _c_void_p]),
    "dldkd_in_proj_f32x3_rows128": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long,
                                             _c_int, _c_int, _c_void_p]),
    "dldkd_in_proj_f32x3_rows128_ok": (_c_int, [_c_int]),
    "dldkd_pack_linear_planes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                                          _c_void_p, _c_void_p]),
    "dldkd_linear_f32x3_rows": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int,
                                         _c_int, _c_int, _c_int, _c_void_p]),
    "dldkd_debug_in_proj_rows128_timeline": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long,
                                                       _c_int, _c_float, _c_int, _c_void_p, _c_void_p]),
    "dldkd_attention_fwd_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "dldkd_pack_linear_h16_frag": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_linear_rows_h16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_int, _c_int,
                                         _c_int, _c_int, _c_void_p]),
    "dldkd_pack_gallery_chunk_bf16": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_int,
                                                _c_int, _c_void_p]),
    "dldkd_dropout_fwd_f32": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_float, ctypes.c_uint64, ctypes.c_uint64, _c_void_p,
                                        _c_void_p]),
    "dldkd_mask_scale_f32": (_c_int, [_c_void_p, _c_void_p, _c_float, _c_void_p, _c_long, _c_void_p]),
    "dldkd_gemm_f32x3": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    "dldkd_gemm_f32x3_flags": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p, _c_void_p]),
    "dldkd_order_by_len_desc": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_blob_bytes": (_c_size_t, [_c_int]),
    "dldkd_tower_pack_h16": (_c_int, [_c_void_p] * 16 + [_c_int, _c_void_p, _c_void_p]),
    "dldkd_tower_seq_h16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                      _c_int, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "dldkd_tower_seq_h16_rows16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                          _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p]),
    "dldkd_debug_tower_seq_timeline": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                                _c_void_p, _c_int, _c_void_p, _c_int, _c_void_p]),
    "dldkd_gemm_bf16": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         _c_int, _c_int, _c_int, _c_void_p, _c_size_t, _c_void_p]),
    # collectives (RCCL, driven directly; dldkd_amd.comm.RcclComm)
    "dldkd_comm_rccl_version": (_c_int, []),
    "dldkd_comm_unique_id": (_c_int, [_c_void_p]),
    "dldkd_comm_init": (_c_int, [ctypes.POINTER(_c_void_p), _c_int, _c_int, _c_void_p]),
    "dldkd_comm_info": (_c_int, [_c_void_p, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "dldkd_comm_destroy": (_c_int, [_c_void_p]),
    "dldkd_comm_abort": (_c_int, [_c_void_p]),
    "dldkd_comm_async_error": (_c_int, [_c_void_p]),
    "dldkd_comm_all_reduce": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_void_p]),
    "dldkd_comm_all_gather": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_void_p]),
    "dldkd_comm_broadcast": (_c_int, [_c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_void_p]),
    "dldkd_comm_group_begin": (_c_int, []),
    "dldkd_comm_group_end": (_c_int, []),
}

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise NativeError if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `make -C dl-dkd_amd/csrc` (or __graft_entry__.build()). "
                "dldkd_amd has no CPU/eager fallback by design.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)      # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if h.dldkd_abi_version() != ABI_VERSION:
            raise NativeError(f"libdldkd_hip.so ABI {h.dldkd_abi_version()} != expected {ABI_VERSION}")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed (rc={rc}): {lib().dldkd_last_error().decode()}")


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NativeError("dldkd_amd kernels need tensors on the GPU (no CPU path)")
    if not t.is_contiguous():
        raise NativeError("non-contiguous tensor passed to a dldkd_amd kernel")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The caller's current HIP stream as a void*.  torch.cuda.current_stream() builds a Stream object per call (~1 us,
    0.8 ms per step of a launch-bound training step); the raw-stream query is a plain C call."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr
