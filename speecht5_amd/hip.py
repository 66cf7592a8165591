"""ctypes binding of libspeecht5_hip.so (C ABI declared in include/speecht5_hip.h).

PyTorch is used only for device memory (`tensor.data_ptr()`), the current HIP stream and
`torch.distributed`; every compute call of the hot path goes through this binding.  There is NO
fallback: if the shared library is missing or a kernel returns an error the call raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

import torch

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_LRELU_01, ACT_LRELU_001 = 0, 1, 2, 3, 4, 5
A_KSTRIDED, B_KSTRIDED, OUT_F32, DACT = 1, 2, 4, 8
DEFERRABLE = 16  # output not read before the data-parallel wrapper flushes the batched split-K reductions

_LIB_PATH = os.environ.get("ST5_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspeecht5_hip.so")


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class Operand(Structure):
    _fields_ = [("ptr", c_void_p), ("ld", c_int64), ("rpb", c_int32), ("seg", c_int32),
                ("bstride", c_int64), ("seg_stride", c_int64), ("zs0", c_int64), ("zs1", c_int64)]


class GemmParams(Structure):
    _fields_ = [("A", Operand), ("B", Operand), ("C", Operand), ("R", Operand), ("P", Operand), ("Cpre", Operand),
                ("bias", c_void_p), ("bias_zs", c_int64),
                ("M", c_int32), ("N", c_int32), ("K", c_int32), ("batch", c_int32), ("zdiv", c_int32),
                ("act", c_int32), ("flags", c_int32),
                ("alpha", c_float), ("beta", c_float), ("dropout_p", c_float), ("seed", c_uint64), ("asum", c_void_p)]


_SIGS = {
    "st5_gemm": (c_int, [POINTER(GemmParams), c_int, c_void_p]),
    "st5_gemm_set_glds": (c_int, [c_int]),
    "st5_stream_fork": (c_int, [c_void_p, c_void_p]),
    "st5_gemm_set_nt_tile": (c_int, [c_int]),
    "st5_gemm_set_m64_max_tiles": (c_int, [c_int]),
    "st5_gemm_set_nt_longk": (c_int, [c_int, c_int]),
    "st5_gemm_set_mx8_tile": (c_int, [c_int]),
    "st5_gemm_set_mx8_heavy_nk": (c_int, [c_int]),
    "st5_gemm_set_tn_group_tile": (c_int, [c_int]),
    "st5_gemm_tn_group_is_phased": (c_int, [c_int32, c_int32, c_int32]),
    "st5_gemm_set_splitk_target": (c_int, [c_int]),
    "st5_gemm_set_deep_ring": (c_int, [c_int, c_int]),
    "st5_gemm_set_nt_slots": (c_int, [c_int]),
    "st5_gemm_tn_group": (c_int, [c_void_p, c_int32, c_int, c_void_p]),
    "st5_gemm_defer_splitk": (c_int, [c_int, c_void_p]),
    "st5_gemm_flush_splitk": (c_int, [c_void_p]),
    "st5_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                  c_float, c_int, c_void_p]),
    "st5_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_int32, c_void_p, c_float, c_uint64, c_int, c_void_p]),
    "st5_layernorm_gated_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                        c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "st5_layernorm_gated_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int32, c_void_p, c_float, c_uint64, c_void_p, c_int, c_void_p]),
    "st5_layernorm_defer": (c_int, [c_int, c_void_p]),
    "st5_layernorm_flush": (c_int, [c_void_p]),
    "st5_layernorm_bwd_ws_bytes": (c_int64, [c_int64, c_int32]),
    "st5_softmax_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                c_int32, c_int32, c_int32, c_int32, c_float, c_uint64, c_int, c_void_p]),
    "st5_softmax_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                c_int32, c_float, c_uint64, c_int, c_void_p]),
    "st5_flash_attn_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                   c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_float, c_float, c_uint64, c_int, c_void_p]),
    "st5_flash_attn_fwd_qp": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_float, c_float, c_uint64, c_void_p, c_int, c_void_p]),
    "st5_flash_attn_bwd_2s": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      c_int32, c_float, c_float, c_uint64, c_int, c_void_p, c_void_p]),
    "st5_flash_attn_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int32, c_float, c_float, c_uint64, c_int, c_void_p]),
    "st5_flash_attn_set_impl": (c_int, [c_int]),
    "st5_gemm_set_tn_phased": (c_int, [c_int]),
    "st5_gemm_mxfp8": (c_int, [POINTER(GemmParams), c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "st5_quant_mxfp8": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_void_p]),
    "st5_multi_quant_mxfp8": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "st5_gemm_mxfp8_q": (c_int, [POINTER(GemmParams), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "st5_layernorm_bwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                      c_void_p, c_int, c_void_p]),
    "st5_layernorm_gelu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int, c_void_p]),
    "st5_layernorm_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int32, c_int, c_void_p]),
    "st5_layernorm_fwd_q8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                                     c_void_p]),
    "st5_flash_attn_qp_row": (c_int32, [c_int32]),
    "st5_flash_attn_qp_table": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_int, c_void_p]),
    "st5_conv0_gn_gelu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                      c_int32, c_int32, c_int32, c_int32, c_float, c_int, c_void_p]),
    "st5_conv0_gn_gelu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int,
                                      c_void_p]),
    "st5_conv0_gn_gelu_fwd_m": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                        c_int32, c_int32, c_int32, c_int32, c_float, c_int, c_void_p]),
    "st5_conv0_gn_gelu_bwd_m": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int,
                                        c_void_p]),
    "st5_conv0_mom_count": (c_int32, [c_int32]),
    "st5_conv0_ws_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "st5_conv0_set_mfma": (c_int, [c_int]),
    "st5_conv0_set_fold": (c_int, [c_int]),
    "st5_conv0_set_gelu_table": (c_int, [c_int]),
    "st5_cast_from_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int, c_void_p]),
    "st5_cast_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "st5_colsum_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_float, c_int32, c_int,
                              c_void_p]),
    "st5_colsum_ws_bytes": (c_int64, [c_int64, c_int32]),
    "st5_sumsq": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_int32, c_int, c_void_p]),
    "st5_axpby": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_int, c_void_p]),
    "st5_select": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "st5_select_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "st5_skip_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "st5_act_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "st5_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "st5_channel_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    "st5_dropout": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_uint64, c_int, c_void_p]),
    "st5_masked_fill_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "st5_masked_fill_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "st5_add_table_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int,
                                   c_void_p]),
    "st5_add_table_rows_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int, c_void_p]),
    "st5_embed_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float,
                               c_int, c_void_p]),
    "st5_embed_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int, c_void_p]),
    "st5_embed_rows_bwd_det": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float, c_int, c_void_p]),
    "st5_unfold_rows": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_pad_time": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_pad_time_act": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_zero_halo": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_conv1d_narrow": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_int32,
                                  c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int32, c_void_p]),
    "st5_conv1d_cout1": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_float, c_int32, c_void_p]),
    "st5_ragged_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint64, c_void_p]),
    "st5_tail_mask": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "st5_stft_frames": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "st5_stft_magnitude": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "st5_log10_floor": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "st5_cross_entropy": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                  c_float, c_int32, c_float, c_int, c_void_p]),
    "st5_cross_entropy_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                  c_float, c_int32, c_float, c_int, c_void_p]),
    "st5_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                              c_int32, c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "st5_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                  c_int32, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "st5_tacotron_loss_ws_bytes": (c_int64, []),
    "st5_tacotron_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32,
                                      c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "st5_tacotron_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32,
                                      c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "st5_embed_rows_bwd_det_w": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_int32, c_int32, c_int, c_void_p]),
    "st5_ctc_loss_ws_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "st5_ctc_loss_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "st5_ctc_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "st5_guided_attn_ws_bytes": (c_int64, []),
    "st5_guided_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p,
                                    c_void_p]),
    "st5_guided_attn_bwd": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "st5_vq_ws_bytes": (c_int64, []),
    "st5_vq_vpad": (c_int32, []),
    "st5_vq_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_vq_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p,
                           c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_norm_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "st5_norm_rows_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    "st5_canon_rows": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "st5_nce_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "st5_nce_logits_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "st5_gather3": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64, c_int64, c_int64, c_int32, c_int, c_void_p]),
    "st5_zero_time_edges": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "st5_layernorm_set_max_blocks": (c_int, [c_int]),
    "st5_adam_step_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                   c_int32, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "st5_sumsq_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_int32, c_void_p]),
    "st5_ctc_initial_state": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "st5_ctc_prefix_score": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p]),
    "st5_multi_transpose_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "st5_batchnorm_ws_bytes": (c_int64, [c_int32]),
    "st5_batchnorm_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int32, c_int32,
                                      c_float, c_uint64, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                      c_int64, c_int64, c_int32, c_int, c_void_p]),
    "st5_batchnorm_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float,
                                      c_uint64, c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int64, c_int64, c_int32, c_int, c_void_p]),
    "st5_version": (c_char_p, []),
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises HipLibraryMissing if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipLibraryMissing(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C speecht5_amd/csrc`). The SpeechT5 hot path has no CPU/PyTorch fallback.")
        L = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def dt(t):
    if isinstance(t, torch.Tensor):
        t = t.dtype
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  Called once per kernel launch (~3000 per training
    step): the raw C getter costs ~0.3 us, `torch.cuda.current_stream().cuda_stream` ~3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise HipKernelError(f"{what} failed with ST5 error {rc} "
                             "(1=bad argument, 2=alignment, 3=launch)")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def operand(t, ld, off=0, rpb=0, bstride=0, seg=0, seg_stride=0, zs0=0, zs1=0):
    """Describe a GEMM operand living in tensor `t` starting `off` elements in."""
    o = Operand()
    o.ptr = t.data_ptr() + off * t.element_size()
    o.ld, o.rpb, o.seg, o.bstride, o.seg_stride, o.zs0, o.zs1 = ld, rpb, seg, bstride, seg_stride, zs0, zs1
    return o


_NULL_OP = Operand()


class GemmProfiler:
    """Optional per-launch HIP-event timing of st5_gemm (bench.py's roofline leg).  Events are recorded on the
    stream the kernel is launched on, around every launch, while the timed region runs."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (variant, flops, start_event, end_event)
        self.regions = []  # (name, algorithmic_bytes, start_event, end_event): HBM-bound kernels timed the same way

    def reset(self):
        self.records = []
        self.regions = []

    def region(self, name, nbytes, flops=0.0):
        """Context manager: HIP events around a launch sequence on the current stream (no-op unless enabled)."""
        prof = self

        class _R:
            def __enter__(self_):
                if prof.enabled:
                    self_.e0 = torch.cuda.Event(enable_timing=True); self_.e1 = torch.cuda.Event(enable_timing=True)
                    self_.e0.record()
                return self_

            def __exit__(self_, *exc):
                if prof.enabled:
                    self_.e1.record()
                    prof.regions.append((name, float(nbytes), self_.e0, self_.e1, float(flops)))
                return False
        return _R()

    def region_summary(self):
        out = {}
        for name, nbytes, e0, e1, _fl in self.regions:
            n, b, t = out.get(name, (0, 0.0, 0.0))
            out[name] = (n + 1, b + nbytes, t + e0.elapsed_time(e1) * 1e-3)
        return out

    def region_flops(self):
        out = {}
        for name, _b, _e0, _e1, fl in self.regions:
            out[name] = out.get(name, 0.0) + fl
        return out

    def summary(self):
        """{variant: (launches, total_flops, total_seconds)} -- call after torch.cuda.synchronize()."""
        out = {}
        for variant, flops, e0, e1, _shape in self.records:
            n, f, t = out.get(variant, (0, 0.0, 0.0))
            out[variant] = (n + 1, f + flops, t + e0.elapsed_time(e1) * 1e-3)
        return out


    def nt_bytes_per_launch(self, variant="bf16_NT"):
        """Mean algorithmic HBM bytes (A + B read once, C written once) of the recorded launches of one variant."""
        tot, n = 0, 0
        es = 2 if variant.startswith("bf16") else 4
        for v, _f, _e0, _e1, shape in self.records:
            if v == variant:
                M, N, K, b = shape
                tot += b * (M * K + N * K + M * N) * es; n += 1
        return int(tot / n) if n else None

    def by_shape(self):
        out = {}
        for variant, flops, e0, e1, shape in self.records:
            k = (variant,) + shape
            n, f, t = out.get(k, (0, 0.0, 0.0))
            out[k] = (n + 1, f + flops, t + e0.elapsed_time(e1) * 1e-3)
        return out


profiler = GemmProfiler()


def gemm(A, B, C, M, N, K, dtype, *, R=None, P=None, Cpre=None, bias=None, bias_zs=0, batch=1, zdiv=1, act=ACT_NONE,
         flags=0, alpha=1.0, beta=0.0, dropout_p=0.0, seed=0, asum=None, on=None):
    """on: torch.cuda.Stream to launch on instead of the current one (the caller orders it: st5_stream_fork)."""
    p = GemmParams()
    p.A, p.B, p.C = A, B, C
    p.R = R if R is not None else _NULL_OP
    p.P = P if P is not None else _NULL_OP
    p.Cpre = Cpre if Cpre is not None else _NULL_OP
    p.bias = ptr(bias)
    p.bias_zs = bias_zs
    p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, batch, zdiv
    p.act, p.flags = act, flags
    p.alpha, p.beta, p.dropout_p, p.seed = alpha, beta, dropout_p, seed
    p.asum = ptr(asum)
    if profiler.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(on)
        check(lib().st5_gemm(byref(p), dtype, stream() if on is None else on.cuda_stream), "st5_gemm")
        e1.record(on)
        variant = ("bf16" if dtype == BF16 else "f32") + "_" + ("T" if flags & A_KSTRIDED else "N") + ("N" if flags & B_KSTRIDED else "T")
        profiler.records.append((variant, 2.0 * M * N * K * batch, e0, e1, (M, N, K, batch)))
        return
    check(lib().st5_gemm(byref(p), dtype, stream() if on is None else on.cuda_stream), "st5_gemm")


def gemm_tn_group(problems, dtype, on=None):
    """problems: [(A, B, C, M, N, K, flags, beta, asum)] weight-gradient GEMMs -> ONE launch without split-K (st5_gemm_tn_group;
    the library falls back to one st5_gemm per problem when the group does not qualify).  The caller keeps the operands alive until
    the call returns (the launch is enqueued on the current stream by then).  on: torch.cuda.Stream to launch on instead of the
    current one (the caller orders it behind the producers: st5_stream_fork, and keeps the operands alive until the streams join)."""
    n = len(problems)
    arr = (GemmParams * n)()
    flops = 0.0
    for p, (A, B, C, M, N, K, flags, beta, asum) in zip(arr, problems):
        p.A, p.B, p.C = A, B, C
        p.R = p.P = p.Cpre = _NULL_OP
        p.bias, p.bias_zs = None, 0
        p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, 1, 1
        p.act, p.flags = ACT_NONE, flags
        p.alpha, p.beta, p.dropout_p, p.seed = 1.0, beta, 0.0, 0
        p.asum = ptr(asum)
        flops += 2.0 * M * N * K
    raw = stream() if on is None else on.cuda_stream
    if profiler.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(on)
        check(lib().st5_gemm_tn_group(arr, n, dtype, raw), "st5_gemm_tn_group")
        e1.record(on)
        profiler.records.append((("bf16" if dtype == BF16 else "f32") + "_TN", flops, e0, e1, (problems[0][3], problems[0][4], problems[0][5], n)))
        return
    check(lib().st5_gemm_tn_group(arr, n, dtype, raw), "st5_gemm_tn_group")


def quant_mxfp8(x2):
    """MX-fp8 image of a contiguous bf16 matrix [rows, cols] (cols % 32 == 0): (q uint8 [rows, cols], s uint8 [rows, cols/32])."""
    rows, cols = x2.shape
    assert x2.dtype == torch.bfloat16 and x2.stride(1) == 1 and cols % 32 == 0
    q = torch.empty(rows, cols, dtype=torch.uint8, device=x2.device)
    sc = torch.empty(rows, cols // 32, dtype=torch.uint8, device=x2.device)
    check(lib().st5_quant_mxfp8(x2.data_ptr(), x2.stride(0), q.data_ptr(), cols, sc.data_ptr(), cols // 32, rows, cols, stream()), "st5_quant_mxfp8")
    return q, sc


def gemm_mxfp8(Aq, As, Bq, Bs, C, M, N, K, *, R=None, P=None, Cpre=None, bias=None, act=ACT_NONE, flags=0, alpha=1.0, beta=0.0,
               dropout_p=0.0, seed=0, out_q=None):
    """C (bf16 operand) = epilogue(Aq . Bq^T) on the block-scaled fp8 MFMA; Aq / Bq uint8 [rows, K] with scales As / Bs [rows, K/32].
    out_q = (q uint8 [M, N], s uint8 [M, N/32]): the epilogue also writes C's MX-fp8 image (st5_gemm_mxfp8_q)."""
    p = GemmParams()
    p.A, p.B, p.C = operand(Aq, Aq.stride(0)), operand(Bq, Bq.stride(0)), C
    p.R = R if R is not None else _NULL_OP
    p.P = P if P is not None else _NULL_OP
    p.Cpre = Cpre if Cpre is not None else _NULL_OP
    p.bias = ptr(bias)
    p.bias_zs = 0
    p.M, p.N, p.K, p.batch, p.zdiv = M, N, K, 1, 1
    p.act, p.flags = act, flags
    p.alpha, p.beta, p.dropout_p, p.seed = alpha, beta, dropout_p, seed
    p.asum = 0
    def launch():
        if out_q is not None:
            oq, os_ = out_q
            check(lib().st5_gemm_mxfp8_q(byref(p), As.data_ptr(), As.stride(0), Bs.data_ptr(), Bs.stride(0), oq.data_ptr(), oq.stride(0),
                                         os_.data_ptr(), os_.stride(0), stream()), "st5_gemm_mxfp8_q")
        else:
            check(lib().st5_gemm_mxfp8(byref(p), As.data_ptr(), As.stride(0), Bs.data_ptr(), Bs.stride(0), stream()), "st5_gemm_mxfp8")
    if profiler.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        profiler.records.append(("fp8_NT", 2.0 * M * N * K, e0, e1, (M, N, K, 1)))
        return
    launch()


# ---------------------------------------------------------------------------------------------
# workspace: one grow-only byte buffer per device for the two-stage reductions
_ws = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer of the CURRENT stream (two micro-batches may run forward side by side on two streams: a
    buffer shared between streams would be written by both)."""
    key = (device.type, device.index, stream())
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf
