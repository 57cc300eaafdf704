"""ctypes binding of libscl_hip.so — the only way the Python host reaches the HIP kernels.

The product path has no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, importing / calling raises immediately.
"""
import ctypes
import os

import torch  # noqa: F401  — MUST precede CDLL: torch ships its own libamdhip64; loading ours first would put two HIP
#                             runtimes in the process and the kernels would see "no ROCm-capable device"

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCL_LIB_PATH") or os.path.join(HERE, "libscl_hip.so")      # SCL_LIB_PATH: an A/B build of build.py (SCL_BUILD_TAG)
FLAT = 0x7FFFFFFF

_lib = None


class SclError(RuntimeError):
    pass


class SclOperand(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("bs1", ctypes.c_int64), ("bs2", ctypes.c_int64),
                ("rbstride", ctypes.c_int64), ("cout", ctypes.c_int64), ("rpb", ctypes.c_int32),
                ("ld", ctypes.c_int32), ("cin", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class SclReduceJob(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out2", ctypes.c_void_p), ("pstride", ctypes.c_int64),
                ("nparts", ctypes.c_int32), ("C", ctypes.c_int32), ("split", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class SclSlabJob(ctypes.Structure):
    _fields_ = [("slabs", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_int64), ("stride", ctypes.c_int64),
                ("nslabs", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class SclRsGeom(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("r_lo", ctypes.c_int32), ("r_hi", ctypes.c_int32),
                ("_pad", ctypes.c_int32)]


class SclRsPackJob(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("out", ctypes.c_void_p), ("Co", ctypes.c_int32), ("Ci", ctypes.c_int32), ("ntaps", ctypes.c_int32),
                ("CINp", ctypes.c_int32), ("COUTp", ctypes.c_int32), ("transposed", ctypes.c_int32), ("ld", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class SclRsConv(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("inp", "wpk", "bias", "addend", "out", "act_a", "y1", "bnstats", "acc", "ticket", "gamma", "beta",
                                              "run_mean", "run_var", "nbt", "stats_out", "dgamma", "dbeta")] + \
               [("nvalid", ctypes.c_double), ("geom", SclRsGeom), ("shift", ctypes.c_int32 * 6), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("ntaps", ctypes.c_int32), ("stat_mode", ctypes.c_int32), ("training", ctypes.c_int32), ("epi_act", ctypes.c_int32),
                ("eps", ctypes.c_float), ("momentum", ctypes.c_float)]


class SclBtseBio(ctypes.Structure):
    """include/scl_hip.h SclBtseBio: the bio transformer of the wav2vec2_btse plugin (csrc/btse.hip)."""
    _fields_ = [("emb", ctypes.c_void_p), ("lw", (ctypes.c_void_p * 18) * 8), ("Ws", ctypes.c_void_p), ("bs", ctypes.c_void_p),
                ("bio", ctypes.c_void_p), ("lens", ctypes.c_void_p), ("ws", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("d_out", ctypes.c_void_p), ("slab", ctypes.c_void_p), ("ws_stride", ctypes.c_int64), ("slab_ld", ctypes.c_int64),
                ("go", (ctypes.c_int32 * 18) * 8), ("go_emb", ctypes.c_int32), ("go_Ws", ctypes.c_int32), ("go_bs", ctypes.c_int32)] + \
               [(n, ctypes.c_int32) for n in ("n_layers", "n_bios", "bio_out", "L", "B", "out_ld", "dout_ld", "bio_dim", "n_heads", "pf_dim", "window")]


class SclGemmDesc(ctypes.Structure):
    _fields_ = [("A", SclOperand), ("B", SclOperand), ("C", ctypes.c_void_p), ("C2", ctypes.c_void_p),
                ("R", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("c_bs1", ctypes.c_int64), ("c_bs2", ctypes.c_int64), ("c_rbstride", ctypes.c_int64),
                ("c_split_stride", ctypes.c_int64), ("bias_bs2", ctypes.c_int64),
                ("c_rpb", ctypes.c_int32), ("ldc", ctypes.c_int32),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("nb1", ctypes.c_int32), ("nb2", ctypes.c_int32), ("splitk", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("alpha", ctypes.c_float), ("drop_p", ctypes.c_float),
                ("drop_seed", ctypes.c_uint32), ("_pad", ctypes.c_int32), ("colsum_part", ctypes.c_void_p)]


# flags (include/scl_hip.h)
GEMM_A_T, GEMM_B_T, GEMM_C_F32, GEMM_C2_F32, GEMM_R_F32 = 1, 2, 4, 8, 16
GEMM_HAS_BIAS, GEMM_HAS_C2, GEMM_DROPOUT = 0x20, 0x40, 0x80
GEMM_NO_DMA = 0x00100000
GEMM_NO_BIG = 0x00200000
GEMM_FORCE_BIG = 0x04000000
GEMM_NO_P8 = 0x00400000
GEMM_FORCE_P8 = 0x00800000
GEMM_NO_W8 = 0x01000000
GEMM_FORCE_W8 = 0x02000000
GEMM_FORCE_X2 = 0x08000000
GEMM_NO_X2 = 0x10000000
GEMM_AB_F32 = 0x40000000
GEMM_C_SPLIT3 = -0x80000000      # bit 31 of the int32 flag word: bf16 C as [hi | hi | lo] planes (wide tiles only)
GEMM_F32X3 = 0x20000000
ACT_SHIFT, RMODE_SHIFT, RACT_SHIFT = 8, 12, 16
ACT_NONE, ACT_GELU, ACT_RELU, ACT_LEAKY = 0, 1, 2, 3
ACT_GELU_DC2 = 5      # gelu whose second output is gelu'(pre-activation); the matching backward epilogue is rmode 2 with ract = RACT_STORED
RACT_STORED = 4
KID_GEMM = 0
KID_AUG = 1
KID_GEMM_F32 = 2

_vp, _i32, _i64, _f32, _f64, _u32 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float,
                                     ctypes.c_double, ctypes.c_uint32)


def _protos():
    """name -> argtypes.  Every symbol declared in include/scl_hip.h must appear here
    (tests/test_abi.py checks both directions)."""
    P = ctypes.POINTER
    return {
        "scl_version": ([], _i32),
        "scl_last_error": ([], ctypes.c_char_p),
        "scl_build_flags": ([], _i32),
        "scl_prof_enable": ([_i32, _i32], _i32),
        "scl_prof_reserve": ([_i32, _i32], _i32),
        "scl_prof_read": ([_i32, P(_i64), P(_f64), P(_f64)], _i32),
        "scl_prof_read_launches": ([_i32, _i32, P(ctypes.c_float), P(_i32), P(_i64)], _i32),
        "scl_gemm_bf16": ([P(SclGemmDesc), _vp], _i32),
        "scl_gemm_bf16_group_ok": ([P(SclGemmDesc), _i32], _i32),
        "scl_gemm_bf16_group": ([P(SclGemmDesc), _i32, _vp], _i32),
        "scl_gemm_bf16_group_tiles": ([P(SclGemmDesc)], _i32),
        "scl_gemm_bf16_group_part": ([P(SclGemmDesc), P(_i32), P(_i32), _i32, _vp], _i32),
        "scl_reduce_slabs_f32": ([_vp, _vp, _i64, _i32, _i64, _vp], _i32),
        "scl_posconv_supported": ([_i32, _i32, _i32, _i32], _i32),
        "scl_posconv_wgrad_supported": ([_i32, _i32, _i32, _i32], _i32),
        "scl_posconv_wgrad": ([_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_posconv_mfma": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_debug_gemm_stamps": ([_vp, _i32], _i32),
        "scl_debug_gemm_persistent_launches": ([], ctypes.c_longlong),
        "scl_gemm_uses_wide_tiles": ([P(SclGemmDesc)], _i32),
        "scl_gemm_colsum_rows": ([P(SclGemmDesc)], _i32),
        # nn.hip
        "scl_bn_nslabs": ([_i32], _i32),
        "scl_bn_fwd": ([_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64,
                        _i64, _i64, _vp], _i32),
        "scl_bn_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp], _i32),
        "scl_pad_nhwc_f32": ([_vp, _i64, _i32, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _vp], _i32),
        "scl_conv_pack_weights": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_conv_wgrad_finish": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_maxpool3_fwd": ([_vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp], _i32),
        "scl_maxpool3_bwd": ([_vp, _vp, _i32, _i32, _i32, _vp, _i64, _i64, _i64, _vp], _i32),
        "scl_avgpool_fwd": ([_vp, _i32, _i32, _i32, _vp, _vp], _i32),
        "scl_avgpool_bwd": ([_vp, _i32, _i32, _i32, _vp, _vp], _i32),
        # norm.hip
        "scl_layernorm_fwd": ([_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _f32, _i32, _vp], _i32),
        "scl_layernorm_bwd_nparts": ([_i32], _i32),
        "scl_layernorm_bwd": ([_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64,
                               _i64, _i32, _i32, _i32, _i64, _i64, _u32, _f32, _u32, _f32, _vp], _i32),
        "scl_colreduce_f32": ([_vp, _vp, _i32, _i32, _i64, _i32, _vp], _i32),
        "scl_colsum_nparts": ([_i32], _i32),
        "scl_colsum_reduce_nparts": ([_i32, _i32], _i32),
        "scl_colsum": ([_vp, _i32, _vp, _i32, _i32, _i64, _vp], _i32),
        "scl_colreduce_multi": ([P(SclReduceJob), _i32, _vp], _i32),
        "scl_reduce_slabs_multi": ([P(SclSlabJob), _i32, _vp], _i32),
        "scl_colreduce_seg_f32": ([_vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _i32, _vp], _i32),
        "scl_colsum_reduce": ([_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _vp], _i32),
        # elementwise.hip
        "scl_cast_f32_bf16": ([_vp, _vp, _i64, _vp], _i32),
        "scl_split3_f32_bf16": ([_vp, _i64, _i32, _i64, _vp, _i32, _vp], _i32),
        "scl_add_f32": ([_vp, _vp, _vp, _vp, _i64, _vp], _i32),
        "scl_pad_rows_bf16": ([_vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_col2im_bf16": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_conv_weight_pack": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_conv_weight_unpack_grad": ([_vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_posconv_weight_pack": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_posconv_weight_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_dropout_f32": ([_vp, _vp, _vp, _i64, _u32, _f32, _vp], _i32),
        "scl_meanpool_fwd": ([_vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_meanpool_bwd": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _u32, _vp], _i32),
        "scl_meanpool_fwd_f32": ([_vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_meanpool_bwd_f32": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _u32, _vp], _i32),
        "scl_utt_head_fwd": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_utt_head_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
        # attention.hip
        "scl_softmax_fwd": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
        "scl_softmax_bwd": ([_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
        "scl_softmax_fwd_f32": ([_vp, _vp, _i64, _i32, _i32, _i32, _vp], _i32),
        "scl_attn_fwd": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp], _i32),
        "scl_attn_fwd_fp8": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        "scl_attn_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u32, _vp], _i32),
        # conv0.hip
        "scl_conv0_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        "scl_conv0_fwd_f32": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        "scl_conv0_bwd_nparts": ([_i32, _i32, _i32, _i32], _i32),
        "scl_conv0_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        # gat.hip
        "scl_gat_score_nblocks": ([_i32], _i32),
        "scl_gat_score_fwd": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_gat_score_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_dropout_rows": ([_vp, _vp, _i64, _i32, _i32, _i32, _u32, _f32, _vp], _i32),
        # resstack.hip
        "scl_rs_conv": ([P(SclRsConv), _vp], _i32),
        "scl_rs_pack_weights": ([P(SclRsPackJob), _i32, _vp], _i32),
        "scl_rs_wgrad_nslabs": ([_i32, _i32], _i32),
        "scl_rs_wgrad": ([_vp, _vp, _i32, _i32, _i32, P(_i32), P(SclRsGeom), _vp, _vp, _vp, _vp, _vp], _i32),
        "scl_rs_wgrad_reduce": ([_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp], _i32),
        "scl_rs_bn_act": ([_vp, _vp, _vp, _i32, _i32, P(SclRsGeom), _vp], _i32),
        "scl_rs_bn_bwd_apply": ([_vp, _vp, _vp, _vp, _i32, _i32, P(SclRsGeom), _vp], _i32),
        "scl_rs_attn_pool_fwd": ([_vp, _vp, _vp, _vp, _vp, _i32, P(SclRsGeom), _vp], _i32),
        "scl_rs_attn_pool_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, P(SclRsGeom), _vp], _i32),
        "scl_rs_bn_eval_stats": ([_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp], _i32),
        "scl_rs_copy": ([_vp, _vp, _i32, _i32, _i32, P(SclRsGeom), _vp], _i32),
        # graph.hip
        "scl_graph_post_fwd": ([_vp, _i32, _i32, _vp], _i32),
        "scl_graph_post_bwd": ([_vp, _i32, _i32, _vp], _i32),
        "scl_graph_pre_fwd": ([_vp, _i32, _i32, _vp], _i32),
        "scl_graph_pre_bwd": ([_vp, _i32, _i32, _vp], _i32),
        "scl_graph_drop": ([_vp, _vp, _i64, _u32, _vp, _vp, _i64, _u32, _f32, _vp], _i32),
        "scl_graph_drop_bwd": ([_vp, _vp, _vp, _i64, _u32, _vp, _vp, _vp, _i64, _u32, _f32, _vp], _i32),
        "scl_graph_final_fwd": ([_vp, _i32, _vp], _i32),
        "scl_graph_final_bwd": ([_vp, _i32, _vp], _i32),
        "scl_graph_reduce": ([_vp, _i32, _vp], _i32),
        # loss.hip
        "scl_supcon_nchunks": ([_i64], _i32),
        "scl_supcon_ws_floats": ([_i32, _i64], _i64),
        "scl_supcon_fwd": ([_vp, _vp, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _vp], _i32),
        "scl_supcon_bwd": ([_vp, _vp, _vp, _f32, _i32, _i64, _i64, _i32, _f32, _vp, _vp, _i32, _vp], _i32),
        "scl_nll_fwd": ([_vp, _vp, _i32, _i32, _vp, _vp, _vp], _i32),
        # optim.hip
        "scl_adamw_flat": ([_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp], _i32),
        # augment.hip
        "scl_fir_nblocks": ([_i32], _i32),
        "scl_fir_multi_f32": ([_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp, _vp], _i32),
        "scl_clip_stats_f32": ([_vp, _i64, _i32, _i32, _vp, _vp], _i32),
        "scl_isd_scatter_f32": ([_vp, _i64, _vp, _vp, _vp, _i32, _i32, _f32, _vp], _i32),
        "scl_clip_affine_f32": ([_i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp], _i32),
        "scl_f32_to_i16_wrap": ([_vp, _vp, _i64, _vp], _i32),
        "scl_i16_sumsq": ([_vp, _i64, _vp, _i32, _vp], _i32),
        "scl_i16_gain_overlay": ([_vp, _i64, _vp, _i64, _f64, _vp, _vp, _vp], _i32),
        "scl_multiview_crop_f32": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _vp], _i32),
        "scl_stream_wait_stream": ([_vp, _vp], _i32),
        "scl_flac_info": ([_vp, _i64, _vp, _vp, _vp, _vp], _i32),
        "scl_flac_decode_i32": ([_vp, _i64, _vp, _i64, _vp, _i32], _i32),
        "scl_flac_decode_mono_f32": ([_vp, _i64, _vp, _i64, _vp, _i32], _i32),
        "scl_gemm_splitk_finish": ([P(SclGemmDesc), _vp, _i32, _i64, _vp], _i32),
        "scl_i16_append_xfade": ([_vp, _i32, _vp, _i32, _i32, _i32, _i32, _f64, _f64, _i32, _i32, _f64, _f64, _i32, _i32, _i32, _vp], _i32),
        "scl_stft_nframes": ([_i32], _i32),
        "scl_stft_f32": ([_vp, _i32, _vp, _i32, _vp], _i32),
        "scl_phase_vocoder_c64": ([_vp, _i32, _f64, _vp, _i32, _vp], _i32),
        "scl_istft_f32": ([_vp, _i32, _vp, _vp, _i32, _vp], _i32),
        "scl_resample_sinc_f32": ([_vp, _i32, _f64, _vp, _i32, _vp], _i32),
        "scl_swish_fwd": ([_vp, _vp, _i64, _vp], _i32),
        "scl_swish_bwd": ([_vp, _vp, _vp, _i64, _vp], _i32),
        "scl_glu_fwd": ([_vp, _vp, _i64, _i32, _vp], _i32),
        "scl_glu_bwd": ([_vp, _vp, _vp, _i64, _i32, _vp], _i32),
        "scl_axpby_f32": ([_vp, _vp, _f32, _f32, _vp, _i64, _vp], _i32),
        "scl_dwconv1d_fwd": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_dwconv1d_wgrad_nslabs": ([_i32, _i32], _i32),
        "scl_dwconv1d_wgrad": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_relpos_gather": ([_vp, _vp, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_relpos_scatter_grad": ([_vp, _vp, _i32, _i32, _i32, _vp], _i32),
        "scl_relpos_softmax_fwd": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        "scl_relpos_softmax_bwd": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp], _i32),
        # btse.hip
        "scl_btse_bio_supported": ([_i32, _i32, _i32, _i32, _i32, _i32, _i32], _i32),
        "scl_btse_bio_ws_floats": ([_i32, _i32], _i64),
        "scl_btse_bio_fwd": ([P(SclBtseBio), _vp], _i32),
        "scl_btse_bio_bwd": ([P(SclBtseBio), _vp], _i32),
        "scl_btse_join_fwd": ([_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp], _i32),
        "scl_btse_join_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp], _i32),
    }


def _bind(lib):
    for name, (argtypes, restype) in _protos().items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


def load():
    """The library through ctypes.CDLL: every call RELEASES the interpreter lock — right for the calls that run long on the host (the FLAC
    decoder, profiling reads that wait for events)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SclError("libscl_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the product path)" % LIB_PATH)
    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


_launch_lib = None


def load_launch():
    """The same library (same dlopen handle underneath, same state) through ctypes.PyDLL: calls KEEP the interpreter lock.  For the
    kernel-launch entry points — microseconds each, ~760 per train step: with CDLL every one of them dropped the lock and had to win it
    back, and beside pack-builder / decoder threads (scl_amd/prefetch.py, main._score_loop) each hand-back could take a whole switch
    interval — 2 builder threads made a PACKS=3 step 37 % slower, 8 decoder threads left the scoring loop at 1.39 x its forward time
    (profiles/r6_pack_builder.txt).  SCL_CTYPES_GIL=release: the round-5 behaviour (A/B)."""
    global _launch_lib
    if _launch_lib is None:
        if os.environ.get("SCL_CTYPES_GIL", "hold") == "release":
            _launch_lib = load()
        else:
            load()      # existence check + error message
            _launch_lib = _bind(ctypes.PyDLL(LIB_PATH))
    return _launch_lib


def all_symbol_names():
    return sorted(_protos())


def check(rc, what=""):
    if rc != 0:
        msg = load().scl_last_error()
        raise SclError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


# ---- csrc/graph.hip --------------------------------------------------------------------------------------------------------------------
_P = ctypes.c_void_p


class SclGraphBn(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("acc", "ticket", "gamma", "beta", "run_mean", "run_var", "nbt", "stats", "bstats", "dgamma", "dbeta")] + \
               [("nvalid", ctypes.c_double), ("eps", ctypes.c_float), ("momentum", ctypes.c_float), ("training", ctypes.c_int32), ("_pad", ctypes.c_int32)]


class SclGraphLayer(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("xd", "S", "g", "y", "Wa", "ba", "Wb", "bb", "min")] + [("min_bs", ctypes.c_int64)] + \
               [(n, _P) for n in ("WM", "bM", "aM", "WaM", "baM", "WbM", "bbM", "am", "gm", "tM", "mout")] + [("bn", SclGraphBn)] + \
               [("N", ctypes.c_int32), ("D", ctypes.c_int32), ("Do", ctypes.c_int32), ("has_master", ctypes.c_int32), ("inv_temp", ctypes.c_float), ("_pad", ctypes.c_int32)] + \
               [(n, _P) for n in ("dz", "d_mout", "d_mout2", "dS", "dxd", "d_min", "slab")] + [("slab_bs", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("o_Wa", "o_ba", "o_Wb", "o_bb", "o_WM", "o_bM", "o_aM", "o_WaM", "o_baM", "o_WbM", "o_bbM", "_pad2")]


class SclGraphPoolUnit(ctypes.Structure):
    _fields_ = [("ysrc", _P), ("src_n", ctypes.c_int32), ("row0", ctypes.c_int32), ("n_in", ctypes.c_int32), ("K", ctypes.c_int32), ("stats", _P),
                ("pw", _P), ("pb", _P), ("pool_seed", ctypes.c_uint32), ("pool_p", ctypes.c_float), ("h", _P), ("sc", _P), ("idx", _P), ("pooled", _P),
                ("Wt", _P), ("bt", _P), ("row_out", ctypes.c_int32), ("_pad", ctypes.c_int32), ("d_res", _P), ("dz", _P),
                ("o_pw", ctypes.c_int32), ("o_pb", ctypes.c_int32), ("o_Wt", ctypes.c_int32), ("o_bt", ctypes.c_int32), ("bn", SclGraphBn)]


class SclGraphPre(ctypes.Structure):
    _fields_ = [("u", SclGraphPoolUnit * 2), ("xd", _P), ("dxd_a", _P), ("dxd_b", _P), ("in_seed", ctypes.c_uint32), ("in_p", ctypes.c_float),
                ("Dp", ctypes.c_int32), ("N", ctypes.c_int32), ("store_common", ctypes.c_int32), ("same_bn", ctypes.c_int32), ("slab", _P), ("slab_bs", ctypes.c_int64)]


class SclGraphFinalBranch(ctypes.Structure):
    _fields_ = [(n, _P) for n in ("y2", "stats2", "Tp", "Sp", "m1", "m2")] + [("way_seed", ctypes.c_uint32 * 3), ("_pad", ctypes.c_int32)] + \
               [(n, _P) for n in ("dz2", "dTp", "dSp", "dm1", "dm2")] + [("bn", SclGraphBn)]


class SclGraphFinal(ctypes.Structure):
    _fields_ = [("br", SclGraphFinalBranch * 2)] + [(n, _P) for n in ("Wout", "bout", "logits", "hidden", "d_logits", "d_hidden", "slab")] + \
               [("slab_bs", ctypes.c_int64), ("o_Wout", ctypes.c_int32), ("o_bout", ctypes.c_int32), ("KT", ctypes.c_int32), ("KS", ctypes.c_int32),
                ("D", ctypes.c_int32), ("NC", ctypes.c_int32), ("way_p", ctypes.c_float), ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32), ("_pad", ctypes.c_int32)]


class SclGraphReduceJob(ctypes.Structure):
    _fields_ = [("src", _P), ("dst", _P), ("stride", ctypes.c_int64), ("n", ctypes.c_int32), ("nparts", ctypes.c_int32)]
