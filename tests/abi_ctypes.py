"""ctypes mirror of include/mi355_abi.h (the reference's DSP pointer tables).

Used by every parity test to call through the C ABI exactly as the reference's
decoder would: fill a context with an init function, then call the pointers.
"""
import ctypes as C

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i16p = C.POINTER(C.c_int16)
intp = C.POINTER(C.c_int)
ptrdiff = C.c_ssize_t
F = C.CFUNCTYPE

weight_fn = F(None, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
biweight_fn = F(None, u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
lf_fn = F(None, u8p, C.c_int, C.c_int, C.c_int, i8p)
lfi_fn = F(None, u8p, C.c_int, C.c_int, C.c_int)
idct_fn = F(None, u8p, i16p, C.c_int)
idctn_fn = F(None, u8p, intp, i16p, C.c_int, u8p)
idct8n_fn = F(None, C.POINTER(u8p), intp, i16p, C.c_int, u8p)


class H264DSPContext(C.Structure):
    _fields_ = [
        ("weight_h264_pixels_tab", weight_fn * 4),
        ("biweight_h264_pixels_tab", biweight_fn * 4),
        ("h264_v_loop_filter_luma", lf_fn),
        ("h264_h_loop_filter_luma", lf_fn),
        ("h264_h_loop_filter_luma_mbaff", lf_fn),
        ("h264_v_loop_filter_luma_intra", lfi_fn),
        ("h264_h_loop_filter_luma_intra", lfi_fn),
        ("h264_h_loop_filter_luma_mbaff_intra", lfi_fn),
        ("h264_v_loop_filter_chroma", lf_fn),
        ("h264_h_loop_filter_chroma", lf_fn),
        ("h264_h_loop_filter_chroma_mbaff", lf_fn),
        ("h264_v_loop_filter_chroma_intra", lfi_fn),
        ("h264_h_loop_filter_chroma_intra", lfi_fn),
        ("h264_h_loop_filter_chroma_mbaff_intra", lfi_fn),
        ("h264_loop_filter_strength", C.c_void_p),
        ("h264_idct_add", idct_fn),
        ("h264_idct8_add", idct_fn),
        ("h264_idct_dc_add", idct_fn),
        ("h264_idct8_dc_add", idct_fn),
        ("h264_idct_add16", idctn_fn),
        ("h264_idct8_add4", idctn_fn),
        ("h264_idct_add8", idct8n_fn),
        ("h264_idct_add16intra", idctn_fn),
        ("h264_luma_dc_dequant_idct", F(None, i16p, i16p, C.c_int)),
        ("h264_chroma_dc_dequant_idct", F(None, i16p, C.c_int)),
        ("h264_add_pixels8_clear", idct_fn),
        ("h264_add_pixels4_clear", idct_fn),
        ("startcode_find_candidate", F(C.c_int, u8p, C.c_int)),
    ]


qpel_fn = F(None, u8p, u8p, ptrdiff)


class H264QpelContext(C.Structure):
    _fields_ = [("put_h264_qpel_pixels_tab", (qpel_fn * 16) * 4),
                ("avg_h264_qpel_pixels_tab", (qpel_fn * 16) * 4)]


chroma_fn = F(None, u8p, u8p, ptrdiff, C.c_int, C.c_int, C.c_int)


class H264ChromaContext(C.Structure):
    _fields_ = [("put_h264_chroma_pixels_tab", chroma_fn * 3),
                ("avg_h264_chroma_pixels_tab", chroma_fn * 3)]


class H264PredContext(C.Structure):
    _fields_ = [
        ("pred4x4", F(None, u8p, u8p, ptrdiff) * 15),
        ("pred8x8l", F(None, u8p, C.c_int, C.c_int, ptrdiff) * 12),
        ("pred8x8", F(None, u8p, ptrdiff) * 11),
        ("pred16x16", F(None, u8p, ptrdiff) * 9),
        ("pred4x4_add", F(None, u8p, i16p, ptrdiff) * 2),
        ("pred8x8l_add", F(None, u8p, i16p, ptrdiff) * 2),
        ("pred8x8l_filter_add", F(None, u8p, i16p, C.c_int, C.c_int, ptrdiff) * 2),
        ("pred8x8_add", F(None, u8p, intp, i16p, ptrdiff) * 3),
        ("pred16x16_add", F(None, u8p, intp, i16p, ptrdiff) * 3),
    ]


class VideoDSPContext(C.Structure):
    _fields_ = [
        ("emulated_edge_mc", F(None, u8p, u8p, ptrdiff, ptrdiff, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)),
        ("prefetch", F(None, u8p, ptrdiff, C.c_int)),
    ]


class GetBitContext(C.Structure):   # libavcodec/get_bits.h:54-61 (CONFIG_SAFE_BITSTREAM_READER layout)
    _fields_ = [("buffer", C.c_void_p), ("buffer_end", C.c_void_p), ("index", C.c_int), ("size_in_bits", C.c_int),
                ("size_in_bits_plus8", C.c_int)]


class SAOParams(C.Structure):
    _fields_ = [("offset_abs", (C.c_int * 4) * 3), ("offset_sign", (C.c_int * 4) * 3),
                ("band_position", C.c_int * 3), ("eo_class", C.c_int * 3),
                ("offset_val", (C.c_int * 5) * 3), ("type_idx", C.c_uint8 * 3)]


hevc_mc_fn = F(None, i16p, ptrdiff, u8p, ptrdiff, C.c_int, C.c_int, C.c_int, i16p)
hevc_put_fn = F(None, u8p, ptrdiff, i16p, ptrdiff, C.c_int)
hevc_avg_fn = F(None, u8p, ptrdiff, i16p, i16p, ptrdiff, C.c_int)
hevc_w_fn = F(None, C.c_uint8, C.c_int16, C.c_int16, u8p, ptrdiff, i16p, ptrdiff, C.c_int)
hevc_wavg_fn = F(None, C.c_uint8, C.c_int16, C.c_int16, C.c_int16, C.c_int16, u8p, ptrdiff, i16p, i16p, ptrdiff, C.c_int)
hevc_lfl_fn = F(None, u8p, ptrdiff, C.c_int, intp, u8p, u8p)
hevc_lfc_fn = F(None, u8p, ptrdiff, intp, u8p, u8p)
hevc_pcm_fn = F(None, u8p, ptrdiff, C.c_int, C.POINTER(GetBitContext), C.c_int)


class HEVCDSPContext(C.Structure):
    _fields_ = [
        ("put_pcm", hevc_pcm_fn),
        ("add_residual", F(None, u8p, i16p, ptrdiff) * 4),
        ("dequant", F(None, i16p)),
        ("transform_4x4_luma", F(None, i16p)),
        ("idct", F(None, i16p, C.c_int) * 4),
        ("idct_dc", F(None, i16p) * 4),
        ("sao_band_filter", F(None, u8p, u8p, ptrdiff, C.POINTER(SAOParams), intp, C.c_int, C.c_int, C.c_int) * 4),
        ("sao_edge_filter", F(None, u8p, u8p, ptrdiff, C.POINTER(SAOParams), intp, C.c_int, C.c_int, C.c_int,
                              C.c_uint8, C.c_uint8, C.c_uint8) * 4),
        ("put_hevc_qpel", ((hevc_mc_fn * 8) * 2) * 2),
        ("put_hevc_epel", ((hevc_mc_fn * 8) * 2) * 2),
        ("put_unweighted_pred", hevc_put_fn * 8),
        ("put_unweighted_pred_chroma", hevc_put_fn * 8),
        ("put_unweighted_pred_avg", hevc_avg_fn * 8),
        ("put_unweighted_pred_avg_chroma", hevc_avg_fn * 8),
        ("weighted_pred", hevc_w_fn * 8),
        ("weighted_pred_chroma", hevc_w_fn * 8),
        ("weighted_pred_avg", hevc_wavg_fn * 8),
        ("weighted_pred_avg_chroma", hevc_wavg_fn * 8),
        ("hevc_h_loop_filter_luma", hevc_lfl_fn),
        ("hevc_v_loop_filter_luma", hevc_lfl_fn),
        ("hevc_h_loop_filter_chroma", hevc_lfc_fn),
        ("hevc_v_loop_filter_chroma", hevc_lfc_fn),
        ("hevc_h_loop_filter_luma_c", hevc_lfl_fn),
        ("hevc_v_loop_filter_luma_c", hevc_lfl_fn),
        ("hevc_h_loop_filter_chroma_c", hevc_lfc_fn),
        ("hevc_v_loop_filter_chroma_c", hevc_lfc_fn),
    ]


class HEVCPredContext(C.Structure):
    _fields_ = [
        ("intra_pred", C.c_void_p * 4),
        ("pred_planar", F(None, u8p, u8p, u8p, ptrdiff) * 4),
        ("pred_dc", F(None, u8p, u8p, u8p, ptrdiff, C.c_int, C.c_int)),
        ("pred_angular", F(None, u8p, u8p, u8p, ptrdiff, C.c_int, C.c_int) * 4),
    ]

AV_CODEC_ID_H264 = 27
