/*
 * ref_glue.c — the few symbols the reference's DSP objects expect from files we
 * do not compile (TEST INFRASTRUCTURE ONLY; links into oracle/_ref/libref.so).
 *  - ff_hevc_qpel_extra*: three 4-entry tables that live in libavcodec/hevcdec.c:45-47
 *    (rows of context the 8-tap HEVC luma filter needs before/after/total for a
 *    fractional position: 0 for integer, 3/4/7 otherwise).
 *  - ref_layout(): sizeof/offsetof of the ABI structs as the REFERENCE headers
 *    define them, for tests/test_abi_layout.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264qpel.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/h264pred.h"
#include "libavcodec/videodsp.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcdec.h"

const uint8_t ff_hevc_qpel_extra_before[4] = { 0, 3, 3, 3 };
const uint8_t ff_hevc_qpel_extra_after[4]  = { 0, 4, 4, 4 };
const uint8_t ff_hevc_qpel_extra[4]        = { 0, 7, 7, 7 };

#define F(S, f) n += snprintf(buf + n, (size_t)(cap - n), #S "." #f "=%zu\n", offsetof(S, f))
#define Z(S)    n += snprintf(buf + n, (size_t)(cap - n), #S "=%zu\n", sizeof(S))
int ref_layout(char *buf, int cap)
{
    int n = 0;
    Z(H264DSPContext); F(H264DSPContext, biweight_h264_pixels_tab); F(H264DSPContext, h264_v_loop_filter_luma);
    F(H264DSPContext, h264_loop_filter_strength); F(H264DSPContext, h264_idct_add); F(H264DSPContext, h264_idct_add16);
    F(H264DSPContext, h264_luma_dc_dequant_idct); F(H264DSPContext, h264_add_pixels8_clear); F(H264DSPContext, startcode_find_candidate);
    Z(H264QpelContext); F(H264QpelContext, avg_h264_qpel_pixels_tab);
    Z(H264ChromaContext); F(H264ChromaContext, avg_h264_chroma_pixels_tab);
    Z(H264PredContext); F(H264PredContext, pred8x8l); F(H264PredContext, pred8x8); F(H264PredContext, pred16x16);
    F(H264PredContext, pred4x4_add); F(H264PredContext, pred16x16_add);
    Z(VideoDSPContext); F(VideoDSPContext, prefetch);
    Z(SAOParams); F(SAOParams, band_position); F(SAOParams, eo_class); F(SAOParams, offset_val); F(SAOParams, type_idx);
    Z(HEVCDSPContext); F(HEVCDSPContext, add_residual); F(HEVCDSPContext, idct); F(HEVCDSPContext, idct_dc);
    F(HEVCDSPContext, sao_band_filter); F(HEVCDSPContext, put_hevc_qpel); F(HEVCDSPContext, put_hevc_epel);
    F(HEVCDSPContext, put_unweighted_pred); F(HEVCDSPContext, weighted_pred); F(HEVCDSPContext, weighted_pred_avg_chroma);
    F(HEVCDSPContext, hevc_h_loop_filter_luma); F(HEVCDSPContext, hevc_v_loop_filter_chroma_c);
    Z(HEVCPredContext); F(HEVCPredContext, pred_planar); F(HEVCPredContext, pred_dc); F(HEVCPredContext, pred_angular);
    n += snprintf(buf + n, (size_t)(cap - n), "AV_CODEC_ID_H264=%d\n", (int)AV_CODEC_ID_H264);
    return n;
}
