/*
 * mi355_abi.h — the reference's DSP function-pointer tables, restated for ABI
 * compatibility (field ORDER and function SIGNATURES only; no code).
 *
 * These are the drop-in boundary of SURVEY.md §8(b).  Each struct below must be
 * layout-identical to the reference struct of the same name, because the
 * reference's own decoder calls through them after our `ff_*_init_mi355x()`
 * hook has overwritten the pointers:
 *
 *   H264DSPContext     libavcodec/h264dsp.h:41-117
 *   H264QpelContext    libavcodec/h264qpel.h:27-30   (qpel_mc_func: qpeldsp.h:65)
 *   H264ChromaContext  libavcodec/h264chroma.h:27-32
 *   H264PredContext    libavcodec/h264pred.h:89-112
 *   VideoDSPContext    libavcodec/videodsp.h:31-64
 *   HEVCDSPContext     libavcodec/hevcdsp.h:41-114   (SAOParams: hevcdsp.h:27-39)
 *   HEVCPredContext    libavcodec/hevcdec.h:399-409
 *
 * Every block is wrapped in the reference header's own include guard, so a
 * translation unit that already included the reference header (the `--wrap`
 * glue of INTEGRATION.md) uses the reference's definition and this file adds
 * nothing.  tests/test_abi_layout.py checks sizeof/offsetof of every field
 * against a probe compiled from the reference headers (ref_layout() in oracle/ref_glue.c, golden copy tests/golden/abi_layout_ref.json).
 */
#ifndef MI355_ABI_H
#define MI355_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- H264DSPContext (libavcodec/h264dsp.h:31-117) ------------------------ */
#ifndef AVCODEC_H264DSP_H
typedef void (*h264_weight_func)(uint8_t *block, int stride, int height,
                                 int log2_denom, int weight, int offset);
typedef void (*h264_biweight_func)(uint8_t *dst, uint8_t *src, int stride,
                                   int height, int log2_denom, int weightd,
                                   int weights, int offset);
typedef struct H264DSPContext {
    h264_weight_func   weight_h264_pixels_tab[4];
    h264_biweight_func biweight_h264_pixels_tab[4];

    void (*h264_v_loop_filter_luma)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_h_loop_filter_luma)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_h_loop_filter_luma_mbaff)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_v_loop_filter_luma_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_h_loop_filter_luma_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_h_loop_filter_luma_mbaff_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_v_loop_filter_chroma)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_h_loop_filter_chroma)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_h_loop_filter_chroma_mbaff)(uint8_t *pix, int stride, int alpha, int beta, int8_t *tc0);
    void (*h264_v_loop_filter_chroma_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_h_loop_filter_chroma_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_h_loop_filter_chroma_mbaff_intra)(uint8_t *pix, int stride, int alpha, int beta);
    void (*h264_loop_filter_strength)(int16_t bS[2][4][4], uint8_t nnz[40],
                                      int8_t ref[2][40], int16_t mv[2][40][2],
                                      int bidir, int edges, int step,
                                      int mask_mv0, int mask_mv1, int field);

    void (*h264_idct_add)(uint8_t *dst, int16_t *block, int stride);
    void (*h264_idct8_add)(uint8_t *dst, int16_t *block, int stride);
    void (*h264_idct_dc_add)(uint8_t *dst, int16_t *block, int stride);
    void (*h264_idct8_dc_add)(uint8_t *dst, int16_t *block, int stride);
    void (*h264_idct_add16)(uint8_t *dst, const int *blockoffset, int16_t *block,
                            int stride, const uint8_t nnzc[15 * 8]);
    void (*h264_idct8_add4)(uint8_t *dst, const int *blockoffset, int16_t *block,
                            int stride, const uint8_t nnzc[15 * 8]);
    void (*h264_idct_add8)(uint8_t **dst, const int *blockoffset, int16_t *block,
                           int stride, const uint8_t nnzc[15 * 8]);
    void (*h264_idct_add16intra)(uint8_t *dst, const int *blockoffset, int16_t *block,
                                 int stride, const uint8_t nnzc[15 * 8]);
    void (*h264_luma_dc_dequant_idct)(int16_t *output, int16_t *input, int qmul);
    void (*h264_chroma_dc_dequant_idct)(int16_t *block, int qmul);

    void (*h264_add_pixels8_clear)(uint8_t *dst, int16_t *block, int stride);
    void (*h264_add_pixels4_clear)(uint8_t *dst, int16_t *block, int stride);

    int (*startcode_find_candidate)(const uint8_t *buf, int size);
} H264DSPContext;
#endif

/* ---- H264QpelContext (libavcodec/h264qpel.h:27-30) ------------------------ */
#ifndef AVCODEC_QPELDSP_H
typedef void (*qpel_mc_func)(uint8_t *dst, const uint8_t *src, ptrdiff_t stride);
#endif
#ifndef AVCODEC_H264QPEL_H
typedef struct H264QpelContext {
    qpel_mc_func put_h264_qpel_pixels_tab[4][16];
    qpel_mc_func avg_h264_qpel_pixels_tab[4][16];
} H264QpelContext;
#endif

/* ---- H264ChromaContext (libavcodec/h264chroma.h:27-32) -------------------- */
#ifndef AVCODEC_H264CHROMA_H
typedef void (*h264_chroma_mc_func)(uint8_t *dst, uint8_t *src, ptrdiff_t srcStride,
                                    int h, int x, int y);
typedef struct H264ChromaContext {
    h264_chroma_mc_func put_h264_chroma_pixels_tab[3];
    h264_chroma_mc_func avg_h264_chroma_pixels_tab[3];
} H264ChromaContext;
#endif

/* ---- H264PredContext (libavcodec/h264pred.h:34-112) ----------------------- */
#ifndef AVCODEC_H264PRED_H
/* prediction-mode indices are part of the interface (table slots) */
#define VERT_PRED              0
#define HOR_PRED               1
#define DC_PRED                2
#define DIAG_DOWN_LEFT_PRED    3
#define DIAG_DOWN_RIGHT_PRED   4
#define VERT_RIGHT_PRED        5
#define HOR_DOWN_PRED          6
#define VERT_LEFT_PRED         7
#define HOR_UP_PRED            8
#define LEFT_DC_PRED           9
#define TOP_DC_PRED           10
#define DC_128_PRED           11

#define DC_PRED8x8             0
#define HOR_PRED8x8            1
#define VERT_PRED8x8           2
#define PLANE_PRED8x8          3
#define LEFT_DC_PRED8x8        4
#define TOP_DC_PRED8x8         5
#define DC_128_PRED8x8         6
#define ALZHEIMER_DC_L0T_PRED8x8  7
#define ALZHEIMER_DC_0LT_PRED8x8  8
#define ALZHEIMER_DC_L00_PRED8x8  9
#define ALZHEIMER_DC_0L0_PRED8x8 10

typedef struct H264PredContext {
    void (*pred4x4[9 + 3 + 3])(uint8_t *src, const uint8_t *topright, ptrdiff_t stride);
    void (*pred8x8l[9 + 3])(uint8_t *src, int topleft, int topright, ptrdiff_t stride);
    void (*pred8x8[4 + 3 + 4])(uint8_t *src, ptrdiff_t stride);
    void (*pred16x16[4 + 3 + 2])(uint8_t *src, ptrdiff_t stride);

    void (*pred4x4_add[2])(uint8_t *pix, int16_t *block, ptrdiff_t stride);
    void (*pred8x8l_add[2])(uint8_t *pix, int16_t *block, ptrdiff_t stride);
    void (*pred8x8l_filter_add[2])(uint8_t *pix, int16_t *block, int topleft,
                                   int topright, ptrdiff_t stride);
    void (*pred8x8_add[3])(uint8_t *pix, const int *block_offset, int16_t *block,
                           ptrdiff_t stride);
    void (*pred16x16_add[3])(uint8_t *pix, const int *block_offset, int16_t *block,
                             ptrdiff_t stride);
} H264PredContext;
#endif

/* ---- VideoDSPContext (libavcodec/videodsp.h:31-64) ------------------------ */
#ifndef AVCODEC_VIDEODSP_H
typedef struct VideoDSPContext {
    void (*emulated_edge_mc)(uint8_t *buf, const uint8_t *src,
                             ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                             int block_w, int block_h, int src_x, int src_y,
                             int w, int h);
    void (*prefetch)(uint8_t *buf, ptrdiff_t stride, int h);
} VideoDSPContext;
#endif

/* ---- HEVCDSPContext (libavcodec/hevcdsp.h:27-114) ------------------------- */
#ifndef AVCODEC_HEVCDSP_H
/* GetBitContext (libavcodec/get_bits.h:54-61) as the reference's default configuration builds it
 * (CONFIG_SAFE_BITSTREAM_READER: the fifth field exists and get_bits() clamps the position to it) */
#ifndef AVCODEC_GET_BITS_H
typedef struct GetBitContext {
    const uint8_t *buffer, *buffer_end;
    int index;
    int size_in_bits;
    int size_in_bits_plus8;
} GetBitContext;
#endif
typedef struct SAOParams {
    int offset_abs[3][4];
    int offset_sign[3][4];
    int band_position[3];
    int eo_class[3];
    int offset_val[3][5];
    uint8_t type_idx[3];
} SAOParams;

typedef struct HEVCDSPContext {
    void (*put_pcm)(uint8_t *dst, ptrdiff_t stride, int size,
                    GetBitContext *gb, int pcm_bit_depth);
    void (*add_residual[4])(uint8_t *dst, int16_t *res, ptrdiff_t stride);
    void (*dequant)(int16_t *coeffs);
    void (*transform_4x4_luma)(int16_t *coeffs);
    void (*idct[4])(int16_t *coeffs, int col_limit);
    void (*idct_dc[4])(int16_t *coeffs);
    void (*sao_band_filter[4])(uint8_t *dst, uint8_t *src, ptrdiff_t stride,
                               struct SAOParams *sao, int *borders,
                               int width, int height, int c_idx);
    void (*sao_edge_filter[4])(uint8_t *dst, uint8_t *src, ptrdiff_t stride,
                               struct SAOParams *sao, int *borders, int width,
                               int height, int c_idx, uint8_t vert_edge,
                               uint8_t horiz_edge, uint8_t diag_edge);
    void (*put_hevc_qpel[2][2][8])(int16_t *dst, ptrdiff_t dststride, uint8_t *src,
                                   ptrdiff_t srcstride, int height,
                                   int mx, int my, int16_t *mcbuffer);
    void (*put_hevc_epel[2][2][8])(int16_t *dst, ptrdiff_t dststride, uint8_t *src,
                                   ptrdiff_t srcstride, int height,
                                   int mx, int my, int16_t *mcbuffer);
    void (*put_unweighted_pred[8])(uint8_t *dst, ptrdiff_t dststride, int16_t *src,
                                   ptrdiff_t srcstride, int height);
    void (*put_unweighted_pred_chroma[8])(uint8_t *dst, ptrdiff_t dststride, int16_t *src,
                                          ptrdiff_t srcstride, int height);
    void (*put_unweighted_pred_avg[8])(uint8_t *dst, ptrdiff_t dststride,
                                       int16_t *src1, int16_t *src2,
                                       ptrdiff_t srcstride, int height);
    void (*put_unweighted_pred_avg_chroma[8])(uint8_t *dst, ptrdiff_t dststride,
                                              int16_t *src1, int16_t *src2,
                                              ptrdiff_t srcstride, int height);
    void (*weighted_pred[8])(uint8_t denom, int16_t wlxFlag, int16_t olxFlag,
                             uint8_t *dst, ptrdiff_t dststride, int16_t *src,
                             ptrdiff_t srcstride, int height);
    void (*weighted_pred_chroma[8])(uint8_t denom, int16_t wlxFlag, int16_t olxFlag,
                                    uint8_t *dst, ptrdiff_t dststride, int16_t *src,
                                    ptrdiff_t srcstride, int height);
    void (*weighted_pred_avg[8])(uint8_t denom, int16_t wl0Flag, int16_t wl1Flag,
                                 int16_t ol0Flag, int16_t ol1Flag, uint8_t *dst,
                                 ptrdiff_t dststride, int16_t *src1, int16_t *src2,
                                 ptrdiff_t srcstride, int height);
    void (*weighted_pred_avg_chroma[8])(uint8_t denom, int16_t wl0Flag, int16_t wl1Flag,
                                        int16_t ol0Flag, int16_t ol1Flag, uint8_t *dst,
                                        ptrdiff_t dststride, int16_t *src1, int16_t *src2,
                                        ptrdiff_t srcstride, int height);
    void (*hevc_h_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc,
                                    uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_luma)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc,
                                    uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int *tc,
                                      uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_chroma)(uint8_t *pix, ptrdiff_t stride, int *tc,
                                      uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_luma_c)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc,
                                      uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_luma_c)(uint8_t *pix, ptrdiff_t stride, int beta, int *tc,
                                      uint8_t *no_p, uint8_t *no_q);
    void (*hevc_h_loop_filter_chroma_c)(uint8_t *pix, ptrdiff_t stride, int *tc,
                                        uint8_t *no_p, uint8_t *no_q);
    void (*hevc_v_loop_filter_chroma_c)(uint8_t *pix, ptrdiff_t stride, int *tc,
                                        uint8_t *no_p, uint8_t *no_q);
} HEVCDSPContext;
#endif

/* ---- HEVCPredContext (libavcodec/hevcdec.h:399-409) ----------------------- */
#ifndef AVCODEC_HEVCDEC_H
struct HEVCContext;
typedef struct HEVCPredContext {
    void (*intra_pred[4])(struct HEVCContext *s, int x0, int y0, int c_idx);
    void (*pred_planar[4])(uint8_t *src, const uint8_t *top, const uint8_t *left,
                           ptrdiff_t stride);
    void (*pred_dc)(uint8_t *src, const uint8_t *top, const uint8_t *left,
                    ptrdiff_t stride, int log2_size, int c_idx);
    void (*pred_angular[4])(uint8_t *src, const uint8_t *top, const uint8_t *left,
                            ptrdiff_t stride, int c_idx, int mode);
} HEVCPredContext;
#endif

/* AV_CODEC_ID_H264 as passed to ff_h264_pred_init (libavcodec/avcodec.h enum AVCodecID);
 * our hook only specialises the H.264 flavour of the table. */
#define MI355_AV_CODEC_ID_H264 27

#ifdef __cplusplus
}
#endif
#endif /* MI355_ABI_H */
