/*
 * mi355_sws.h — C ABI of the libswscale part of the hot path:
 * horizontal 8->15 bit FIR, vertical FIR to planar 8 bit or through the yuv->rgb LUTs to RGB24,
 * and the unscaled yuv420p -> rgb24 converter.
 *
 * The reference keeps these behind function pointers of the (private) SwsContext
 * (libswscale/swscale_internal.h:253-540): hyScale/hcScale :526-531, yuv2plane1/yuv2planeX/
 * yuv2packed1/2/X :437-443, swscale :263.  SwsContext is not a public type, so this ABI carries
 * the handful of fields those functions consume in a plain descriptor; the few lines of glue that
 * fill it from a SwsContext inside the reference tree are in INTEGRATION.md (and compiled for the
 * tests as oracle/ref_sws_glue.c).  Filter banks and LUTs stay the product of the reference's own
 * init code (initFilter utils.c:249-632, ff_yuv2rgb_c_init_tables yuv2rgb.c:671-896).
 */
#ifndef MI355_SWS_H
#define MI355_SWS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* yuv->rgb tables of the 24-bpp case (yuv2rgb.c:850-863) as offsets instead of pointers:
 *   r = y_table[rV[V] + Y], g = y_table[gU[U] + gV[V] + Y], b = y_table[bU[U] + Y]
 * with rV[v] = c->table_rV[v] - c->yuvTable, gU[u] = c->table_gU[u] - c->yuvTable,
 * gV[v] = c->table_gV[v], bU[u] = c->table_bU[u] - c->yuvTable. */
typedef struct mi355_sws_luts {
    uint8_t y_table[1024];
    int16_t rV[256], gU[256], gV[256], bU[256];
} mi355_sws_luts;

/* one filter bank as built by initFilter: n outputs, `size` taps each */
typedef struct mi355_sws_filter {
    const int16_t *coef;   /* [n * size]; horizontal: 1.0 = 1<<14, vertical: 1.0 = 1<<12 */
    const int32_t *pos;    /* [n] first input sample / line */
    int size, n;
} mi355_sws_filter;

typedef struct mi355_sws_desc {
    int srcW, srcH, dstW, dstH;
    int chrSrcW, chrSrcH, chrDstW;          /* SwsContext :267-269 (yuv420p in, rgb24 out: chrDstH = dstH) */
    int unscaled_special;                   /* c->swscale is yuv2rgb_c_24_rgb (swscale_unscaled.c:1050-1055) */
    mi355_sws_filter hLum, hChr, vLum, vChr;
    mi355_sws_luts luts;
} mi355_sws_desc;

typedef struct mi355_sws_ctx mi355_sws_ctx;   /* descriptor + filter banks resident in HBM */

/* one picture of a batch, device pointers.  A source plane whose pointer and stride are multiples of 16 is fetched in aligned 16-byte
 * pieces: every line must be readable over its whole STRIDE (src_stride[k] bytes, also the last line's — the bytes between the width
 * and the stride may hold anything; the reference's own buffers are allocated that way, libavutil/frame.c).  Planes with other
 * pointers or strides are read sample-exactly. */
typedef struct mi355_sws_frame {
    const uint8_t *src[3];
    int src_stride[3];
    uint8_t *dst;          /* packed RGB24 */
    int dst_stride;
} mi355_sws_frame;

mi355_sws_ctx *mi355_sws_create(const mi355_sws_desc *desc);
void mi355_sws_destroy(mi355_sws_ctx *ctx);

/* Tier 1: replaces c->swscale(c, src, srcStride, 0, srcH, dst, dstStride) for a whole picture
 * (SwsFunc, swscale_internal.h:62-64; generic swscale() swscale.c:343-722 or yuv2rgb_c_24_rgb
 * yuv2rgb.c:335-363 according to desc->unscaled_special).  Host pointers, synchronous.
 * Returns the number of output lines. */
int mi355_sws_scale(mi355_sws_ctx *ctx, const uint8_t *const src[3], const int src_stride[3],
                    uint8_t *dst, int dst_stride);

/* Tier 2: a batch of pictures resident in HBM, one launch; d_frames is a device array. */
int mi355_sws_scale_frames_dev(mi355_sws_ctx *ctx, const mi355_sws_frame *d_frames, int nframes, void *stream);   /* 0, -1 bad argument, -2 launch failure */

/* ---- the individual inner loops (Tier 1, host pointers), argument lists of the reference's
 * function-pointer types minus the SwsContext ------------------------------------------------ */
/* hScale8To15_c swscale.c:133-147 (c->hyScale / c->hcScale) */
void mi355_sws_hscale8to15(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter,
                           const int32_t *filterPos, int filterSize);
/* yuv2planeX_8_c output.c:242-255, yuv2plane1_8_c :257-266 */
void mi355_sws_yuv2planeX_8(const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW,
                            const uint8_t *dither, int offset);
void mi355_sws_yuv2plane1_8(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset);
/* yuv2rgb24_X_c / _2_c / _1_c  output.c:937-1110 with target AV_PIX_FMT_RGB24, no alpha */
void mi355_sws_yuv2rgb24_X(const mi355_sws_luts *luts, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                           const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                           uint8_t *dest, int dstW);
void mi355_sws_yuv2rgb24_2(const mi355_sws_luts *luts, const int16_t *buf[2], const int16_t *ubuf[2], const int16_t *vbuf[2],
                           uint8_t *dest, int dstW, int yalpha, int uvalpha);
void mi355_sws_yuv2rgb24_1(const mi355_sws_luts *luts, const int16_t *buf0, const int16_t *ubuf[2], const int16_t *vbuf[2],
                           uint8_t *dest, int dstW, int uvalpha);
/* yuv2rgb_c_24_rgb yuv2rgb.c:335-363 (SwsFunc slice interface; returns srcSliceH) */
int mi355_sws_yuv2rgb_c_24_rgb(const mi355_sws_luts *luts, int dstW, const uint8_t *const src[3], const int srcStride[3],
                               int srcSliceY, int srcSliceH, uint8_t *dst, int dstStride);

#ifdef __cplusplus
}
#endif
#endif
