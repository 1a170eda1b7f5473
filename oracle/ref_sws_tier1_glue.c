/*
 * ref_sws_tier1_glue.c — TEST INFRASTRUCTURE.  The reference's libswscale with the inner loops of its
 * generic scaler replaced, per call, by this project's Tier-1 swscale entry points (include/mi355_sws.h):
 * ff_getSwsFunc (libswscale/swscale.c:773, the last step of sws_init_context, utils.c:1336) is interposed
 * by the linker; after the reference has filled the context, c->hyScale / c->hcScale and
 * c->yuv2packed{1,2,X} are pointed at shims that forward to mi355_sws_*.  swscale()'s own line-pull loop,
 * ring buffers and filter banks stay the reference's.  Linked against the emulated build of the product
 * sources, so sws_scale() runs through the product's kernels on a machine without a GPU.
 */
#include <string.h>
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "mi355_sws.h"
#include "mi355dsp.h"

static unsigned long n_calls;
unsigned long ref_sws_tier1_calls(void) { return n_calls; }

static void luts_of(const SwsContext *c, mi355_sws_luts *t)
{
    memcpy(t->y_table, c->yuvTable, 1024);
    for (int i = 0; i < 256; i++) {
        t->rV[i] = c->table_rV[i] - (uint8_t *)c->yuvTable;
        t->gU[i] = c->table_gU[i] - (uint8_t *)c->yuvTable;
        t->gV[i] = c->table_gV[i];
        t->bU[i] = c->table_bU[i] - (uint8_t *)c->yuvTable;
    }
}
static void t1_hscale(SwsContext *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{
    (void)c; n_calls++;
    mi355_sws_hscale8to15(dst, dstW, src, filter, filterPos, filterSize);
}
static void t1_packedX(SwsContext *c, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize, const int16_t *chrFilter,
                       const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize, const int16_t **alpSrc, uint8_t *dest, int dstW, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_X(&t, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, dest, dstW);
}
static void t1_packed2(SwsContext *c, const int16_t *lumSrc[2], const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], const int16_t *alpSrc[2],
                       uint8_t *dest, int dstW, int yalpha, int uvalpha, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_2(&t, lumSrc, chrUSrc, chrVSrc, dest, dstW, yalpha, uvalpha);
}
static void t1_packed1(SwsContext *c, const int16_t *lumSrc, const int16_t *chrUSrc[2], const int16_t *chrVSrc[2], const int16_t *alpSrc,
                       uint8_t *dest, int dstW, int uvalpha, int y)
{
    mi355_sws_luts t;
    (void)alpSrc; (void)y; n_calls++;
    luts_of(c, &t);
    mi355_sws_yuv2rgb24_1(&t, lumSrc, chrUSrc, chrVSrc, dest, dstW, uvalpha);
}

SwsFunc __real_ff_getSwsFunc(SwsContext *c);
SwsFunc __wrap_ff_getSwsFunc(SwsContext *c)
{
    SwsFunc f = __real_ff_getSwsFunc(c);
    if (c->srcFormat == AV_PIX_FMT_YUV420P && c->dstFormat == AV_PIX_FMT_RGB24 && !c->hyscale_fast) {
        mi355_init(0);
        c->hyScale = c->hcScale = t1_hscale;
        c->yuv2packedX = t1_packedX;
        c->yuv2packed2 = t1_packed2;
        c->yuv2packed1 = t1_packed1;
    }
    return f;
}
