/*
 * mi355_sws_glue.c — libswscale binding: PRODUCT glue that lives beside the reference's libswscale.
 *
 * Compiled against the reference's own (private) swscale_internal.h; nothing in the reference tree changes.
 *
 *   mi355_sws_describe()   fills the plain descriptor of include/mi355_sws.h from a live SwsContext (filter banks and
 *                          yuv->rgb tables exactly as the reference built them): what mi355_sws_create() takes for the
 *                          whole-picture and batched entry points.
 *   ff_sws_init_mi355x()   the arch-hook form (cf. ff_sws_init_swscale_x86, swscale.c:773-790): after the reference has
 *                          chosen its C inner loops, c->hyScale / c->hcScale (hScale8To15_c, swscale.c:133) and
 *                          c->yuv2packed1 / 2 / X (yuv2rgb24_{1,2,X}_c, output.c:937-1110) are pointed at shims that
 *                          forward each call to the device (mi355_sws_hscale8to15, mi355_sws_yuv2rgb24_*).  swscale()'s
 *                          line-pull loop, ring buffers and filter banks stay the reference's.
 *   __wrap_ff_getSwsFunc   the same without a patch: link with -Wl,--wrap=ff_getSwsFunc (ff_getSwsFunc is the last step
 *                          of sws_init_context, utils.c:1336).
 * Only yuv420p -> rgb24 without hyscale_fast is taken; every other conversion keeps the reference's functions.
 */
#include <string.h>
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "mi355_sws.h"
#include "mi355dsp.h"

/* -DMI355_SWS_DESCRIBE_ONLY: only mi355_sws_describe() (a build that wants the descriptor of a context without binding the
 * library: the plain reference library the tests compare against) */
#ifndef MI355_SWS_DESCRIBE_ONLY
static unsigned long n_calls;
unsigned long mi355_sws_glue_calls(void) { return n_calls; }      /* inner-loop calls forwarded so far (diagnostics) */

#endif

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

int mi355_sws_describe(struct SwsContext *c, mi355_sws_desc *d)
{
    if (c->srcFormat != AV_PIX_FMT_YUV420P || c->dstFormat != AV_PIX_FMT_RGB24)
        return -1;
    d->srcW = c->srcW; d->srcH = c->srcH; d->dstW = c->dstW; d->dstH = c->dstH;
    d->chrSrcW = c->chrSrcW; d->chrSrcH = c->chrSrcH; d->chrDstW = c->chrDstW;
    d->unscaled_special = c->swscale != ff_getSwsFunc(c);       /* yuv2rgb_c_24_rgb was selected (yuv2rgb.c:570) */
    d->hLum = (mi355_sws_filter){ c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize, c->dstW };
    d->hChr = (mi355_sws_filter){ c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize, c->chrDstW };
    d->vLum = (mi355_sws_filter){ c->vLumFilter, c->vLumFilterPos, c->vLumFilterSize, c->dstH };
    d->vChr = (mi355_sws_filter){ c->vChrFilter, c->vChrFilterPos, c->vChrFilterSize, c->dstH };
    luts_of(c, &d->luts);
    return 0;
}

#ifndef MI355_SWS_DESCRIBE_ONLY
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

void ff_sws_init_mi355x(SwsContext *c)
{
    if (c->srcFormat != AV_PIX_FMT_YUV420P || c->dstFormat != AV_PIX_FMT_RGB24 || c->hyscale_fast) return;
    if (mi355_init(0) != 0) return;                             /* no usable MI355X: the reference's functions stay */
    c->hyScale = c->hcScale = t1_hscale;
    c->yuv2packedX = t1_packedX;
    c->yuv2packed2 = t1_packed2;
    c->yuv2packed1 = t1_packed1;
}

SwsFunc __real_ff_getSwsFunc(SwsContext *c);
SwsFunc __wrap_ff_getSwsFunc(SwsContext *c)
{
    SwsFunc f = __real_ff_getSwsFunc(c);
    ff_sws_init_mi355x(c);
    return f;
}
#endif
