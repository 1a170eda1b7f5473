/*
 * ref_sws_glue.c — TEST INFRASTRUCTURE.  Compiled against the reference's own headers (from
 * /root/reference, never copied) into oracle/_ref/libswsref.so together with the reference's
 * libswscale objects.  (1) fills the public descriptor of include/mi355_sws.h from a live
 * SwsContext — the same few lines INTEGRATION.md shows for the in-tree binding; (2) exposes the
 * reference's static inner loops through the pointers the context holds, so the tests can call
 * them one at a time.
 */
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "mi355_sws.h"

int ref_sws_describe(struct SwsContext *c, mi355_sws_desc *d)
{
    if (c->srcFormat != AV_PIX_FMT_YUV420P || c->dstFormat != AV_PIX_FMT_RGB24)
        return -1;
    d->srcW = c->srcW; d->srcH = c->srcH; d->dstW = c->dstW; d->dstH = c->dstH;
    d->chrSrcW = c->chrSrcW; d->chrSrcH = c->chrSrcH; d->chrDstW = c->chrDstW;
    d->unscaled_special = c->swscale != ff_getSwsFunc(c);
    d->hLum = (mi355_sws_filter){ c->hLumFilter, c->hLumFilterPos, c->hLumFilterSize, c->dstW };
    d->hChr = (mi355_sws_filter){ c->hChrFilter, c->hChrFilterPos, c->hChrFilterSize, c->chrDstW };
    d->vLum = (mi355_sws_filter){ c->vLumFilter, c->vLumFilterPos, c->vLumFilterSize, c->dstH };
    d->vChr = (mi355_sws_filter){ c->vChrFilter, c->vChrFilterPos, c->vChrFilterSize, c->dstH };
    memcpy(d->luts.y_table, c->yuvTable, 1024);
    for (int i = 0; i < 256; i++) {
        d->luts.rV[i] = c->table_rV[i] - (uint8_t *)c->yuvTable;
        d->luts.gU[i] = c->table_gU[i] - (uint8_t *)c->yuvTable;
        d->luts.gV[i] = c->table_gV[i];
        d->luts.bU[i] = c->table_bU[i] - (uint8_t *)c->yuvTable;
    }
    return 0;
}

int ref_sws_flags_word(int bicubic, int accurate_rnd, int bitexact)
{
    return (bicubic ? SWS_BICUBIC : SWS_BILINEAR) | (accurate_rnd ? SWS_ACCURATE_RND : 0) | (bitexact ? SWS_BITEXACT : 0);
}
int ref_pix_fmt(int which) { return which == 0 ? AV_PIX_FMT_YUV420P : which == 1 ? AV_PIX_FMT_RGB24 : AV_PIX_FMT_YUV444P; }

void ref_sws_hscale(struct SwsContext *c, int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter,
                    const int32_t *filterPos, int filterSize)
{
    c->hyScale(c, dst, dstW, src, filter, filterPos, filterSize);
}
void ref_sws_planeX(struct SwsContext *c, const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW,
                    const uint8_t *dither, int offset)
{
    c->yuv2planeX(filter, filterSize, src, dest, dstW, dither, offset);
}
void ref_sws_plane1(struct SwsContext *c, const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    c->yuv2plane1(src, dest, dstW, dither, offset);
}
void ref_sws_packedX(struct SwsContext *c, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                     const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                     uint8_t *dest, int dstW)
{
    c->yuv2packedX(c, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, NULL, dest, dstW, 0);
}
void ref_sws_packed2(struct SwsContext *c, const int16_t *buf[2], const int16_t *ubuf[2], const int16_t *vbuf[2],
                     uint8_t *dest, int dstW, int yalpha, int uvalpha)
{
    c->yuv2packed2(c, buf, ubuf, vbuf, NULL, dest, dstW, yalpha, uvalpha, 0);
}
void ref_sws_packed1(struct SwsContext *c, const int16_t *buf0, const int16_t *ubuf[2], const int16_t *vbuf[2],
                     uint8_t *dest, int dstW, int uvalpha)
{
    c->yuv2packed1(c, buf0, ubuf, vbuf, NULL, dest, dstW, uvalpha, 0);
}
