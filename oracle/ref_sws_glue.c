/*
 * ref_sws_glue.c — TEST INFRASTRUCTURE.  Compiled against the reference's own headers (from
 * /root/reference, never copied) into oracle/_ref/libswsref.so together with the reference's
 * libswscale objects and with the product's libswscale binding (contrib/libav/mi355_sws_glue.c).
 * (1) hands out that binding's descriptor of a live SwsContext; (2) exposes the
 * reference's static inner loops through the pointers the context holds, so the tests can call
 * them one at a time.
 */
#include "libswscale/swscale.h"
#include "libswscale/swscale_internal.h"
#include "libavutil/log.h"
#include "mi355_sws.h"

/* the reference warns about every context without an accelerated converter ("No accelerated colorspace conversion found"): in bench.py's log those lines
 * pushed everything else out of the tail the driver keeps.  Errors still print. */
__attribute__((constructor)) static void ref_sws_quiet(void) { av_log_set_level(AV_LOG_ERROR); }

/* the descriptor filler is product code: contrib/libav/mi355_sws_glue.c (compiled into this library from there) */
int mi355_sws_describe(struct SwsContext *c, mi355_sws_desc *d);
int ref_sws_describe(struct SwsContext *c, mi355_sws_desc *d) { return mi355_sws_describe(c, d); }

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
