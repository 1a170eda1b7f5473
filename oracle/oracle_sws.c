/*
 * oracle_sws.c — CPU restatement of libswscale's yuv420p -> {rgb24, planar 8 bit} inner loops and
 * of the generic line-pull scaler around them.
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench tools as the
 * checker / CPU baseline; never linked into libmi355dsp.so.
 *
 * Pinned by tests/test_oracle_sws.py against the reference's own libswscale objects
 * (oracle/_ref/libswsref.so, built in place by oracle/Makefile) function by function and for
 * whole pictures, and against golden vectors those objects produced (tests/golden/sws_ref_sha1.json).
 * Filter banks are inputs here, as they are for the reference's inner loops (built by initFilter,
 * libswscale/utils.c:249-632, which stays on the host side of the boundary).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/mi355_sws.h"

static inline int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

/* a19: hScale8To15_c, swscale.c:133-147 */
void oracle_sws_hscale8to15(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{
    for (int i = 0; i < dstW; i++) {
        int val = 0;
        for (int j = 0; j < filterSize; j++) val += (int)src[filterPos[i] + j] * filter[filterSize * i + j];
        val >>= 7;
        dst[i] = (int16_t)(val < 32767 ? val : 32767);
    }
}

/* a20: yuv2planeX_8_c output.c:242-255, yuv2plane1_8_c :257-266 */
void oracle_sws_yuv2planeX_8(const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW,
                             const uint8_t *dither, int offset)
{
    for (int i = 0; i < dstW; i++) {
        int val = dither[(i + offset) & 7] << 12;
        for (int j = 0; j < filterSize; j++) val += src[j][i] * filter[j];
        dest[i] = (uint8_t)clip_u8(val >> 19);
    }
}
void oracle_sws_yuv2plane1_8(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    for (int i = 0; i < dstW; i++) dest[i] = (uint8_t)clip_u8((src[i] + dither[(i + offset) & 7]) >> 7);
}

/* yuv2rgb_write, rgb24 branch output.c:853-866 with the LUT pointers as offsets (mi355_sws.h) */
static inline void write_pair(const mi355_sws_luts *t, uint8_t *dest, int i, int Y1, int Y2, int U, int V)
{
    const int r = t->rV[V], g = t->gU[U] + t->gV[V], b = t->bU[U];
    dest[i * 6 + 0] = t->y_table[r + Y1]; dest[i * 6 + 1] = t->y_table[g + Y1]; dest[i * 6 + 2] = t->y_table[b + Y1];
    dest[i * 6 + 3] = t->y_table[r + Y2]; dest[i * 6 + 4] = t->y_table[g + Y2]; dest[i * 6 + 5] = t->y_table[b + Y2];
}

/* yuv2rgb_X_c_template output.c:937-996: values are clipped only when one of them has bit 8 set */
void oracle_sws_yuv2rgb24_X(const mi355_sws_luts *t, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                            const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                            uint8_t *dest, int dstW)
{
    for (int i = 0; i < ((dstW + 1) >> 1); i++) {
        int Y1 = 1 << 18, Y2 = 1 << 18, U = 1 << 18, V = 1 << 18;
        for (int j = 0; j < lumFilterSize; j++) { Y1 += lumSrc[j][i * 2] * lumFilter[j]; Y2 += lumSrc[j][i * 2 + 1] * lumFilter[j]; }
        for (int j = 0; j < chrFilterSize; j++) { U += chrUSrc[j][i] * chrFilter[j]; V += chrVSrc[j][i] * chrFilter[j]; }
        Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
        if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
        write_pair(t, dest, i, Y1, Y2, U, V);
    }
}
/* yuv2rgb_2_c_template output.c:998-1041 */
void oracle_sws_yuv2rgb24_2(const mi355_sws_luts *t, const int16_t *buf[2], const int16_t *ubuf[2], const int16_t *vbuf[2],
                            uint8_t *dest, int dstW, int yalpha, int uvalpha)
{
    const int yalpha1 = 4096 - yalpha, uvalpha1 = 4096 - uvalpha;
    for (int i = 0; i < ((dstW + 1) >> 1); i++) {
        const int Y1 = clip_u8((buf[0][i * 2] * yalpha1 + buf[1][i * 2] * yalpha) >> 19);
        const int Y2 = clip_u8((buf[0][i * 2 + 1] * yalpha1 + buf[1][i * 2 + 1] * yalpha) >> 19);
        const int U = clip_u8((ubuf[0][i] * uvalpha1 + ubuf[1][i] * uvalpha) >> 19);
        const int V = clip_u8((vbuf[0][i] * uvalpha1 + vbuf[1][i] * uvalpha) >> 19);
        write_pair(t, dest, i, Y1, Y2, U, V);
    }
}
/* yuv2rgb_1_c_template output.c:1043-1110 */
void oracle_sws_yuv2rgb24_1(const mi355_sws_luts *t, const int16_t *buf0, const int16_t *ubuf[2], const int16_t *vbuf[2],
                            uint8_t *dest, int dstW, int uvalpha)
{
    for (int i = 0; i < ((dstW + 1) >> 1); i++) {
        const int Y1 = clip_u8(buf0[i * 2] >> 7), Y2 = clip_u8(buf0[i * 2 + 1] >> 7);
        int U, V;
        if (uvalpha < 2048) { U = clip_u8(ubuf[0][i] >> 7); V = clip_u8(vbuf[0][i] >> 7); }
        else { U = clip_u8((ubuf[0][i] + ubuf[1][i]) >> 8); V = clip_u8((vbuf[0][i] + vbuf[1][i]) >> 8); }
        write_pair(t, dest, i, Y1, Y2, U, V);
    }
}

/* a22: yuv2rgb_c_24_rgb yuv2rgb.c:335-363 through YUV2RGBFUNC :129-160 / ENDYUV2RGBLINE :162-171:
 * two lines at a time, 8 + 4 + 2 sample groups == every even-aligned pair below dstW; the chroma
 * sample of pair i is used for both lines (nearest, no interpolation). */
int oracle_sws_yuv2rgb_c_24_rgb(const mi355_sws_luts *t, int dstW, const uint8_t *const src[3], const int srcStride[3],
                                int srcSliceY, int srcSliceH, uint8_t *dst, int dstStride)
{
    for (int y = 0; y < srcSliceH; y += 2) {
        uint8_t *d1 = dst + (ptrdiff_t)(y + srcSliceY) * dstStride, *d2 = d1 + dstStride;
        const uint8_t *py1 = src[0] + (ptrdiff_t)y * srcStride[0], *py2 = py1 + srcStride[0];
        const uint8_t *pu = src[1] + (ptrdiff_t)(y >> 1) * srcStride[1], *pv = src[2] + (ptrdiff_t)(y >> 1) * srcStride[2];
        for (int i = 0; i < (dstW >> 1); i++) {
            write_pair(t, d1, i, py1[2 * i], py1[2 * i + 1], pu[i], pv[i]);
            write_pair(t, d2, i, py2[2 * i], py2[2 * i + 1], pu[i], pv[i]);
        }
    }
    return srcSliceH;
}

/* a21: ff_yuv2rgb_c_init_tables yuv2rgb.c:671-896, the 24-bpp case :850-863 with fill_table :632-643
 * and fill_gv_table :645-655.  inv_table: one row of ff_yuv2rgb_coeffs (:49-58). */
void oracle_sws_init_luts(mi355_sws_luts *t, const int inv_table[4], int fullRange, int brightness, int contrast, int saturation)
{
    int64_t crv = inv_table[0], cbu = inv_table[1], cgu = -inv_table[2], cgv = -inv_table[3];
    int64_t cy = 1 << 16, oy = 0, yb;
    const int yoffs = fullRange ? 384 : 326;
    if (!fullRange) { cy = (cy * 255) / 219; oy = 16 << 16; }
    else { crv = (crv * 224) / 255; cbu = (cbu * 224) / 255; cgu = (cgu * 224) / 255; cgv = (cgv * 224) / 255; }
    cy = (cy * contrast) >> 16;
    crv = (crv * contrast * saturation) >> 32; cbu = (cbu * contrast * saturation) >> 32;
    cgu = (cgu * contrast * saturation) >> 32; cgv = (cgv * contrast * saturation) >> 32;
    oy -= 256 * brightness;
    crv = ((crv << 16) + 0x8000) / cy; cbu = ((cbu << 16) + 0x8000) / cy;
    cgu = ((cgu << 16) + 0x8000) / cy; cgv = ((cgv << 16) + 0x8000) / cy;
    yb = -(384 << 16) - oy;
    for (int i = 0; i < 1024; i++) { t->y_table[i] = (uint8_t)clip_u8((int)((yb + 0x8000) >> 16)); yb += cy; }
    const int64_t inc[3] = { crv, cgu, cbu };
    int16_t *tab[3] = { t->rV, t->gU, t->bU };
    for (int k = 0; k < 3; k++) {
        int64_t cb = 0;
        const int base = yoffs - (int)(inc[k] >> 9);
        for (int i = 0; i < 256; i++) { tab[k][i] = (int16_t)(base + (int)(cb >> 16)); cb += inc[k]; }
    }
    {
        int64_t cb = 0;
        const int off = -(int)(cgv >> 9);
        for (int i = 0; i < 256; i++) { t->gV[i] = (int16_t)(off + (int)(cb >> 16)); cb += cgv; }
    }
}

/* The generic scaler for a whole picture (srcSliceY = 0, srcSliceH = srcH), swscale.c:343-722:
 * every source line goes through the horizontal filter once (:497-538); output line y takes
 * vLumFilterSize lines from firstLumSrcY = max(1 - size, vLumFilterPos[y]) with lines outside
 * the picture replaced by the first / last one (:571-616); packed output picks the _1 / _2 / _X
 * template by the filter sizes (:658-682). */
int oracle_sws_scale(const mi355_sws_desc *d, const uint8_t *const src[3], const int srcStride[3], uint8_t *dst, int dstStride)
{
    if (d->unscaled_special) return oracle_sws_yuv2rgb_c_24_rgb(&d->luts, d->dstW, src, srcStride, 0, d->srcH, dst, dstStride);
    const int dstW = d->dstW, cw = d->chrDstW;
    int16_t *lum = malloc(sizeof(int16_t) * (size_t)d->srcH * dstW);
    int16_t *cu = malloc(sizeof(int16_t) * (size_t)d->chrSrcH * cw), *cv = malloc(sizeof(int16_t) * (size_t)d->chrSrcH * cw);
    for (int y = 0; y < d->srcH; y++)
        oracle_sws_hscale8to15(lum + (size_t)y * dstW, dstW, src[0] + (ptrdiff_t)y * srcStride[0], d->hLum.coef, d->hLum.pos, d->hLum.size);
    for (int y = 0; y < d->chrSrcH; y++) {
        oracle_sws_hscale8to15(cu + (size_t)y * cw, cw, src[1] + (ptrdiff_t)y * srcStride[1], d->hChr.coef, d->hChr.pos, d->hChr.size);
        oracle_sws_hscale8to15(cv + (size_t)y * cw, cw, src[2] + (ptrdiff_t)y * srcStride[2], d->hChr.coef, d->hChr.pos, d->hChr.size);
    }
    const int ls = d->vLum.size, cs = d->vChr.size;
    const int16_t **lp = malloc(sizeof(*lp) * (ls + 1)), **up = malloc(sizeof(*up) * (cs + 1)), **vp = malloc(sizeof(*vp) * (cs + 1));
    for (int y = 0; y < d->dstH; y++) {
        const int fl = d->vLum.pos[y] > 1 - ls ? d->vLum.pos[y] : 1 - ls;
        const int fc = d->vChr.pos[y] > 1 - cs ? d->vChr.pos[y] : 1 - cs;
        for (int j = 0; j < ls; j++) {
            int l = fl + j; l = l < 0 ? 0 : l > d->srcH - 1 ? d->srcH - 1 : l;
            lp[j] = lum + (size_t)l * dstW;
        }
        for (int j = 0; j < cs; j++) {
            int l = fc + j; l = l < 0 ? 0 : l > d->chrSrcH - 1 ? d->chrSrcH - 1 : l;
            up[j] = cu + (size_t)l * cw; vp[j] = cv + (size_t)l * cw;
        }
        up[cs] = up[cs - 1]; vp[cs] = vp[cs - 1];   /* never read: ubuf[1] of the _1 template when cs == 1 */
        uint8_t *dest = dst + (ptrdiff_t)y * dstStride;
        if (ls == 1 && cs <= 2)
            oracle_sws_yuv2rgb24_1(&d->luts, lp[0], up, vp, dest, dstW, cs == 1 ? 0 : d->vChr.coef[2 * y + 1]);
        else if (ls == 2 && cs == 2)
            oracle_sws_yuv2rgb24_2(&d->luts, lp, up, vp, dest, dstW, d->vLum.coef[2 * y + 1], d->vChr.coef[2 * y + 1]);
        else
            oracle_sws_yuv2rgb24_X(&d->luts, d->vLum.coef + (size_t)y * ls, lp, ls, d->vChr.coef + (size_t)y * cs, up, vp, cs, dest, dstW);
    }
    free(lp); free(up); free(vp); free(lum); free(cu); free(cv);
    return d->dstH;
}
