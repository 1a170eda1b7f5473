/*
 * oracle_hevcdsp.c — CPU restatement of the reference's HEVC DSP arithmetic
 * (HEVCDSPContext + the pure predictors of HEVCPredContext), bit depths 8/9/10.
 * TEST INFRASTRUCTURE ONLY (see oracle_h264dsp.c header).
 *
 * Follows libavcodec/hevcdsp_template.c (add_residual :43-82, dequant :84-98,
 * transform_4x4_luma :103-136, idct :140-236 with its col_limit pruning, idct_dc :238-251,
 * SAO band :270-324 / edge :362-718, qpel :729-937, epel :939-1089, (un)weighted
 * prediction :1091-1242, deblocking :1264-1422) and hevcpred_template.c (planar :349-374,
 * dc :378-407, angular :409-516).  Written from the standard's formulas; the inverse DCT
 * matrix is generated from its 31 distinct magnitudes instead of being tabulated.
 * Pinned bit-exact to the reference objects (oracle/_ref) by tests/test_oracle_hevcdsp.py and
 * the golden vectors tests/golden/hevcdsp_ref_sha1.json.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "../include/mi355_abi.h"
#include "oracle.h"

static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int clip_i16(int v) { return clip3(v, -32768, 32767); }
static inline int clip_px(int v, int bd) { return clip3(v, 0, (1 << bd) - 1); }
static inline int ldp(const uint8_t *p, ptrdiff_t i, int bd) { return bd > 8 ? ((const uint16_t *)p)[i] : p[i]; }
static inline void stp(uint8_t *p, ptrdiff_t i, int v, int bd) { if (bd > 8) ((uint16_t *)p)[i] = (uint16_t)v; else p[i] = (uint8_t)v; }
static inline ptrdiff_t pxs(ptrdiff_t stride_bytes, int bd) { return bd > 8 ? stride_bytes / 2 : stride_bytes; }

/* ---- inverse DCT matrix ---------------------------------------------------------------
 * transMatrix[k][n] = c(k) * cos((2n+1) k pi / 64) in the standard's integer approximation:
 * 64 for k = 0 and the magnitudes below for angle index a = (2n+1)k mod 128 (units of pi/64). */
static int trans(int k, int n)
{
    static const int8_t mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                    61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    if (!k) return 64;
    int a = ((2 * n + 1) * k) & 127;
    if (a <= 32) return mag[a];
    if (a <= 64) return -mag[64 - a];
    if (a <= 96) return -mag[a - 64];
    return mag[128 - a];
}

/* a13 */
static void add_residual(uint8_t *dst, int16_t *res, ptrdiff_t stride, int size, int bd)
{
    const ptrdiff_t st = pxs(stride, bd);
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            stp(dst, x + y * st, clip_px(ldp(dst, x + y * st, bd) + res[x + y * size], bd), bd);
}

/* transform-skip scaling, hevcdsp_template.c:84-98 */
static void dequant(int16_t *c, int bd)
{
    const int shift = 13 - bd, off = 1 << (shift - 1);
    for (int i = 0; i < 16; i++) c[i] = (int16_t)((c[i] + off) >> shift);
}

/* 4x4 DST-VII for intra luma, :103-136 */
static void dst4_1d(const int in[4], int out[4])
{
    int c0 = in[0] + in[2], c1 = in[2] + in[3], c2 = in[0] - in[3], c3 = 74 * in[1];
    out[0] = 29 * c0 + 55 * c1 + c3;
    out[1] = 55 * c2 - 29 * c1 + c3;
    out[2] = 74 * (in[0] - in[2] + in[3]);
    out[3] = 55 * c0 + 29 * c2 - c3;
}
static void transform_4x4_luma(int16_t *c, int bd)
{
    int in[4], out[4];
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 4; k++) in[k] = c[i + 4 * k];
        dst4_1d(in, out);
        for (int k = 0; k < 4; k++) c[i + 4 * k] = (int16_t)clip_i16((out[k] + 64) >> 7);
    }
    const int shift = 20 - bd, add = 1 << (shift - 1);
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 4; k++) in[k] = c[4 * i + k];
        dst4_1d(in, out);
        for (int k = 0; k < 4; k++) c[4 * i + k] = (int16_t)clip_i16((out[k] + add) >> shift);
    }
}

/* Which input rows j a 1-D pass of size H looks at when told `end` (the partial butterflies of
 * :140-206 prune odd rows at every level except the innermost 4-point one):
 * H=8: odd j < end; H=16: odd j < end; H=32: odd j < end and j = 2*odd with j/2 < end/2. */
static int row_used(int H, int j, int end)
{
    if (H == 4) return 1;
    if (j & 1) return j < end;
    if (H == 32 && ((j >> 1) & 1)) return (j >> 1) < (end >> 1);
    return 1;
}
static void idct_1d(const int *in, int *out, int H, int end)
{
    const int step = 32 / H;
    for (int n = 0; n < H; n++) {
        int s = 0;
        for (int j = 0; j < H; j++)
            if (row_used(H, j, end)) s += trans(j * step, n) * in[j];
        out[n] = s;
    }
}
static void idct(int16_t *c, int col_limit, int H, int bd)
{
    int in[32], out[32];
    int limit = col_limit < H ? col_limit : H, limit2 = col_limit + 4 < H ? col_limit + 4 : H;
    for (int i = 0; i < H; i++) {
        for (int k = 0; k < H; k++) in[k] = c[i + H * k];
        idct_1d(in, out, H, limit2);
        for (int k = 0; k < H; k++) c[i + H * k] = (int16_t)clip_i16((out[k] + 64) >> 7);
        if (limit2 < H && i % 4 == 0 && i) limit2 -= 4;
    }
    const int shift = 20 - bd, add = 1 << (shift - 1);
    for (int i = 0; i < H; i++) {
        for (int k = 0; k < H; k++) in[k] = c[H * i + k];
        idct_1d(in, out, H, limit);
        for (int k = 0; k < H; k++) c[H * i + k] = (int16_t)clip_i16((out[k] + add) >> shift);
    }
}
static void idct_dc(int16_t *c, int H, int bd)
{
    const int shift = 14 - bd, add = 1 << (shift - 1);
    const int v = (((c[0] + 1) >> 1) + add) >> shift;
    for (int i = 0; i < H * H; i++) c[i] = (int16_t)v;
}

/* ---- SAO (a17) ------------------------------------------------------------------------ */
static void sao_band(uint8_t *dst, uint8_t *src, ptrdiff_t stride, SAOParams *sao, int *borders,
                     int width, int height, int c_idx, int cls, int bd)
{
    const ptrdiff_t st = pxs(stride, bd);
    const int chroma = !!c_idx, cw = (8 >> chroma) + 2, ch = (4 >> chroma) + 2, shift = bd - 5;
    int table[32] = { 0 }, x0 = 0, y0 = 0;
    if (cls & 1) { y0 = -ch; height = ch; } else if (!borders[3]) height -= ch;
    if (cls & 2) { x0 = -cw; width = cw; } else if (!borders[2]) width -= cw;
    for (int k = 0; k < 4; k++) table[(k + sao->band_position[c_idx]) & 31] = sao->offset_val[c_idx][k + 1];
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            ptrdiff_t o = (y0 + y) * st + x0 + x;
            int v = ldp(src, o, bd);
            stp(dst, o, clip_px(v + table[v >> shift], bd), bd);
        }
}

static int sgn3(int a, int b) { return a > b ? 1 : (a == b ? 0 : -1); }
static void sao_edge(uint8_t *dst_, uint8_t *src_, ptrdiff_t stride, SAOParams *sao, int *borders,
                     int width, int height, int c_idx, int vert_edge, int horiz_edge, int diag_edge,
                     int cls, int bd)
{
    static const int8_t pos[4][2][2] = { { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } },
                                         { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } } };
    static const uint8_t edge_idx[5] = { 1, 2, 0, 3, 4 };
    const ptrdiff_t st = pxs(stride, bd);
    const int chroma = !!c_idx, cw = (8 >> chroma) + 2, ch = (4 >> chroma) + 2;
    const int *ov = sao->offset_val[c_idx];
    const int eo = sao->eo_class[c_idx];
    int x0 = 0, y0 = 0, init_x = 0, init_y = 0;
    if (cls & 1) { y0 = -ch; height = ch; } else if (!borders[3]) height -= ch;
    if (cls & 2) { x0 = -cw; width = cw; } else if (!borders[2]) width -= cw;
    /* everything below addresses relative to the region origin */
    uint8_t *dst = dst_ + (bd > 8 ? 2 : 1) * (y0 * st + x0);
    uint8_t *src = src_ + (bd > 8 ? 2 : 1) * (y0 * st + x0);
#define S(x, y) ldp(src, (x) + (y) * st, bd)
#define D(x, y, v) stp(dst, (x) + (y) * st, (v), bd)
    /* picture-border columns/rows get SaoOffsetVal[0]; only the classes that own them look */
    if (!(cls & 2) && eo != 1) {
        if (borders[0]) { for (int y = 0; y < height; y++) D(0, y, clip_px(S(0, y) + ov[0], bd)); init_x = 1; }
        if (borders[2]) { for (int y = 0; y < height; y++) D(width - 1, y, clip_px(S(width - 1, y) + ov[0], bd)); width--; }
    }
    if (!(cls & 1) && eo != 0) {
        if (borders[1]) { for (int x = init_x; x < width; x++) D(x, 0, clip_px(S(x, 0) + ov[0], bd)); init_y = 1; }
        if (borders[3]) { for (int x = init_x; x < width; x++) D(x, height - 1, clip_px(S(x, height - 1) + ov[0], bd)); height--; }
    }
    for (int y = init_y; y < height; y++)
        for (int x = init_x; x < width; x++) {
            int c = S(x, y);
            int d0 = sgn3(c, S(x + pos[eo][0][0], y + pos[eo][0][1]));
            int d1 = sgn3(c, S(x + pos[eo][1][0], y + pos[eo][1][1]));
            D(x, y, clip_px(c + ov[edge_idx[2 + d0 + d1]], bd));
        }
    /* samples that must keep their deblocked value when in-loop filtering may not cross the
     * slice/tile edge: each class owns one corner of the CTB */
    {
        const int ex = (cls & 2) ? width - 1 : 0;      /* column restored by vert_edge */
        const int ey = (cls & 1) ? height - 1 : 0;     /* row restored by horiz_edge */
        const int diag_class = (cls == 0 || cls == 3) ? 2 : 3;   /* 135 degree for classes 0/3, 45 for 1/2 */
        int save;
        if (cls == 0) save = !diag_edge && eo == 2 && !borders[0] && !borders[1];
        else if (cls == 1) save = !diag_edge && eo == 3 && !borders[0];
        else if (cls == 2) save = !diag_edge && eo == 3 && !borders[1];
        else save = !diag_edge && eo == 2;
        if (vert_edge && eo != 1) {
            int ya = init_y + ((cls & 1) ? 0 : save), yb = height - ((cls & 1) ? save : 0);
            for (int y = ya; y < yb; y++) D(ex, y, S(ex, y));
        }
        if (horiz_edge && eo != 0) {
            int xa = init_x + ((cls & 2) ? 0 : save), xb = width - ((cls & 2) ? save : 0);
            for (int x = xa; x < xb; x++) D(x, ey, S(x, ey));
        }
        if (diag_edge && eo == diag_class) D(ex, ey, S(ex, ey));
    }
#undef S
#undef D
}

/* ---- MC (a14) ------------------------------------------------------------------------- */
static const int8_t qpel_c[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                     { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int8_t epel_c[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                     { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

/* taps: 8 (luma, offsets -3..4) or 4 (chroma, offsets -1..2) */
static void mc_generic(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride, int width, int height,
                       int mx, int my, int bd, int taps)
{
    const ptrdiff_t ds = dststride / 2, ss = pxs(srcstride, bd);
    const int before = taps == 8 ? 3 : 1, extra = taps == 8 ? 7 : 3;
    const int8_t *fh = taps == 8 ? qpel_c[mx] : epel_c[mx], *fv = taps == 8 ? qpel_c[my] : epel_c[my];
    if (!mx && !my) {
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) dst[x + y * ds] = (int16_t)(ldp(src, x + y * ss, bd) << (14 - bd));
    } else if (!my) {
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                int s = 0;
                for (int k = 0; k < taps; k++) if (fh[k]) s += fh[k] * ldp(src, x + k - before + y * ss, bd);
                dst[x + y * ds] = (int16_t)(s >> (bd - 8));
            }
    } else if (!mx) {
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                int s = 0;
                for (int k = 0; k < taps; k++) if (fv[k]) s += fv[k] * ldp(src, x + (y + k - before) * ss, bd);
                dst[x + y * ds] = (int16_t)(s >> (bd - 8));
            }
    } else {
        int16_t *tmp = malloc(sizeof(int16_t) * 64 * (64 + 7));
        for (int y = 0; y < height + extra; y++)
            for (int x = 0; x < width; x++) {
                int s = 0;
                for (int k = 0; k < taps; k++) if (fh[k]) s += fh[k] * ldp(src, x + k - before + (y - before) * ss, bd);
                tmp[x + y * 64] = (int16_t)(s >> (bd - 8));
            }
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++) {
                int s = 0;
                for (int k = 0; k < taps; k++) s += fv[k] * tmp[x + (y + k) * 64];
                dst[x + y * ds] = (int16_t)(s >> 6);
            }
        free(tmp);
    }
}

/* ---- (un)weighted prediction (a15) ------------------------------------------------------ */
static void put_unweighted(uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride, int w, int h, int bd)
{
    const ptrdiff_t ds = pxs(dststride, bd), ss = srcstride / 2;
    const int shift = 14 - bd, off = 1 << (shift - 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) stp(dst, x + y * ds, clip_px((src[x + y * ss] + off) >> shift, bd), bd);
}
static void put_unweighted_avg(uint8_t *dst, ptrdiff_t dststride, int16_t *s1, int16_t *s2, ptrdiff_t srcstride, int w, int h, int bd)
{
    const ptrdiff_t ds = pxs(dststride, bd), ss = srcstride / 2;
    const int shift = 15 - bd, off = 1 << (shift - 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) stp(dst, x + y * ds, clip_px((s1[x + y * ss] + s2[x + y * ss] + off) >> shift, bd), bd);
}
static void weighted(int denom, int wx, int ox, uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride, int w, int h, int bd)
{
    const ptrdiff_t ds = pxs(dststride, bd), ss = srcstride / 2;
    const int log2Wd = denom + 14 - bd, off = log2Wd >= 1 ? 1 << (log2Wd - 1) : 0;
    ox *= 1 << (bd - 8);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = log2Wd >= 1 ? ((src[x + y * ss] * wx + off) >> log2Wd) + ox : src[x + y * ss] * wx + ox;
            stp(dst, x + y * ds, clip_px(v, bd), bd);
        }
}
static void weighted_avg(int denom, int w0, int w1, int o0, int o1, uint8_t *dst, ptrdiff_t dststride,
                         int16_t *s1, int16_t *s2, ptrdiff_t srcstride, int w, int h, int bd)
{
    const ptrdiff_t ds = pxs(dststride, bd), ss = srcstride / 2;
    const int log2Wd = denom + 14 - bd;
    o0 *= 1 << (bd - 8); o1 *= 1 << (bd - 8);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            stp(dst, x + y * ds, clip_px((s1[x + y * ss] * w0 + s2[x + y * ss] * w1 + ((o0 + o1 + 1) << log2Wd)) >> (log2Wd + 1), bd), bd);
}

/* ---- deblocking (a16) ------------------------------------------------------------------- */
static void lf_luma(uint8_t *pix, ptrdiff_t xs_b, ptrdiff_t ys_b, int beta, int *tc_, uint8_t *no_p_, uint8_t *no_q_, int bd)
{
    const ptrdiff_t xs = pxs(xs_b, bd), ys = pxs(ys_b, bd);
#define P(k, l) ldp(pix, -(k + 1) * xs + (l) * ys, bd)
#define Q(k, l) ldp(pix, (k) * xs + (l) * ys, bd)
#define SP(k, l, v) stp(pix, -(k + 1) * xs + (l) * ys, (v), bd)
#define SQ(k, l, v) stp(pix, (k) * xs + (l) * ys, (v), bd)
    beta <<= bd - 8;
    for (int j = 0; j < 2; j++) {
        const int l0 = 4 * j, l3 = 4 * j + 3;
        const int dp0 = abs(P(2, l0) - 2 * P(1, l0) + P(0, l0)), dq0 = abs(Q(2, l0) - 2 * Q(1, l0) + Q(0, l0));
        const int dp3 = abs(P(2, l3) - 2 * P(1, l3) + P(0, l3)), dq3 = abs(Q(2, l3) - 2 * Q(1, l3) + Q(0, l3));
        const int d0 = dp0 + dq0, d3 = dp3 + dq3;
        const int tc = tc_[j] << (bd - 8), no_p = no_p_[j], no_q = no_q_[j];
        if (d0 + d3 >= beta) continue;
        const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
        if (abs(P(3, l0) - P(0, l0)) + abs(Q(3, l0) - Q(0, l0)) < beta_3 && abs(P(0, l0) - Q(0, l0)) < tc25 &&
            abs(P(3, l3) - P(0, l3)) + abs(Q(3, l3) - Q(0, l3)) < beta_3 && abs(P(0, l3) - Q(0, l3)) < tc25 &&
            (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
            const int tc2 = tc << 1;
            for (int l = l0; l < l0 + 4; l++) {
                const int p3 = P(3, l), p2 = P(2, l), p1 = P(1, l), p0 = P(0, l);
                const int q0 = Q(0, l), q1 = Q(1, l), q2 = Q(2, l), q3 = Q(3, l);
                if (!no_p) {
                    SP(0, l, p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc2, tc2));
                    SP(1, l, p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
                    SP(2, l, p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc2, tc2));
                }
                if (!no_q) {
                    SQ(0, l, q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc2, tc2));
                    SQ(1, l, q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
                    SQ(2, l, q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc2, tc2));
                }
            }
        } else {
            const int tc_2 = tc >> 1, thr = (beta + (beta >> 1)) >> 3;
            const int nd_p = dp0 + dp3 < thr ? 2 : 1, nd_q = dq0 + dq3 < thr ? 2 : 1;
            for (int l = l0; l < l0 + 4; l++) {
                const int p2 = P(2, l), p1 = P(1, l), p0 = P(0, l), q0 = Q(0, l), q1 = Q(1, l), q2 = Q(2, l);
                int delta0 = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                if (abs(delta0) >= 10 * tc) continue;
                delta0 = clip3(delta0, -tc, tc);
                if (!no_p) SP(0, l, clip_px(p0 + delta0, bd));
                if (!no_q) SQ(0, l, clip_px(q0 - delta0, bd));
                if (!no_p && nd_p > 1) SP(1, l, clip_px(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta0) >> 1, -tc_2, tc_2), bd));
                if (!no_q && nd_q > 1) SQ(1, l, clip_px(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta0) >> 1, -tc_2, tc_2), bd));
            }
        }
    }
}
static void lf_chroma(uint8_t *pix, ptrdiff_t xs_b, ptrdiff_t ys_b, int *tc_, uint8_t *no_p_, uint8_t *no_q_, int bd)
{
    const ptrdiff_t xs = pxs(xs_b, bd), ys = pxs(ys_b, bd);
    for (int j = 0; j < 2; j++) {
        const int tc = tc_[j] << (bd - 8);
        if (tc <= 0) continue;
        for (int l = 4 * j; l < 4 * j + 4; l++) {
            const int p1 = P(1, l), p0 = P(0, l), q0 = Q(0, l), q1 = Q(1, l);
            const int delta0 = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
            if (!no_p_[j]) SP(0, l, clip_px(p0 + delta0, bd));
            if (!no_q_[j]) SQ(0, l, clip_px(q0 - delta0, bd));
        }
    }
#undef P
#undef Q
#undef SP
#undef SQ
}

/* ---- intra prediction (a18) ---------------------------------------------------------------
 * NB: unlike every other table entry these take `stride` in SAMPLES, not bytes: the caller divides
 * the line size before the call (hevcpred_template.c:31) and POS() indexes pixels (:349-374). */
static void pred_planar(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int bd)
{
    const int size = 1 << log2;
    const ptrdiff_t st = stride;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            stp(src, x + y * st, ((size - 1 - x) * ldp(left, y, bd) + (x + 1) * ldp(top, size, bd) +
                                  (size - 1 - y) * ldp(top, x, bd) + (y + 1) * ldp(left, size, bd) + size) >> (log2 + 1), bd);
}
static void pred_dc(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int c_idx, int bd)
{
    const int size = 1 << log2;
    const ptrdiff_t st = stride;
    int dc = size;
    for (int i = 0; i < size; i++) dc += ldp(left, i, bd) + ldp(top, i, bd);
    dc >>= log2 + 1;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) stp(src, x + y * st, dc, bd);
    if (c_idx == 0 && size < 32) {
        stp(src, 0, (ldp(left, 0, bd) + 2 * dc + ldp(top, 0, bd) + 2) >> 2, bd);
        for (int x = 1; x < size; x++) stp(src, x, (ldp(top, x, bd) + 3 * dc + 2) >> 2, bd);
        for (int y = 1; y < size; y++) stp(src, y * st, (ldp(left, y, bd) + 3 * dc + 2) >> 2, bd);
    }
}
static void pred_angular(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx, int mode, int size, int bd)
{
    static const int8_t angle_tab[33] = { 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                          -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int16_t inv_tab[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
    const ptrdiff_t st = stride;
    const int angle = angle_tab[mode - 2], last = (size * angle) >> 5;
    const int vertical = mode >= 18;
    const uint8_t *main_e = vertical ? top : left, *side_e = vertical ? left : top;
    int refbuf[3 * 32 + 1], *ref = refbuf + 32;   /* ref[k] = main[k-1], extended to the left from the side edge */
    for (int k = 0; k <= (angle < 0 ? size : 2 * size); k++) ref[k] = ldp(main_e, k - 1, bd);
    if (angle < 0 && last < -1)
        for (int k = last; k <= -1; k++) ref[k] = ldp(side_e, -1 + ((k * inv_tab[mode - 11] + 128) >> 8), bd);
    for (int a = 0; a < size; a++) {          /* a: index along the prediction direction's minor axis */
        const int idx = ((a + 1) * angle) >> 5, fact = ((a + 1) * angle) & 31;
        for (int b = 0; b < size; b++) {
            int v = fact ? ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
            if (vertical) stp(src, b + a * st, v, bd); else stp(src, a + b * st, v, bd);
        }
    }
    if (c_idx == 0 && size < 32) {
        if (mode == 26) for (int y = 0; y < size; y++) stp(src, y * st, clip_px(ldp(top, 0, bd) + ((ldp(left, y, bd) - ldp(left, -1, bd)) >> 1), bd), bd);
        if (mode == 10) for (int x = 0; x < size; x++) stp(src, x, clip_px(ldp(left, 0, bd) + ((ldp(top, x, bd) - ldp(top, -1, bd)) >> 1), bd), bd);
    }
}

/* ---- table plumbing: one set of entry points per bit depth -------------------------------- */
#define DEPTH_FUNCS(D)                                                                                           \
static void addres4_##D(uint8_t *d, int16_t *r, ptrdiff_t s) { add_residual(d, r, s, 4, D); }                     \
static void addres8_##D(uint8_t *d, int16_t *r, ptrdiff_t s) { add_residual(d, r, s, 8, D); }                     \
static void addres16_##D(uint8_t *d, int16_t *r, ptrdiff_t s) { add_residual(d, r, s, 16, D); }                   \
static void addres32_##D(uint8_t *d, int16_t *r, ptrdiff_t s) { add_residual(d, r, s, 32, D); }                   \
static void dequant_##D(int16_t *c) { dequant(c, D); }                                                            \
static void dst4_##D(int16_t *c) { transform_4x4_luma(c, D); }                                                    \
static void idct4_##D(int16_t *c, int l) { idct(c, l, 4, D); }   static void idct8_##D(int16_t *c, int l) { idct(c, l, 8, D); }     \
static void idct16_##D(int16_t *c, int l) { idct(c, l, 16, D); } static void idct32_##D(int16_t *c, int l) { idct(c, l, 32, D); }   \
static void idctdc4_##D(int16_t *c) { idct_dc(c, 4, D); }   static void idctdc8_##D(int16_t *c) { idct_dc(c, 8, D); }     \
static void idctdc16_##D(int16_t *c) { idct_dc(c, 16, D); } static void idctdc32_##D(int16_t *c) { idct_dc(c, 32, D); }   \
static void saob0_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c) { sao_band(d, s, st, p, b, w, h, c, 0, D); } \
static void saob1_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c) { sao_band(d, s, st, p, b, w, h, c, 1, D); } \
static void saob2_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c) { sao_band(d, s, st, p, b, w, h, c, 2, D); } \
static void saob3_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c) { sao_band(d, s, st, p, b, w, h, c, 3, D); } \
static void saoe0_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c, uint8_t v, uint8_t hz, uint8_t dg) { sao_edge(d, s, st, p, b, w, h, c, v, hz, dg, 0, D); } \
static void saoe1_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c, uint8_t v, uint8_t hz, uint8_t dg) { sao_edge(d, s, st, p, b, w, h, c, v, hz, dg, 1, D); } \
static void saoe2_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c, uint8_t v, uint8_t hz, uint8_t dg) { sao_edge(d, s, st, p, b, w, h, c, v, hz, dg, 2, D); } \
static void saoe3_##D(uint8_t *d, uint8_t *s, ptrdiff_t st, SAOParams *p, int *b, int w, int h, int c, uint8_t v, uint8_t hz, uint8_t dg) { sao_edge(d, s, st, p, b, w, h, c, v, hz, dg, 3, D); } \
static void lfl_h_##D(uint8_t *p, ptrdiff_t s, int be, int *tc, uint8_t *np, uint8_t *nq) { lf_luma(p, s, D > 8 ? 2 : 1, be, tc, np, nq, D); }  \
static void lfl_v_##D(uint8_t *p, ptrdiff_t s, int be, int *tc, uint8_t *np, uint8_t *nq) { lf_luma(p, D > 8 ? 2 : 1, s, be, tc, np, nq, D); }  \
static void lfc_h_##D(uint8_t *p, ptrdiff_t s, int *tc, uint8_t *np, uint8_t *nq) { lf_chroma(p, s, D > 8 ? 2 : 1, tc, np, nq, D); }            \
static void lfc_v_##D(uint8_t *p, ptrdiff_t s, int *tc, uint8_t *np, uint8_t *nq) { lf_chroma(p, D > 8 ? 2 : 1, s, tc, np, nq, D); }            \
static void planar0_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st) { pred_planar(s, t, l, st, 2, D); }   \
static void planar1_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st) { pred_planar(s, t, l, st, 3, D); }   \
static void planar2_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st) { pred_planar(s, t, l, st, 4, D); }   \
static void planar3_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st) { pred_planar(s, t, l, st, 5, D); }   \
static void dc_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int lg, int c) { pred_dc(s, t, l, st, lg, c, D); } \
static void ang0_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int c, int m) { pred_angular(s, t, l, st, c, m, 4, D); }   \
static void ang1_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int c, int m) { pred_angular(s, t, l, st, c, m, 8, D); }   \
static void ang2_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int c, int m) { pred_angular(s, t, l, st, c, m, 16, D); }  \
static void ang3_##D(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int c, int m) { pred_angular(s, t, l, st, c, m, 32, D); }

#define WIDTH_FUNCS(W, D)                                                                                                                                              \
static void qp_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, 0, 0, D, 8); }   \
static void qh_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, mx, 0, D, 8); }  \
static void qv_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, 0, my, D, 8); }  \
static void qhv_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, mx, my, D, 8); } \
static void up_##W##_##D(uint8_t *d, ptrdiff_t ds, int16_t *s, ptrdiff_t ss, int h) { put_unweighted(d, ds, s, ss, W, h, D); }                                          \
static void ua_##W##_##D(uint8_t *d, ptrdiff_t ds, int16_t *a, int16_t *b, ptrdiff_t ss, int h) { put_unweighted_avg(d, ds, a, b, ss, W, h, D); }                        \
static void wp_##W##_##D(uint8_t dn, int16_t w, int16_t o, uint8_t *d, ptrdiff_t ds, int16_t *s, ptrdiff_t ss, int h) { weighted(dn, w, o, d, ds, s, ss, W, h, D); }     \
static void wa_##W##_##D(uint8_t dn, int16_t w0, int16_t w1, int16_t o0, int16_t o1, uint8_t *d, ptrdiff_t ds, int16_t *a, int16_t *b, ptrdiff_t ss, int h) { weighted_avg(dn, w0, w1, o0, o1, d, ds, a, b, ss, W, h, D); }
#define EPEL_FUNCS(W, D)                                                                                                                                               \
static void ep_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, 0, 0, D, 4); }   \
static void eh_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, mx, 0, D, 4); }  \
static void ev_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, 0, my, D, 4); }  \
static void ehv_##W##_##D(int16_t *d, ptrdiff_t ds, uint8_t *s, ptrdiff_t ss, int h, int mx, int my, int16_t *mc) { (void)mc; mc_generic(d, ds, s, ss, W, h, mx, my, D, 4); }

#define ALL_FOR_DEPTH(D) DEPTH_FUNCS(D) \
    WIDTH_FUNCS(2, D) WIDTH_FUNCS(4, D) WIDTH_FUNCS(6, D) WIDTH_FUNCS(8, D) WIDTH_FUNCS(12, D) WIDTH_FUNCS(16, D) \
    WIDTH_FUNCS(24, D) WIDTH_FUNCS(32, D) WIDTH_FUNCS(48, D) WIDTH_FUNCS(64, D) \
    EPEL_FUNCS(2, D) EPEL_FUNCS(4, D) EPEL_FUNCS(6, D) EPEL_FUNCS(8, D) EPEL_FUNCS(12, D) EPEL_FUNCS(16, D) EPEL_FUNCS(24, D) EPEL_FUNCS(32, D)
ALL_FOR_DEPTH(8)
ALL_FOR_DEPTH(9)
ALL_FOR_DEPTH(10)

#define SETW(i, W, D) \
    c->put_hevc_qpel[0][0][i] = qp_##W##_##D; c->put_hevc_qpel[0][1][i] = qh_##W##_##D; \
    c->put_hevc_qpel[1][0][i] = qv_##W##_##D; c->put_hevc_qpel[1][1][i] = qhv_##W##_##D; \
    c->put_unweighted_pred[i] = up_##W##_##D; c->put_unweighted_pred_avg[i] = ua_##W##_##D; \
    c->weighted_pred[i] = wp_##W##_##D; c->weighted_pred_avg[i] = wa_##W##_##D;
#define SETC(i, W, D) \
    c->put_hevc_epel[0][0][i] = ep_##W##_##D; c->put_hevc_epel[0][1][i] = eh_##W##_##D; \
    c->put_hevc_epel[1][0][i] = ev_##W##_##D; c->put_hevc_epel[1][1][i] = ehv_##W##_##D; \
    c->put_unweighted_pred_chroma[i] = up_##W##_##D; c->put_unweighted_pred_avg_chroma[i] = ua_##W##_##D; \
    c->weighted_pred_chroma[i] = wp_##W##_##D; c->weighted_pred_avg_chroma[i] = wa_##W##_##D;
#define SET_DEPTH(D) \
    c->put_pcm = put_pcm_##D; \
    c->add_residual[0] = addres4_##D; c->add_residual[1] = addres8_##D; c->add_residual[2] = addres16_##D; c->add_residual[3] = addres32_##D; \
    c->dequant = dequant_##D; c->transform_4x4_luma = dst4_##D; \
    c->idct[0] = idct4_##D; c->idct[1] = idct8_##D; c->idct[2] = idct16_##D; c->idct[3] = idct32_##D; \
    c->idct_dc[0] = idctdc4_##D; c->idct_dc[1] = idctdc8_##D; c->idct_dc[2] = idctdc16_##D; c->idct_dc[3] = idctdc32_##D; \
    c->sao_band_filter[0] = saob0_##D; c->sao_band_filter[1] = saob1_##D; c->sao_band_filter[2] = saob2_##D; c->sao_band_filter[3] = saob3_##D; \
    c->sao_edge_filter[0] = saoe0_##D; c->sao_edge_filter[1] = saoe1_##D; c->sao_edge_filter[2] = saoe2_##D; c->sao_edge_filter[3] = saoe3_##D; \
    SETW(0, 4, D) SETW(1, 8, D) SETW(2, 12, D) SETW(3, 16, D) SETW(4, 24, D) SETW(5, 32, D) SETW(6, 48, D) SETW(7, 64, D) \
    SETC(0, 2, D) SETC(1, 4, D) SETC(2, 6, D) SETC(3, 8, D) SETC(4, 12, D) SETC(5, 16, D) SETC(6, 24, D) SETC(7, 32, D) \
    c->hevc_h_loop_filter_luma = c->hevc_h_loop_filter_luma_c = lfl_h_##D; c->hevc_v_loop_filter_luma = c->hevc_v_loop_filter_luma_c = lfl_v_##D; \
    c->hevc_h_loop_filter_chroma = c->hevc_h_loop_filter_chroma_c = lfc_h_##D; c->hevc_v_loop_filter_chroma = c->hevc_v_loop_filter_chroma_c = lfc_v_##D;

/* put_pcm hevcdsp_template.c:28-41 over get_bits() get_bits.h:228-237 (big-endian reader, safe variant:
 * index = FFMIN(size_in_bits_plus8, index + n)) */
static void put_pcm(uint8_t *dst, ptrdiff_t stride, int size, GetBitContext *gb, int pcm_bit_depth, int bd)
{
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) {
            const uint8_t *p = gb->buffer + ((unsigned)gb->index >> 3);
            const uint32_t cache = (((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]) << (gb->index & 7);
            const unsigned v = cache >> (32 - pcm_bit_depth);
            const unsigned next = (unsigned)gb->index + (unsigned)pcm_bit_depth;
            gb->index = (int)(next < (unsigned)gb->size_in_bits_plus8 ? next : (unsigned)gb->size_in_bits_plus8);
            if (bd > 8) ((uint16_t *)(dst + y * stride))[x] = (uint16_t)(v << (bd - pcm_bit_depth));
            else dst[y * stride + x] = (uint8_t)(v << (bd - pcm_bit_depth));
        }
}
static void put_pcm_8(uint8_t *d, ptrdiff_t s, int n, GetBitContext *gb, int pb) { put_pcm(d, s, n, gb, pb, 8); }
static void put_pcm_9(uint8_t *d, ptrdiff_t s, int n, GetBitContext *gb, int pb) { put_pcm(d, s, n, gb, pb, 9); }
static void put_pcm_10(uint8_t *d, ptrdiff_t s, int n, GetBitContext *gb, int pb) { put_pcm(d, s, n, gb, pb, 10); }
void oracle_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)
{
    switch (bit_depth) {
    case 9:  SET_DEPTH(9)  break;
    case 10: SET_DEPTH(10) break;
    default: SET_DEPTH(8)  break;
    }
}
#define SET_PRED(D) \
    h->pred_planar[0] = planar0_##D; h->pred_planar[1] = planar1_##D; h->pred_planar[2] = planar2_##D; h->pred_planar[3] = planar3_##D; \
    h->pred_dc = dc_##D; h->pred_angular[0] = ang0_##D; h->pred_angular[1] = ang1_##D; h->pred_angular[2] = ang2_##D; h->pred_angular[3] = ang3_##D;
/* intra_pred[] (needs HEVCContext) is not restated */
void oracle_hevc_pred_init(HEVCPredContext *h, int bit_depth)
{
    switch (bit_depth) {
    case 9:  SET_PRED(9)  break;
    case 10: SET_PRED(10) break;
    default: SET_PRED(8)  break;
    }
}
