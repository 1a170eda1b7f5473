/*
 * oracle_h264dsp.c — CPU restatement of the reference's H.264 DSP arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libav_amd/, the C-ABI
 * library) may call, link or import this file; only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Scalar, one-sample-at-a-time loops written from the formulas recorded in
 * SURVEY.md §8(a); each function cites the reference lines it must agree with.
 * 8-bit samples, int16 coefficients (the BIT_DEPTH 8 instantiation).
 * Pinned bit-exact against the reference's own C objects (oracle/_ref, built
 * from /root/reference by oracle/Makefile) in tests/test_oracle_h264dsp.py and
 * by the committed golden vectors under tests/golden/.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "../include/mi355_abi.h"
#include "oracle.h"

static inline int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int iabs(int v) { return v < 0 ? -v : v; }

/* scan8[] of libavcodec/h264dec.h:631-645 in closed form: luma/chroma block i
 * of plane p sits at column 4+x, row 1+y+5p of the 8-wide nnz cache. */
int oracle_scan8(int i)
{
    int p = i >> 4, b = i & 15;
    int x = (b & 1) + 2 * ((b >> 2) & 1);
    int y = ((b >> 1) & 1) + 2 * (b >> 3);
    return 4 + x + 8 * (1 + y + 5 * p);
}

/* ------------------------------------------------------------------------- */
/* a1: inverse transforms — libavcodec/h264idct_template.c:33-171             */
/* ------------------------------------------------------------------------- */

/* 4x4: h264idct_template.c:33-67.  The coefficient block is stored transposed
 * (first pass runs down i, second pass writes column i of dst), intermediates
 * live in the int16 block itself (so they wrap to 16 bits), block is cleared. */
static void idct4_add(uint8_t *dst, int16_t *blk, int stride)
{
    int16_t t[16];
    blk[0] = (int16_t)(blk[0] + 32);
    for (int i = 0; i < 4; i++) {
        int a = blk[i], b = blk[i + 4], c = blk[i + 8], d = blk[i + 12];
        int e0 = a + c, e1 = a - c, e2 = (b >> 1) - d, e3 = b + (d >> 1);
        t[i]      = (int16_t)(e0 + e3);
        t[i + 4]  = (int16_t)(e1 + e2);
        t[i + 8]  = (int16_t)(e1 - e2);
        t[i + 12] = (int16_t)(e0 - e3);
    }
    for (int i = 0; i < 4; i++) {
        int a = t[4 * i], b = t[4 * i + 1], c = t[4 * i + 2], d = t[4 * i + 3];
        int e0 = a + c, e1 = a - c, e2 = (b >> 1) - d, e3 = b + (d >> 1);
        int r[4] = { e0 + e3, e1 + e2, e1 - e2, e0 - e3 };
        for (int k = 0; k < 4; k++)
            dst[i + k * stride] = (uint8_t)clip_u8(dst[i + k * stride] + (r[k] >> 6));
    }
    memset(blk, 0, 16 * sizeof(*blk));
}

/* one 8-point pass of h264idct_template.c:80-108 / :111-136 */
static void idct8_1d(const int in[8], int out[8])
{
    int a0 = in[0] + in[4], a2 = in[0] - in[4];
    int a4 = (in[2] >> 1) - in[6], a6 = (in[6] >> 1) + in[2];
    int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int a1 = -in[3] + in[5] - in[7] - (in[7] >> 1);
    int a3 =  in[1] + in[7] - in[3] - (in[3] >> 1);
    int a5 = -in[1] + in[7] + in[5] + (in[5] >> 1);
    int a7 =  in[3] + in[5] + in[1] + (in[1] >> 1);
    int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2);
    int b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7;
    out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3;
    out[3] = b6 + b1; out[4] = b6 - b1;
}

/* 8x8: h264idct_template.c:69-141 */
static void idct8_add(uint8_t *dst, int16_t *blk, int stride)
{
    int16_t t[64];
    int in[8], out[8];
    blk[0] = (int16_t)(blk[0] + 32);
    for (int i = 0; i < 8; i++) {
        for (int k = 0; k < 8; k++) in[k] = blk[i + 8 * k];
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++) t[i + 8 * k] = (int16_t)out[k];
    }
    for (int i = 0; i < 8; i++) {
        for (int k = 0; k < 8; k++) in[k] = t[8 * i + k];
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++)
            dst[i + k * stride] = (uint8_t)clip_u8(dst[i + k * stride] + (out[k] >> 6));
    }
    memset(blk, 0, 64 * sizeof(*blk));
}

/* dc-only: h264idct_template.c:144-171 (only block[0] is cleared) */
static void idct_dc_add_n(uint8_t *dst, int16_t *blk, int stride, int n)
{
    int dc = (blk[0] + 32) >> 6;
    blk[0] = 0;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[x + y * stride] = (uint8_t)clip_u8(dst[x + y * stride] + dc);
}
static void idct4_dc_add(uint8_t *dst, int16_t *blk, int stride) { idct_dc_add_n(dst, blk, stride, 4); }
static void idct8_dc_add(uint8_t *dst, int16_t *blk, int stride) { idct_dc_add_n(dst, blk, stride, 8); }

/* a2: per-MB dispatchers — h264idct_template.c:174-239 */
static void idct_add16(uint8_t *dst, const int *off, int16_t *blk, int stride, const uint8_t nnzc[15 * 8])
{
    for (int i = 0; i < 16; i++) {
        int nnz = nnzc[oracle_scan8(i)];
        if (!nnz) continue;
        if (nnz == 1 && blk[i * 16]) idct4_dc_add(dst + off[i], blk + i * 16, stride);
        else                         idct4_add(dst + off[i], blk + i * 16, stride);
    }
}
static void idct_add16intra(uint8_t *dst, const int *off, int16_t *blk, int stride, const uint8_t nnzc[15 * 8])
{
    for (int i = 0; i < 16; i++) {
        if (nnzc[oracle_scan8(i)]) idct4_add(dst + off[i], blk + i * 16, stride);
        else if (blk[i * 16])      idct4_dc_add(dst + off[i], blk + i * 16, stride);
    }
}
static void idct8_add4(uint8_t *dst, const int *off, int16_t *blk, int stride, const uint8_t nnzc[15 * 8])
{
    for (int i = 0; i < 16; i += 4) {
        int nnz = nnzc[oracle_scan8(i)];
        if (!nnz) continue;
        if (nnz == 1 && blk[i * 16]) idct8_dc_add(dst + off[i], blk + i * 16, stride);
        else                         idct8_add(dst + off[i], blk + i * 16, stride);
    }
}
static void idct_add8(uint8_t **dest, const int *off, int16_t *blk, int stride, const uint8_t nnzc[15 * 8])
{
    for (int j = 1; j < 3; j++)
        for (int i = j * 16; i < j * 16 + 4; i++) {
            if (nnzc[oracle_scan8(i)]) idct4_add(dest[j - 1] + off[i], blk + i * 16, stride);
            else if (blk[i * 16])      idct4_dc_add(dest[j - 1] + off[i], blk + i * 16, stride);
        }
}
/* 4:2:2 flavour, h264idct_template.c:216-239 */
static void idct_add8_422(uint8_t **dest, const int *off, int16_t *blk, int stride, const uint8_t nnzc[15 * 8])
{
    idct_add8(dest, off, blk, stride, nnzc);
    for (int j = 1; j < 3; j++)
        for (int i = j * 16 + 4; i < j * 16 + 8; i++) {
            if (nnzc[oracle_scan8(i + 4)]) idct4_add(dest[j - 1] + off[i + 4], blk + i * 16, stride);
            else if (blk[i * 16])          idct4_dc_add(dest[j - 1] + off[i + 4], blk + i * 16, stride);
        }
}

/* a3: DC transforms — h264idct_template.c:242-324.
 * Luma: 4x4 Hadamard of the 16 DC levels, rows first (butterfly outputs in the
 * order s+t, s-t, d-e, d+e), then columns; scaled (v*qmul+128)>>8 and scattered
 * to the DC slot of the 16 luma blocks. */
static void luma_dc_dequant_idct(int16_t *out, int16_t *in, int qmul)
{
    int t[16];
    static const int col_off[4] = { 0, 2 * 16, 8 * 16, 10 * 16 };
    static const int row_off[4] = { 0, 1 * 16, 4 * 16, 5 * 16 };
    for (int i = 0; i < 4; i++) {
        int s = in[4 * i] + in[4 * i + 1], d = in[4 * i] - in[4 * i + 1];
        int e = in[4 * i + 2] - in[4 * i + 3], u = in[4 * i + 2] + in[4 * i + 3];
        t[4 * i] = s + u; t[4 * i + 1] = s - u; t[4 * i + 2] = d - e; t[4 * i + 3] = d + e;
    }
    for (int i = 0; i < 4; i++) {
        int s = t[i] + t[8 + i], d = t[i] - t[8 + i];
        int e = t[4 + i] - t[12 + i], u = t[4 + i] + t[12 + i];
        int r[4] = { s + u, d + e, d - e, s - u };
        for (int k = 0; k < 4; k++)
            out[col_off[i] + row_off[k]] = (int16_t)((r[k] * qmul + 128) >> 8);
    }
}
/* Chroma 4:2:0: 2x2 Hadamard in place at DC slots 0,16,32,48; (v*qmul)>>7. */
static void chroma_dc_dequant_idct(int16_t *blk, int qmul)
{
    int a = blk[0], b = blk[16], c = blk[32], d = blk[48];
    int s0 = a + b, d0 = a - b, s1 = c + d, d1 = c - d;
    blk[0]  = (int16_t)(((s0 + s1) * qmul) >> 7);
    blk[16] = (int16_t)(((d0 + d1) * qmul) >> 7);
    blk[32] = (int16_t)(((s0 - s1) * qmul) >> 7);
    blk[48] = (int16_t)(((d0 - d1) * qmul) >> 7);
}
/* Chroma 4:2:2: 2x4, h264idct_template.c:277-302 */
static void chroma422_dc_dequant_idct(int16_t *blk, int qmul)
{
    int t[8];
    for (int i = 0; i < 4; i++) {
        t[2 * i]     = blk[32 * i] + blk[32 * i + 16];
        t[2 * i + 1] = blk[32 * i] - blk[32 * i + 16];
    }
    for (int i = 0; i < 2; i++) {
        int s = t[i] + t[4 + i], d = t[i] - t[4 + i];
        int e = t[2 + i] - t[6 + i], u = t[2 + i] + t[6 + i];
        int o = i * 16;
        blk[o]      = (int16_t)(((s + u) * qmul + 128) >> 8);
        blk[32 + o] = (int16_t)(((d + e) * qmul + 128) >> 8);
        blk[64 + o] = (int16_t)(((d - e) * qmul + 128) >> 8);
        blk[96 + o] = (int16_t)(((s - u) * qmul + 128) >> 8);
    }
}

/* a4: transform bypass — h264addpx_template.c:30-72: dst += residual, no clip */
static void add_pixels_clear_n(uint8_t *dst, int16_t *blk, int stride, int n)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[x + y * stride] = (uint8_t)(dst[x + y * stride] + blk[x + y * n]);
    memset(blk, 0, (size_t)n * n * sizeof(*blk));
}
static void add_pixels4_clear(uint8_t *dst, int16_t *blk, int stride) { add_pixels_clear_n(dst, blk, stride, 4); }
static void add_pixels8_clear(uint8_t *dst, int16_t *blk, int stride) { add_pixels_clear_n(dst, blk, stride, 8); }

/* ------------------------------------------------------------------------- */
/* a7: weighted prediction — h264dsp_template.c:30-98                          */
/* ------------------------------------------------------------------------- */
static void weight_n(uint8_t *p, int stride, int h, int ld, int w, int o, int width)
{
    o = (int)((unsigned)o << ld);
    if (ld) o += 1 << (ld - 1);
    for (int y = 0; y < h; y++, p += stride)
        for (int x = 0; x < width; x++)
            p[x] = (uint8_t)clip_u8((p[x] * w + o) >> ld);
}
static void biweight_n(uint8_t *d, uint8_t *s, int stride, int h, int ld, int wd, int ws, int o, int width)
{
    o = (int)((unsigned)((o + 1) | 1) << ld);
    for (int y = 0; y < h; y++, d += stride, s += stride)
        for (int x = 0; x < width; x++)
            d[x] = (uint8_t)clip_u8((s[x] * ws + d[x] * wd + o) >> (ld + 1));
}
#define WFUNCS(W) \
static void weight##W(uint8_t *p, int st, int h, int ld, int w, int o) { weight_n(p, st, h, ld, w, o, W); } \
static void biweight##W(uint8_t *d, uint8_t *s, int st, int h, int ld, int wd, int ws, int o) { biweight_n(d, s, st, h, ld, wd, ws, o, W); }
WFUNCS(16) WFUNCS(8) WFUNCS(4) WFUNCS(2)

/* ------------------------------------------------------------------------- */
/* a8: deblocking edge filters — h264dsp_template.c:104-328                    */
/* xs = step across the edge, ys = step along it; `inner` lines per tc0 entry. */
/* ------------------------------------------------------------------------- */
static void lf_luma(uint8_t *pix, int xs, int ys, int inner, int alpha, int beta, const int8_t *tc0)
{
    for (int i = 0; i < 4; i++) {
        int tc_o = tc0[i];
        if (tc_o < 0) { pix += inner * ys; continue; }
        for (int d = 0; d < inner; d++, pix += ys) {
            int p0 = pix[-xs], p1 = pix[-2 * xs], p2 = pix[-3 * xs];
            int q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs];
            if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
                continue;
            int tc = tc_o;
            if (iabs(p2 - p0) < beta) {
                if (tc_o)
                    pix[-2 * xs] = (uint8_t)(p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc_o, tc_o));
                tc++;
            }
            if (iabs(q2 - q0) < beta) {
                if (tc_o)
                    pix[xs] = (uint8_t)(q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc_o, tc_o));
                tc++;
            }
            int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            pix[-xs] = (uint8_t)clip_u8(p0 + delta);
            pix[0]   = (uint8_t)clip_u8(q0 - delta);
        }
    }
}
static void lf_luma_intra(uint8_t *pix, int xs, int ys, int inner, int alpha, int beta)
{
    for (int d = 0; d < 4 * inner; d++, pix += ys) {
        int p3 = pix[-4 * xs], p2 = pix[-3 * xs], p1 = pix[-2 * xs], p0 = pix[-xs];
        int q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs], q3 = pix[3 * xs];
        if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
            continue;
        if (iabs(p0 - q0) < ((alpha >> 2) + 2)) {
            if (iabs(p2 - p0) < beta) {
                pix[-xs]     = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3 * xs] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else {
                pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            }
            if (iabs(q2 - q0) < beta) {
                pix[0]      = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                pix[xs]     = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2 * xs] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else {
                pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
            }
        } else {
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0]   = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}
static void lf_chroma(uint8_t *pix, int xs, int ys, int inner, int alpha, int beta, const int8_t *tc0)
{
    for (int i = 0; i < 4; i++) {
        int tc = tc0[i];          /* ((tc0-1)<<0)+1 at 8 bits */
        if (tc <= 0) { pix += inner * ys; continue; }
        for (int d = 0; d < inner; d++, pix += ys) {
            int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
            if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
                continue;
            int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            pix[-xs] = (uint8_t)clip_u8(p0 + delta);
            pix[0]   = (uint8_t)clip_u8(q0 - delta);
        }
    }
}
static void lf_chroma_intra(uint8_t *pix, int xs, int ys, int inner, int alpha, int beta)
{
    for (int d = 0; d < 4 * inner; d++, pix += ys) {
        int p0 = pix[-xs], p1 = pix[-2 * xs], q0 = pix[0], q1 = pix[xs];
        if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
            continue;
        pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        pix[0]   = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}
/* "v" = edge is horizontal (samples across it are `stride` apart); "h" = vertical edge */
static void v_lf_luma(uint8_t *p, int st, int a, int b, int8_t *tc)         { lf_luma(p, st, 1, 4, a, b, tc); }
static void h_lf_luma(uint8_t *p, int st, int a, int b, int8_t *tc)         { lf_luma(p, 1, st, 4, a, b, tc); }
static void h_lf_luma_mbaff(uint8_t *p, int st, int a, int b, int8_t *tc)   { lf_luma(p, 1, st, 2, a, b, tc); }
static void v_lf_luma_intra(uint8_t *p, int st, int a, int b)               { lf_luma_intra(p, st, 1, 4, a, b); }
static void h_lf_luma_intra(uint8_t *p, int st, int a, int b)               { lf_luma_intra(p, 1, st, 4, a, b); }
static void h_lf_luma_mbaff_intra(uint8_t *p, int st, int a, int b)         { lf_luma_intra(p, 1, st, 2, a, b); }
static void v_lf_chroma(uint8_t *p, int st, int a, int b, int8_t *tc)       { lf_chroma(p, st, 1, 2, a, b, tc); }
static void h_lf_chroma(uint8_t *p, int st, int a, int b, int8_t *tc)       { lf_chroma(p, 1, st, 2, a, b, tc); }
static void h_lf_chroma_mbaff(uint8_t *p, int st, int a, int b, int8_t *tc) { lf_chroma(p, 1, st, 1, a, b, tc); }
static void h_lf_chroma422(uint8_t *p, int st, int a, int b, int8_t *tc)    { lf_chroma(p, 1, st, 4, a, b, tc); }
static void h_lf_chroma422_mbaff(uint8_t *p, int st, int a, int b, int8_t *tc) { lf_chroma(p, 1, st, 2, a, b, tc); }
static void v_lf_chroma_intra(uint8_t *p, int st, int a, int b)             { lf_chroma_intra(p, st, 1, 2, a, b); }
static void h_lf_chroma_intra(uint8_t *p, int st, int a, int b)             { lf_chroma_intra(p, 1, st, 2, a, b); }
static void h_lf_chroma_mbaff_intra(uint8_t *p, int st, int a, int b)       { lf_chroma_intra(p, 1, st, 1, a, b); }
static void h_lf_chroma422_intra(uint8_t *p, int st, int a, int b)          { lf_chroma_intra(p, 1, st, 4, a, b); }
static void h_lf_chroma422_mbaff_intra(uint8_t *p, int st, int a, int b)    { lf_chroma_intra(p, 1, st, 2, a, b); }

/* startcode_find_candidate — libavcodec/startcode.c: index of first zero byte */
static int startcode_find_candidate(const uint8_t *buf, int size)
{
    int i = 0;
    for (; i < size; i++)
        if (!buf[i]) break;
    return i;
}

void oracle_h264dsp_init(H264DSPContext *c, int bit_depth, int chroma_format_idc)
{
    (void)bit_depth; /* 8-bit instantiation only */
    c->weight_h264_pixels_tab[0] = weight16;   c->weight_h264_pixels_tab[1] = weight8;
    c->weight_h264_pixels_tab[2] = weight4;    c->weight_h264_pixels_tab[3] = weight2;
    c->biweight_h264_pixels_tab[0] = biweight16; c->biweight_h264_pixels_tab[1] = biweight8;
    c->biweight_h264_pixels_tab[2] = biweight4;  c->biweight_h264_pixels_tab[3] = biweight2;
    c->h264_v_loop_filter_luma = v_lf_luma;
    c->h264_h_loop_filter_luma = h_lf_luma;
    c->h264_h_loop_filter_luma_mbaff = h_lf_luma_mbaff;
    c->h264_v_loop_filter_luma_intra = v_lf_luma_intra;
    c->h264_h_loop_filter_luma_intra = h_lf_luma_intra;
    c->h264_h_loop_filter_luma_mbaff_intra = h_lf_luma_mbaff_intra;
    c->h264_v_loop_filter_chroma = v_lf_chroma;
    c->h264_h_loop_filter_chroma = chroma_format_idc <= 1 ? h_lf_chroma : h_lf_chroma422;
    c->h264_h_loop_filter_chroma_mbaff = chroma_format_idc <= 1 ? h_lf_chroma_mbaff : h_lf_chroma422_mbaff;
    c->h264_v_loop_filter_chroma_intra = v_lf_chroma_intra;
    c->h264_h_loop_filter_chroma_intra = chroma_format_idc <= 1 ? h_lf_chroma_intra : h_lf_chroma422_intra;
    c->h264_h_loop_filter_chroma_mbaff_intra = chroma_format_idc <= 1 ? h_lf_chroma_mbaff_intra : h_lf_chroma422_mbaff_intra;
    c->h264_loop_filter_strength = NULL;
    c->h264_idct_add = idct4_add;
    c->h264_idct8_add = idct8_add;
    c->h264_idct_dc_add = idct4_dc_add;
    c->h264_idct8_dc_add = idct8_dc_add;
    c->h264_idct_add16 = idct_add16;
    c->h264_idct8_add4 = idct8_add4;
    c->h264_idct_add8 = chroma_format_idc <= 1 ? idct_add8 : idct_add8_422;
    c->h264_idct_add16intra = idct_add16intra;
    c->h264_luma_dc_dequant_idct = luma_dc_dequant_idct;
    c->h264_chroma_dc_dequant_idct = chroma_format_idc <= 1 ? chroma_dc_dequant_idct : chroma422_dc_dequant_idct;
    c->h264_add_pixels8_clear = add_pixels8_clear;
    c->h264_add_pixels4_clear = add_pixels4_clear;
    c->startcode_find_candidate = startcode_find_candidate;
}

/* ------------------------------------------------------------------------- */
/* a5: quarter-pel luma MC — h264qpel_template.c:77-531                        */
/* b = horizontal half-pel, h = vertical half-pel, j = centre (6-tap of the    */
/* unclipped horizontal sums, rounded once with +512 >> 10).                   */
/* ------------------------------------------------------------------------- */
static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
static inline int hsum(const uint8_t *s) { return tap6(s[-2], s[-1], s[0], s[1], s[2], s[3]); }
static inline int vsum(const uint8_t *s, ptrdiff_t st) { return tap6(s[-2 * st], s[-st], s[0], s[st], s[2 * st], s[3 * st]); }
static inline int half_h(const uint8_t *s) { return clip_u8((hsum(s) + 16) >> 5); }
static inline int half_v(const uint8_t *s, ptrdiff_t st) { return clip_u8((vsum(s, st) + 16) >> 5); }
static inline int half_hv(const uint8_t *s, ptrdiff_t st)
{
    return clip_u8((tap6(hsum(s - 2 * st), hsum(s - st), hsum(s), hsum(s + st), hsum(s + 2 * st), hsum(s + 3 * st)) + 512) >> 10);
}
static inline int ravg(int a, int b) { return (a + b + 1) >> 1; }

void oracle_h264_qpel2(uint8_t *dst, ptrdiff_t dst_st, const uint8_t *src, ptrdiff_t st, int size, int mx, int my, int avg)
{
    uint8_t out[16 * 16];
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++) {
            const uint8_t *s = src + x + y * st;
            int v;
            switch (mx + 4 * my) {
            default:
            case 0:  v = s[0]; break;
            case 1:  v = ravg(s[0], half_h(s)); break;
            case 2:  v = half_h(s); break;
            case 3:  v = ravg(s[1], half_h(s)); break;
            case 4:  v = ravg(s[0], half_v(s, st)); break;
            case 8:  v = half_v(s, st); break;
            case 12: v = ravg(s[st], half_v(s, st)); break;
            case 5:  v = ravg(half_h(s), half_v(s, st)); break;
            case 7:  v = ravg(half_h(s), half_v(s + 1, st)); break;
            case 13: v = ravg(half_h(s + st), half_v(s, st)); break;
            case 15: v = ravg(half_h(s + st), half_v(s + 1, st)); break;
            case 10: v = half_hv(s, st); break;
            case 6:  v = ravg(half_h(s), half_hv(s, st)); break;
            case 14: v = ravg(half_h(s + st), half_hv(s, st)); break;
            case 9:  v = ravg(half_v(s, st), half_hv(s, st)); break;
            case 11: v = ravg(half_v(s + 1, st), half_hv(s, st)); break;
            }
            out[x + y * 16] = (uint8_t)v;
        }
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            dst[x + y * dst_st] = avg ? (uint8_t)ravg(dst[x + y * dst_st], out[x + y * 16]) : out[x + y * 16];
}
void oracle_h264_qpel(uint8_t *dst, const uint8_t *src, ptrdiff_t st, int size, int mx, int my, int avg)
{
    oracle_h264_qpel2(dst, st, src, st, size, mx, my, avg);
}

#define QP1(op, avg, N, mx, my) \
static void op##N##_mc##mx##my(uint8_t *d, const uint8_t *s, ptrdiff_t st) { oracle_h264_qpel(d, s, st, N, mx, my, avg); }
#define QP16(op, avg, N) \
    QP1(op, avg, N, 0, 0) QP1(op, avg, N, 1, 0) QP1(op, avg, N, 2, 0) QP1(op, avg, N, 3, 0) \
    QP1(op, avg, N, 0, 1) QP1(op, avg, N, 1, 1) QP1(op, avg, N, 2, 1) QP1(op, avg, N, 3, 1) \
    QP1(op, avg, N, 0, 2) QP1(op, avg, N, 1, 2) QP1(op, avg, N, 2, 2) QP1(op, avg, N, 3, 2) \
    QP1(op, avg, N, 0, 3) QP1(op, avg, N, 1, 3) QP1(op, avg, N, 2, 3) QP1(op, avg, N, 3, 3)
QP16(put, 0, 16) QP16(put, 0, 8) QP16(put, 0, 4) QP16(put, 0, 2)
QP16(avg, 1, 16) QP16(avg, 1, 8) QP16(avg, 1, 4)
#define QTAB(op, N) { \
    op##N##_mc00, op##N##_mc10, op##N##_mc20, op##N##_mc30, op##N##_mc01, op##N##_mc11, op##N##_mc21, op##N##_mc31, \
    op##N##_mc02, op##N##_mc12, op##N##_mc22, op##N##_mc32, op##N##_mc03, op##N##_mc13, op##N##_mc23, op##N##_mc33 }

void oracle_h264qpel_init(H264QpelContext *c, int bit_depth)
{
    static const qpel_mc_func put[4][16] = { QTAB(put, 16), QTAB(put, 8), QTAB(put, 4), QTAB(put, 2) };
    static const qpel_mc_func avg[3][16] = { QTAB(avg, 16), QTAB(avg, 8), QTAB(avg, 4) };
    (void)bit_depth;
    memcpy(c->put_h264_qpel_pixels_tab, put, sizeof(put));
    memcpy(c->avg_h264_qpel_pixels_tab, avg, sizeof(avg)); /* avg[3] stays as the caller left it (h264qpel.c:61-68) */
}

/* ------------------------------------------------------------------------- */
/* a6: 1/8-pel bilinear chroma MC — h264chroma_template.c:27-173               */
/* The reference skips zero-weight taps (so it never reads them); results are  */
/* identical to the 4-tap formula, but we mirror the read pattern to stay      */
/* in-bounds on the same inputs.                                               */
/* ------------------------------------------------------------------------- */
void oracle_h264_chroma_mc2(uint8_t *dst, ptrdiff_t dst_st, const uint8_t *src, ptrdiff_t st, int h, int x, int y, int w, int avg)
{
    int A = (8 - x) * (8 - y), B = x * (8 - y), C = (8 - x) * y, D = x * y;
    for (int j = 0; j < h; j++, dst += dst_st, src += st)
        for (int i = 0; i < w; i++) {
            int v = A * src[i];
            if (B) v += B * src[i + 1];
            if (C) v += C * src[i + st];
            if (D) v += D * src[i + st + 1];
            v = (v + 32) >> 6;
            dst[i] = (uint8_t)(avg ? ravg(dst[i], v) : v);
        }
}
#define CFUNCS(W) \
static void put_chroma##W(uint8_t *d, uint8_t *s, ptrdiff_t st, int h, int x, int y) { oracle_h264_chroma_mc2(d, st, s, st, h, x, y, W, 0); } \
static void avg_chroma##W(uint8_t *d, uint8_t *s, ptrdiff_t st, int h, int x, int y) { oracle_h264_chroma_mc2(d, st, s, st, h, x, y, W, 1); }
CFUNCS(8) CFUNCS(4) CFUNCS(2)

void oracle_h264chroma_init(H264ChromaContext *c, int bit_depth)
{
    (void)bit_depth;
    c->put_h264_chroma_pixels_tab[0] = put_chroma8; c->put_h264_chroma_pixels_tab[1] = put_chroma4;
    c->put_h264_chroma_pixels_tab[2] = put_chroma2;
    c->avg_h264_chroma_pixels_tab[0] = avg_chroma8; c->avg_h264_chroma_pixels_tab[1] = avg_chroma4;
    c->avg_h264_chroma_pixels_tab[2] = avg_chroma2;
}

/* ------------------------------------------------------------------------- */
/* a11: border-replicating block fetch — videodsp_template.c:24-96             */
/* == read every sample at coordinates clamped into the w x h plane.           */
/* ------------------------------------------------------------------------- */
static void emulated_edge_mc(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_ls, ptrdiff_t src_ls,
                             int bw, int bh, int sx, int sy, int w, int h)
{
    if (!w || !h) return;
    /* the reference first pulls a fully-outside block back to the nearest edge
     * row/column (videodsp_template.c:34-47); clamping per sample is the same */
    const uint8_t *origin = src - sy * src_ls - sx; /* sample (0,0) of the plane */
    for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
            int cx = clip3(sx + x, 0, w - 1), cy = clip3(sy + y, 0, h - 1);
            buf[x + y * buf_ls] = origin[cx + cy * src_ls];
        }
}
static void prefetch_nop(uint8_t *buf, ptrdiff_t stride, int h) { (void)buf; (void)stride; (void)h; }

void oracle_videodsp_init(VideoDSPContext *c, int bpc)
{
    (void)bpc;
    c->emulated_edge_mc = emulated_edge_mc;
    c->prefetch = prefetch_nop;
}
