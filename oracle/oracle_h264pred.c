/*
 * oracle_h264pred.c — CPU restatement of the reference's H.264 intra predictors
 * (8-bit).  TEST INFRASTRUCTURE ONLY (see oracle_h264dsp.c header).
 *
 * Follows libavcodec/h264pred_template.c: pred4x4 :34-327, pred16x16 :329-486,
 * pred8x8 (chroma) :488-802, pred8x8l with the (1,2,1) edge pre-filter
 * :846-1125; slot numbering from libavcodec/h264pred.h:34-88; table filling
 * for codec_id == H264 from libavcodec/h264pred.c:402-560.
 *
 * Written from the standard's formulation: every directional mode is a
 * function of two reference vectors T[-1..2N-1] (row above, T[-1] = corner)
 * and L[-1..N-1] (column to the left), for N = 4 (raw edge samples) and N = 8
 * (edge samples after the low-pass pre-filter).
 * The lossless `*_add` slots (transform bypass, :1127-1354) are at the end of the file.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "../include/mi355_abi.h"
#include "oracle.h"

static inline int clip_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static inline int f3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
static inline int f2(int a, int b) { return (a + b + 1) >> 1; }

/* T and L are indexed from -1: pass pointers to element 0. */
static void pred_dir(uint8_t *dst, ptrdiff_t st, int N, int mode, const int *T, const int *L)
{
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            int v = 0;
            switch (mode) {
            case VERT_PRED: v = T[x]; break;
            case HOR_PRED:  v = L[y]; break;
            case DIAG_DOWN_LEFT_PRED:
                v = (x == N - 1 && y == N - 1) ? (T[2 * N - 2] + 3 * T[2 * N - 1] + 2) >> 2
                                               : f3(T[x + y], T[x + y + 1], T[x + y + 2]);
                break;
            case DIAG_DOWN_RIGHT_PRED:
                if (x > y)      v = f3(T[x - y - 2], T[x - y - 1], T[x - y]);
                else if (x < y) v = f3(L[y - x - 2], L[y - x - 1], L[y - x]);
                else            v = f3(T[0], T[-1], L[0]);
                break;
            case VERT_RIGHT_PRED: {
                int z = 2 * x - y, i = x - (y >> 1);
                if (z >= 0 && !(z & 1)) v = f2(T[i - 1], T[i]);
                else if (z > 0)         v = f3(T[i - 2], T[i - 1], T[i]);
                else if (z == -1)       v = f3(L[0], T[-1], T[0]);
                else                    v = f3(L[y - 2 * x - 1], L[y - 2 * x - 2], L[y - 2 * x - 3]);
                break;
            }
            case HOR_DOWN_PRED: {
                int z = 2 * y - x, i = y - (x >> 1);
                if (z >= 0 && !(z & 1)) v = f2(L[i - 1], L[i]);
                else if (z > 0)         v = f3(L[i - 2], L[i - 1], L[i]);
                else if (z == -1)       v = f3(L[0], T[-1], T[0]);
                else                    v = f3(T[x - 2 * y - 1], T[x - 2 * y - 2], T[x - 2 * y - 3]);
                break;
            }
            case VERT_LEFT_PRED: {
                int i = x + (y >> 1);
                v = (y & 1) ? f3(T[i], T[i + 1], T[i + 2]) : f2(T[i], T[i + 1]);
                break;
            }
            case HOR_UP_PRED: {
                int z = x + 2 * y, i = y + (x >> 1);
                if (z > 2 * N - 3)       v = L[N - 1];
                else if (z == 2 * N - 3) v = (L[N - 2] + 3 * L[N - 1] + 2) >> 2;
                else if (z & 1)          v = f3(L[i], L[i + 1], L[i + 2]);
                else                     v = f2(L[i], L[i + 1]);
                break;
            }
            }
            dst[x + y * st] = (uint8_t)v;
        }
}

static void fill(uint8_t *dst, ptrdiff_t st, int w, int h, int v)
{
    for (int y = 0; y < h; y++) memset(dst + y * st, v, (size_t)w);
}

/* ---- 4x4: h264pred_template.c:34-327; t4..t7 come from `topright` -------- */
static void pred4x4(uint8_t *src, const uint8_t *topright, ptrdiff_t st, int mode)
{
    int Tb[10] = {0}, Lb[6] = {0};
    int *T = Tb + 1, *L = Lb + 1;
    int need_top = 0, need_left = 0, need_lt = 0, need_tr = 0, sum;
    switch (mode) {
    case VERT_PRED: case TOP_DC_PRED: need_top = 1; break;
    case HOR_PRED: case LEFT_DC_PRED: case HOR_UP_PRED: need_left = 1; break;
    case DC_PRED: need_top = need_left = 1; break;
    case DIAG_DOWN_LEFT_PRED: case VERT_LEFT_PRED: need_top = need_tr = 1; break;
    case DIAG_DOWN_RIGHT_PRED: case VERT_RIGHT_PRED: case HOR_DOWN_PRED: need_top = need_left = need_lt = 1; break;
    default: break;
    }
    if (need_top)  for (int i = 0; i < 4; i++) T[i] = src[i - st];
    if (need_tr)   for (int i = 0; i < 4; i++) T[4 + i] = topright[i];
    if (need_left) for (int i = 0; i < 4; i++) L[i] = src[-1 + i * st];
    if (need_lt)   T[-1] = L[-1] = src[-1 - st];
    switch (mode) {
    case DC_PRED:      sum = T[0] + T[1] + T[2] + T[3] + L[0] + L[1] + L[2] + L[3]; fill(src, st, 4, 4, (sum + 4) >> 3); break;
    case LEFT_DC_PRED: sum = L[0] + L[1] + L[2] + L[3]; fill(src, st, 4, 4, (sum + 2) >> 2); break;
    case TOP_DC_PRED:  sum = T[0] + T[1] + T[2] + T[3]; fill(src, st, 4, 4, (sum + 2) >> 2); break;
    case DC_128_PRED:  fill(src, st, 4, 4, 128); break;
    default:           pred_dir(src, st, 4, mode, T, L); break;
    }
}

/* ---- 8x8 luma: h264pred_template.c:846-1125 ------------------------------ */
static void pred8x8l(uint8_t *src, int has_topleft, int has_topright, ptrdiff_t st, int mode)
{
    int Tb[18] = {0}, Lb[10] = {0};
    int *T = Tb + 1, *L = Lb + 1;
    int need_top = 0, need_left = 0, need_lt = 0, need_tr = 0, sum = 0;
#define S(x, y) ((int)src[(x) + (y) * st])
    switch (mode) {
    case VERT_PRED: case TOP_DC_PRED: need_top = 1; break;
    case HOR_PRED: case LEFT_DC_PRED: case HOR_UP_PRED: need_left = 1; break;
    case DC_PRED: need_top = need_left = 1; break;
    case DIAG_DOWN_LEFT_PRED: case VERT_LEFT_PRED: need_top = need_tr = 1; break;
    case DIAG_DOWN_RIGHT_PRED: case VERT_RIGHT_PRED: case HOR_DOWN_PRED: need_top = need_left = need_lt = 1; break;
    default: break;
    }
    if (need_left) {   /* PREDICT_8x8_LOAD_LEFT :849-853 */
        L[0] = f3(has_topleft ? S(-1, -1) : S(-1, 0), S(-1, 0), S(-1, 1));
        for (int y = 1; y < 7; y++) L[y] = f3(S(-1, y - 1), S(-1, y), S(-1, y + 1));
        L[7] = (S(-1, 6) + 3 * S(-1, 7) + 2) >> 2;
    }
    if (need_top) {    /* PREDICT_8x8_LOAD_TOP :856-861 */
        T[0] = f3(has_topleft ? S(-1, -1) : S(0, -1), S(0, -1), S(1, -1));
        for (int x = 1; x < 7; x++) T[x] = f3(S(x - 1, -1), S(x, -1), S(x + 1, -1));
        T[7] = f3(has_topright ? S(8, -1) : S(7, -1), S(7, -1), S(6, -1));
    }
    if (need_tr) {     /* PREDICT_8x8_LOAD_TOPRIGHT :864-870 */
        if (has_topright) {
            for (int x = 8; x < 15; x++) T[x] = f3(S(x - 1, -1), S(x, -1), S(x + 1, -1));
            T[15] = (S(14, -1) + 3 * S(15, -1) + 2) >> 2;
        } else {
            for (int x = 8; x < 16; x++) T[x] = S(7, -1);
        }
    }
    if (need_lt)       /* PREDICT_8x8_LOAD_TOPLEFT :872-873 */
        T[-1] = L[-1] = f3(S(-1, 0), S(-1, -1), S(0, -1));
#undef S
    switch (mode) {
    case DC_PRED:      for (int i = 0; i < 8; i++) sum += T[i] + L[i]; fill(src, st, 8, 8, (sum + 8) >> 4); break;
    case LEFT_DC_PRED: for (int i = 0; i < 8; i++) sum += L[i];        fill(src, st, 8, 8, (sum + 4) >> 3); break;
    case TOP_DC_PRED:  for (int i = 0; i < 8; i++) sum += T[i];        fill(src, st, 8, 8, (sum + 4) >> 3); break;
    case DC_128_PRED:  fill(src, st, 8, 8, 128); break;
    default:           pred_dir(src, st, 8, mode, T, L); break;
    }
}

/* ---- plane prediction: 16x16 :434-481, 8x8 chroma :768-802 --------------- */
static void pred_plane(uint8_t *src, ptrdiff_t st, int N)
{
    int half = N / 2, H = 0, V = 0;
    for (int k = 1; k <= half; k++) {
        H += k * (src[half - 1 + k - st] - src[half - 1 - k - st]);
        V += k * (src[-1 + (half - 1 + k) * st] - src[-1 + (half - 1 - k) * st]);
    }
    if (N == 16) { H = (5 * H + 32) >> 6;  V = (5 * V + 32) >> 6; }
    else         { H = (17 * H + 16) >> 5; V = (17 * V + 16) >> 5; }
    int a = 16 * (src[-1 + (N - 1) * st] + src[N - 1 - st] + 1) - (half - 1) * (V + H);
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++)
            src[x + y * st] = (uint8_t)clip_u8((a + x * H + y * V) >> 5);
}

/* ---- 16x16: h264pred_template.c:329-486 ---------------------------------- */
static void pred16x16(uint8_t *src, ptrdiff_t st, int mode)
{
    int sl = 0, stp = 0;
    if (mode == DC_PRED8x8 || mode == LEFT_DC_PRED8x8) for (int i = 0; i < 16; i++) sl += src[-1 + i * st];
    if (mode == DC_PRED8x8 || mode == TOP_DC_PRED8x8)  for (int i = 0; i < 16; i++) stp += src[i - st];
    switch (mode) {
    case VERT_PRED8x8:  for (int y = 0; y < 16; y++) memmove(src + y * st, src - st, 16); break;
    case HOR_PRED8x8:   for (int y = 0; y < 16; y++) memset(src + y * st, src[-1 + y * st], 16); break;
    case DC_PRED8x8:      fill(src, st, 16, 16, (sl + stp + 16) >> 5); break;
    case LEFT_DC_PRED8x8: fill(src, st, 16, 16, (sl + 8) >> 4); break;
    case TOP_DC_PRED8x8:  fill(src, st, 16, 16, (stp + 8) >> 4); break;
    case DC_128_PRED8x8:  fill(src, st, 16, 16, 128); break;
    case PLANE_PRED8x8:   pred_plane(src, st, 16); break;
    }
}

/* ---- 8x8 chroma: h264pred_template.c:488-802 ------------------------------
 * DC works per 4x4 quadrant; quadrant (qx,qy) uses the 4 top samples above it
 * (t) and/or the 4 left samples beside it (l):
 *   full DC:  q00 = t+l, q10 = t, q01 = l, q11 = t+l
 *   left DC:  every quadrant = l of its row;  top DC: = t of its column.
 * The four "mad cow" slots patch quadrants for partially available edges. */
static int sum4_top(const uint8_t *src, ptrdiff_t st, int qx) { int s = 0; for (int i = 0; i < 4; i++) s += src[4 * qx + i - st]; return s; }
static int sum4_left(const uint8_t *src, ptrdiff_t st, int qy) { int s = 0; for (int i = 0; i < 4; i++) s += src[-1 + (4 * qy + i) * st]; return s; }

static void chroma_dc_quads(uint8_t *src, ptrdiff_t st, int use_top, int use_left)
{
    int t[2] = {0, 0}, l[2] = {0, 0}, q[2][2];
    if (use_top)  { t[0] = sum4_top(src, st, 0);  t[1] = sum4_top(src, st, 1); }
    if (use_left) { l[0] = sum4_left(src, st, 0); l[1] = sum4_left(src, st, 1); }
    if (use_top && use_left) {
        q[0][0] = (t[0] + l[0] + 4) >> 3; q[0][1] = (t[1] + 2) >> 2;
        q[1][0] = (l[1] + 2) >> 2;        q[1][1] = (t[1] + l[1] + 4) >> 3;
    } else if (use_left) {
        q[0][0] = q[0][1] = (l[0] + 2) >> 2; q[1][0] = q[1][1] = (l[1] + 2) >> 2;
    } else {
        q[0][0] = q[1][0] = (t[0] + 2) >> 2; q[0][1] = q[1][1] = (t[1] + 2) >> 2;
    }
    for (int qy = 0; qy < 2; qy++)
        for (int qx = 0; qx < 2; qx++)
            fill(src + 4 * qx + 4 * qy * st, st, 4, 4, q[qy][qx]);
}

static void pred8x8(uint8_t *src, ptrdiff_t st, int mode)
{
    int s;
    switch (mode) {
    case VERT_PRED8x8:  for (int y = 0; y < 8; y++) memmove(src + y * st, src - st, 8); break;
    case HOR_PRED8x8:   for (int y = 0; y < 8; y++) memset(src + y * st, src[-1 + y * st], 8); break;
    case DC_PRED8x8:      chroma_dc_quads(src, st, 1, 1); break;
    case LEFT_DC_PRED8x8: chroma_dc_quads(src, st, 0, 1); break;
    case TOP_DC_PRED8x8:  chroma_dc_quads(src, st, 1, 0); break;
    case DC_128_PRED8x8:  fill(src, st, 8, 8, 128); break;
    case PLANE_PRED8x8:   pred_plane(src, st, 8); break;
    case ALZHEIMER_DC_L0T_PRED8x8: /* :716-720: top DC, then quadrant 00 = DC(top+left) */
        s = sum4_top(src, st, 0) + sum4_left(src, st, 0);
        chroma_dc_quads(src, st, 1, 0);
        fill(src, st, 4, 4, (s + 4) >> 3);
        break;
    case ALZHEIMER_DC_0LT_PRED8x8: /* :729-733: full DC, then quadrant 00 = top DC */
        s = sum4_top(src, st, 0);
        chroma_dc_quads(src, st, 1, 1);
        fill(src, st, 4, 4, (s + 2) >> 2);
        break;
    case ALZHEIMER_DC_L00_PRED8x8: /* :742-747: left DC, bottom half = 128 */
        chroma_dc_quads(src, st, 0, 1);
        fill(src + 4 * st, st, 8, 4, 128);
        break;
    case ALZHEIMER_DC_0L0_PRED8x8: /* :756-761: left DC, top half = 128 */
        chroma_dc_quads(src, st, 0, 1);
        fill(src, st, 8, 4, 128);
        break;
    }
}

/* ---- 8x16 chroma (4:2:0's pred8x8[] slots when chroma_format_idc == 2): h264pred_template.c:502-846 ----
 * Same per-4x4-quadrant DC scheme with four quadrant rows: a quadrant in the first row or the first column uses
 * what lies next to it, every other one both sums (:673-720).  left_dc is the 8x8 rule on both halves (:590-595),
 * top_dc the column sums for all sixteen rows (:622-642).  The "mad cow" slots (:722-766) patch the top-left
 * quadrant or rows 4..7 / 0..3 exactly like their 8x8 forms (the 128 patch is 4 rows high, not 8). */
static void chroma422_dc_quads(uint8_t *src, ptrdiff_t st, int use_top, int use_left)
{
    for (int qy = 0; qy < 4; qy++)
        for (int qx = 0; qx < 2; qx++) {
            const int t = use_top ? sum4_top(src, st, qx) : 0, l = use_left ? sum4_left(src, st, qy) : 0;
            int v;
            if (use_top && use_left) v = qy == 0 ? (qx ? (t + 2) >> 2 : (t + l + 4) >> 3) : (qx ? (t + l + 4) >> 3 : (l + 2) >> 2);
            else if (use_left) v = (l + 2) >> 2;
            else v = (t + 2) >> 2;
            fill(src + 4 * qx + 4 * qy * st, st, 4, 4, v);
        }
}
static void pred8x16(uint8_t *src, ptrdiff_t st, int mode)
{
    int s;
    switch (mode) {
    case VERT_PRED8x8:  for (int y = 0; y < 16; y++) memmove(src + y * st, src - st, 8); break;
    case HOR_PRED8x8:   for (int y = 0; y < 16; y++) memset(src + y * st, src[-1 + y * st], 8); break;
    case DC_PRED8x8:      chroma422_dc_quads(src, st, 1, 1); break;
    case LEFT_DC_PRED8x8: chroma422_dc_quads(src, st, 0, 1); break;
    case TOP_DC_PRED8x8:  chroma422_dc_quads(src, st, 1, 0); break;
    case DC_128_PRED8x8:  fill(src, st, 8, 16, 128); break;
    case PLANE_PRED8x8: { /* :804-846 */
        int H = 0, V = 0;
        for (int k = 1; k <= 4; k++) H += k * (src[3 + k - st] - src[3 - k - st]);
        for (int k = 1; k <= 8; k++) V += k * (src[-1 + (7 + k) * st] - src[-1 + (7 - k) * st]);
        H = (17 * H + 16) >> 5; V = (5 * V + 32) >> 6;
        const int a = 16 * (src[-1 + 15 * st] + src[7 - st] + 1) - 7 * V - 3 * H;
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 8; x++)
                src[x + y * st] = (uint8_t)clip_u8((a + x * H + y * V) >> 5);
        break;
    }
    case ALZHEIMER_DC_L0T_PRED8x8:
        s = sum4_top(src, st, 0) + sum4_left(src, st, 0);
        chroma422_dc_quads(src, st, 1, 0);
        fill(src, st, 4, 4, (s + 4) >> 3);
        break;
    case ALZHEIMER_DC_0LT_PRED8x8:
        s = sum4_top(src, st, 0);
        chroma422_dc_quads(src, st, 1, 1);
        fill(src, st, 4, 4, (s + 2) >> 2);
        break;
    case ALZHEIMER_DC_L00_PRED8x8:
        chroma422_dc_quads(src, st, 0, 1);
        fill(src + 4 * st, st, 8, 4, 128);
        break;
    case ALZHEIMER_DC_0L0_PRED8x8:
        chroma422_dc_quads(src, st, 0, 1);
        fill(src, st, 8, 4, 128);
        break;
    }
}

/* ---- table plumbing ------------------------------------------------------- */
#define P4(m)  static void p4_##m(uint8_t *s, const uint8_t *tr, ptrdiff_t st) { pred4x4(s, tr, st, m); }
#define P8L(m) static void p8l_##m(uint8_t *s, int tl, int tr, ptrdiff_t st) { pred8x8l(s, tl, tr, st, m); }
#define P8(m)  static void p8_##m(uint8_t *s, ptrdiff_t st) { pred8x8(s, st, m); } \
               static void p8x16_##m(uint8_t *s, ptrdiff_t st) { pred8x16(s, st, m); }
#define P16(m) static void p16_##m(uint8_t *s, ptrdiff_t st) { pred16x16(s, st, m); }
P4(0) P4(1) P4(2) P4(3) P4(4) P4(5) P4(6) P4(7) P4(8) P4(9) P4(10) P4(11)
P8L(0) P8L(1) P8L(2) P8L(3) P8L(4) P8L(5) P8L(6) P8L(7) P8L(8) P8L(9) P8L(10) P8L(11)
P8(0) P8(1) P8(2) P8(3) P8(4) P8(5) P8(6) P8(7) P8(8) P8(9) P8(10)
P16(0) P16(1) P16(2) P16(3) P16(4) P16(5) P16(6)

/* ---- lossless prediction + residual (h264pred_template.c:1127-1354): a running sum along the prediction
 * direction that starts at the neighbouring sample and wraps like the reference's `pixel` type at every
 * step; the coefficient block is cleared ----------------------------------------------------------------- */
static void run_add(uint8_t *pix, const int16_t *blk, ptrdiff_t stride, int n, int horizontal, const uint8_t *start)
{
    for (int i = 0; i < n; i++) {
        uint8_t v = start[i];
        for (int k = 0; k < n; k++) {
            v = (uint8_t)(v + (horizontal ? blk[i * n + k] : blk[k * n + i]));
            if (horizontal) pix[i * stride + k] = v; else pix[k * stride + i] = v;
        }
    }
}
static void pred_add(uint8_t *pix, int16_t *blk, ptrdiff_t stride, int n, int horizontal)
{
    uint8_t start[8];
    for (int i = 0; i < n; i++) start[i] = horizontal ? pix[i * stride - 1] : pix[i - stride];
    run_add(pix, blk, stride, n, horizontal, start);
    memset(blk, 0, sizeof(*blk) * n * n);
}
static void p4_vadd(uint8_t *p, int16_t *b, ptrdiff_t s) { pred_add(p, b, s, 4, 0); }
static void p4_hadd(uint8_t *p, int16_t *b, ptrdiff_t s) { pred_add(p, b, s, 4, 1); }
static void p8l_vadd(uint8_t *p, int16_t *b, ptrdiff_t s) { pred_add(p, b, s, 8, 0); }
static void p8l_hadd(uint8_t *p, int16_t *b, ptrdiff_t s) { pred_add(p, b, s, 8, 1); }
/* the 8x8 variants that start from the (1,2,1)-filtered edge: PREDICT_8x8_LOAD_TOP :857-862, _LEFT :849-853 */
static void p8l_vfadd(uint8_t *p, int16_t *b, int has_tl, int has_tr, ptrdiff_t s)
{
    uint8_t t[8];
#define SRC(x, y) p[(x) + (y) * s]
    t[0] = (uint8_t)(((has_tl ? SRC(-1, -1) : SRC(0, -1)) + 2 * SRC(0, -1) + SRC(1, -1) + 2) >> 2);
    for (int x = 1; x < 7; x++) t[x] = (uint8_t)((SRC(x - 1, -1) + 2 * SRC(x, -1) + SRC(x + 1, -1) + 2) >> 2);
    t[7] = (uint8_t)(((has_tr ? SRC(8, -1) : SRC(7, -1)) + 2 * SRC(7, -1) + SRC(6, -1) + 2) >> 2);
    run_add(p, b, s, 8, 0, t);
    memset(b, 0, sizeof(*b) * 64);
}
static void p8l_hfadd(uint8_t *p, int16_t *b, int has_tl, int has_tr, ptrdiff_t s)
{
    uint8_t l[8];
    (void)has_tr;
    l[0] = (uint8_t)(((has_tl ? SRC(-1, -1) : SRC(-1, 0)) + 2 * SRC(-1, 0) + SRC(-1, 1) + 2) >> 2);
    for (int y = 1; y < 7; y++) l[y] = (uint8_t)((SRC(-1, y - 1) + 2 * SRC(-1, y) + SRC(-1, y + 1) + 2) >> 2);
    l[7] = (uint8_t)((SRC(-1, 6) + 3 * SRC(-1, 7) + 2) >> 2);
#undef SRC
    run_add(p, b, s, 8, 1, l);
    memset(b, 0, sizeof(*b) * 64);
}
static void multi_add(uint8_t *pix, const int *off, int16_t *blk, ptrdiff_t s, int nblk, int horizontal)
{
    for (int i = 0; i < nblk; i++) pred_add(pix + off[i], blk + i * 16, s, 4, horizontal);
}
static void p16_vadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { multi_add(p, o, b, s, 16, 0); }
static void p16_hadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { multi_add(p, o, b, s, 16, 1); }
static void p8_vadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { multi_add(p, o, b, s, 4, 0); }
static void p8_hadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { multi_add(p, o, b, s, 4, 1); }
/* pred8x16_*_add :1326-1354: blocks 4..7 use block_offset[8..11] */
static void p8x16_add(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s, int horizontal)
{
    for (int i = 0; i < 4; i++) pred_add(p + o[i], b + i * 16, s, 4, horizontal);
    for (int i = 4; i < 8; i++) pred_add(p + o[i + 4], b + i * 16, s, 4, horizontal);
}
static void p8x16_vadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { p8x16_add(p, o, b, s, 0); }
static void p8x16_hadd(uint8_t *p, const int *o, int16_t *b, ptrdiff_t s) { p8x16_add(p, o, b, s, 1); }

/* Fills the H.264 slots (codec_id == AV_CODEC_ID_H264; chroma_format_idc 2 puts the 8x16 forms into the
 * pred8x8 slots, h264pred.c:470-531, :558-565); leaves every other slot (VP8/RV40/SVQ3 flavours) untouched. */
void oracle_h264_pred_init(H264PredContext *h, int codec_id, int bit_depth, int chroma_format_idc)
{
    (void)codec_id; (void)bit_depth;
    h->pred4x4[0] = p4_0; h->pred4x4[1] = p4_1; h->pred4x4[2] = p4_2; h->pred4x4[3] = p4_3;
    h->pred4x4[4] = p4_4; h->pred4x4[5] = p4_5; h->pred4x4[6] = p4_6; h->pred4x4[7] = p4_7;
    h->pred4x4[8] = p4_8; h->pred4x4[9] = p4_9; h->pred4x4[10] = p4_10; h->pred4x4[11] = p4_11;
    h->pred8x8l[0] = p8l_0; h->pred8x8l[1] = p8l_1; h->pred8x8l[2] = p8l_2; h->pred8x8l[3] = p8l_3;
    h->pred8x8l[4] = p8l_4; h->pred8x8l[5] = p8l_5; h->pred8x8l[6] = p8l_6; h->pred8x8l[7] = p8l_7;
    h->pred8x8l[8] = p8l_8; h->pred8x8l[9] = p8l_9; h->pred8x8l[10] = p8l_10; h->pred8x8l[11] = p8l_11;
    h->pred8x8[0] = p8_0; h->pred8x8[1] = p8_1; h->pred8x8[2] = p8_2; h->pred8x8[3] = p8_3;
    h->pred8x8[4] = p8_4; h->pred8x8[5] = p8_5; h->pred8x8[6] = p8_6; h->pred8x8[7] = p8_7;
    h->pred8x8[8] = p8_8; h->pred8x8[9] = p8_9; h->pred8x8[10] = p8_10;
    h->pred16x16[0] = p16_0; h->pred16x16[1] = p16_1; h->pred16x16[2] = p16_2; h->pred16x16[3] = p16_3;
    h->pred16x16[4] = p16_4; h->pred16x16[5] = p16_5; h->pred16x16[6] = p16_6;
    /* VERT_PRED 0 / HOR_PRED 1; VERT_PRED8x8 2 / HOR_PRED8x8 1 (h264pred.h:38-39, :69-70; h264pred.c:551-565) */
    h->pred4x4_add[0] = p4_vadd; h->pred4x4_add[1] = p4_hadd;
    h->pred8x8l_add[0] = p8l_vadd; h->pred8x8l_add[1] = p8l_hadd;
    h->pred8x8l_filter_add[0] = p8l_vfadd; h->pred8x8l_filter_add[1] = p8l_hfadd;
    h->pred8x8_add[2] = p8_vadd; h->pred8x8_add[1] = p8_hadd;
    h->pred16x16_add[2] = p16_vadd; h->pred16x16_add[1] = p16_hadd;
    if (chroma_format_idc == 2) {
        h->pred8x8[0] = p8x16_0; h->pred8x8[1] = p8x16_1; h->pred8x8[2] = p8x16_2; h->pred8x8[3] = p8x16_3;
        h->pred8x8[4] = p8x16_4; h->pred8x8[5] = p8x16_5; h->pred8x8[6] = p8x16_6; h->pred8x8[7] = p8x16_7;
        h->pred8x8[8] = p8x16_8; h->pred8x8[9] = p8x16_9; h->pred8x8[10] = p8x16_10;
        h->pred8x8_add[2] = p8x16_vadd; h->pred8x8_add[1] = p8x16_hadd;
    }
}
