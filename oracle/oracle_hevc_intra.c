/* TEST INFRASTRUCTURE (oracle/): CPU restatement of the reference's HEVC intra prediction WRAPPER,
 * libavcodec/hevcpred_template.c:31-334 (`intra_pred`, reached through HEVCPredContext.intra_pred[log2_size - 2],
 * hevcdec.c:1274-1285): availability, neighbour gather, constrained-intra substitution, inference of unavailable
 * samples, smoothing, then pred_planar / pred_dc / pred_angular (restated in oracle_hevcdsp.c).
 * Pinned to the reference's own function compiled in place (oracle/_ref/libhevcfilterref.so: ref_hevc_intra_pred_blocks)
 * by tests/test_oracle_hevc_intra.py and to the golden sha1s it produced (tests/golden/hevc_intra_ref_sha1.json).
 * Only tests/, smoke() and bench.py's cpu_baseline may call this; the product never does. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "oracle.h"
#include "../include/mi355_hevc_batch.h"

#define NMAX 32                                   /* MAX_TB_SIZE */

typedef struct {
    const mi355_hevc_intra_picture *p;
    int x0, y0, hs, vs;                           /* luma position of the block, plane shifts */
} Where;

/* is_intra of the prediction unit that covers plane sample (dx, dy) relative to the block's first sample
 * (IS_INTRA / MVF_PU, :36-41; a negative offset is scaled like the reference's arithmetic shift) */
static int intra_at(const Where *w, int dx, int dy)
{
    const mi355_hevc_intra_picture *p = w->p;
    const int xp = (w->x0 + dx * (1 << w->hs)) >> p->log2_min_pu_size;
    const int yp = (w->y0 + dy * (1 << w->vs)) >> p->log2_min_pu_size;
    return p->tab_mvf[xp + yp * p->min_pu_width].is_intra;
}
static int any_intra_pu(const mi355_hevc_intra_picture *p, int xp, int yp, int step_x, int count)
{
    int hit = 0;
    for (int i = 0; i < count; i++)
        hit |= p->tab_mvf[(xp + (step_x ? i : 0)) + (yp + (step_x ? 0 : i)) * p->min_pu_width].is_intra;
    return hit;
}
static int imin_(int a, int b) { return a < b ? a : b; }
static void fill(int *d, int v, int n) { for (int i = 0; i < n; i++) d[i] = v; }

void oracle_hevc_intra_pred_block(const mi355_hevc_intra_picture *p, const mi355_hevc_intra_block *b, int bit_depth)
{
    const int c = b->c_idx, hs = c ? p->hshift : 0, vs = c ? p->vshift : 0;
    const int n = 1 << b->log2_size, x0 = b->x0, y0 = b->y0;
    const int nl = n << hs;                                         /* size_in_luma (:74: hshift both ways) */
    const int ntb = nl >> p->log2_min_tb_size;
    const int px = bit_depth > 8 ? 2 : 1;
    const ptrdiff_t st = p->linesize[c] / px;                       /* in samples (:83) */
    uint8_t *org = p->data[c] + ((x0 >> hs) + (y0 >> vs) * st) * px;
#define PIX(dx, dy) (px == 2 ? (int)((const uint16_t *)org)[(dx) + (dy) * st] : (int)org[(dx) + (dy) * st])
#define ZS(xt, yt) p->min_tb_addr_zs[(yt) * p->min_tb_width + (xt)]
    const int xtb = x0 >> p->log2_min_tb_size, ytb = y0 >> p->log2_min_tb_size, here = ZS(xtb, ytb);
    const Where w = { p, x0, y0, hs, vs };

    int Lb[2 * NMAX + 1], Tb[2 * NMAX + 1], FLb[2 * NMAX + 1], FTb[2 * NMAX + 1];
    int *L = Lb + 1, *T = Tb + 1, *FL = FLb + 1, *FT = FTb + 1;    /* [-1] = the corner */

    /* availability: the caller's flags, narrowed by decoding order for the two far neighbours (:93-97) */
    int a_bl = (b->cand & MI355_HEVC_CAND_BOTTOM_LEFT) && here > ZS(xtb - 1, ytb + ntb);
    int a_l = !!(b->cand & MI355_HEVC_CAND_LEFT), a_ul = !!(b->cand & MI355_HEVC_CAND_UP_LEFT), a_u = !!(b->cand & MI355_HEVC_CAND_UP);
    int a_ur = (b->cand & MI355_HEVC_CAND_UP_RIGHT) && here > ZS(xtb + ntb, ytb - 1);
    const int n_bl = (imin_(y0 + 2 * nl, p->height) - (y0 + nl)) >> vs;     /* rows that exist below-left (:99-102) */
    const int n_ur = (imin_(x0 + 2 * nl, p->width) - (x0 + nl)) >> hs;
    const int cip = p->constrained_intra_pred == 1;

    if (cip) {                                                      /* :104-151 */
        const int l2pu = p->log2_min_pu_size, pu_mask = (1 << l2pu) - 1;
        int npu = nl >> l2pu;
        const int on_x = !(x0 & pu_mask), on_y = !(y0 & pu_mask);
        if (!npu) npu = 1;
        const int xl = (x0 - 1) >> l2pu, yt = (y0 - 1) >> l2pu;
        if (a_bl && on_x) { const int yb = (y0 + nl) >> l2pu; a_bl = any_intra_pu(p, xl, yb, 0, imin_(npu, p->min_pu_height - yb)); }
        if (a_l && on_x)  { const int yl = y0 >> l2pu;        a_l  = any_intra_pu(p, xl, yl, 0, imin_(npu, p->min_pu_height - yl)); }
        if (a_ul)         a_ul = p->tab_mvf[xl + yt * p->min_pu_width].is_intra;
        if (a_u && on_y)  { const int xt = x0 >> l2pu;        a_u  = any_intra_pu(p, xt, yt, 1, imin_(npu, p->min_pu_width - xt)); }
        if (a_ur && on_y) { const int xr = (x0 + nl) >> l2pu; a_ur = any_intra_pu(p, xr, yt, 1, imin_(npu, p->min_pu_width - xr)); }
        fill(L, 128, 2 * NMAX); fill(T, 128, 2 * NMAX);
        /* The reference's loop above starts at index 0 (:157-160): its corner left[-1] / top[-1] is NOT set here.  Every path writes it
         * before use — except one: the corner's unit is intra but the corner is not a candidate (cand_up_left 0 because it lies
         * across a slice or tile edge).  The substitution walks (:187-199, :214-221) ask IS_INTRA, not the candidate flag, leave such
         * a corner alone and then copy it into the whole left column: the reference predicts from a sample it never wrote (whatever
         * its stack held; found by tools/hevc_stream_sweep.py, seed 7 case 32: a 16x16 block at a tile edge predicted from 0).
         * Here the corner starts like every other neighbour the reference initialises: 128. */
        L[-1] = T[-1] = 128;
    } else {
        fill(Lb, 0, 2 * NMAX + 1); fill(Tb, 0, 2 * NMAX + 1);
    }

    /* gather (:152-170); rows / columns past the picture repeat the last one inside */
    if (a_bl) for (int i = n; i < 2 * n; i++) L[i] = PIX(-1, imin_(i, n + n_bl - 1));
    if (a_l)  for (int i = 0; i < n; i++) L[i] = PIX(-1, i);
    if (a_ul) L[-1] = T[-1] = PIX(-1, -1);
    if (a_u)  for (int i = 0; i < n; i++) T[i] = PIX(i, -1);
    if (a_ur) for (int i = n; i < 2 * n; i++) T[i] = PIX(imin_(i, n + n_ur - 1), -1);

    if (cip && (a_bl || a_l || a_ul || a_u || a_ur)) {              /* substitution of non-intra neighbours (:172-232) */
        const int room_x = (p->width - x0) >> hs, room_y = (p->height - y0) >> vs;
        const int ex = a_ur ? 2 * n : n, ey = a_bl ? 2 * n : n;
        const int lim_x = x0 + (ex << hs) < p->width ? ex : room_x;
        const int lim_y = y0 + (ey << vs) < p->height ? ey : room_y;
        int j;
        if (a_bl || a_l || a_ul) {
            /* lowest intra sample of the left column, the corner included */
            j = n + (a_bl ? n_bl : 0) - 1;
            while (j > -1 && !intra_at(&w, -1, j)) j--;
            if (!intra_at(&w, -1, j)) {
                /* none: take the first intra sample of the top row and spread it leftwards over the non-intra ones */
                j = 0;
                while (j < lim_x && !intra_at(&w, j, -1)) j++;
                for (int i = j; i > -1; i--) if (!intra_at(&w, i - 1, -1)) T[i - 1] = T[i];
                L[-1] = T[-1];
                j = 0;
            }
        } else {
            j = 0;
            while (j < lim_x && !intra_at(&w, j, -1)) j++;
            if (j > 0) {
                if (x0 > 0) {
                    for (int i = j; i > -1; i--) if (!intra_at(&w, i - 1, -1)) T[i - 1] = T[i];
                } else {                                            /* no column left of the picture to ask about */
                    for (int i = j; i > 0; i--) if (!intra_at(&w, i - 1, -1)) T[i - 1] = T[i];
                    T[-1] = T[0];
                }
            }
            L[-1] = T[-1];
            j = 0;
        }
        if (a_bl || a_l)                                            /* downwards from there */
            for (int i = j; i < lim_y; i++) if (!intra_at(&w, -1, i)) L[i] = L[i - 1];
        if (!a_l)  fill(L, L[-1], n);
        if (!a_bl) fill(L + n, L[n - 1], n);
        if (x0 != 0 && y0 != 0) {                                   /* upwards, into the corner */
            for (int i = lim_y - 1; i > -1; i--) if (!intra_at(&w, -1, i - 1)) L[i - 1] = L[i];
        } else if (x0 == 0) {
            for (int i = lim_y - 1; i > -1; i--) L[i - 1] = L[i];
        } else {
            for (int i = lim_y - 1; i > 0; i--) if (!intra_at(&w, -1, i - 1)) L[i - 1] = L[i];
        }
        T[-1] = L[-1];
        if (y0 != 0)                                                /* and rightwards along the top row */
            for (int i = 0; i < lim_x; i++) if (!intra_at(&w, i, -1)) T[i] = T[i - 1];
    }

    /* unavailable neighbours take the nearest available sample (:233-270) */
    if (!a_bl) {
        if (a_l) fill(L + n, L[n - 1], n);
        else if (a_ul) { fill(L, L[-1], 2 * n); a_l = 1; }
        else if (a_u)  { L[-1] = T[0]; fill(L, L[-1], 2 * n); a_ul = a_l = 1; }
        else if (a_ur) { fill(T, T[n], n); L[-1] = T[n]; fill(L, L[-1], 2 * n); a_u = a_ul = a_l = 1; }
        else           { L[-1] = 1 << (bit_depth - 1); fill(T, L[-1], 2 * n); fill(L, L[-1], 2 * n); }
    }
    if (!a_l)  fill(L, L[n], n);
    if (!a_ul) L[-1] = L[0];
    if (!a_u)  fill(T, L[-1], n);
    if (!a_ur) fill(T + n, T[n - 1], n);
    T[-1] = L[-1];

    /* smoothing of the neighbours (:272-318): luma, not DC, not 4x4, directions far enough from horizontal / vertical */
    const int mode = b->mode;
    const int *top = T, *left = L;
    if (c == 0 && mode != 1 && n != 4) {
        static const int min_dist[3] = { 7, 1, 0 };
        const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
        if (imin_(dv, dh) > min_dist[b->log2_size - 3]) {
            const int thr = 1 << (bit_depth - 5);
            const int bt = T[-1] + T[63] - 2 * T[31], bl = L[-1] + L[63] - 2 * L[31];
            if (p->strong_intra_smoothing && b->log2_size == 5 && (bt < 0 ? -bt : bt) < thr && (bl < 0 ? -bl : bl) < thr) {
                FT[-1] = T[-1]; FT[63] = T[63];                     /* bilinear between the corner and the far end */
                for (int i = 0; i < 63; i++) FT[i] = ((63 - i) * T[-1] + (i + 1) * T[63] + 32) >> 6;
                FL[-1] = L[-1]; FL[63] = L[63];
                for (int i = 0; i < 63; i++) FL[i] = ((63 - i) * L[-1] + (i + 1) * L[63] + 32) >> 6;
            } else {
                FL[2 * n - 1] = L[2 * n - 1]; FT[2 * n - 1] = T[2 * n - 1];
                for (int i = 2 * n - 2; i >= 0; i--) FL[i] = (L[i + 1] + 2 * L[i] + L[i - 1] + 2) >> 2;
                FT[-1] = FL[-1] = (L[0] + 2 * L[-1] + T[0] + 2) >> 2;
                for (int i = 2 * n - 2; i >= 0; i--) FT[i] = (T[i + 1] + 2 * T[i] + T[i - 1] + 2) >> 2;
            }
            top = FT; left = FL;
        }
    }

    /* the prediction proper, through the oracle's HEVCPredContext (oracle_hevcdsp.c) */
    uint16_t t16[2 * NMAX + 1], l16[2 * NMAX + 1];
    uint8_t t8[2 * NMAX + 1], l8[2 * NMAX + 1];
    for (int i = -1; i < 2 * n; i++) { t16[i + 1] = (uint16_t)top[i]; l16[i + 1] = (uint16_t)left[i]; t8[i + 1] = (uint8_t)top[i]; l8[i + 1] = (uint8_t)left[i]; }
    const uint8_t *tp = px == 2 ? (const uint8_t *)(t16 + 1) : t8 + 1, *lp = px == 2 ? (const uint8_t *)(l16 + 1) : l8 + 1;
    HEVCPredContext h;
    oracle_hevc_pred_init(&h, bit_depth);
    if (mode == 0) h.pred_planar[b->log2_size - 2](org, tp, lp, st);
    else if (mode == 1) h.pred_dc(org, tp, lp, st, b->log2_size, c);
    else h.pred_angular[b->log2_size - 2](org, tp, lp, st, c, mode);
#undef PIX
#undef ZS
}

void oracle_hevc_intra_pred_blocks(const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks, int n, int bit_depth)
{
    for (int i = 0; i < n; i++)
        oracle_hevc_intra_pred_block(pics + blocks[i].pic, blocks + i, bit_depth);
}
