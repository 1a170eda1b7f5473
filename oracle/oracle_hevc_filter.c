/*
 * oracle_hevc_filter.c — CPU restatement (TEST INFRASTRUCTURE ONLY) of the deblocking half of the reference's
 * ff_hevc_hls_filter for a whole picture: deblocking_filter_CTB (libavcodec/hevc_filter.c:337-505) with tctable /
 * betatable (:35-45), chroma_tc (:47-72), TC_CALC (:332-335), get_qPy (:166-172), get_pcm (:316-330), written as
 * "all vertical edges of the picture, then all horizontal edges" (the formulation the device driver uses), edge
 * parameters taken from the CTB that contains the edge sample.  The edge filters themselves are the oracle's
 * hevc_{v,h}_loop_filter_{luma,chroma} (oracle_hevcdsp.c, pinned to hevcdsp_template.c:1264-1422).
 * Pinned against the reference's own hevc_filter.c compiled in place (oracle/_ref/libhevcfilterref.so,
 * tests/test_oracle_hevc_filter.py) and against tests/golden/hevc_filter_ref_sha1.json where the reference is absent.
 * The pointers of mi355_hevc_lf_picture are HOST pointers here.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "oracle.h"
#include "../include/mi355_hevc_batch.h"

static const uint8_t tctable[54] = {     /* Table 8-12 of the standard, tC' (hevc_filter.c:35-39) */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static const uint8_t betatable[52] = {   /* beta' (hevc_filter.c:41-45) */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28,
    30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int qpy(const mi355_hevc_lf_picture *p, int x, int y)              /* get_qPy */
{
    return p->qp_y_tab[(x >> p->log2_min_cb_size) + (y >> p->log2_min_cb_size) * p->min_cb_width];
}
static int pcm(const mi355_hevc_lf_picture *p, int x, int y)              /* get_pcm */
{
    if (x < 0 || y < 0) return 2;
    const int xp = x >> p->log2_min_pu_size, yp = y >> p->log2_min_pu_size;
    if (xp >= p->min_pu_width || yp >= p->min_pu_height) return 2;
    return p->is_pcm[yp * p->min_pu_width + xp];
}
static const mi355_hevc_db_params *dbp(const mi355_hevc_lf_picture *p, int x, int y)   /* s->deblock[ctb containing (x, y)] */
{
    return &p->deblock[(x >> p->log2_ctb_size) + (y >> p->log2_ctb_size) * p->ctb_width];
}
static int tc_calc(int qp, int bs, int tc_offset)                         /* TC_CALC: DEFAULT_INTRA_TC_OFFSET 2, MAX_QP 51 */
{
    return tctable[clipi(qp + 2 * (bs - 1) + (tc_offset >> 1 << 1), 0, 53)];
}
static int chroma_tc(const mi355_hevc_lf_picture *p, int qp_y, int c_idx, int tc_offset)
{
    static const int qp_c[] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
    const int qp_i = clipi(qp_y + (c_idx == 1 ? p->cb_qp_offset : p->cr_qp_offset), 0, 57);
    const int qp = qp_i < 30 ? qp_i : (qp_i > 43 ? qp_i - 6 : qp_c[qp_i - 30]);
    return tctable[clipi(qp + 2 + tc_offset, 0, 53)];
}

void oracle_hevc_deblock_picture(const mi355_hevc_lf_picture *p, int bit_depth)
{
    HEVCDSPContext dsp;
    oracle_hevc_dsp_init(&dsp, bit_depth);
    const int ps = bit_depth > 8;
    const int W = p->width, H = p->height;
    /* ---- vertical edges: luma (:372-401), 8-sample segments on the 8x8 grid, x >= 8 */
    for (int y = 0; y < H; y += 8)
        for (int x = 8; x < W; x += 8) {
            const int bs0 = p->vertical_bs[(x >> 3) + (y >> 2) * p->bs_width], bs1 = p->vertical_bs[(x >> 3) + ((y + 4) >> 2) * p->bs_width];
            if (!bs0 && !bs1) continue;
            const mi355_hevc_db_params *d = dbp(p, x, y);
            const int qp = (qpy(p, x - 1, y) + qpy(p, x, y) + 1) >> 1;
            const int beta = betatable[clipi(qp + d->beta_offset, 0, 51)];
            int tc[2] = { bs0 ? tc_calc(qp, bs0, d->tc_offset) : 0, bs1 ? tc_calc(qp, bs1, d->tc_offset) : 0 };
            uint8_t no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
            if (p->pcmf) {
                no_p[0] = (uint8_t)pcm(p, x - 1, y); no_p[1] = (uint8_t)pcm(p, x - 1, y + 4);
                no_q[0] = (uint8_t)pcm(p, x, y);     no_q[1] = (uint8_t)pcm(p, x, y + 4);
            }
            dsp.hevc_v_loop_filter_luma(p->data[0] + (ptrdiff_t)y * p->linesize[0] + (x << ps), p->linesize[0], beta, tc, no_p, no_q);
        }
    /* ---- vertical edges: chroma (:403-432), 16-luma-sample grid, bS 2 only */
    for (int c = 1; c <= 2; c++)
        for (int y = 0; y < H; y += 16)
            for (int x = 16; x < W; x += 16) {
                const int bs0 = p->vertical_bs[(x >> 3) + (y >> 2) * p->bs_width], bs1 = p->vertical_bs[(x >> 3) + ((y + 8) >> 2) * p->bs_width];
                if (bs0 != 2 && bs1 != 2) continue;
                const mi355_hevc_db_params *d = dbp(p, x, y);
                const int qp0 = (qpy(p, x - 1, y) + qpy(p, x, y) + 1) >> 1, qp1 = (qpy(p, x - 1, y + 8) + qpy(p, x, y + 8) + 1) >> 1;
                int tc[2] = { bs0 == 2 ? chroma_tc(p, qp0, c, d->tc_offset) : 0, bs1 == 2 ? chroma_tc(p, qp1, c, d->tc_offset) : 0 };
                uint8_t no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
                if (p->pcmf) {
                    no_p[0] = (uint8_t)pcm(p, x - 1, y); no_p[1] = (uint8_t)pcm(p, x - 1, y + 8);
                    no_q[0] = (uint8_t)pcm(p, x, y);     no_q[1] = (uint8_t)pcm(p, x, y + 8);
                }
                dsp.hevc_v_loop_filter_chroma(p->data[c] + (ptrdiff_t)(y / 2) * p->linesize[c] + ((x / 2) << ps), p->linesize[c], tc, no_p, no_q);
            }
    /* ---- horizontal edges: luma (:434-467), y >= 8; parameters of the CTB containing (x, y) */
    for (int y = 8; y < H; y += 8)
        for (int x = 0; x < W; x += 8) {
            const int bs0 = p->horizontal_bs[(x + y * p->bs_width) >> 2], bs1 = p->horizontal_bs[(x + 4 + y * p->bs_width) >> 2];
            if (!bs0 && !bs1) continue;
            const mi355_hevc_db_params *d = dbp(p, x, y);
            const int qp = (qpy(p, x, y - 1) + qpy(p, x, y) + 1) >> 1;
            const int beta = betatable[clipi(qp + d->beta_offset, 0, 51)];
            int tc[2] = { bs0 ? tc_calc(qp, bs0, d->tc_offset) : 0, bs1 ? tc_calc(qp, bs1, d->tc_offset) : 0 };
            uint8_t no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
            if (p->pcmf) {
                no_p[0] = (uint8_t)pcm(p, x, y - 1); no_p[1] = (uint8_t)pcm(p, x + 4, y - 1);
                no_q[0] = (uint8_t)pcm(p, x, y);     no_q[1] = (uint8_t)pcm(p, x + 4, y);
            }
            dsp.hevc_h_loop_filter_luma(p->data[0] + (ptrdiff_t)y * p->linesize[0] + (x << ps), p->linesize[0], beta, tc, no_p, no_q);
        }
    /* ---- horizontal edges: chroma (:469-504): the reference pairs the 8-luma-sample halves at x = 8 (mod 16) and
     * gives each half the tc of the CTB it lies in; a half outside the picture (x < 0 or x >= width) has bS 0 */
    for (int c = 1; c <= 2; c++)
        for (int y = 16; y < H; y += 16)
            for (int x = -8; x < W; x += 16) {
                const int bs0 = x < 0 ? 0 : p->horizontal_bs[(x + y * p->bs_width) >> 2];
                const int bs1 = x + 8 >= W ? 0 : p->horizontal_bs[(x + 8 + y * p->bs_width) >> 2];
                if (bs0 != 2 && bs1 != 2) continue;
                const int qp0 = bs0 == 2 ? (qpy(p, x, y - 1) + qpy(p, x, y) + 1) >> 1 : 0;
                const int qp1 = bs1 == 2 ? (qpy(p, x + 8, y - 1) + qpy(p, x + 8, y) + 1) >> 1 : 0;
                int tc[2] = { bs0 == 2 ? chroma_tc(p, qp0, c, dbp(p, x, y)->tc_offset) : 0, bs1 == 2 ? chroma_tc(p, qp1, c, dbp(p, x + 8, y)->tc_offset) : 0 };
                uint8_t no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
                if (p->pcmf) {
                    no_p[0] = (uint8_t)pcm(p, x, y - 1); no_p[1] = (uint8_t)pcm(p, x + 8, y - 1);
                    no_q[0] = (uint8_t)pcm(p, x, y);     no_q[1] = (uint8_t)pcm(p, x + 8, y);
                }
                dsp.hevc_h_loop_filter_chroma(p->data[c] + (ptrdiff_t)(y / 2) * p->linesize[c] + ((x / 2) * (1 << ps)), p->linesize[c], tc, no_p, no_q);
            }
}

/* ---- boundary strengths: boundary_strength (hevc_filter.c:507-583) per marked 4-sample cell side; see
 * include/mi355_hevc_batch.h for what the marks mean.  Pinned against ff_hevc_deblocking_boundary_strengths itself,
 * called for every block of a random tiling (tests/test_oracle_hevc_filter.py). */
static int far4(const int16_t a[2], const int16_t b[2]) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }
static int bs_pair(const mi355_hevc_bs_picture *p, const mi355_hevc_mvfield *c, int c_cbf, const mi355_hevc_mvfield *n, int n_cbf, int tu_border)
{
    const int mvs = c->pred_flag[0] + c->pred_flag[1];
    if (tu_border) {
        if (c->is_intra || n->is_intra) return 2;
        if (c_cbf || n_cbf) return 1;
    }
    if (mvs != n->pred_flag[0] + n->pred_flag[1]) return 1;
    if (mvs == 2) {
        const int c0 = p->ref_poc[0][c->ref_idx[0]], c1 = p->ref_poc[1][c->ref_idx[1]];
        const int n0 = p->ref_poc[0][n->ref_idx[0]], n1 = p->ref_poc[1][n->ref_idx[1]];
        if (c0 == n0 && c0 == c1 && n0 == n1)
            return (far4(n->mv[0], c->mv[0]) || far4(n->mv[1], c->mv[1])) && (far4(n->mv[1], c->mv[0]) || far4(n->mv[0], c->mv[1]));
        if (n0 == c0 && n1 == c1) return far4(n->mv[0], c->mv[0]) || far4(n->mv[1], c->mv[1]);
        if (n1 == c0 && n0 == c1) return far4(n->mv[1], c->mv[0]) || far4(n->mv[0], c->mv[1]);
        return 1;
    }
    {   /* one vector each */
        const int lc = c->pred_flag[0] ? 0 : 1, ln = n->pred_flag[0] ? 0 : 1;
        if (p->ref_poc[lc][c->ref_idx[lc]] != p->ref_poc[ln][n->ref_idx[ln]]) return 1;
        return far4(c->mv[lc], n->mv[ln]);
    }
}
void oracle_hevc_boundary_strengths(const mi355_hevc_bs_picture *p)
{
    const int cw = p->width >> 2;
    for (int y = 0; y < p->height; y += 4)
        for (int x = 0; x < p->width; x += 4) {
            const int fl = p->edge_flags[(y >> 2) * cw + (x >> 2)];
            const mi355_hevc_mvfield *c = &p->tab_mvf[(y >> p->log2_min_pu_size) * p->min_pu_width + (x >> p->log2_min_pu_size)];
            const int c_cbf = p->cbf_luma[(y >> p->log2_min_tb_size) * p->min_tb_width + (x >> p->log2_min_tb_size)];
            if (!(x & 7)) {
                int bs = 0;
                if (x > 0 && (fl & (MI355_HEVC_EDGE_L_BLOCK | MI355_HEVC_EDGE_L_INNER))) {
                    const mi355_hevc_mvfield *n = &p->tab_mvf[(y >> p->log2_min_pu_size) * p->min_pu_width + ((x - 1) >> p->log2_min_pu_size)];
                    const int n_cbf = p->cbf_luma[(y >> p->log2_min_tb_size) * p->min_tb_width + ((x - 1) >> p->log2_min_tb_size)];
                    bs = bs_pair(p, c, c_cbf, n, n_cbf, (fl & MI355_HEVC_EDGE_L_BLOCK) != 0);
                }
                p->vertical_bs[(x >> 3) + (y >> 2) * p->bs_width] = (uint8_t)bs;
            }
            if (!(y & 7)) {
                int bs = 0;
                if (y > 0 && (fl & (MI355_HEVC_EDGE_T_BLOCK | MI355_HEVC_EDGE_T_INNER))) {
                    const mi355_hevc_mvfield *n = &p->tab_mvf[((y - 1) >> p->log2_min_pu_size) * p->min_pu_width + (x >> p->log2_min_pu_size)];
                    const int n_cbf = p->cbf_luma[((y - 1) >> p->log2_min_tb_size) * p->min_tb_width + (x >> p->log2_min_tb_size)];
                    bs = bs_pair(p, c, c_cbf, n, n_cbf, (fl & MI355_HEVC_EDGE_T_BLOCK) != 0);
                }
                p->horizontal_bs[(x + y * p->bs_width) >> 2] = (uint8_t)bs;
            }
        }
}
