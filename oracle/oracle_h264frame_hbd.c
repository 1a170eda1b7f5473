/*
 * oracle_h264frame_hbd.c — the frame-level checker of oracle_h264frame.c for 9 / 10-bit pictures, 4:2:0 and (round 6) 4:2:2: the same per-macroblock
 * reconstruction and loop-filter drivers on 16-bit samples and 32-bit coefficients (`dctcoef` is int32 above 8 bits, h264dec.h), calling
 * THE REFERENCE'S OWN tables at that bit depth — ff_h264dsp_init(c, bd, 1), ff_h264qpel_init, ff_h264chroma_init, ff_h264_pred_init from
 * oracle/_ref/libref.so (the reference's C files compiled where they lie; libavcodec/h264dsp.c:37-47,57-137, h264qpel.c:37-89), bound
 * with oracle_h264frame_hbd_bind().  TEST INFRASTRUCTURE ONLY; there is no restated arithmetic above 8 bits, so without libref.so this
 * checker does not exist (the tests that use it skip).
 *
 * What differs from the 8-bit driver, and where the reference does it:
 *   - samples are two bytes: every column offset doubles, strides stay bytes (the tables take byte strides, h264dsp_template.c);
 *   - motion compensation goes through the qpel / chroma tables with ONE stride for source and destination, as mc_dir_part does
 *     (h264_mb.c:204-318): the clamped window (emulated_edge_mc's result) is built with the picture's stride;
 *   - the loop filter's table indices take the 8-bit QP: index_a = qp + a with a = 52 + slice_alpha_c0_offset - qp_bd_offset
 *     (h264_loopfilter.c:104-236 with h264_slice.c's qp_bd_offset); alpha, beta and tc0 are scaled to the depth inside the tables' functions;
 *   - a macroblock's chroma QPs come from its record (mi355_h264_mb.qpc, what the second kernel set reads).
 * With chroma_format_idc 2 (oracle_h264frame_hbd_bind_cf): the tables are initialised for it, which makes h264_chroma_dc_dequant_idct the 2x4 form
 * (h264idct_template.c:275-310), h264_idct_add8 the eight-blocks-a-plane form (:216-238), h264_h_loop_filter_chroma* the sixteen-line forms
 * (h264dsp.c:57-137) and pred8x8[] the 8x16 predictors (h264pred.c:448-507); the drivers follow the CHROMA422 branches: chroma blocks keep the luma's
 * height and vertical vector resolution (mc_dir_part h264_mb.c:284-315: ysh = 2, (my << 1) & 7), weights run over h lines, horizontal chroma edges lie
 * at chroma rows 0, 4, 8, 12 with the strengths of luma edges 0..3 — also where an 8x8 transform leaves the luma edge out (h264_loopfilter.c:633, :693-700).
 * A chroma block's non_zero_count_cache entry is read off its coefficients (any AC level), as the second kernel set does (include/mi355_h264_frame.h).
 * Restated drivers: hl_decode_mb (h264_mb_template.c:41-257), hl_motion (h264_mc_template.c:64-163), mc_part_* (h264_mb.c:320-471),
 * hl_decode_mb_predict_luma / _idct_luma (h264_mb.c:612-795), ff_h264_filter_mb (h264_loopfilter.c:716-847).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/mi355_abi.h"
#include "../include/mi355_h264_frame.h"
#include "oracle.h"

#define PX 2                         /* bytes per sample */
#define MCS 64                       /* pitch of the motion code's private tiles: a 21-sample window row and a 16-sample block row fit */
typedef uint16_t pixel;
typedef int32_t dctcoef;

static H264DSPContext dsp;
static H264QpelContext qpel;
static H264ChromaContext chroma;
static H264PredContext pred;
static int bit_depth;                /* 0: not bound */
static int cfi = 1;                  /* chroma_format_idc the tables were bound for: 1 or 2 */
#define CH (cfi == 2 ? 16 : 8)       /* chroma rows of a macroblock */
#define NCB (cfi == 2 ? 8 : 4)       /* 4x4 blocks of a chroma plane */

/* fills the four tables with the reference's functions at `bd` bits (9 or 10), 4:2:0 */
int oracle_h264frame_hbd_bind_cf(void (*dsp_init)(H264DSPContext *, int, int), void (*qpel_init)(H264QpelContext *, int),
                                 void (*chroma_init)(H264ChromaContext *, int), void (*pred_init)(H264PredContext *, int, int, int), int bd, int chroma_format_idc)
{
    if (!dsp_init || !qpel_init || !chroma_init || !pred_init || bd < 9 || bd > 10 || chroma_format_idc < 1 || chroma_format_idc > 2) return -1;
    memset(&dsp, 0, sizeof(dsp)); memset(&qpel, 0, sizeof(qpel)); memset(&chroma, 0, sizeof(chroma)); memset(&pred, 0, sizeof(pred));
    dsp_init(&dsp, bd, chroma_format_idc);
    qpel_init(&qpel, bd);
    chroma_init(&chroma, bd);
    pred_init(&pred, MI355_AV_CODEC_ID_H264, bd, chroma_format_idc);
    bit_depth = bd;
    cfi = chroma_format_idc;
    return 0;
}
int oracle_h264frame_hbd_bind(void (*dsp_init)(H264DSPContext *, int, int), void (*qpel_init)(H264QpelContext *, int),
                              void (*chroma_init)(H264ChromaContext *, int), void (*pred_init)(H264PredContext *, int, int, int), int bd)
{
    return oracle_h264frame_hbd_bind_cf(dsp_init, qpel_init, chroma_init, pred_init, bd, 1);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int blk_x4(int i) { return (i & 1) + 2 * ((i >> 2) & 1); }  /* block index -> column (h264dec.h scan8) */
static inline int blk_y4(int i) { return ((i >> 1) & 1) + 2 * (i >> 3); }

typedef struct Ctx {
    const mi355_h264_frame *f;
    int mb_x, mb_y, mb_xy;
    const mi355_h264_mb *m;
    const mi355_h264_slice *sl;
    dctcoef coef[16 * 48];       /* sl->mb layout: luma 0.., Cb 256.., Cr 512.. */
    uint8_t nnzc[15 * 8];
    uint8_t *emu;                /* 22 rows of MCS bytes: the clamped window (edge_emu_buffer); behind it 16 rows: the block being predicted */
    uint8_t *tmp[3];             /* the second prediction of a bi-predicted partition, picture strides (bipred_scratchpad) */
} Ctx;

/* fetch a w x h block at (x,y) of a plane with border replication into `out` (row pitch `ostride` bytes): what
 * emulated_edge_mc hands to the MC functions (h264_mb.c:239-314) */
static void fetch(uint8_t *out, int ostride, const uint8_t *plane, int stride, int pw, int ph, int x, int y, int w, int h)
{
    for (int j = 0; j < h; j++) {
        const pixel *src = (const pixel *)(plane + (size_t)clampi(y + j, 0, ph - 1) * stride);
        pixel *dst = (pixel *)(out + (size_t)j * ostride);
        for (int i = 0; i < w; i++) dst[i] = src[clampi(x + i, 0, pw - 1)];
    }
}

/* mc_dir_part, h264_mb.c:204-318 (chroma_idc == 1, frame MB).  (bx,by) is the partition origin inside the MB in luma samples,
 * w x h its luma size, n the 4x4 block index whose MV is used.  dy / dcb / dcr: destination of the partition, strides dys / dcs bytes;
 * the reference's functions take ONE stride for source and destination, so the window is staged with the destination's */
static void mc_dir_part(Ctx *c, int list, int n_raster, int refn, int bx, int by, int w, int h,
                        uint8_t *dy, int dys, uint8_t *dcb, uint8_t *dcr, int dcs, int avg)
{
    const mi355_h264_frame *f = c->f;
    const int16_t *mv = f->mv[list] + ((size_t)c->mb_xy * 16 + n_raster) * 2;
    const int slot = c->sl->ref_slot[list][refn];
    const int pw = 16 * f->mb_width, ph = 16 * f->mb_height;
    const int mx = mv[0] + (c->mb_x * 16 + bx) * 4;
    const int my = mv[1] + (c->mb_y * 16 + by) * 4;
    const int myc = my + (f->field_picture ? f->mb[c->mb_xy].u.inter.chroma_dy[list][(bx >> 3) + 2 * (by >> 3)] : 0);
    /* window and destination block in private tiles of ONE pitch (MCS bytes: the tables' functions take one stride for both, as mc_dir_part's calls do with
     * sl->mb_linesize; a picture one macroblock wide has rows shorter than a window) */
    uint8_t *win = c->emu, *td = c->emu + 22 * MCS;
    fetch(win, MCS, f->ref[slot][0], f->dst_stride[0], pw, ph, (mx >> 2) - 2, (my >> 2) - 2, w + 5, h + 5);
    for (int j = 0; j < h; j++) memcpy(td + j * MCS, dy + (size_t)j * dys, (size_t)w * PX);
    {
        /* the reference issues square calls (16, 8, 4 wide) side by side / stacked: qpix_op[luma_xy], table 0 = 16 wide */
        const int s = w < h ? w : h, tab = s == 16 ? 0 : (s == 8 ? 1 : 2);
        const qpel_mc_func *op = avg ? qpel.avg_h264_qpel_pixels_tab[tab] : qpel.put_h264_qpel_pixels_tab[tab];
        for (int oy = 0; oy < h; oy += s)
            for (int ox = 0; ox < w; ox += s)
                op[(mx & 3) + 4 * (my & 3)](td + ox * PX + oy * MCS, win + 2 * MCS + (2 + ox) * PX + oy * MCS, MCS);
    }
    for (int j = 0; j < h; j++) memcpy(dy + (size_t)j * dys, td + j * MCS, (size_t)w * PX);
    for (int p = 1; p < 3; p++) {
        uint8_t *d = p == 1 ? dcb : dcr;
        /* h264_mb.c:284-315: 4:2:2 keeps the luma's lines (ysh = 2, the fraction (my << 1) & 7) and has no field offset */
        const int v422 = cfi == 2, ysh = v422 ? 2 : 3, myv = v422 ? my : myc;
        const int cw = w >> 1, ch = v422 ? h : h >> 1, tab = cw == 8 ? 0 : (cw == 4 ? 1 : 2);
        fetch(win, MCS, f->ref[slot][p], f->dst_stride[1], pw >> 1, v422 ? ph : ph >> 1, mx >> 3, myv >> ysh, cw + 1, ch + 1);
        for (int j = 0; j < ch; j++) memcpy(td + j * MCS, d + (size_t)j * dcs, (size_t)cw * PX);
        (avg ? chroma.avg_h264_chroma_pixels_tab : chroma.put_h264_chroma_pixels_tab)[tab](td, win, MCS, ch, mx & 7, (v422 ? myv << 1 : myv) & 7);
        for (int j = 0; j < ch; j++) memcpy(d + (size_t)j * dcs, td + j * MCS, (size_t)cw * PX);
    }
}

/* mc_part (h264_mc_template.c:44-62) + mc_part_std / mc_part_weighted (h264_mb.c:320-471) */
static void mc_part(Ctx *c, int n_raster, int quadrant, int bx, int by, int w, int h, int list0, int list1,
                    uint8_t *dy, uint8_t *dcb, uint8_t *dcr)
{
    const mi355_h264_frame *f = c->f;
    const mi355_h264_slice *sl = c->sl;
    const int ys = f->recon_stride[0], cs = f->recon_stride[1];
    const int hc = cfi == 2 ? h : h >> 1, cby = cfi == 2 ? by : by >> 1;      /* chroma lines of the partition, its first chroma line */
    uint8_t *py = dy + bx * PX + by * ys, *pcb = dcb + (bx >> 1) * PX + cby * cs, *pcr = dcr + (bx >> 1) * PX + cby * cs;
    const int r0 = c->m->ref_idx[0][quadrant], r1 = c->m->ref_idx[1][quadrant];
    const int weighted = (sl->use_weight == 2 && list0 && list1 && sl->implicit_weight[r0][r1] != 32) || sl->use_weight == 1;
    const int widx = w == 16 ? 0 : (w == 8 ? 1 : (w == 4 ? 2 : 3)), cwidx = widx + 1;
    if (!weighted) {
        int avg = 0;
        if (list0) { mc_dir_part(c, 0, n_raster, r0, bx, by, w, h, py, ys, pcb, pcr, cs, 0); avg = 1; }
        if (list1) mc_dir_part(c, 1, n_raster, r1, bx, by, w, h, py, ys, pcb, pcr, cs, avg);
        return;
    }
    if (list0 && list1) {
        /* the second prediction into the scratch planes at the same strides (the biweight functions take one stride for both operands) */
        uint8_t *ty = c->tmp[0], *tcb = c->tmp[1], *tcr = c->tmp[2];
        mc_dir_part(c, 0, n_raster, r0, bx, by, w, h, py, ys, pcb, pcr, cs, 0);
        mc_dir_part(c, 1, n_raster, r1, bx, by, w, h, ty, ys, tcb, tcr, cs, 0);
        if (sl->use_weight == 2) {
            int w0 = sl->implicit_weight[r0][r1], w1 = 64 - w0;
            dsp.biweight_h264_pixels_tab[widx](py, ty, ys, h, 5, w0, w1, 0);
            dsp.biweight_h264_pixels_tab[cwidx](pcb, tcb, cs, hc, 5, w0, w1, 0);
            dsp.biweight_h264_pixels_tab[cwidx](pcr, tcr, cs, hc, 5, w0, w1, 0);
        } else {
            dsp.biweight_h264_pixels_tab[widx](py, ty, ys, h, sl->luma_log2_weight_denom,
                                               sl->luma_weight[r0][0][0], sl->luma_weight[r1][1][0],
                                               sl->luma_weight[r0][0][1] + sl->luma_weight[r1][1][1]);
            for (int p = 0; p < 2; p++)
                dsp.biweight_h264_pixels_tab[cwidx](p ? pcr : pcb, p ? tcr : tcb, cs, hc, sl->chroma_log2_weight_denom,
                                                    sl->chroma_weight[r0][0][p][0], sl->chroma_weight[r1][1][p][0],
                                                    sl->chroma_weight[r0][0][p][1] + sl->chroma_weight[r1][1][p][1]);
        }
    } else {
        int list = list1 ? 1 : 0, refn = list ? r1 : r0;
        mc_dir_part(c, list, n_raster, refn, bx, by, w, h, py, ys, pcb, pcr, cs, 0);
        dsp.weight_h264_pixels_tab[widx](py, ys, h, sl->luma_log2_weight_denom,
                                         sl->luma_weight[refn][list][0], sl->luma_weight[refn][list][1]);
        if (sl->use_weight_chroma) {
            dsp.weight_h264_pixels_tab[cwidx](pcb, cs, hc, sl->chroma_log2_weight_denom,
                                              sl->chroma_weight[refn][list][0][0], sl->chroma_weight[refn][list][0][1]);
            dsp.weight_h264_pixels_tab[cwidx](pcr, cs, hc, sl->chroma_log2_weight_denom,
                                              sl->chroma_weight[refn][list][1][0], sl->chroma_weight[refn][list][1][1]);
        }
    }
}

/* hl_motion, h264_mc_template.c:64-163 */
static void hl_motion(Ctx *c, uint8_t *dy, uint8_t *dcb, uint8_t *dcr)
{
    const uint32_t t = c->m->mb_type;
#define DIR(part, list) ((t >> (12 + (part) + 2 * (list))) & 1)
    if (t & MI355_MB_16x16) {
        mc_part(c, 0, 0, 0, 0, 16, 16, DIR(0, 0), DIR(0, 1), dy, dcb, dcr);
    } else if (t & MI355_MB_16x8) {
        mc_part(c, 0, 0, 0, 0, 16, 8, DIR(0, 0), DIR(0, 1), dy, dcb, dcr);
        mc_part(c, 8, 2, 0, 8, 16, 8, DIR(1, 0), DIR(1, 1), dy, dcb, dcr);
    } else if (t & MI355_MB_8x16) {
        mc_part(c, 0, 0, 0, 0, 8, 16, DIR(0, 0), DIR(0, 1), dy, dcb, dcr);
        mc_part(c, 2, 1, 8, 0, 8, 16, DIR(1, 0), DIR(1, 1), dy, dcb, dcr);
    } else {
        for (int i = 0; i < 4; i++) {
            const int st = c->m->sub_mb_type[i], shape = st & 3;
            const int l0 = (st & MI355_SUB_L0) != 0, l1 = (st & MI355_SUB_L1) != 0;
            const int x = (i & 1) * 8, y = (i >> 1) * 8, n = (x >> 2) + 4 * (y >> 2);
            if (shape == MI355_SUB_8x8) mc_part(c, n, i, x, y, 8, 8, l0, l1, dy, dcb, dcr);
            else if (shape == MI355_SUB_8x4) {
                mc_part(c, n, i, x, y, 8, 4, l0, l1, dy, dcb, dcr);
                mc_part(c, n + 4, i, x, y + 4, 8, 4, l0, l1, dy, dcb, dcr);
            } else if (shape == MI355_SUB_4x8) {
                mc_part(c, n, i, x, y, 4, 8, l0, l1, dy, dcb, dcr);
                mc_part(c, n + 1, i, x + 4, y, 4, 8, l0, l1, dy, dcb, dcr);
            } else
                for (int j = 0; j < 4; j++)
                    mc_part(c, n + (j & 1) + 4 * (j >> 1), i, x + 4 * (j & 1), y + 4 * (j >> 1), 4, 4, l0, l1, dy, dcb, dcr);
        }
    }
#undef DIR
}

static void block_offsets(int *off, int ys, int cs)
{
    for (int i = 0; i < 16; i++) {
        off[i] = 4 * blk_x4(i) * PX + 4 * blk_y4(i) * ys;
        off[16 + i] = off[32 + i] = 4 * blk_x4(i) * PX + 4 * blk_y4(i) * cs;
    }
}

/* one macroblock of hl_decode_mb (h264_mb_template.c:41-257) */
static void recon_mb(Ctx *c)
{
    const mi355_h264_frame *f = c->f;
    const mi355_h264_mb *m = c->m;
    const int ys = f->recon_stride[0], cs = f->recon_stride[1];
    uint8_t *dy = f->recon[0] + (size_t)c->mb_y * 16 * ys + c->mb_x * 16 * PX;
    uint8_t *dcb = f->recon[1] + (size_t)c->mb_y * CH * cs + c->mb_x * 8 * PX;
    uint8_t *dcr = f->recon[2] + (size_t)c->mb_y * CH * cs + c->mb_x * 8 * PX;
    const int ncc = 16 * NCB;                                 /* coefficients of a chroma plane: 64 / 128 */
    const dctcoef *src = (const dctcoef *)f->coef + (size_t)c->mb_xy * (256 + 2 * ncc);
    const uint32_t t = m->mb_type;
    int off[48];
    block_offsets(off, ys, cs);

    if (t & MI355_MB_INTRA_PCM) {   /* h264_mb_template.c:139-153: one sample per coefficient slot (the second kernel set's convention) */
        for (int i = 0; i < 16; i++) for (int x = 0; x < 16; x++) ((pixel *)(dy + i * ys))[x] = (pixel)src[16 * i + x];
        for (int i = 0; i < CH; i++)
            for (int x = 0; x < 8; x++) {
                ((pixel *)(dcb + i * cs))[x] = (pixel)src[256 + 8 * i + x];
                ((pixel *)(dcr + i * cs))[x] = (pixel)src[256 + 8 * CH + 8 * i + x];
            }
        return;
    }
    /* sl->mb image + nnz cache */
    memset(c->coef, 0, sizeof(c->coef));
    memcpy(c->coef, src, 256 * sizeof(dctcoef));
    memcpy(c->coef + 256, src + 256, (size_t)ncc * sizeof(dctcoef));
    memcpy(c->coef + 512, src + 256 + ncc, (size_t)ncc * sizeof(dctcoef));
    memset(c->nnzc, 0, sizeof(c->nnzc));
    for (int i = 0; i < 16; i++) c->nnzc[oracle_scan8(i)] = (m->nnz_mask >> i) & 1 ? 2 : 0;
    if (cfi == 1)
        for (int j = 0; j < 4; j++) {
            c->nnzc[oracle_scan8(16 + j)] = (m->nnz_mask >> (16 + j)) & 1 ? 2 : 0;
            c->nnzc[oracle_scan8(32 + j)] = (m->nnz_mask >> (20 + j)) & 1 ? 2 : 0;
        }
    else
        /* eight blocks a plane: block j < 4 at scan8[16 + j], block 4 + j at scan8[16 + j + 8] — where idct_add8_422 looks (h264idct_template.c:222-236:
         * nnzc[scan8[i]] for i = 16..19, nnzc[scan8[i + 4]] for i = 20..23); the entry is the count of the block's AC levels (decode_residual on the 15
         * coefficients behind the DC), read off the coefficients here */
        for (int p = 0; p < 2; p++)
            for (int j = 0; j < 8; j++) {
                int ac = 0;
                for (int k = 1; k < 16; k++) ac |= c->coef[256 * (1 + p) + 16 * j + k] != 0;
                c->nnzc[oracle_scan8(16 * (1 + p) + (j < 4 ? j : j + 4))] = ac ? 2 : 0;
            }
#define COEF(i) ((int16_t *)(c->coef + (i)))          /* the tables' prototypes say int16_t; above 8 bits their functions read dctcoef = int32 */
    if (t & MI355_MB_INTRA) {
        pred.pred8x8[m->chroma_pred_mode](dcb, cs);
        pred.pred8x8[m->chroma_pred_mode](dcr, cs);
        if (t & MI355_MB_INTRA4x4) {      /* hl_decode_mb_predict_luma, h264_mb.c:626-700 */
            if (t & MI355_MB_8x8DCT) {
                for (int i = 0; i < 16; i += 4) {
                    uint8_t *p = dy + off[i];
                    pred.pred8x8l[m->u.intra4x4_pred_mode[i]](p, (m->topleft_samples_available << i) & 0x8000,
                                                            (m->topright_samples_available << i) & 0x4000, ys);
                    if (c->nnzc[oracle_scan8(i)]) dsp.h264_idct8_add(p, COEF(i * 16), ys);
                }
            } else {
                for (int i = 0; i < 16; i++) {
                    uint8_t *p = dy + off[i];
                    const int dir = m->u.intra4x4_pred_mode[i];
                    pixel trbuf[4];
                    const uint8_t *tr = NULL;
                    if (dir == DIAG_DOWN_LEFT_PRED || dir == VERT_LEFT_PRED) {
                        if ((m->topright_samples_available << i) & 0x8000) tr = p + 4 * PX - ys;
                        else {
                            const pixel v = ((const pixel *)(p - ys))[3];
                            trbuf[0] = trbuf[1] = trbuf[2] = trbuf[3] = v;
                            tr = (const uint8_t *)trbuf;
                        }
                    }
                    pred.pred4x4[dir](p, tr, ys);
                    if (c->nnzc[oracle_scan8(i)]) dsp.h264_idct_add(p, COEF(i * 16), ys);
                }
            }
        } else {                          /* Intra16x16, h264_mb.c:701-722 */
            pred.pred16x16[m->intra16x16_pred_mode](dy, ys);
            if ((m->nnz_mask >> MI355_NNZ_LUMA_DC) & 1) {
                dctcoef dcin[16];
                for (int k = 0; k < 16; k++) dcin[k] = c->coef[mi355_luma_dc_slot(k)];
                dsp.h264_luma_dc_dequant_idct(COEF(0), (int16_t *)dcin, (int)m->dc_qmul[0]);
            }
        }
    } else {
        hl_motion(c, dy, dcb, dcr);
    }
    /* hl_decode_mb_idct_luma, h264_mb.c:726-795 */
    if (!(t & MI355_MB_INTRA4x4)) {
        if (t & MI355_MB_INTRA16x16) dsp.h264_idct_add16intra(dy, off, COEF(0), ys, c->nnzc);
        else if (m->cbp & 15) {
            if (t & MI355_MB_8x8DCT) dsp.h264_idct8_add4(dy, off, COEF(0), ys, c->nnzc);
            else dsp.h264_idct_add16(dy, off, COEF(0), ys, c->nnzc);
        }
    }
    if (m->cbp & 0x30) {                  /* h264_mb_template.c:196-247 */
        uint8_t *dest[2] = { dcb, dcr };
        if ((m->nnz_mask >> MI355_NNZ_CB_DC) & 1) dsp.h264_chroma_dc_dequant_idct(COEF(256), (int)m->dc_qmul[1]);
        if ((m->nnz_mask >> MI355_NNZ_CR_DC) & 1) dsp.h264_chroma_dc_dequant_idct(COEF(512), (int)m->dc_qmul[2]);
        dsp.h264_idct_add8(dest, off, COEF(0), cs, c->nnzc);
    }
#undef COEF
}

int oracle_h264_recon_frame_hbd(const mi355_h264_frame *f)
{
    if (!bit_depth) return -1;
    Ctx *c = (Ctx *)calloc(1, sizeof(Ctx));
    const int ys = f->recon_stride[0], cs = f->recon_stride[1];
    c->emu = (uint8_t *)calloc(1, (size_t)(22 + 16) * MCS);
    c->tmp[0] = (uint8_t *)calloc(1, (size_t)16 * ys);
    c->tmp[1] = (uint8_t *)calloc(1, (size_t)16 * cs);
    c->tmp[2] = (uint8_t *)calloc(1, (size_t)16 * cs);
    c->f = f;
    for (int y = 0; y < f->mb_height; y++)
        for (int x = 0; x < f->mb_width; x++) {
            c->mb_x = x; c->mb_y = y; c->mb_xy = x + y * f->mb_width;
            c->m = &f->mb[c->mb_xy];
            c->sl = &f->slices[c->m->slice_id];
            recon_mb(c);
        }
    free(c->emu); free(c->tmp[0]); free(c->tmp[1]); free(c->tmp[2]);
    free(c);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* loop filter                                                                 */
/* ------------------------------------------------------------------------- */
/* Tables 8-16 / 8-17 of the standard (h264_loopfilter.c:41-101): index = qp + offset clamps to 0..51 */
static const uint8_t alpha_tab[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
    32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
static const uint8_t beta_tab[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
    9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
static const int8_t tc0_tab[52][3] = {
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},
    {1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},
    {3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},{9,12,18},{10,13,20},
    {11,15,23},{13,17,25} };

typedef struct MbView {      /* what fill_filter_caches gathers for one MB (h264_slice.c:2056-2196) */
    const mi355_h264_mb *m;
    const mi355_h264_slice *sl;
    const int16_t *mv[2];    /* 16 raster entries x 2 */
} MbView;

static MbView view(const mi355_h264_frame *f, int mb_xy)
{
    MbView v;
    v.m = &f->mb[mb_xy];
    v.sl = &f->slices[v.m->slice_id];
    v.mv[0] = f->mv[0] ? f->mv[0] + (size_t)mb_xy * 32 : NULL;
    v.mv[1] = f->mv[1] ? f->mv[1] + (size_t)mb_xy * 32 : NULL;
    return v;
}
static int ref_id(const MbView *v, int list, int x4, int y4)
{
    if (v->m->mb_type & MI355_MB_INTRA) return -1;
    int r = v->m->ref_idx[list][(x4 >> 1) + 2 * (y4 >> 1)];
    return r < 0 ? -1 : v->sl->ref_slot[list][r];
}
static void mv_of(const MbView *v, int list, int x4, int y4, int out[2])
{
    if (!v->mv[list] || ref_id(v, list, x4, y4) < 0) { out[0] = out[1] = 0; return; }
    out[0] = v->mv[list][(x4 + 4 * y4) * 2];
    out[1] = v->mv[list][(x4 + 4 * y4) * 2 + 1];
}
static int mvy_limit = 4;
static int mv_far(const int a[2], const int b[2]) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= mvy_limit; }

/* check_mv, h264_loopfilter.c:442-470 */
static int check_mv(const MbView *p, int px, int py, const MbView *q, int qx, int qy, int list_count)
{
    int r0p = ref_id(p, 0, px, py), r0q = ref_id(q, 0, qx, qy), mp[2], mq[2];
    int v = r0p != r0q;
    if (!v && r0p != -1) { mv_of(p, 0, px, py, mp); mv_of(q, 0, qx, qy, mq); v = mv_far(mp, mq); }
    if (list_count == 2) {
        int r1p = ref_id(p, 1, px, py), r1q = ref_id(q, 1, qx, qy), np[2], nq[2];
        mv_of(p, 1, px, py, np); mv_of(q, 1, qx, qy, nq);
        if (!v) v = r1p != r1q || mv_far(np, nq);
        if (v) {
            if (r0p != r1q || r1p != r0q) return 1;
            mv_of(p, 0, px, py, mp); mv_of(q, 0, qx, qy, mq);
            return mv_far(mp, nq) || mv_far(np, mq);
        }
    }
    return v;
}

/* filter one 16-sample luma edge + the matching chroma edges: filter_mb_edge{v,h,cv,ch}, h264_loopfilter.c:104-236.  qp, qpc0, qpc1 carry
 * qp_bd_offset (the record's QPs do, above 8 bits); the tables are indexed without it (a = 52 + offset - qp_bd_offset there) */
static void filter_edge(const mi355_h264_mb *m, const int16_t bS[4], int dir, int edge, int intra_ok, int luma_on,
                        int qp, int qpc0, int qpc1, uint8_t *y, int ys, uint8_t *cb, uint8_t *cr, int cs)
{
    const int bdo = 6 * (bit_depth - 8);
    const int a = m->slice_alpha_c0_offset - bdo, b = m->slice_beta_offset - bdo;
    const int c422h = cfi == 2 && dir == 1;           /* :693-700: horizontal chroma edges of 4:2:2 at chroma rows 4 * edge, every edge */
    for (int plane = 0; plane < 3; plane++) {
        if (plane == 0 && !luma_on) continue;
        if (plane && (edge & 1) && !c422h) break;
        const int q = plane == 0 ? qp : (plane == 1 ? qpc0 : qpc1);
        const int ia = clampi(q + a, 0, 51), ib = clampi(q + b, 0, 51);
        const int alpha = alpha_tab[ia], beta = beta_tab[ib];
        if (!alpha || !beta) continue;
        uint8_t *pix = plane == 0 ? y + (dir ? 4 * edge * ys : 4 * edge * PX)
                                  : (plane == 1 ? cb : cr) + (dir ? (c422h ? 4 : 2) * edge * cs : 2 * edge * PX);
        const int st = plane ? cs : ys;
        if (bS[0] < 4 || !intra_ok) {
            int8_t tc[4];
            for (int i = 0; i < 4; i++) tc[i] = (int8_t)((bS[i] ? tc0_tab[ia][bS[i] - 1] : -1) + (plane ? 1 : 0));
            if (plane == 0) (dir ? dsp.h264_v_loop_filter_luma : dsp.h264_h_loop_filter_luma)(pix, st, alpha, beta, tc);
            else (dir ? dsp.h264_v_loop_filter_chroma : dsp.h264_h_loop_filter_chroma)(pix, st, alpha, beta, tc);
        } else {
            if (plane == 0) (dir ? dsp.h264_v_loop_filter_luma_intra : dsp.h264_h_loop_filter_luma_intra)(pix, st, alpha, beta);
            else (dir ? dsp.h264_v_loop_filter_chroma_intra : dsp.h264_h_loop_filter_chroma_intra)(pix, st, alpha, beta);
        }
    }
}

/* ff_h264_filter_mb (h264_loopfilter.c:716) for one frame MB */
static void filter_mb(const mi355_h264_frame *f, int mb_x, int mb_y)
{
    const int mb_xy = mb_x + mb_y * f->mb_width;
    MbView cur = view(f, mb_xy);
    const mi355_h264_mb *m = cur.m;
    if (m->flags & MI355_MBF_NO_DEBLOCK) return;
    const int ys = f->dst_stride[0], cs = f->dst_stride[1];
    uint8_t *y = f->dst[0] + (size_t)mb_y * 16 * ys + mb_x * 16 * PX;
    uint8_t *cb = f->dst[1] + (size_t)mb_y * CH * cs + mb_x * 8 * PX;
    uint8_t *cr = f->dst[2] + (size_t)mb_y * CH * cs + mb_x * 8 * PX;
    const int intra = (m->mb_type & MI355_MB_INTRA) != 0;
    const int dct8 = (m->mb_type & MI355_MB_8x8DCT) != 0;
    for (int dir = 0; dir < 2; dir++) {
        const int have_n = m->flags & (dir ? MI355_MBF_TOP_EDGE : MI355_MBF_LEFT_EDGE);
        for (int edge = 0; edge < 4; edge++) {
            int16_t bS[4];
            int qp, qc0, qc1;
            if (edge == 0) {
                if (!have_n) continue;
                MbView nb = view(f, dir ? mb_xy - f->mb_width : mb_xy - 1);
                if (intra || (nb.m->mb_type & MI355_MB_INTRA)) {
                    bS[0] = bS[1] = bS[2] = bS[3] = (int16_t)(dir && f->field_picture ? 3 : 4);
                } else {
                    for (int i = 0; i < 4; i++) {
                        int x4 = dir ? i : 0, y4 = dir ? 0 : i;
                        int nx = dir ? i : 3, ny = dir ? 3 : i;
                        int bi_c = (x4 & 1) + 2 * (y4 & 1) + 4 * (x4 >> 1) + 8 * (y4 >> 1);
                        int bi_n = (nx & 1) + 2 * (ny & 1) + 4 * (nx >> 1) + 8 * (ny >> 1);
                        if (((m->nnz_mask >> bi_c) | (nb.m->nnz_mask >> bi_n)) & 1) bS[i] = 2;
                        else bS[i] = (int16_t)check_mv(&cur, x4, y4, &nb, nx, ny, cur.sl->list_count);
                    }
                }
                if (!(bS[0] + bS[1] + bS[2] + bS[3])) continue;
                qp = (m->qp + nb.m->qp + 1) >> 1;
                qc0 = (m->qpc[0] + nb.m->qpc[0] + 1) >> 1;     /* both by the CURRENT slice's table in the reference (:628-629): the records' own values when one table serves the picture */
                qc1 = (m->qpc[1] + nb.m->qpc[1] + 1) >> 1;
            } else {
                if (dct8 && (edge & 1) && !(cfi == 2 && dir == 1)) continue;      /* :633: 4:2:2 still has a horizontal CHROMA edge there */
                if (intra) bS[0] = bS[1] = bS[2] = bS[3] = 3;
                else {
                    for (int i = 0; i < 4; i++) {
                        int x4 = dir ? i : edge, y4 = dir ? edge : i;
                        int nx = dir ? i : edge - 1, ny = dir ? edge - 1 : i;
                        int bi_c = (x4 & 1) + 2 * (y4 & 1) + 4 * (x4 >> 1) + 8 * (y4 >> 1);
                        int bi_n = (nx & 1) + 2 * (ny & 1) + 4 * (nx >> 1) + 8 * (ny >> 1);
                        if (((m->nnz_mask >> bi_c) | (m->nnz_mask >> bi_n)) & 1) bS[i] = 2;
                        else bS[i] = (int16_t)check_mv(&cur, x4, y4, &cur, nx, ny, cur.sl->list_count);
                    }
                    if (!(bS[0] + bS[1] + bS[2] + bS[3])) continue;
                }
                qp = m->qp; qc0 = m->qpc[0]; qc1 = m->qpc[1];
            }
            filter_edge(m, bS, dir, edge, edge == 0, !(edge && dct8 && (edge & 1)), qp, qc0, qc1, y, ys, cb, cr, cs);
        }
    }
}

int oracle_h264_deblock_frame_hbd(const mi355_h264_frame *f)
{
    if (!bit_depth) return -1;
    mvy_limit = f->field_picture ? 2 : 4;
    for (int p = 0; p < 3; p++) {
        int rows = (p ? CH : 16) * f->mb_height, w = (p ? 8 : 16) * f->mb_width * PX;
        for (int r = 0; r < rows; r++)
            memcpy(f->dst[p] + (size_t)r * f->dst_stride[p ? 1 : 0], f->recon[p] + (size_t)r * f->recon_stride[p ? 1 : 0], (size_t)w);
    }
    for (int y = 0; y < f->mb_height; y++)
        for (int x = 0; x < f->mb_width; x++)
            filter_mb(f, x, y);
    return 0;
}
