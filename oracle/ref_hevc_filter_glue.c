/* TEST INFRASTRUCTURE (oracle/): drives the REFERENCE's own deblocking_filter_CTB (libavcodec/hevc_filter.c, compiled
 * where it lies into _ref/libhevcfilterref.so together with hevcdsp.c) on a picture described by this project's
 * mi355_hevc_lf_picture, with HOST pointers: a zeroed HEVCContext gets exactly the fields hevc_filter.c:337-505 reads,
 * then ff_hevc_hls_filter() runs for every CTB in raster order (sao_enabled = 0, so it is the deblocking half only).
 * Pins oracle_hevc_filter.c (tests/test_oracle_hevc_filter.py) and generates tests/golden/hevc_filter_ref_sha1.json. */
#include <stdlib.h>
#include <string.h>
#include "libavutil/mem.h"
#include "libavutil/frame.h"
#include "libavcodec/hevcdec.h"
#include "../include/mi355_hevc_batch.h"

/* three 4-entry tables hevcdsp.c expects from libavcodec/hevcdec.c:45-47 (the decoder itself is not built here; the same
 * lines ref_glue.c carries for libref.so) */
const uint8_t ff_hevc_qpel_extra_before[4] = { 0, 3, 3, 3 };
const uint8_t ff_hevc_qpel_extra_after[4]  = { 0, 4, 4, 4 };
const uint8_t ff_hevc_qpel_extra[4]        = { 0, 7, 7, 7 };
/* symbols hevc_filter.c references on paths this driver never takes */
void ff_thread_report_progress(ThreadFrame *f, int progress, int field) { (void)f; (void)progress; (void)field; }
RefPicList *ff_hevc_get_ref_list(HEVCContext *s, HEVCFrame *ref, int x0, int y0) { (void)s; (void)x0; (void)y0; return ref->refPicList; }

int ref_hevc_deblock_picture(const mi355_hevc_lf_picture *p, int bit_depth)
{
    HEVCContext *s = av_mallocz(sizeof(*s));
    HEVCSPS *sps = av_mallocz(sizeof(*sps));
    HEVCPPS *pps = av_mallocz(sizeof(*pps));
    AVFrame *fr = av_frame_alloc();
    if (!s || !sps || !pps || !fr) return -1;
    sps->log2_ctb_size = p->log2_ctb_size;
    sps->ctb_width = p->ctb_width;
    sps->width = p->width;
    sps->height = p->height;
    sps->pixel_shift = bit_depth > 8;
    sps->log2_min_cb_size = p->log2_min_cb_size;
    sps->min_cb_width = p->min_cb_width;
    sps->log2_min_pu_size = p->log2_min_pu_size;
    sps->min_pu_width = p->min_pu_width;
    sps->min_pu_height = p->min_pu_height;
    sps->sao_enabled = 0;
    /* pcmf = (pcm_enabled && pcm.loop_filter_disable) || transquant_bypass_enable: any combination with that value */
    sps->pcm_enabled_flag = 0;
    pps->transquant_bypass_enable_flag = p->pcmf != 0;
    pps->cb_qp_offset = p->cb_qp_offset;
    pps->cr_qp_offset = p->cr_qp_offset;
    s->ps.sps = sps;
    s->ps.pps = pps;
    s->deblock = (DBParams *)p->deblock;           /* same two ints, same order (checked below) */
    s->vertical_bs = (uint8_t *)p->vertical_bs;
    s->horizontal_bs = (uint8_t *)p->horizontal_bs;
    s->bs_width = p->bs_width;
    s->qp_y_tab = (int8_t *)p->qp_y_tab;
    s->is_pcm = (uint8_t *)p->is_pcm;
    for (int i = 0; i < 3; i++) { fr->data[i] = p->data[i]; fr->linesize[i] = p->linesize[i]; }
    s->frame = fr;
    ff_hevc_dsp_init(&s->hevcdsp, bit_depth);
    if (sizeof(DBParams) != sizeof(mi355_hevc_db_params) || offsetof(DBParams, tc_offset) != offsetof(mi355_hevc_db_params, tc_offset)) return -2;
    const int ctb = 1 << p->log2_ctb_size;
    for (int y = 0; y < p->height; y += ctb)
        for (int x = 0; x < p->width; x += ctb)
            ff_hevc_hls_filter(s, x, y);
    fr->data[0] = fr->data[1] = fr->data[2] = NULL;
    av_frame_free(&fr);
    av_free(pps); av_free(sps); av_free(s);
    return 0;
}

/* ---- ff_hevc_deblocking_boundary_strengths itself, called for every block (x0, y0, log2_size) of `blocks` in order, on a
 * context that holds the picture's motion field, cbf_luma and one RefPicList pair (no slice / tile boundaries:
 * lc->boundary_flags = 0).  vertical_bs / horizontal_bs must come in zeroed (the function only writes non-zero values). */
int ref_hevc_boundary_strengths(const mi355_hevc_bs_picture *p, const int32_t *blocks, int nblocks)
{
    HEVCContext *s = av_mallocz(sizeof(*s));
    HEVCSPS *sps = av_mallocz(sizeof(*sps));
    HEVCPPS *pps = av_mallocz(sizeof(*pps));
    HEVCFrame *ref = av_mallocz(sizeof(*ref));
    RefPicList *rpl = av_mallocz(2 * sizeof(*rpl));
    if (!s || !sps || !pps || !ref || !rpl) return -1;
    if (sizeof(MvField) != sizeof(mi355_hevc_mvfield) || offsetof(MvField, ref_idx) != offsetof(mi355_hevc_mvfield, ref_idx) ||
        offsetof(MvField, pred_flag) != offsetof(mi355_hevc_mvfield, pred_flag) || offsetof(MvField, is_intra) != offsetof(mi355_hevc_mvfield, is_intra))
        return -2;
    sps->log2_min_pu_size = p->log2_min_pu_size;
    sps->log2_min_tb_size = p->log2_min_tb_size;
    sps->min_pu_width = p->min_pu_width;
    sps->min_tb_width = p->min_tb_width;
    sps->log2_ctb_size = 6;
    pps->loop_filter_across_tiles_enabled_flag = 1;
    s->sh.slice_loop_filter_across_slices_enabled_flag = 1;
    s->ps.sps = sps; s->ps.pps = pps;
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 16; i++) rpl[l].list[i] = p->ref_poc[l][i];
    ref->tab_mvf = (MvField *)p->tab_mvf;
    ref->refPicList = rpl;
    s->ref = ref;
    s->cbf_luma = (uint8_t *)p->cbf_luma;
    s->vertical_bs = p->vertical_bs; s->horizontal_bs = p->horizontal_bs; s->bs_width = p->bs_width;
    s->HEVClc.boundary_flags = 0;
    for (int i = 0; i < nblocks; i++) ff_hevc_deblocking_boundary_strengths(s, blocks[3 * i], blocks[3 * i + 1], blocks[3 * i + 2]);
    av_free(rpl); av_free(ref); av_free(pps); av_free(sps); av_free(s);
    return 0;
}

/* ---- HEVCPredContext.intra_pred[] itself (hevcpred_template.c:31-334, via ff_hevc_pred_init), called once per block of
 * `blocks` in order on a context that holds exactly the fields the wrapper reads: the SPS / PPS geometry, the frame, the
 * motion field's is_intra, pps->min_tb_addr_zs, lc->na and the two mode fields.  Host pointers. */
int ref_hevc_intra_pred_blocks(const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks, int n, int bit_depth)
{
    HEVCContext *s = av_mallocz(sizeof(*s));
    HEVCSPS *sps = av_mallocz(sizeof(*sps));
    HEVCPPS *pps = av_mallocz(sizeof(*pps));
    HEVCFrame *ref = av_mallocz(sizeof(*ref));
    AVFrame *fr = av_frame_alloc();
    if (!s || !sps || !pps || !ref || !fr) return -1;
    if (sizeof(MvField) != sizeof(mi355_hevc_mvfield) || offsetof(MvField, is_intra) != offsetof(mi355_hevc_mvfield, is_intra) ||
        sizeof(*pps->min_tb_addr_zs) != sizeof(int32_t))
        return -2;
    ff_hevc_pred_init(&s->hpc, bit_depth);
    s->ps.sps = sps; s->ps.pps = pps; s->ref = ref; s->frame = fr;
    for (int i = 0; i < n; i++) {
        const mi355_hevc_intra_picture *p = pics + blocks[i].pic;
        const mi355_hevc_intra_block *b = blocks + i;
        sps->width = p->width; sps->height = p->height;
        sps->pixel_shift = bit_depth > 8;
        sps->hshift[0] = sps->vshift[0] = 0;
        sps->hshift[1] = sps->hshift[2] = p->hshift;
        sps->vshift[1] = sps->vshift[2] = p->vshift;
        sps->log2_min_pu_size = p->log2_min_pu_size;
        sps->log2_min_tb_size = p->log2_min_tb_size;
        sps->min_pu_width = p->min_pu_width; sps->min_pu_height = p->min_pu_height;
        sps->min_tb_width = p->min_tb_width;
        sps->sps_strong_intra_smoothing_enable_flag = p->strong_intra_smoothing != 0;
        pps->constrained_intra_pred_flag = p->constrained_intra_pred != 0;
        pps->min_tb_addr_zs = (int *)p->min_tb_addr_zs;
        ref->tab_mvf = (MvField *)p->tab_mvf;
        for (int c = 0; c < 3; c++) { fr->data[c] = p->data[c]; fr->linesize[c] = p->linesize[c]; }
        s->HEVClc.na.cand_bottom_left = !!(b->cand & MI355_HEVC_CAND_BOTTOM_LEFT);
        s->HEVClc.na.cand_left        = !!(b->cand & MI355_HEVC_CAND_LEFT);
        s->HEVClc.na.cand_up_left     = !!(b->cand & MI355_HEVC_CAND_UP_LEFT);
        s->HEVClc.na.cand_up          = !!(b->cand & MI355_HEVC_CAND_UP);
        s->HEVClc.na.cand_up_right    = !!(b->cand & MI355_HEVC_CAND_UP_RIGHT);
        s->HEVClc.tu.cur_intra_pred_mode = b->mode;
        s->HEVClc.pu.intra_pred_mode_c = b->mode;
        s->hpc.intra_pred[b->log2_size - 2](s, b->x0, b->y0, b->c_idx);
    }
    fr->data[0] = fr->data[1] = fr->data[2] = NULL;
    pps->min_tb_addr_zs = NULL;
    av_frame_free(&fr);
    av_free(ref); av_free(pps); av_free(sps); av_free(s);
    return 0;
}
