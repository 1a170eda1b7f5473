/*
 * mi355_h264_bridge.c — the H.264 Tier-2 bridge: PRODUCT glue that lives beside the reference's decoder.
 *
 * Compiled against the reference's own headers and linked into its decoder with
 *   -Wl,--wrap=ff_h264_hl_decode_mb,--wrap=ff_h264_field_end,--wrap=ff_h264_filter_mb,--wrap=ff_h264_filter_mb_fast
 * (nothing in the reference tree is modified), it turns the decoder into: host = parsing and entropy decoding only,
 * MI355X = everything the per-macroblock DSP did.
 *
 *   ff_h264_hl_decode_mb (libavcodec/h264_mb.c:798; called per macroblock from h264_slice.c:2375,2384,2442,2450)
 *       -> the macroblock's record, vectors and coefficients are packed into pinned staging memory exactly as
 *          include/mi355_h264_frame.h specifies; the host reconstructs nothing; sl->mb is cleared like the reference's
 *          idct_add functions would have left it.
 *   ff_h264_filter_mb / _fast (h264_loopfilter.c:716, :420; called per macroblock from loop_filter, h264_slice.c:2198)
 *       -> nothing: the device derives bS / alpha / beta / tc0 from the records and filters the whole picture.
 *   ff_h264_field_end (h264_picture.c:145; end of every coded picture)
 *       -> staging -> HBM on the stream, mi355_h264_decode_frames_levels_dev(), decoded picture -> the AVFrame the decoder will
 *          output, all enqueued asynchronously; the host waits only when the picture the decoder is about to output is
 *          not finished (always, unless MI355_BRIDGE_LAZY=1, which trusts h->output_frame).
 *
 * The decoded picture buffer lives in HBM: one device picture per H264Picture the decoder uses, found again through the
 * reference lists' parent pointers; reference samples never cross PCIe.  Two staging sets alternate, so the host can pack
 * picture n + 1 while the copies and kernels of picture n run.  One bridge state per decoding thread (N decoder threads =
 * N streams = N HIP streams).  Streams outside the Tier-2 scope (MBAFF / field pictures, more than 8 bits, not 4:2:0)
 * and any runtime failure make the bridge step aside for that decoder: the reference's own C path continues.
 * Errors are reported once on stderr; nothing here calls abort().
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/mpegutils.h"
#include "mi355dsp.h"
#include "mi355_h264_frame.h"

void __real_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl);
int __real_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup);
void __real_ff_h264_filter_mb(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                              uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize);
void __real_ff_h264_filter_mb_fast(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                                   uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize);

#define BR_MAX_PICS 40        /* H264_MAX_PICTURE_COUNT (36) + slack */
#define BR_MAX_SLICES 64

typedef struct DevPic {
    const H264Picture *owner;
    uint8_t *plane[3];
    void *done;                 /* event: kernels and the copy into the AVFrame are complete */
    int pending;
} DevPic;

typedef struct Staging {        /* pinned host images and their device mirrors */
    mi355_h264_mb *mb, *d_mb;
    int16_t *mv[2], *d_mv[2];
    int16_t *coef, *d_coef;
    mi355_h264_slice *slices, *d_slices;
    uint32_t *ilist, *d_ilist;
    int32_t *istart, *d_istart;
    mi355_h264_frame *desc, *d_desc;
    void *free_again;           /* event: the device no longer reads this set */
    int in_flight;
} Staging;

typedef struct Bridge {
    int state;                  /* 0 new, 1 active, -1 stepped aside */
    int lazy;
    int mb_w, mb_h, nmb;
    void *stream;
    Staging st[2];
    int cur;                    /* staging set being packed */
    int open;                   /* a picture is being packed */
    DevPic pics[BR_MAX_PICS];
    uint8_t *recon[3];
    int stride[2];
    /* per picture */
    int nslices, slice_num_of[BR_MAX_SLICES], uses_l1;
    const H264Picture *slot_pic[MI355_H264_MAX_SLOTS];
    int nslots;
    unsigned long pictures, waits;
} Bridge;

static __thread Bridge *br_tls;

static void br_fail(Bridge *b, const char *what)
{
    if (b->state >= 0) fprintf(stderr, "mi355 bridge: %s — this decoder continues on the reference's C path\n", what);
    b->state = -1;
}

static void *dalloc(size_t n) { return mi355_malloc(n); }

static int staging_alloc(Bridge *b, Staging *s)
{
    const size_t n = (size_t)b->nmb;
    s->mb = mi355_host_alloc(n * sizeof(*s->mb));           s->d_mb = dalloc(n * sizeof(*s->mb));
    for (int l = 0; l < 2; l++) { s->mv[l] = mi355_host_alloc(n * 64); s->d_mv[l] = dalloc(n * 64); }
    s->coef = mi355_host_alloc(n * 768);                    s->d_coef = dalloc(n * 768);
    s->slices = mi355_host_alloc(BR_MAX_SLICES * sizeof(*s->slices)); s->d_slices = dalloc(BR_MAX_SLICES * sizeof(*s->slices));
    s->ilist = mi355_host_alloc(n * 4);                     s->d_ilist = dalloc(n * 4);
    /* second half of istart: the per-level widths handed to mi355_h264_decode_frames_levels_dev */
    s->istart = mi355_host_alloc((size_t)(b->mb_w + 2 * b->mb_h + 2) * 8); s->d_istart = dalloc((size_t)(b->mb_w + 2 * b->mb_h + 2) * 4);
    s->desc = mi355_host_alloc(sizeof(*s->desc));           s->d_desc = dalloc(sizeof(*s->desc));
    s->free_again = mi355_event_create();
    return s->mb && s->d_mb && s->mv[0] && s->d_mv[0] && s->mv[1] && s->d_mv[1] && s->coef && s->d_coef && s->slices && s->d_slices &&
           s->ilist && s->d_ilist && s->istart && s->d_istart && s->desc && s->d_desc && s->free_again;
}

static Bridge *bridge_get(const H264Context *h)
{
    Bridge *b = br_tls;
    if (!b) {
        b = br_tls = calloc(1, sizeof(*b));
        if (!b) return NULL;
        b->lazy = getenv("MI355_BRIDGE_LAZY") != NULL;
    }
    if (b->state) return b;
    if (FRAME_MBAFF(h) || FIELD_PICTURE(h) || h->pixel_shift || h->ps.sps->chroma_format_idc != 1 || h->ps.sps->transform_bypass) {
        br_fail(b, "stream outside the batched path (needs progressive 8-bit 4:2:0 without transform bypass)");
        return b;
    }
    const char *dev = getenv("MI355_DEVICE");
    if (mi355_init(dev ? atoi(dev) : 0) != 0) { br_fail(b, "no usable MI355X"); return b; }
    b->mb_w = h->mb_width; b->mb_h = h->mb_height; b->nmb = b->mb_w * b->mb_h;
    b->stride[0] = (16 * b->mb_w + 63) & ~63; b->stride[1] = b->stride[0] / 2;
    b->stream = mi355_stream_create();
    int ok = b->stream != NULL && staging_alloc(b, &b->st[0]) && staging_alloc(b, &b->st[1]);
    for (int p = 0; p < 3 && ok; p++) ok = (b->recon[p] = dalloc((size_t)b->stride[p > 0] * (p ? 8 : 16) * b->mb_h)) != NULL;
    if (!ok) { br_fail(b, "device or pinned memory allocation failed"); return b; }
    b->state = 1;
    return b;
}

static DevPic *devpic_of(Bridge *b, const H264Picture *p, int create)
{
    DevPic *slot = NULL;
    for (int i = 0; i < BR_MAX_PICS; i++) {
        if (b->pics[i].owner == p) return &b->pics[i];
        if (!slot && !b->pics[i].owner) slot = &b->pics[i];
    }
    if (!create || !slot) return NULL;
    if (!slot->plane[0]) {
        for (int k = 0; k < 3; k++)
            if (!(slot->plane[k] = dalloc((size_t)b->stride[k > 0] * (k ? 8 : 16) * b->mb_h))) return NULL;
        if (!(slot->done = mi355_event_create())) return NULL;
    }
    slot->owner = p;
    return slot;
}

static int slot_of(Bridge *b, const H264Picture *p)
{
    for (int i = 0; i < b->nslots; i++)
        if (b->slot_pic[i] == p) return i;
    if (b->nslots >= MI355_H264_MAX_SLOTS) return -1;
    b->slot_pic[b->nslots] = p;
    return b->nslots++;
}

static void begin_picture(Bridge *b, const H264Context *h)
{
    /* the staging set must be free again: the kernels that read its device mirror two pictures ago have finished */
    b->cur ^= 1;
    Staging *s = &b->st[b->cur];
    if (s->in_flight) { mi355_event_sync(s->free_again); s->in_flight = 0; b->waits++; }
    memset(s->mb, 0, (size_t)b->nmb * sizeof(*s->mb));
    memset(s->mv[0], 0, (size_t)b->nmb * 64);
    memset(s->mv[1], 0, (size_t)b->nmb * 64);
    memset(s->slices, 0, BR_MAX_SLICES * sizeof(*s->slices));
    b->nslices = b->nslots = b->uses_l1 = 0;
    b->open = 1;
    (void)h;
}

static int slice_index(Bridge *b, const H264Context *h, const H264SliceContext *sl)
{
    for (int i = 0; i < b->nslices; i++)
        if (b->slice_num_of[i] == sl->slice_num) return i;
    if (b->nslices >= BR_MAX_SLICES) return -1;
    mi355_h264_slice *s = &b->st[b->cur].slices[b->nslices];
    b->slice_num_of[b->nslices] = sl->slice_num;
    s->use_weight = sl->pwt.use_weight;
    s->use_weight_chroma = sl->pwt.use_weight_chroma;
    s->luma_log2_weight_denom = sl->pwt.luma_log2_weight_denom;
    s->chroma_log2_weight_denom = sl->pwt.chroma_log2_weight_denom;
    s->list_count = sl->list_count;
    for (unsigned list = 0; list < sl->list_count; list++)
        for (unsigned i = 0; i < sl->ref_count[list] && i < MI355_H264_MAX_REFS; i++) {
            const int slot = slot_of(b, sl->ref_list[list][i].parent);
            if (slot < 0) return -1;
            s->ref_slot[list][i] = (uint8_t)slot;
        }
    for (int r = 0; r < MI355_H264_MAX_REFS; r++) {
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 2; k++) {
                s->luma_weight[r][l][k] = (int16_t)sl->pwt.luma_weight[r][l][k];
                for (int c = 0; c < 2; c++) s->chroma_weight[r][l][c][k] = (int16_t)sl->pwt.chroma_weight[r][l][c][k];
            }
        for (int r1 = 0; r1 < MI355_H264_MAX_REFS; r1++) s->implicit_weight[r][r1] = (int16_t)sl->pwt.implicit_weight[r][r1][0];
    }
    for (int t = 0; t < 2; t++)
        for (int q = 0; q < 52; q++) s->chroma_qp_table[t][q] = h->ps.pps->chroma_qp_table[t][q];
    return b->nslices++;
}

void __wrap_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl)
{
    Bridge *b = bridge_get(h);
    if (!b || b->state < 0) { __real_ff_h264_hl_decode_mb(h, sl); return; }
    if (!b->open) begin_picture(b, h);
    Staging *st = &b->st[b->cur];
    const int mb_xy = sl->mb_xy, idx = sl->mb_x + sl->mb_y * b->mb_w;
    const int mb_type = h->cur_pic.mb_type[mb_xy];
    mi355_h264_mb *m = &st->mb[idx];
    const int si = slice_index(b, h, sl);
    if (si < 0) { br_fail(b, "more slices or reference pictures than the batched path holds"); __real_ff_h264_hl_decode_mb(h, sl); return; }
    const int intra = IS_INTRA(mb_type);
    /* skipped macroblocks leave sl->cbp and the count caches stale (h264_cabac.c:1935-1941: only cbp_table is reset); their
     * residual is empty, which is what the loop filter sees through h->cbp_table / h->non_zero_count (h264_mvpred.h:808) */
    const int cbp = IS_SKIP(mb_type) ? 0 : sl->cbp;
    memset(m, 0, sizeof(*m));
    m->mb_type = (uint32_t)mb_type;
    m->cbp = (uint16_t)cbp;
    m->qp = h->cur_pic.qscale_table[mb_xy];
    m->qpc[0] = h->ps.pps->chroma_qp_table[0][m->qp & 0xff];
    m->qpc[1] = h->ps.pps->chroma_qp_table[1][m->qp & 0xff];
    m->slice_alpha_c0_offset = (int8_t)sl->slice_alpha_c0_offset;
    m->slice_beta_offset = (int8_t)sl->slice_beta_offset;
    m->slice_id = (uint8_t)si;
    /* which macroblock edges the loop filter sees a neighbour across: fill_filter_caches, h264_slice.c:2131-2145 */
    if (!sl->deblocking_filter) m->flags |= MI355_MBF_NO_DEBLOCK;
    else {
        if (sl->mb_x > 0 && (sl->deblocking_filter != 2 || h->slice_table[mb_xy - 1] == sl->slice_num)) m->flags |= MI355_MBF_LEFT_EDGE;
        if (sl->mb_y > 0 && (sl->deblocking_filter != 2 || h->slice_table[mb_xy - h->mb_stride] == sl->slice_num)) m->flags |= MI355_MBF_TOP_EDGE;
    }
    if (sl->pwt.use_weight) m->flags |= MI355_MBF_WEIGHTED;
    m->intra16x16_pred_mode = (uint8_t)sl->intra16x16_pred_mode;
    m->chroma_pred_mode = (uint8_t)sl->chroma_pred_mode;
    m->topleft_samples_available = (uint16_t)sl->topleft_samples_available;
    m->topright_samples_available = (uint16_t)sl->topright_samples_available;
    m->dc_qmul[0] = h->ps.pps->dequant4_coeff[0][sl->qscale][0];
    m->dc_qmul[1] = h->ps.pps->dequant4_coeff[intra ? 1 : 4][sl->chroma_qp[0]][0];
    m->dc_qmul[2] = h->ps.pps->dequant4_coeff[intra ? 2 : 5][sl->chroma_qp[1]][0];
    memset(m->ref_idx, -1, sizeof(m->ref_idx));

    int16_t *cf = st->coef + (size_t)idx * 384;
    memset(cf, 0, 768);
    if (IS_INTRA_PCM(mb_type)) {
        memcpy(cf, sl->intra_pcm_ptr, 384);
        m->nnz_mask = 0xFFFFFF;
        memset(m->u.intra4x4_pred_mode, 0, 16);
    } else {
        /* coefficient masks: the count caches are only meaningful where cbp says something was coded */
        const int luma_coded = IS_INTRA16x16(mb_type) || (cbp & 15);
        if (luma_coded) {
            for (int i = 0; i < 16; i++) {
                const int src = IS_8x8DCT(mb_type) ? (i & ~3) : i;
                if (sl->non_zero_count_cache[scan8[src]]) m->nnz_mask |= 1u << i;
            }
            memcpy(cf, sl->mb, 256 * 2);
        }
        if (IS_INTRA16x16(mb_type) && sl->non_zero_count_cache[scan8[LUMA_DC_BLOCK_INDEX]]) {
            m->nnz_mask |= 1u << MI355_NNZ_LUMA_DC;
            for (int k = 0; k < 16; k++) cf[mi355_luma_dc_slot(k)] = sl->mb_luma_dc[0][k];
        }
        if (cbp & 0x30) {
            memcpy(cf + 256, sl->mb + 256, 64 * 2);
            memcpy(cf + 320, sl->mb + 512, 64 * 2);
            if (cbp & 0x20)
                for (int j = 0; j < 4; j++) {
                    if (sl->non_zero_count_cache[scan8[16 + j]]) m->nnz_mask |= 1u << MI355_NNZ_CB(j);
                    if (sl->non_zero_count_cache[scan8[32 + j]]) m->nnz_mask |= 1u << MI355_NNZ_CR(j);
                }
            if (sl->non_zero_count_cache[scan8[CHROMA_DC_BLOCK_INDEX + 0]]) m->nnz_mask |= 1u << MI355_NNZ_CB_DC;
            if (sl->non_zero_count_cache[scan8[CHROMA_DC_BLOCK_INDEX + 1]]) m->nnz_mask |= 1u << MI355_NNZ_CR_DC;
        }
        if (intra) {
            for (int i = 0; i < 16; i++) m->u.intra4x4_pred_mode[i] = sl->intra4x4_pred_mode_cache[scan8[i]];
        } else {
            memset(m->u.inter.ref_pic, 0xFF, sizeof(m->u.inter.ref_pic));
            for (unsigned list = 0; list < sl->list_count; list++) {
                if (!USES_LIST(mb_type, list)) continue;
                if (list) b->uses_l1 = 1;
                for (int q = 0; q < 4; q++) {
                    const int r = sl->ref_cache[list][scan8[4 * q]];
                    m->ref_idx[list][q] = (int8_t)(r < 0 ? -1 : r);
                    if (r >= 0) m->u.inter.ref_pic[list][q] = st->slices[si].ref_slot[list][r];
                }
                for (int i = 0; i < 16; i++) {
                    const int x4 = (i & 1) + 2 * ((i >> 2) & 1), y4 = ((i >> 1) & 1) + 2 * (i >> 3);
                    int16_t *d = st->mv[list] + ((size_t)idx * 16 + x4 + 4 * y4) * 2;
                    d[0] = sl->mv_cache[list][scan8[i]][0];
                    d[1] = sl->mv_cache[list][scan8[i]][1];
                }
            }
            if (IS_8X8(mb_type))
                for (int q = 0; q < 4; q++) {
                    const int t = sl->sub_mb_type[q];
                    const int shape = IS_SUB_8X8(t) ? MI355_SUB_8x8 : IS_SUB_8X4(t) ? MI355_SUB_8x4 : IS_SUB_4X8(t) ? MI355_SUB_4x8 : MI355_SUB_4x4;
                    m->sub_mb_type[q] = (uint8_t)(shape | (IS_DIR(t, 0, 0) ? MI355_SUB_L0 : 0) | (IS_DIR(t, 0, 1) ? MI355_SUB_L1 : 0));
                }
        }
        /* what the reference's idct_add / dc_dequant functions leave behind (h264idct_template.c:66,:140,:150): the residual
         * decoders rely on finding the block array zeroed */
        if (luma_coded || (cbp & 0x30)) memset(sl->mb, 0, 16 * 48 * sizeof(int16_t));
    }
}

void __wrap_ff_h264_filter_mb(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                              uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize)
{
    if (br_tls && br_tls->state > 0) return;
    __real_ff_h264_filter_mb(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize);
}
void __wrap_ff_h264_filter_mb_fast(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                                   uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize)
{
    if (br_tls && br_tls->state > 0) return;
    __real_ff_h264_filter_mb_fast(h, sl, mb_x, mb_y, img_y, img_cb, img_cr, linesize, uvlinesize);
}

static int submit_picture(Bridge *b, H264Context *h)
{
    Staging *s = &b->st[b->cur];
    DevPic *cur = devpic_of(b, h->cur_pic_ptr, 1);
    if (!cur) return -1;
    int lw = 0;
    const int maxl = mi355_h264_intra_schedule(s->mb, b->mb_w, b->mb_h, s->ilist, s->istart, &lw);
    if (maxl < 0) return -1;
    mi355_h264_frame *f = s->desc;
    memset(f, 0, sizeof(*f));
    f->mb_width = b->mb_w; f->mb_height = b->mb_h;
    for (int k = 0; k < 3; k++) { f->dst[k] = cur->plane[k]; f->recon[k] = b->recon[k]; }
    f->dst_stride[0] = f->recon_stride[0] = b->stride[0];
    f->dst_stride[1] = f->recon_stride[1] = b->stride[1];
    for (int i = 0; i < b->nslots; i++) {
        DevPic *r = devpic_of(b, b->slot_pic[i], 0);
        if (!r) return -2;                       /* a reference this bridge never decoded (a stream joined mid-way) */
        for (int k = 0; k < 3; k++) f->ref[i][k] = r->plane[k];
    }
    f->mb = s->d_mb; f->mv[0] = s->d_mv[0]; f->mv[1] = b->uses_l1 ? s->d_mv[1] : NULL; f->coef = s->d_coef;
    f->slices = s->d_slices; f->nslices = b->nslices;
    f->max_intra_level = maxl; f->intra_list = s->d_ilist; f->intra_level_start = s->d_istart; f->max_level_width = lw;
    const size_t n = (size_t)b->nmb;
    int rc = mi355_memcpy_h2d_async(s->d_mb, s->mb, n * sizeof(*s->mb), b->stream);
    rc |= mi355_memcpy_h2d_async(s->d_mv[0], s->mv[0], n * 64, b->stream);
    if (b->uses_l1) rc |= mi355_memcpy_h2d_async(s->d_mv[1], s->mv[1], n * 64, b->stream);
    rc |= mi355_memcpy_h2d_async(s->d_coef, s->coef, n * 768, b->stream);
    rc |= mi355_memcpy_h2d_async(s->d_slices, s->slices, (size_t)b->nslices * sizeof(*s->slices), b->stream);
    if (maxl > 0) {
        rc |= mi355_memcpy_h2d_async(s->d_ilist, s->ilist, (size_t)s->istart[maxl] * 4, b->stream);
        rc |= mi355_memcpy_h2d_async(s->d_istart, s->istart, (size_t)(maxl + 1) * 4, b->stream);
    }
    rc |= mi355_memcpy_h2d_async(s->d_desc, s->desc, sizeof(*s->desc), b->stream);
    if (rc) return -3;
    int32_t *widths = s->istart + (b->mb_w + 2 * b->mb_h + 2);
    for (int l = 0; l < maxl; l++) widths[l] = s->istart[l + 1] - s->istart[l];
    if (mi355_h264_decode_frames_levels_dev(s->d_desc, 1, b->mb_w, b->mb_h, maxl, widths, b->stream) != 0) return -4;
    mi355_event_record(s->free_again, b->stream);
    s->in_flight = 1;
    /* the finished picture -> the frame the decoder hands out (coded size; the reference crops on output) */
    const AVFrame *fr = h->cur_pic_ptr->f;
    for (int k = 0; k < 3; k++)
        if (mi355_memcpy2d_d2h_async(fr->data[k], (size_t)fr->linesize[k], cur->plane[k], (size_t)b->stride[k > 0],
                                     (size_t)(k ? 8 : 16) * b->mb_w, (size_t)(k ? 8 : 16) * b->mb_h, b->stream)) return -5;
    mi355_event_record(cur->done, b->stream);
    cur->pending = 1;
    b->pictures++;
    return 0;
}

int __wrap_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup)
{
    Bridge *b = br_tls;
    if (b && b->state > 0 && b->open) {
        b->open = 0;
        if (submit_picture(b, h) != 0) {
            /* the picture is lost for this path; what was enqueued must drain before the host touches the frames again */
            mi355_sync(b->stream);
            br_fail(b, "submitting a picture to the device failed");
        } else {
            /* wait only for what the decoder is about to hand out: h->output_frame was chosen when the picture started
             * (h264_select_output_frame, h264_slice.c:1173-1290, called from h264_field_start :1528) and shares its buffers
             * with the H264Picture it refers to; without MI355_BRIDGE_LAZY every picture is complete before this returns */
            const uint8_t *out0 = h->output_frame && h->output_frame->buf[0] ? h->output_frame->data[0] : NULL;
            for (int i = 0; i < BR_MAX_PICS; i++) {
                DevPic *p = &b->pics[i];
                if (p->pending && (!b->lazy || (out0 && p->owner && p->owner->f && p->owner->f->data[0] == out0))) {
                    mi355_event_sync(p->done);
                    p->pending = 0;
                }
            }
        }
    }
    return __real_ff_h264_field_end(h, sl, in_setup);
}

/* for hosts that want the numbers (the throughput harness prints them) */
void mi355_h264_bridge_stats(unsigned long *pictures, unsigned long *staging_waits, int *active)
{
    Bridge *b = br_tls;
    if (pictures) *pictures = b ? b->pictures : 0;
    if (staging_waits) *staging_waits = b ? b->waits : 0;
    if (active) *active = b ? b->state : 0;
}
/* a decoder thread that ends (or flushes with MI355_BRIDGE_LAZY) calls this: everything enqueued is complete afterwards */
void mi355_h264_bridge_drain(void)
{
    Bridge *b = br_tls;
    if (b && b->state > 0) {
        mi355_sync(b->stream);
        for (int i = 0; i < BR_MAX_PICS; i++) b->pics[i].pending = 0;
    }
}
