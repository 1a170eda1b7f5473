/*
 * ref_h264_export.c — run the REFERENCE's own H.264 decoder on a real bitstream and export,
 * for every macroblock, the Tier-2 record (include/mi355_h264_frame.h) that a bridge would
 * build, together with the reference's decoded pictures.  TEST INFRASTRUCTURE ONLY: it is the
 * generator of tests/golden/h264_stream_*.npz (see tests/golden/make_stream_golden.py), which
 * pins oracle_h264frame.c and the Tier-2 kernels to the reference decoder's output.
 *
 * Links the reference's decoder objects (built in place by oracle/Makefile, no configure run)
 * with two linker wraps, so nothing in /root/reference is modified:
 *   --wrap=ff_h264_hl_decode_mb  (libavcodec/h264_mb.c:798; called per MB from h264_slice.c:2375/2443)
 *   --wrap=ff_h264_field_end     (libavcodec/h264_picture.c:145; end of every coded picture)
 *
 * usage: ref_h264_export <in.samples> <out.bin>
 *   in.samples: u32 extradata_len, extradata (avcC), u32 n, then n x {u32 len, bytes}
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/mpegutils.h"
#include "../include/mi355_h264_frame.h"

extern AVCodec ff_h264_decoder;
void __real_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl);
int __real_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup);

static FILE *out;
static int n_decoded;                 /* pictures finished so far = id of the picture being decoded */

/* state of the picture being decoded */
static mi355_h264_mb *mbs;
static int16_t *mvs[2];
static int16_t *coefs;
static mi355_h264_slice slices[64];
static int slice_num_of[64];
static int nslices, nmb, mb_w, mb_h, uses_l1;
static int slot_pic[MI355_H264_MAX_SLOTS], nslots;     /* slot -> decoded picture id */
static const H264Picture *pic_ptr[64];                 /* H264Picture* currently holding decoded id */
static int pic_id[64], npics;

static int id_of_picture(const H264Picture *p)
{
    for (int i = npics - 1; i >= 0; i--)
        if (pic_ptr[i] == p) return pic_id[i];
    fprintf(stderr, "export: reference picture not seen before\n");
    exit(2);
}
static int slot_of(int id)
{
    for (int i = 0; i < nslots; i++)
        if (slot_pic[i] == id) return i;
    if (nslots >= MI355_H264_MAX_SLOTS) { fprintf(stderr, "export: too many reference pictures\n"); exit(2); }
    slot_pic[nslots] = id;
    return nslots++;
}

static void begin_picture(const H264Context *h)
{
    mb_w = h->mb_width; mb_h = h->mb_height; nmb = mb_w * mb_h;
    mbs = calloc(nmb, sizeof(*mbs));
    mvs[0] = calloc(nmb, 64); mvs[1] = calloc(nmb, 64);
    coefs = calloc(nmb, 768);
    nslices = 0; nslots = 0; uses_l1 = 0;
    memset(slices, 0, sizeof(slices));
    if (npics == 64) { memmove(pic_ptr, pic_ptr + 1, 63 * sizeof(*pic_ptr)); memmove(pic_id, pic_id + 1, 63 * sizeof(*pic_id)); npics--; }
    pic_ptr[npics] = h->cur_pic_ptr; pic_id[npics++] = n_decoded;
}

static int slice_index(const H264Context *h, const H264SliceContext *sl)
{
    for (int i = 0; i < nslices; i++)
        if (slice_num_of[i] == sl->slice_num) return i;
    if (nslices >= 64) { fprintf(stderr, "export: too many slices\n"); exit(2); }
    mi355_h264_slice *s = &slices[nslices];
    slice_num_of[nslices] = sl->slice_num;
    s->use_weight = sl->pwt.use_weight;
    s->use_weight_chroma = sl->pwt.use_weight_chroma;
    s->luma_log2_weight_denom = sl->pwt.luma_log2_weight_denom;
    s->chroma_log2_weight_denom = sl->pwt.chroma_log2_weight_denom;
    s->list_count = sl->list_count;
    for (unsigned list = 0; list < sl->list_count; list++)
        for (unsigned i = 0; i < sl->ref_count[list] && i < MI355_H264_MAX_REFS; i++)
            s->ref_slot[list][i] = (uint8_t)slot_of(id_of_picture(sl->ref_list[list][i].parent));
    for (int r = 0; r < MI355_H264_MAX_REFS; r++) {
        for (int l = 0; l < 2; l++) {
            for (int k = 0; k < 2; k++) {
                s->luma_weight[r][l][k] = (int16_t)sl->pwt.luma_weight[r][l][k];
                for (int c = 0; c < 2; c++) s->chroma_weight[r][l][c][k] = (int16_t)sl->pwt.chroma_weight[r][l][c][k];
            }
        }
        for (int r1 = 0; r1 < MI355_H264_MAX_REFS; r1++) s->implicit_weight[r][r1] = (int16_t)sl->pwt.implicit_weight[r][r1][0];
    }
    for (int t = 0; t < 2; t++)
        for (int q = 0; q < 52; q++) s->chroma_qp_table[t][q] = h->ps.pps->chroma_qp_table[t][q];
    return nslices++;
}

void __wrap_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl)
{
    if (FRAME_MBAFF(h) || FIELD_PICTURE(h) || h->pixel_shift || h->ps.sps->chroma_format_idc != 1) {
        fprintf(stderr, "export: stream is outside the Tier-2 scope (needs progressive 8-bit 4:2:0)\n");
        exit(3);
    }
    if (!mbs) begin_picture(h);
    const int mb_xy = sl->mb_xy, idx = sl->mb_x + sl->mb_y * mb_w;
    const int mb_type = h->cur_pic.mb_type[mb_xy];
    mi355_h264_mb *m = &mbs[idx];
    const int si = slice_index(h, sl);
    const int intra = IS_INTRA(mb_type);
    /* skipped MBs leave sl->cbp and the count caches stale (h264_cabac.c:1935-1941: only cbp_table is
     * reset); their residual is empty, which is what the loop filter sees through h->cbp_table /
     * h->non_zero_count (h264_mvpred.h:808) */
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
    /* which MB edges the loop filter will see a neighbour across: fill_filter_caches, h264_slice.c:2131-2145 */
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

    int16_t *cf = coefs + (size_t)idx * 384;
    if (IS_INTRA_PCM(mb_type)) {
        memcpy(cf, sl->intra_pcm_ptr, 384);
        m->nnz_mask = 0xFFFFFF;
        memset(m->u.intra4x4_pred_mode, 0, 16);
    } else {
        /* coefficient masks: count caches are only meaningful where cbp says something was coded */
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
                if (list) uses_l1 = 1;
                for (int q = 0; q < 4; q++) {
                    const int r = sl->ref_cache[list][scan8[4 * q]];
                    m->ref_idx[list][q] = (int8_t)(r < 0 ? -1 : r);
                    if (r >= 0) m->u.inter.ref_pic[list][q] = slices[si].ref_slot[list][r];
                }
                for (int i = 0; i < 16; i++) {
                    const int x4 = (i & 1) + 2 * ((i >> 2) & 1), y4 = ((i >> 1) & 1) + 2 * (i >> 3);
                    int16_t *d = mvs[list] + ((size_t)idx * 16 + x4 + 4 * y4) * 2;
                    d[0] = sl->mv_cache[list][scan8[i]][0];
                    d[1] = sl->mv_cache[list][scan8[i]][1];
                }
            }
            if (IS_8X8(mb_type))
                for (int q = 0; q < 4; q++) {
                    const int st = sl->sub_mb_type[q];
                    int shape = IS_SUB_8X8(st) ? MI355_SUB_8x8 : IS_SUB_8X4(st) ? MI355_SUB_8x4 : IS_SUB_4X8(st) ? MI355_SUB_4x8 : MI355_SUB_4x4;
                    m->sub_mb_type[q] = (uint8_t)(shape | (IS_DIR(st, 0, 0) ? MI355_SUB_L0 : 0) | (IS_DIR(st, 0, 1) ? MI355_SUB_L1 : 0));
                }
        }
    }
    __real_ff_h264_hl_decode_mb(h, sl);
}

static void put_u32(uint32_t v) { fwrite(&v, 4, 1, out); }

int __wrap_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup)
{
    if (mbs) {
        int lw = 0;
        uint32_t *list = calloc(nmb, 4);
        int32_t *start = calloc(mb_w + 2 * mb_h + 2, 4);
        /* intra levels are a pure function of mb_type: recompute them here with the same rule as
         * mi355_h264_intra_schedule() so the fixture is self-contained */
        int maxl = 0;
        int32_t *level = calloc(nmb, sizeof(*level));   /* int levels: the record's 8-bit field only gets the saturated value */
        for (int y = 0; y < mb_h; y++)
            for (int x = 0; x < mb_w; x++) {
                mi355_h264_mb *m = &mbs[x + y * mb_w];
                if (!(m->mb_type & MI355_MB_INTRA)) { m->intra_level = 0; continue; }
                int lv = 0;
                const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
                for (int k = 0; k < 4; k++) {
                    int nx = x + dx[k], ny = y + dy[k];
                    if (nx >= 0 && nx < mb_w && ny >= 0 && ny < mb_h && level[nx + ny * mb_w] > lv) lv = level[nx + ny * mb_w];
                }
                level[x + y * mb_w] = lv + 1;
                m->intra_level = (uint8_t)(lv + 1 > 255 ? 255 : lv + 1);
                if (lv + 1 > maxl) maxl = lv + 1;
            }
        free(level);
        (void)lw; (void)list; (void)start;
        const AVFrame *f = h->cur_pic_ptr->f;
        put_u32(0x46523634);     /* "FR64" */
        put_u32((uint32_t)mb_w); put_u32((uint32_t)mb_h); put_u32((uint32_t)nslices); put_u32((uint32_t)nslots);
        put_u32((uint32_t)uses_l1); put_u32((uint32_t)maxl); put_u32((uint32_t)h->cur_pic_ptr->f->pict_type);
        for (int i = 0; i < MI355_H264_MAX_SLOTS; i++) put_u32(i < nslots ? (uint32_t)slot_pic[i] : 0xFFFFFFFFu);
        fwrite(mbs, sizeof(*mbs), nmb, out);
        fwrite(mvs[0], 64, nmb, out);
        fwrite(mvs[1], 64, nmb, out);
        fwrite(coefs, 768, nmb, out);
        fwrite(slices, sizeof(slices[0]), nslices, out);
        free(list); free(start);
    }
    int ret = __real_ff_h264_field_end(h, sl, in_setup);
    if (mbs) {
        /* the finished (deblocked) picture, full coded size */
        const AVFrame *f = h->cur_pic_ptr ? h->cur_pic_ptr->f : NULL;
        const H264Picture *p = pic_ptr[npics - 1];
        f = p->f;
        for (int pl = 0; pl < 3; pl++) {
            int w = (pl ? 8 : 16) * mb_w, hh = (pl ? 8 : 16) * mb_h;
            for (int y = 0; y < hh; y++) fwrite(f->data[pl] + (size_t)y * f->linesize[pl], 1, w, out);
        }
        free(mbs); free(mvs[0]); free(mvs[1]); free(coefs);
        mbs = NULL;
        n_decoded++;
    }
    return ret;
}

static uint32_t get_u32(FILE *f) { uint32_t v = 0; if (fread(&v, 4, 1, f) != 1) exit(4); return v; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.bin\n", argv[0]); return 1; }
    FILE *in = fopen(argv[1], "rb");
    out = fopen(argv[2], "wb");
    if (!in || !out) return 1;
    AVCodecContext *c = avcodec_alloc_context3(&ff_h264_decoder);
    uint32_t el = get_u32(in);
    c->extradata = av_mallocz(el + AV_INPUT_BUFFER_PADDING_SIZE);
    c->extradata_size = (int)el;
    if (fread(c->extradata, 1, el, in) != el) return 4;
    c->thread_count = 1;
    c->flags |= AV_CODEC_FLAG_BITEXACT;
    if (avcodec_open2(c, &ff_h264_decoder, NULL) < 0) { fprintf(stderr, "open failed\n"); return 5; }
    uint32_t n = get_u32(in);
    AVFrame *fr = av_frame_alloc();
    int shown = 0;
    for (uint32_t i = 0; i <= n; i++) {
        AVPacket pkt;
        av_init_packet(&pkt);
        pkt.data = NULL; pkt.size = 0;
        if (i < n) {
            uint32_t len = get_u32(in);
            if (av_new_packet(&pkt, (int)len) < 0) return 6;
            if (fread(pkt.data, 1, len, in) != len) return 4;
        }
        int r = avcodec_send_packet(c, i < n ? &pkt : NULL);
        if (r < 0) { fprintf(stderr, "send_packet failed %d\n", r); return 7; }
        while (avcodec_receive_frame(c, fr) >= 0) { shown++; av_frame_unref(fr); }
        if (i < n) av_packet_unref(&pkt);
    }
    fprintf(stderr, "export: %u packets, %d pictures decoded, %d output\n", n, n_decoded, shown);
    fclose(out);
    return 0;
}
