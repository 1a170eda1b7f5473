/*
 * ref_hevc_tier1_main.c — TEST INFRASTRUCTURE.  The REFERENCE's own HEVC decoder (its objects built in place by
 * oracle/Makefile) with its three DSP init functions (hevcdec.c:443-445) interposed by the linker as INTEGRATION.md §2
 * describes: every table is filled by the reference's C init and then overridden by this project's ff_*_init_mi355x hooks.
 * Linked against the SIMT-emulated build of the product sources (tests/_emu/libmi355dsp_emu.so) or the real library; the
 * decoded pictures are compared with the unmodified reference decoder's (MI355_TIER1_PLAIN=1: the comparison run).
 *
 * usage: ref_hevc_tier1 <in.samples> <out.yuv>     (samples format: see ref_h264_export.c)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavutil/mem.h"
#include "libavcodec/avcodec.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcdec.h"
#include "libavcodec/videodsp.h"
#include "../include/mi355dsp.h"      /* the table structs are skipped: the reference's headers came first */
#include "../include/mi355_hevc_batch.h"
#include "../include/mi355_h264_frame.h"      /* mi355_malloc / mi355_memcpy_* / mi355_sync */

#include "libavutil/pixdesc.h"

extern AVCodec ff_hevc_decoder;
/* the wraps themselves are PRODUCT code: contrib/libav/mi355_wrap.c, compiled into this binary from there (with
 * -DMI355_WRAP_NO_H264; ff_hevc_pred_init is wrapped HERE because of the pin below and applies the hook the same way) */
void mi355_wrap_stats(unsigned long *tables_hooked, unsigned long *entries_replaced);
static unsigned long n_pred_hooks, n_pred_replaced;
static int plain;     /* MI355_TIER1_PLAIN=1: leave the tables as the reference filled them (the comparison run) */

/* MI355_HEVC_INTRA_DEVICE=1 (a PIN, not a binding: one launch and two picture copies per block): HEVCPredContext.intra_pred[]
 * — the context-walking wrapper the Tier-1 hook leaves to the reference's C (hevcpred_template.c:31-334) — is replaced by
 * mi355_hevc_intra_pred_blocks_dev() called for ONE block against the decoder's own state: the picture so far, lc->na,
 * tab_mvf (constrained intra prediction) and pps->min_tb_addr_zs.  The batched entry point is thereby run on every
 * intra block of every generated stream with the availability patterns a real coding tree produces. */
static unsigned long n_intra_dev;
static struct { uint8_t *plane[3]; size_t plane_bytes[3]; uint8_t *mvf, *zs; size_t mvf_bytes, zs_bytes; void *desc, *blk; } idev;
static int idev_ensure(uint8_t **p, size_t *have, size_t want)
{
    if (*p && *have >= want) return 0;
    if (*p) mi355_free(*p);
    *p = mi355_malloc(want);
    *have = *p ? want : 0;
    return *p ? 0 : -1;
}
static void intra_dev(HEVCContext *s, int x0, int y0, int c_idx, int log2_size)
{
    const HEVCSPS *sps = s->ps.sps;
    const HEVCLocalContext *lc = &s->HEVClc;
    const int h[3] = { sps->height, sps->height >> sps->vshift[1], sps->height >> sps->vshift[2] };
    const size_t mvf = (size_t)sps->min_pu_width * sps->min_pu_height * sizeof(MvField);
    const size_t zs = (size_t)sps->min_tb_width * sps->min_tb_height * sizeof(int);
    int rc = 0;
    mi355_hevc_intra_picture d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < 3; i++) {
        const size_t sz = (size_t)s->frame->linesize[i] * h[i];
        if (idev_ensure(&idev.plane[i], &idev.plane_bytes[i], sz)) abort();
        if (i == c_idx) rc |= mi355_memcpy_h2d(idev.plane[i], s->frame->data[i], sz);
        d.data[i] = idev.plane[i]; d.linesize[i] = s->frame->linesize[i];
    }
    if (idev_ensure(&idev.mvf, &idev.mvf_bytes, mvf) || idev_ensure(&idev.zs, &idev.zs_bytes, zs)) abort();
    if (!idev.desc) { idev.desc = mi355_malloc(sizeof(d)); idev.blk = mi355_malloc(sizeof(mi355_hevc_intra_block)); }
    rc |= mi355_memcpy_h2d(idev.mvf, s->ref->tab_mvf, mvf) | mi355_memcpy_h2d(idev.zs, s->ps.pps->min_tb_addr_zs, zs);
    d.width = sps->width; d.height = sps->height; d.hshift = sps->hshift[1]; d.vshift = sps->vshift[1];
    d.log2_min_pu_size = sps->log2_min_pu_size; d.log2_min_tb_size = sps->log2_min_tb_size;
    d.min_pu_width = sps->min_pu_width; d.min_pu_height = sps->min_pu_height; d.min_tb_width = sps->min_tb_width;
    d.constrained_intra_pred = s->ps.pps->constrained_intra_pred_flag;
    d.strong_intra_smoothing = sps->sps_strong_intra_smoothing_enable_flag;
    d.tab_mvf = (const mi355_hevc_mvfield *)idev.mvf; d.min_tb_addr_zs = (const int32_t *)idev.zs;
    mi355_hevc_intra_block b;
    memset(&b, 0, sizeof(b));
    b.x0 = (uint16_t)x0; b.y0 = (uint16_t)y0; b.log2_size = (uint8_t)log2_size; b.c_idx = (uint8_t)c_idx;
    b.mode = (uint8_t)(c_idx ? lc->pu.intra_pred_mode_c : lc->tu.cur_intra_pred_mode);
    b.cand = (uint8_t)((lc->na.cand_bottom_left ? MI355_HEVC_CAND_BOTTOM_LEFT : 0) | (lc->na.cand_left ? MI355_HEVC_CAND_LEFT : 0) |
                       (lc->na.cand_up_left ? MI355_HEVC_CAND_UP_LEFT : 0) | (lc->na.cand_up ? MI355_HEVC_CAND_UP : 0) |
                       (lc->na.cand_up_right ? MI355_HEVC_CAND_UP_RIGHT : 0));
    rc |= mi355_memcpy_h2d(idev.desc, &d, sizeof(d)) | mi355_memcpy_h2d(idev.blk, &b, sizeof(b));
    rc |= mi355_hevc_intra_pred_blocks_dev(idev.desc, idev.blk, 1, sps->bit_depth, NULL);
    rc |= mi355_sync(NULL);
    rc |= mi355_memcpy_d2h(s->frame->data[c_idx], idev.plane[c_idx], (size_t)s->frame->linesize[c_idx] * h[c_idx]);
    if (rc) { fprintf(stderr, "intra pin: device call failed\n"); abort(); }
    n_intra_dev++;
}
static void intra_dev_2(HEVCContext *s, int x0, int y0, int c_idx) { intra_dev(s, x0, y0, c_idx, 2); }
static void intra_dev_3(HEVCContext *s, int x0, int y0, int c_idx) { intra_dev(s, x0, y0, c_idx, 3); }
static void intra_dev_4(HEVCContext *s, int x0, int y0, int c_idx) { intra_dev(s, x0, y0, c_idx, 4); }
static void intra_dev_5(HEVCContext *s, int x0, int y0, int c_idx) { intra_dev(s, x0, y0, c_idx, 5); }

void __real_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth);
void __wrap_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth)
{
    __real_ff_hevc_pred_init(c, bit_depth);
    HEVCPredContext was = *c;
    if (!plain) ff_hevc_pred_init_mi355x(c, bit_depth);
    for (size_t i = 0; i < sizeof(was) / sizeof(void *); i++) n_pred_replaced += ((void **)&was)[i] != ((void **)c)[i];
    if (getenv("MI355_HEVC_INTRA_DEVICE")) {
        if (plain && mi355_init(0) != 0) { fprintf(stderr, "mi355_init failed\n"); exit(2); }
        c->intra_pred[0] = intra_dev_2; c->intra_pred[1] = intra_dev_3; c->intra_pred[2] = intra_dev_4; c->intra_pred[3] = intra_dev_5;
    }
    n_pred_hooks++;
}

/* the writer's streams are open loop: a stream whose arithmetic decoding ran out of step would still decode to SOMETHING.
 * Counting the coding tree units and the slice ends the decoder sees catches that: a slice that misses its end flag runs
 * on into the CTUs of the next one (tests/golden/make_hevc_streams.py compares both counts with what it wrote). */
static unsigned long n_ctus, n_slice_ends;
int __real_ff_hevc_end_of_slice_flag_decode(HEVCContext *s);
int __wrap_ff_hevc_end_of_slice_flag_decode(HEVCContext *s)
{
    const int r = __real_ff_hevc_end_of_slice_flag_decode(s);
    n_ctus++;
    n_slice_ends += r != 0;
    if (getenv("MI355_HEVC_TRACE_SLICES")) fprintf(stderr, "ctu %lu end_of_slice %d\n", n_ctus, r);
    return r;
}

/* present when contrib/libav/mi355_hevc_lf_bridge.c is linked in (_ref/hevc_lf_*): pictures deblocked by the picture-level pass */
extern unsigned long mi355_hevc_lf_bridge_pictures(void) __attribute__((weak));
extern unsigned long mi355_hevc_lf_bridge_bs_pictures(void) __attribute__((weak));

static uint32_t get_u32(FILE *f) { uint32_t v = 0; if (fread(&v, 4, 1, f) != 1) exit(4); return v; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.yuv\n", argv[0]); return 1; }
    plain = getenv("MI355_TIER1_PLAIN") != NULL;
    if (!plain && mi355_init(0) != 0) { fprintf(stderr, "mi355_init failed\n"); return 2; }
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!in || !out) return 1;
    AVCodecContext *c = avcodec_alloc_context3(&ff_hevc_decoder);
    uint32_t el = get_u32(in);
    c->extradata = av_mallocz(el + AV_INPUT_BUFFER_PADDING_SIZE);
    c->extradata_size = (int)el;
    if (fread(c->extradata, 1, el, in) != el) return 4;
    c->thread_count = 1;
    c->flags |= AV_CODEC_FLAG_BITEXACT;
    if (avcodec_open2(c, &ff_hevc_decoder, NULL) < 0) { fprintf(stderr, "open failed\n"); return 5; }
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
        if (avcodec_send_packet(c, i < n ? &pkt : NULL) < 0) { fprintf(stderr, "send_packet failed\n"); return 7; }
        while (avcodec_receive_frame(c, fr) >= 0) {
            for (int pl = 0; pl < 3; pl++) {
                const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fr->format);
                const int w = pl ? fr->width >> d->log2_chroma_w : fr->width, h = pl ? fr->height >> d->log2_chroma_h : fr->height;
                const int bps = (d->comp[0].depth + 7) >> 3;          /* 9 / 10-bit pictures: two bytes per sample */
                for (int y = 0; y < h; y++) fwrite(fr->data[pl] + (size_t)y * fr->linesize[pl], bps, w, out);
            }
            shown++;
            av_frame_unref(fr);
        }
        if (i < n) av_packet_unref(&pkt);
    }
    unsigned long n_hooks = 0, n_replaced = 0;
    mi355_wrap_stats(&n_hooks, &n_replaced);
    n_hooks += n_pred_hooks; n_replaced += n_pred_replaced;
    fprintf(stderr, "tier1: %u packets, %d pictures, %lu table initialisations hooked (%lu entries replaced), %dx%d %s, %lu pictures deblocked per picture (%lu with strengths from the device), %lu coding tree units in %lu slices, %lu intra blocks predicted by the batched wrapper\n", n, shown, n_hooks, n_replaced, c->width, c->height,
            av_get_pix_fmt_name(c->pix_fmt), mi355_hevc_lf_bridge_pictures ? mi355_hevc_lf_bridge_pictures() : 0ul, mi355_hevc_lf_bridge_bs_pictures ? mi355_hevc_lf_bridge_bs_pictures() : 0ul, n_ctus, n_slice_ends, n_intra_dev);
    fclose(out);
    return n_hooks >= 3 ? 0 : 8;
}
