/*
 * ref_h264_tier1_main.c — TEST INFRASTRUCTURE.  The REFERENCE's own H.264 decoder (its objects built in
 * place by oracle/Makefile) with its five DSP init functions interposed by the linker exactly as
 * INTEGRATION.md §2 describes: every table is filled by the reference's C init and then overridden by
 * this project's ff_*_init_mi355x hooks.  Linked against the SIMT-emulated build of the product sources
 * (tests/_emu/libmi355dsp_emu.so) so that the whole decoder runs through the Tier-1 entry points on a
 * machine without a GPU; the decoded pictures are compared with the unmodified reference decoder's.
 *
 * usage: ref_h264_tier1 <in.samples> <out.yuv>     (samples format: see ref_h264_export.c)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavutil/mem.h"
#include "libavcodec/avcodec.h"
#include "libavcodec/h264dsp.h"
#include "libavcodec/h264qpel.h"
#include "libavcodec/h264chroma.h"
#include "libavcodec/h264pred.h"
#include "libavcodec/videodsp.h"
#include "../include/mi355dsp.h"      /* the table structs are skipped: the reference's headers came first */

#include "libavutil/pixdesc.h"

extern AVCodec ff_h264_decoder;
/* the wraps themselves are PRODUCT code: contrib/libav/mi355_wrap.c, compiled into this binary from there */
void mi355_wrap_stats(unsigned long *tables_hooked, unsigned long *entries_replaced);

static uint32_t get_u32(FILE *f) { uint32_t v = 0; if (fread(&v, 4, 1, f) != 1) exit(4); return v; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.yuv\n", argv[0]); return 1; }
    if (!getenv("MI355_TIER1_PLAIN") && mi355_init(0) != 0) { fprintf(stderr, "mi355_init failed\n"); return 2; }
    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
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
    unsigned long n_hooks = 0;
    mi355_wrap_stats(&n_hooks, NULL);
    fprintf(stderr, "tier1: %u packets, %d pictures, %lu table initialisations hooked, %dx%d %s\n", n, shown, n_hooks, c->width, c->height,
            av_get_pix_fmt_name(c->pix_fmt));
    fclose(out);
    return n_hooks >= 5 ? 0 : 8;
}
