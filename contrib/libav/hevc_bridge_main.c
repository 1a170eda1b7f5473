/*
 * hevc_bridge_main.c — a small host for the HEVC Tier-2 bridge (contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c): the
 * reference's own HEVC decoder, linked with the bridges' wraps, decodes a demuxed stream; every picture it outputs is written as
 * raw planes and the bridge's counters are printed as one JSON line.  MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1: the
 * comparison run (the reference's own reconstruction and filters).
 *   usage: hevc_bridge <in.samples> <out.yuv | -> [loops]
 *   samples: u32 extradata size, extradata, u32 packet count, count x (u32 size, bytes)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"
#include "libavcodec/avcodec.h"

extern AVCodec ff_hevc_decoder;
void mi355_hevc_bridge_stats(unsigned long *pictures, unsigned long *on_device, unsigned long *uploads, unsigned long *launches, unsigned long *levels);
unsigned long mi355_hevc_lf_bridge_pictures(void);

static uint32_t rd32(const uint8_t **p) { uint32_t v; memcpy(&v, *p, 4); *p += 4; return v; }

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.yuv|- [loops]\n", argv[0]); return 1; }
    const int loops = argc > 3 ? atoi(argv[3]) : 1;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 1;
    fseek(in, 0, SEEK_END);
    const long size = ftell(in);
    fseek(in, 0, SEEK_SET);
    uint8_t *data = malloc((size_t)size);
    if (!data || fread(data, 1, (size_t)size, in) != (size_t)size) return 1;
    fclose(in);
    FILE *out = strcmp(argv[2], "-") ? fopen(argv[2], "wb") : NULL;
    long shown = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int loop = 0; loop < loops; loop++) {
        const uint8_t *p = data;
        AVCodecContext *c = avcodec_alloc_context3(&ff_hevc_decoder);
        const uint32_t el = rd32(&p);
        c->extradata = av_mallocz(el + AV_INPUT_BUFFER_PADDING_SIZE);
        c->extradata_size = (int)el;
        memcpy(c->extradata, p, el); p += el;
        c->thread_count = 1;
        c->flags |= AV_CODEC_FLAG_BITEXACT;
        if (avcodec_open2(c, &ff_hevc_decoder, NULL) < 0) { fprintf(stderr, "open failed\n"); return 5; }
        const uint32_t n = rd32(&p);
        AVFrame *fr = av_frame_alloc();
        for (uint32_t i = 0; i <= n; i++) {
            AVPacket pkt;
            av_init_packet(&pkt);
            pkt.data = NULL; pkt.size = 0;
            if (i < n) {
                const uint32_t len = rd32(&p);
                if (av_new_packet(&pkt, (int)len) < 0) return 6;
                memcpy(pkt.data, p, len); p += len;
            }
            if (avcodec_send_packet(c, i < n ? &pkt : NULL) < 0) { fprintf(stderr, "send_packet failed\n"); return 7; }
            while (avcodec_receive_frame(c, fr) >= 0) {
                if (out && loop == 0)
                    for (int pl = 0; pl < 3; pl++) {
                        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fr->format);
                        const int w = pl ? fr->width >> d->log2_chroma_w : fr->width, h = pl ? fr->height >> d->log2_chroma_h : fr->height;
                        const size_t bps = (size_t)(d->comp[0].depth + 7) >> 3;
                        for (int y = 0; y < h; y++) fwrite(fr->data[pl] + (size_t)y * fr->linesize[pl], bps, (size_t)w, out);
                    }
                shown++;
                av_frame_unref(fr);
            }
            if (i < n) av_packet_unref(&pkt);
        }
        av_frame_free(&fr);
        avcodec_free_context(&c);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out) fclose(out);
    unsigned long pics = 0, dev = 0, up = 0, launches = 0, levels = 0;
    mi355_hevc_bridge_stats(&pics, &dev, &up, &launches, &levels);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("{\"loops\": %d, \"pictures_output\": %ld, \"pictures_decoded\": %lu, \"pictures_reconstructed_on_device\": %lu, \"pictures_filtered_on_device\": %lu, "
           "\"reference_uploads\": %lu, \"reconstruction_launches\": %lu, \"dependency_levels\": %lu, \"seconds\": %.4f, \"pictures_per_s\": %.1f}\n",
           loops, shown, pics, dev, mi355_hevc_lf_bridge_pictures(), up, launches, levels, sec, (double)shown / sec);
    return 0;
}
