/*
 * hevc_bridge_main.c — a small host for the HEVC Tier-2 bridge (contrib/libav/mi355_hevc_bridge.c + mi355_hevc_lf_bridge.c): the
 * reference's own HEVC decoder, linked with the bridges' wraps, decodes a demuxed stream; every picture it outputs is written as
 * raw planes and the bridge's counters are printed as one JSON line.  MI355_HEVC_RECON_PLAIN=1 MI355_HEVC_LF_PLAIN=1: the
 * comparison run (the reference's own reconstruction and filters).
 *   usage: hevc_bridge <in.samples> <out.yuv | -> [loops [threads]]      (threads: that many decoders at once, one per thread, same stream)
 *   samples: u32 extradata size, extradata, u32 packet count, count x (u32 size, bytes)
 */
#include <pthread.h>
#include <stdint.h>
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
void mi355_hevc_bridge_batch_stats(unsigned long *sets, unsigned long *pictures);

static uint32_t rd32(const uint8_t **p) { uint32_t v; memcpy(&v, *p, 4); *p += 4; return v; }

/* the reference's avcodec_open2 / close are serialised by the application when no lock manager is registered (static tables are built on first use) */
static pthread_mutex_t open_lock = PTHREAD_MUTEX_INITIALIZER;

typedef struct Job { const uint8_t *data; int loops; FILE *out; long shown; uint64_t hash; int rc; } Job;

/* one decoder after the other over the stream, `loops` times; the pictures of the first pass are hashed (and written, for the thread that has a file) */
static void *decode_thread(void *arg)
{
    Job *j = arg;
    j->hash = 1469598103934665603ull;
    for (int loop = 0; loop < j->loops; loop++) {
        const uint8_t *p = j->data;
        AVCodecContext *c = avcodec_alloc_context3(&ff_hevc_decoder);
        const uint32_t el = rd32(&p);
        c->extradata = av_mallocz(el + AV_INPUT_BUFFER_PADDING_SIZE);
        c->extradata_size = (int)el;
        memcpy(c->extradata, p, el); p += el;
        c->thread_count = 1;
        c->flags |= AV_CODEC_FLAG_BITEXACT;
        pthread_mutex_lock(&open_lock);
        const int opened = avcodec_open2(c, &ff_hevc_decoder, NULL);
        pthread_mutex_unlock(&open_lock);
        if (opened < 0) { fprintf(stderr, "open failed\n"); j->rc = 5; return NULL; }
        const uint32_t n = rd32(&p);
        AVFrame *fr = av_frame_alloc();
        for (uint32_t i = 0; i <= n; i++) {
            AVPacket pkt;
            av_init_packet(&pkt);
            pkt.data = NULL; pkt.size = 0;
            if (i < n) {
                const uint32_t len = rd32(&p);
                if (av_new_packet(&pkt, (int)len) < 0) { j->rc = 6; return NULL; }
                memcpy(pkt.data, p, len); p += len;
            }
            if (avcodec_send_packet(c, i < n ? &pkt : NULL) < 0) { fprintf(stderr, "send_packet failed\n"); j->rc = 7; return NULL; }
            while (avcodec_receive_frame(c, fr) >= 0) {
                if (loop == 0)
                    for (int pl = 0; pl < 3; pl++) {
                        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fr->format);
                        const int w = pl ? fr->width >> d->log2_chroma_w : fr->width, h = pl ? fr->height >> d->log2_chroma_h : fr->height;
                        const size_t bps = (size_t)(d->comp[0].depth + 7) >> 3;
                        for (int y = 0; y < h; y++) {
                            const uint8_t *row = fr->data[pl] + (size_t)y * fr->linesize[pl];
                            if (j->out) fwrite(row, bps, (size_t)w, j->out);
                            for (size_t k = 0; k < bps * (size_t)w; k++) j->hash = (j->hash ^ row[k]) * 1099511628211ull;
                        }
                    }
                j->shown++;
                av_frame_unref(fr);
            }
            if (i < n) av_packet_unref(&pkt);
        }
        av_frame_free(&fr);
        pthread_mutex_lock(&open_lock);
        avcodec_free_context(&c);
        pthread_mutex_unlock(&open_lock);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.yuv|- [loops [threads]]\n", argv[0]); return 1; }
    const int loops = argc > 3 ? atoi(argv[3]) : 1;
    int threads = argc > 4 ? atoi(argv[4]) : 1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 1;
    fseek(in, 0, SEEK_END);
    const long size = ftell(in);
    fseek(in, 0, SEEK_SET);
    uint8_t *data = malloc((size_t)size);
    if (!data || fread(data, 1, (size_t)size, in) != (size_t)size) return 1;
    fclose(in);
    FILE *out = strcmp(argv[2], "-") ? fopen(argv[2], "wb") : NULL;
    Job *jobs = calloc((size_t)threads, sizeof(*jobs));
    pthread_t *tid = calloc((size_t)threads, sizeof(*tid));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* every thread is a decoder of its own (one context, thread_count 1) over the same stream: what a host with many streams looks like to the bridge */
    for (int t = 0; t < threads; t++) {
        jobs[t].data = data; jobs[t].loops = loops; jobs[t].out = t == 0 ? out : NULL;
        if (threads == 1) decode_thread(&jobs[t]);
        else if (pthread_create(&tid[t], NULL, decode_thread, &jobs[t])) return 8;
    }
    long shown = 0;
    int same = 1, rc = 0;
    for (int t = 0; t < threads; t++) {
        if (threads > 1) pthread_join(tid[t], NULL);
        shown += jobs[t].shown;
        same &= jobs[t].hash == jobs[0].hash && jobs[t].shown == jobs[0].shown;
        rc |= jobs[t].rc;
    }
    if (rc) return rc;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out) fclose(out);
    unsigned long pics = 0, dev = 0, up = 0, launches = 0, levels = 0;
    mi355_hevc_bridge_stats(&pics, &dev, &up, &launches, &levels);
    unsigned long sets = 0, set_pics = 0;
    mi355_hevc_bridge_batch_stats(&sets, &set_pics);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("{\"loops\": %d, \"threads\": %d, \"outputs_identical\": %s, \"launch_sets\": %lu, \"pictures_per_launch_set\": %.2f, \"pictures_output\": %ld, \"pictures_decoded\": %lu, \"pictures_reconstructed_on_device\": %lu, \"pictures_filtered_on_device\": %lu, "
           "\"reference_uploads\": %lu, \"reconstruction_launches\": %lu, \"dependency_levels\": %lu, \"seconds\": %.4f, \"pictures_per_s\": %.1f}\n",
           loops, threads, same ? "true" : "false", sets, sets ? (double)set_pics / (double)sets : 0.0, shown, pics, dev, mi355_hevc_lf_bridge_pictures(), up, launches, levels, sec, (double)shown / sec);
    return 0;
}
