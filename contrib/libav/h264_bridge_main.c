/*
 * h264_bridge_main.c — a small host around the reference's H.264 decoder + the Tier-2 bridge (mi355_h264_bridge.c):
 * decodes a demuxed elementary stream on N threads (N independent decoder instances = N streams, each with its own
 * bridge state and HIP stream), optionally K times in a row, and reports end-to-end pictures per second.
 *   usage: h264_bridge <in.samples[,second.samples]> <out.yuv | -> [threads [loops]]
 *   two inputs: thread t decodes input t % 2 (streams of different sizes and formats meet in the dispatcher's launch
 *   sets); thread 1 writes its pictures to <out.yuv>.1
 *   in.samples: u32 extradata_len, extradata (avcC), u32 n, then n x {u32 len, bytes}   (tests/golden/mp4_samples.py)
 *   MI355_BRIDGE_PLAIN=1: the bridge steps aside at once (the reference's C path: the comparison run);
 *   MI355_BRIDGE_DIRECT=1: every thread drives its own HIP stream instead of handing pictures to the dispatcher;
 *   MI355_BRIDGE_LAZY=1: a thread waits for a picture only when the decoder is about to output it.
 * Thread 0 of loop 0 writes the decoded pictures (coded size, planar) to out.yuv.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "libavutil/mem.h"
#include "libavutil/pixdesc.h"
#include "libavcodec/avcodec.h"

extern AVCodec ff_h264_decoder;
void mi355_h264_bridge_stats(unsigned long *pictures, unsigned long *staging_waits, int *active);
void mi355_h264_bridge_drain(void);
void mi355_h264_bridge_batch_stats(unsigned long *batches, unsigned long *pictures);

static uint8_t *file_data[2];
static size_t file_size[2];
static int nfiles = 1;
static int loops = 1;
static const char *out_path;

typedef struct { int id; long shown; unsigned long dev_pictures, waits; int active; int rc; } Arg;

/* the reference's avcodec_open2 / close are serialised by the application when no lock manager is registered
 * (libavcodec/utils.c: its one-time static table set-up is not re-entrant) */
static pthread_mutex_t open_lock = PTHREAD_MUTEX_INITIALIZER;

static uint32_t rd32(const uint8_t **p) { uint32_t v; memcpy(&v, *p, 4); *p += 4; return v; }

static void *decode_thread(void *vp)
{
    Arg *a = vp;
    FILE *out = NULL;
    if (a->id < nfiles && out_path && strcmp(out_path, "-")) {
        char name[4096];
        snprintf(name, sizeof(name), a->id ? "%s.%d" : "%s", out_path, a->id);
        out = fopen(name, "wb");
    }
    for (int loop = 0; loop < loops; loop++) {
        const uint8_t *p = file_data[a->id % nfiles];
        AVCodecContext *c = avcodec_alloc_context3(&ff_h264_decoder);
        const uint32_t el = rd32(&p);
        c->extradata = av_mallocz(el + AV_INPUT_BUFFER_PADDING_SIZE);
        c->extradata_size = (int)el;
        memcpy(c->extradata, p, el); p += el;
        c->thread_count = 1;
        if (getenv("MI355_HARNESS_SKIP_LOOP_FILTER")) c->skip_loop_filter = AVDISCARD_ALL;      /* developer switch: both sides without the in-loop filter */
        c->flags |= AV_CODEC_FLAG_BITEXACT;
        pthread_mutex_lock(&open_lock);
        const int opened = avcodec_open2(c, &ff_h264_decoder, NULL);
        pthread_mutex_unlock(&open_lock);
        if (opened < 0) { a->rc = 5; return NULL; }
        const uint32_t n = rd32(&p);
        AVFrame *fr = av_frame_alloc();
        for (uint32_t i = 0; i <= n; i++) {
            AVPacket pkt;
            av_init_packet(&pkt);
            pkt.data = NULL; pkt.size = 0;
            if (i < n) {
                const uint32_t len = rd32(&p);
                if (av_new_packet(&pkt, (int)len) < 0) { a->rc = 6; return NULL; }
                memcpy(pkt.data, p, len); p += len;
            } else {
                mi355_h264_bridge_drain();
            }
            if (avcodec_send_packet(c, i < n ? &pkt : NULL) < 0) { a->rc = 7; return NULL; }
            while (avcodec_receive_frame(c, fr) >= 0) {
                if (out && loop == 0)
                    for (int pl = 0; pl < 3; pl++) {
                        const AVPixFmtDescriptor *d = av_pix_fmt_desc_get(fr->format);
                        const int w = pl ? fr->width >> d->log2_chroma_w : fr->width, h = pl ? fr->height >> d->log2_chroma_h : fr->height;
                        const size_t bps = (size_t)(d->comp[0].depth + 7) >> 3;      /* 9 / 10-bit pictures: two bytes per sample */
                        for (int y = 0; y < h; y++) fwrite(fr->data[pl] + (size_t)y * fr->linesize[pl], bps, (size_t)w, out);
                    }
                a->shown++;
                av_frame_unref(fr);
            }
            if (i < n) av_packet_unref(&pkt);
        }
        av_frame_free(&fr);
        pthread_mutex_lock(&open_lock);
        avcodec_free_context(&c);
        pthread_mutex_unlock(&open_lock);
    }
    mi355_h264_bridge_stats(&a->dev_pictures, &a->waits, &a->active);
    if (out) fclose(out);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.samples out.yuv|- [threads [loops]]\n", argv[0]); return 1; }
    const int nthreads = argc > 3 ? atoi(argv[3]) : 1;
    loops = argc > 4 ? atoi(argv[4]) : 1;
    out_path = argv[2];
    char *names = strdup(argv[1]), *second = strchr(names, ',');
    if (second) { *second++ = 0; nfiles = 2; }
    for (int k = 0; k < nfiles; k++) {
        FILE *in = fopen(k ? second : names, "rb");
        if (!in) return 1;
        fseek(in, 0, SEEK_END); file_size[k] = (size_t)ftell(in); fseek(in, 0, SEEK_SET);
        file_data[k] = malloc(file_size[k]);
        if (fread(file_data[k], 1, file_size[k], in) != file_size[k]) return 4;
        fclose(in);
    }
    pthread_t *th = calloc((size_t)nthreads, sizeof(*th));
    Arg *args = calloc((size_t)nthreads, sizeof(*args));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) { args[t].id = t; pthread_create(&th[t], NULL, decode_thread, &args[t]); }
    long shown = 0; unsigned long dev = 0, waits = 0; int rc = 0, active = 0;
    for (int t = 0; t < nthreads; t++) {
        pthread_join(th[t], NULL);
        shown += args[t].shown; dev += args[t].dev_pictures; waits += args[t].waits; rc |= args[t].rc; active += args[t].active > 0;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    unsigned long batches = 0, batched = 0;
    mi355_h264_bridge_batch_stats(&batches, &batched);
    printf("{\"threads\": %d, \"loops\": %d, \"pictures_output\": %ld, \"pictures_on_device\": %lu, \"bridges_active\": %d, "
           "\"staging_waits\": %lu, \"launch_sets\": %lu, \"pictures_per_launch_set\": %.2f, \"seconds\": %.4f, \"pictures_per_s\": %.1f}\n",
           nthreads, loops, shown, dev, active, waits, batches, batches ? (double)batched / (double)batches : 0.0, s, (double)shown / s);
    return rc;
}
