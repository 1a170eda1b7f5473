/*
 * h264_pipelines.hip — how a host should run LARGE batches through the three passes (include/mi355_h264_frame.h, mi355_h264_pipelines_*): the batch as a few
 * shares, each through reconstruction / intra / loop filter on a HIP stream of its own, the shares' reconstruction launches taking turns.
 *
 * Why: the reconstruction (k_recon_inter_tiled) waits on the memory pipeline, the loop filter and the intra pass (k_deblock_tiled, k_recon_intra) on the vector
 * pipe.  One after the other over the whole batch they leave the other resource idle: 13.4 - 13.7 ms per 2048 1080p pictures.  Shares on streams overlap them, and a
 * stream-wait between the shares' reconstruction launches keeps two reconstructions from running side by side (both would wait on the memory pipeline):
 * three shares 12.0 - 12.2 ms (profiles/r05_experiments.md 12; two 12.6 - 12.7, four and more slower again: smaller launches, longer chains).
 * The streams and the turn live across calls: the first share's reconstruction of a call waits for the last share's of the call before — and, on its own stream, for
 * its own share's loop filter of the call before; nothing else joins the calls, so consecutive batches run into each other the same way.
 * Host code only: no kernel here.  What the reference does at this place: nothing comparable — its frame threads (libavcodec/pthread_frame.c) overlap whole
 * pictures of ONE stream on CPU cores; this overlaps passes of MANY streams' pictures on one device.
 */
#include "mi355_rt.h"
#include "mi355_h264_frame.h"

#include <vector>

struct mi355_h264_pipelines {
    int shares = 0, turns = 1, device = -1;
    bool started = false;                       /* a turn event has been recorded: there is something to wait for */
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> turn;               /* recorded behind share i's reconstruction launch */
    std::vector<hipEvent_t> pool;               /* timing events of the calls since the last collect(): four per share and call */
    std::vector<hipEvent_t> spare;
    bool timing = false;
};

extern "C" mi355_h264_pipelines *mi355_h264_pipelines_create(int shares, int turns)
{
    if (shares < 1 || shares > 16 || !mi355::bind()) return nullptr;
    mi355_h264_pipelines *p = new mi355_h264_pipelines;
    p->shares = shares; p->turns = turns != 0; p->device = mi355::current_device();
    for (int i = 0; i < shares; i++) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            if (st) (void)hipStreamDestroy(st);
            for (hipStream_t s : p->stream) (void)hipStreamDestroy(s);
            for (hipEvent_t e : p->turn) (void)hipEventDestroy(e);
            delete p;
            return nullptr;
        }
        p->stream.push_back(st); p->turn.push_back(ev);
    }
    return p;
}

extern "C" int mi355_h264_pipelines_sync(mi355_h264_pipelines *p)
{
    if (!p) return -1;
    int rc = 0;
    for (hipStream_t st : p->stream) if (hipStreamSynchronize(st) != hipSuccess) rc = -2;
    return rc;
}

extern "C" void mi355_h264_pipelines_destroy(mi355_h264_pipelines *p)
{
    if (!p) return;
    (void)mi355_h264_pipelines_sync(p);
    for (hipStream_t st : p->stream) (void)hipStreamDestroy(st);
    for (hipEvent_t e : p->turn) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->spare) (void)hipEventDestroy(e);
    delete p;
}

extern "C" void mi355_h264_pipelines_timing(mi355_h264_pipelines *p, int on) { if (p) p->timing = on != 0; }

extern "C" int mi355_h264_pipelines_share(const mi355_h264_pipelines *p, int nframes, int share, int *first, int *count)
{
    if (!p || share < 0 || share >= p->shares || nframes < 0) return -1;
    const int q = nframes / p->shares, r = nframes % p->shares;             /* 2048 as 683 + 683 + 682 */
    if (first) *first = share * q + (share < r ? share : r);
    if (count) *count = q + (share < r ? 1 : 0);
    return 0;
}

extern "C" int mi355_h264_pipelines_decode_dev(mi355_h264_pipelines *p, const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                               int max_intra_level, const int32_t *level_widths, int layouts)
{
    if (!p || !d_frames || nframes <= 0 || !mi355::bind()) return -1;
    auto stamp = [&](hipStream_t st) -> int {
        if (!p->timing) return 0;
        hipEvent_t e = nullptr;
        if (!p->spare.empty()) { e = p->spare.back(); p->spare.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) return -4;
        p->pool.push_back(e);
        return hipEventRecord(e, st) == hipSuccess ? 0 : -4;
    };
    for (int i = 0; i < p->shares; i++) {
        int first = 0, count = 0;
        (void)mi355_h264_pipelines_share(p, nframes, i, &first, &count);
        hipStream_t st = p->stream[(size_t)i];
        if (count <= 0) {                       /* fewer pictures than shares: the turn still passes through */
            if (p->turns && p->started && p->shares > 1) MI355_TRY(hipStreamWaitEvent(st, p->turn[(size_t)((i + p->shares - 1) % p->shares)], 0), -4);
            if (p->turns) { MI355_TRY(hipEventRecord(p->turn[(size_t)i], st), -4); p->started = true; }
            if (p->timing) for (int k = 0; k < 4; k++) { const int rc = stamp(st); if (rc) return rc; }
            continue;
        }
        const mi355_h264_frame *d = d_frames + first;
        if (p->turns && p->shares > 1 && (i > 0 || p->started)) MI355_TRY(hipStreamWaitEvent(st, p->turn[(size_t)((i + p->shares - 1) % p->shares)], 0), -4);
        int rc = stamp(st);
        if (rc) return rc;
        rc = mi355_h264_recon_inter_layouts_dev(d, count, max_mb_width, max_mb_height, layouts, st);
        if (rc) return rc;
        if (p->turns) { MI355_TRY(hipEventRecord(p->turn[(size_t)i], st), -4); p->started = true; }
        if ((rc = stamp(st))) return rc;
        rc = mi355_h264_recon_intra_all_dev(d, count, max_mb_width, max_mb_height, max_intra_level, level_widths, st);
        if (rc) return rc;
        if ((rc = stamp(st))) return rc;
        rc = mi355_h264_deblock_layouts_dev(d, count, max_mb_width, max_mb_height, layouts, st);
        if (rc) return rc;
        if ((rc = stamp(st))) return rc;
    }
    return 0;
}

/* waits for everything submitted; sums[0..2] += the durations (ms) of the reconstruction / intra / loop-filter passes of every share and call since the last collect
 * (each measured on its own stream: they overlap), *launches += the number of (share, call) pairs.  Needs mi355_h264_pipelines_timing(p, 1) before the calls. */
extern "C" int mi355_h264_pipelines_collect(mi355_h264_pipelines *p, double sums[3], int *launches)
{
    if (!p) return -1;
    const int rc = mi355_h264_pipelines_sync(p);
    if (rc) return rc;
    for (size_t i = 0; i + 3 < p->pool.size(); i += 4) {
        for (int k = 0; k < 3; k++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p->pool[i + (size_t)k], p->pool[i + (size_t)k + 1]) != hipSuccess) return -4;
            if (sums) sums[k] += ms;
        }
        if (launches) ++*launches;
    }
    p->spare.insert(p->spare.end(), p->pool.begin(), p->pool.end());
    p->pool.clear();
    return 0;
}
