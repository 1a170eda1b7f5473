/*
 * h264_pipelines.hip — how a host should run LARGE batches through the three passes (include/mi355_h264_frame.h, mi355_h264_pipelines_*): the batch as a few
 * shares, each through reconstruction / intra / loop filter on a HIP stream of its own, the shares' reconstruction launches taking turns.
 *
 * Why: the reconstruction (k_recon_inter_tiled) waits on the memory pipeline, the loop filter and the intra pass (k_deblock_tiled, k_recon_intra) on the vector
 * pipe.  One after the other over the whole batch they leave the other resource idle: 13.4 - 13.7 ms per 2048 1080p pictures.  Shares on streams overlap them, and a
 * stream-wait between the shares' reconstruction launches keeps two reconstructions from running side by side (both would wait on the memory pipeline):
 * three shares 12.0 - 12.2 ms (profiles/r05_experiments.md 12; two 12.6 - 12.7, four and more slower again: smaller launches, longer chains).
 * The streams and the turn live across calls: the first share's reconstruction of a call waits for the last share's of the call before — and, on its own stream, for
 * its own share's loop filter of the call before; consecutive calls on the SAME batch (array and count: every picture stays in its share) are joined by nothing else
 * and run into each other the same way.  A call on another batch first waits, in every share, for every share's loop filter of the call before (a picture may
 * reference one that another share decoded): mi355_h264_pipelines_join().
 * Host code only: no kernel here.  What the reference does at this place: nothing comparable — its frame threads (libavcodec/pthread_frame.c) overlap whole
 * pictures of ONE stream on CPU cores; this overlaps passes of MANY streams' pictures on one device.
 */
#include "mi355_rt.h"
#include "mi355_h264_frame.h"

#include <vector>

struct mi355_h264_pipelines {
    int shares = 0, turns = 1, device = -1;
    bool started = false;                       /* a turn event has been recorded: there is something to wait for */
    bool started_done = false;                  /* ... a done event */
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> turn;               /* recorded behind share i's reconstruction launch */
    std::vector<hipEvent_t> done;               /* recorded behind share i's loop filter */
    int join = 1;                               /* mi355_h264_pipelines_join() */
    const void *last_frames = nullptr;          /* the previous call's batch: the same array and count = the same pictures in the same shares */
    int last_nframes = -1;
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
        hipEvent_t ev = nullptr, dn = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&dn, hipEventDisableTiming) != hipSuccess) {
            if (st) (void)hipStreamDestroy(st);
            if (ev) (void)hipEventDestroy(ev);
            for (hipStream_t s : p->stream) (void)hipStreamDestroy(s);
            for (hipEvent_t e : p->turn) (void)hipEventDestroy(e);
            for (hipEvent_t e : p->done) (void)hipEventDestroy(e);
            delete p;
            return nullptr;
        }
        p->stream.push_back(st); p->turn.push_back(ev); p->done.push_back(dn);
    }
    return p;
}

extern "C" int mi355_h264_pipelines_sync(mi355_h264_pipelines *p)
{
    if (!p) return -1;
    int rc = 0;
    for (hipStream_t st : p->stream) if (hipStreamSynchronize(st) != hipSuccess) rc = -2;
    return rc ? rc : mi355::fault_after_wait();         /* a kernel that gave up said so in the device's error word (include/mi355dsp.h) */
}

extern "C" void mi355_h264_pipelines_destroy(mi355_h264_pipelines *p)
{
    if (!p) return;
    (void)mi355_h264_pipelines_sync(p);
    for (hipStream_t st : p->stream) { mi355::sync_words_release(st); (void)hipStreamDestroy(st); }     /* the streams' counter buffers go with them */
    for (hipEvent_t e : p->turn) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->done) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->spare) (void)hipEventDestroy(e);
    delete p;
}

extern "C" void mi355_h264_pipelines_timing(mi355_h264_pipelines *p, int on) { if (p) p->timing = on != 0; }
extern "C" void mi355_h264_pipelines_join(mi355_h264_pipelines *p, int mode) { if (p && mode >= 0 && mode <= 2) p->join = mode; }

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
    /* Which calls are joined.  A picture of this call may reference a picture the call before decoded in ANOTHER share; that share's loop filter is ordered before
     * nothing here but its own stream's next launches.  So: when the batch is not the one of the call before (another array or count: pictures change shares), or
     * always (join 2), every share first waits for every share's loop filter of the call before.  join 0: never — the caller keeps a stream's pictures in one share. */
    const bool joined = p->started_done && (p->join == 2 || (p->join == 1 && (d_frames != p->last_frames || nframes != p->last_nframes)));
    const size_t pool0 = p->pool.size();
    auto fail = [&](int rc) -> int {          /* the stamps of a call that failed leave the pool (collect() pairs them four by four) */
        while (p->pool.size() > pool0) { p->spare.push_back(p->pool.back()); p->pool.pop_back(); }
        return rc;
    };
    auto stamp = [&](hipStream_t st) -> int {
        if (!p->timing) return 0;
        hipEvent_t e = nullptr;
        if (!p->spare.empty()) { e = p->spare.back(); p->spare.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) return -4;
        if (hipEventRecord(e, st) != hipSuccess) { p->spare.push_back(e); return -4; }
        p->pool.push_back(e);
        return 0;
    };
#define PIPE_TRY(call) do { if ((call) != hipSuccess) return fail(-4); } while (0)
    if (joined)
        for (int i = 0; i < p->shares; i++)
            for (int k = 0; k < p->shares; k++)
                if (k != i) PIPE_TRY(hipStreamWaitEvent(p->stream[(size_t)i], p->done[(size_t)k], 0));
    for (int i = 0; i < p->shares; i++) {
        int first = 0, count = 0;
        (void)mi355_h264_pipelines_share(p, nframes, i, &first, &count);
        hipStream_t st = p->stream[(size_t)i];
        if (count <= 0) {                       /* fewer pictures than shares: the turn still passes through */
            if (p->turns && p->started && p->shares > 1) PIPE_TRY(hipStreamWaitEvent(st, p->turn[(size_t)((i + p->shares - 1) % p->shares)], 0));
            if (p->turns) { PIPE_TRY(hipEventRecord(p->turn[(size_t)i], st)); p->started = true; }
            if (p->timing) for (int k = 0; k < 4; k++) { const int rc = stamp(st); if (rc) return fail(rc); }
            PIPE_TRY(hipEventRecord(p->done[(size_t)i], st));
            continue;
        }
        const mi355_h264_frame *d = d_frames + first;
        if (p->turns && p->shares > 1 && (i > 0 || p->started)) PIPE_TRY(hipStreamWaitEvent(st, p->turn[(size_t)((i + p->shares - 1) % p->shares)], 0));
        int rc = stamp(st);
        if (rc) return fail(rc);
        rc = mi355_h264_recon_inter_layouts_dev(d, count, max_mb_width, max_mb_height, layouts, st);
        if (rc) return fail(rc);
        if (p->turns) { PIPE_TRY(hipEventRecord(p->turn[(size_t)i], st)); p->started = true; }
        if ((rc = stamp(st))) return fail(rc);
        rc = mi355_h264_recon_intra_all_dev(d, count, max_mb_width, max_mb_height, max_intra_level, level_widths, st);
        if (rc) return fail(rc);
        if ((rc = stamp(st))) return fail(rc);
        rc = mi355_h264_deblock_layouts_dev(d, count, max_mb_width, max_mb_height, layouts, st);
        if (rc) return fail(rc);
        if ((rc = stamp(st))) return fail(rc);
        PIPE_TRY(hipEventRecord(p->done[(size_t)i], st));
    }
#undef PIPE_TRY
    p->started_done = true;
    p->last_frames = d_frames; p->last_nframes = nframes;
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
