/*
 * mi355_h264_bridge.c — the H.264 Tier-2 bridge: PRODUCT glue that lives beside the reference's decoder.
 *
 * Compiled against the reference's own headers and linked into its decoder with
 *   -Wl,--wrap=ff_h264_hl_decode_mb,--wrap=ff_h264_field_end,--wrap=ff_h264_filter_mb,--wrap=ff_h264_filter_mb_fast,--wrap=ff_h264_flush_change
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
 *       -> the picture is SUBMITTED: the kernels read the staging block in place (pinned, device-visible memory: no copy
 *          into HBM first), mi355_h264_decode_frames_levels_dev(), decoded picture -> a pinned buffer, from which the
 *          decoder thread copies it into the AVFrame the decoder will output.  The host waits only when the picture the
 *          decoder is about to output is not finished (always, unless MI355_BRIDGE_LAZY=1, which trusts
 *          h->output_frame), or when it needs the staging set again.
 *
 * Two ways to submit:
 *   batched (default)   N decoder threads = N streams; a picture is handed to ONE dispatcher thread, which owns the only
 *                       HIP stream: it takes what the decoder threads have queued (at most one picture per stream) and
 *                       issues, for the whole batch, one descriptor copy, ONE set of kernel launches
 *                       (mi355_h264_decode_frames_levels_dev over the pictures of all streams) and one launch that
 *                       writes the finished pictures to the streams' pinned buffers (mi355_copy_batch_dev); up to four
 *                       such launch sets in flight, each on its own HIP stream (a small set is a chain of short
 *                       launches: several chains overlap on the device).  Decoder threads make no runtime calls after set-up, and the dispatcher makes a fixed
 *                       number per batch, so the per-picture cost of launches and runtime locks is shared by the streams
 *                       that are decoding at the same time — the regime the batched kernels are built for.
 *   direct (MI355_BRIDGE_DIRECT=1)  every decoder thread drives its own HIP stream, one picture per launch set.
 *   sessions (MI355_BRIDGE_SESSION=1)  the reference-side caller of the whole-frame façade (include/mi355_h264_session.h, the
 *                       AVHWAccel-shaped boundary: start_frame / decode_slice / end_frame, libavcodec/avcodec.h:3062-3086,
 *                       call sites h264_slice.c:1534, h264dec.c:591, h264_picture.c:166): at the end of a picture the bridge
 *                       names the surfaces (one per H264Picture), hands the picture's slices — the macroblocks it packed,
 *                       run by run — to mi355_h264_decode_slice(), and takes the result back with mi355_h264_get_frame().
 *                       The session owns surfaces, staging copies, intra schedule and launches; the bridge only parses.
 *                       4:2:0 frame and field pictures (4:4:4 is submitted plane by plane, which sessions do not do).
 *
 * The decoded picture buffer lives in HBM: one device picture per H264Picture the decoder uses, found again through the
 * reference lists' parent pointers; reference samples never cross PCIe.  Two staging sets per stream alternate, so the
 * host can pack picture n + 1 while the copies and kernels of picture n run.
 * Sequences of frame pictures only (frame_mbs_only_flag, 4:2:0) keep their device pictures MACROBLOCK-TILED
 * (mi355_h264_frame.h, MI355_SURFACE_TILED: a macroblock is a run of whole cache lines for the reference fetch, the
 * reconstruction stores and the loop filter); a picture is turned back into lines by the launch that brings it to the
 * pinned buffer (mi355_h264_surface_convert_dev instead of a plain copy).  MI355_BRIDGE_LINEAR=1, sequences that may hold
 * field pictures and 4:4:4 keep planes with line strides.
 *
 * Threads: ONE decoder context per thread and no threading INSIDE a decoder (avctx->thread_count = 1): the bridge finds
 * its state per thread and its references per context — with frame threads a reference decoded by another thread would
 * be read from a host frame that is not complete yet, with slice threads one picture would be packed by several threads.
 * A decoder opened with either kind of threading is left to the reference's C path.
 *
 * 4:4:4 streams (hl_decode_mb_444, h264_mb_template.c:259-345: every plane is decoded like luma — luma interpolation,
 * luma intra modes, luma transforms, the luma loop filter with the plane's own QP) are submitted as THREE passes per
 * picture, one per plane: the plane takes the luma role of a descriptor whose chroma planes point at scratch surfaces, with
 * per-plane records (that plane's coefficient flags, QP and DC multiplier), coefficients and weight tables.  The loop filter
 * takes its boundary strengths from the LUMA coefficient flags for every plane (filter_mb_dir, h264_loopfilter.c:482-713),
 * so planes 1 and 2 get a second record array for the filter pass.  No kernel knows about 4:4:4.
 *
 * High 10 / High 4:2:2 (9 / 10-bit samples, chroma_format_idc 2): the same packing with the reference's own widths — `dctcoef` is
 * 32 bits wide when sps->bit_depth_luma > 8 (sl->mb holds int32, h264dec.h), chroma has sixteen rows in 4:2:2 — and the pictures go
 * through mi355_h264_decode_frames_wide_dev (the second kernel set, libav_amd/csrc/h264_frame_wide.hip) on planes with line strides;
 * a launch set of the dispatcher holds pictures of ONE format.
 *
 * MBAFF frames (macroblock pairs coded as frame or field macroblocks) go through the second kernel set too: the bridge names a field macroblock's
 * references by (frame, parity) — what hl_decode_mb's ref_list[16 + ..] remap does — schedules intra macroblocks pair-wise
 * (mi355_h264_intra_schedule_mbaff) and marks the picture MI355_FRAME_MBAFF; the kernels take a field macroblock's rows as every other line of its pair
 * and filter pairs (k_wide_deblock_mbaff).  An MBAFF slice with IMPLICIT weights is packed with the field macroblocks' own table as well
 * (implicit_weight_field, what the reference's implicit_weight[..][..][1 + parity] holds: h264_frame_wide.hip reads it for field macroblocks).
 *
 * Streams outside the Tier-2 scope (more than 10 bits, separate colour planes), MI355_BRIDGE_PLAIN=1 and any runtime failure make the bridge step
 * aside for that decoder: the reference's own C path continues.  Errors are reported once on stderr; nothing here
 * calls abort().
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/h264dec.h"
#include "libavcodec/h264_ps.h"
#include "libavcodec/mpegutils.h"
#include "mi355dsp.h"
#include "mi355_h264_frame.h"
#include "mi355_h264_session.h"

void __real_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl);
void __real_ff_h264_flush_change(H264Context *h);
int __real_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup);
void __real_ff_h264_filter_mb(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                              uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize);
void __real_ff_h264_filter_mb_fast(const H264Context *h, H264SliceContext *sl, int mb_x, int mb_y, uint8_t *img_y, uint8_t *img_cb,
                                   uint8_t *img_cr, unsigned int linesize, unsigned int uvlinesize);

#define BR_MAX_PICS 40        /* H264_MAX_PICTURE_COUNT (36) + slack */
#define BR_MAX_SLICES 64
#define DISP_MAX_BATCH 256    /* descriptors (pictures of 4:2:0 streams, planes of 4:4:4 ones) per launch set */
#define BR_MAX_PASSES 3       /* 4:4:4: one pass per plane */
#define DISP_MAX_LEVELS 8192  /* mb_width + 2 * mb_height of the largest picture the dispatcher takes */

struct Bridge;
struct Staging;
struct Disp;

typedef struct DevPic {
    const H264Picture *owner;
    uint8_t *plane[3];          /* one allocation, planes back to back (tiled: luma tiles, chroma tiles; [2] unused) */
    int frame_num, poc;         /* what the owner held when this copy was made: the decoder recycles its H264Picture entries, */
    const uint8_t *data0;       /* and a frame it made up for a frame_num gap can sit where an older picture of ours sat */
} DevPic;

typedef struct Submission {     /* a packed picture on its way through the dispatcher */
    struct Bridge *b;
    struct Staging *s;
    int done, rc;
    unsigned long after;        /* launch set that must finish first (the same decoder's previous picture), 0 = none */
    struct Submission *next;
} Submission;

typedef struct Staging {        /* one pinned, device-visible block (mi355_host_alloc): descriptor, records, vectors, coefficients,
                                 * slices, intra schedule.  The kernels read them where the decoder thread wrote them: every
                                 * input byte of a picture is read once, a copy into HBM first would only add a runtime call */
    uint8_t *host;
    size_t size;
    mi355_h264_frame *desc;     /* 2 * npass descriptors: [p] reconstruction of pass p, [npass + p] its loop filter */
    mi355_h264_mb *mb[BR_MAX_PASSES];    /* records the reconstruction of pass p reads */
    mi355_h264_mb *mbd[BR_MAX_PASSES];   /* records its loop filter reads (4:4:4 planes 1, 2: the luma coefficient flags) */
    int16_t *mv[2];
    int16_t *coef[BR_MAX_PASSES];
    mi355_h264_slice *slices[BR_MAX_PASSES];
    uint32_t *ilist;
    int32_t *istart;
    int32_t *widths;            /* host only: macroblocks per intra level */
    int maxl;
    uint8_t *out;               /* pinned: the decoded picture as it comes back (planes with line strides, back to back) */
    DevPic *pic;                /* the picture this set was submitted for */
    int field, parity;          /* it was a field picture: only the lines of its parity go to the frame */
    int nmb_pic, nslices, uses_l1;  /* of the picture submitted from this set (side_upload) */
    int cls;                    /* fmt_class of that picture: pictures of one class share a launch set */
    mi355_surface_job *cvt;     /* pinned, direct mode with tiled device pictures: the conversion job of the copy-back */
    uint8_t *frame_data[3];     /* where it goes: the AVFrame the decoder will hand out */
    int frame_linesize[3];
    mi355_h264_frame *d_desc;   /* direct mode: the descriptors on the device */
    /* the second kernel set's loop filter is one launch per anti-diagonal and reads a macroblock's record and vectors (and its neighbours') in
     * every one of them: those go to HBM first (1.5 MB per 1080p picture; over PCIe each of the ~250 launches of a picture waited for three
     * dependent round trips); everything that is read once (coefficients, the reconstruction's records) stays where the decoder wrote it */
    uint8_t *d_side;            /* one allocation: mbd[npass], mv[2], slices[npass] */
    size_t side_off_mbd[BR_MAX_PASSES], side_off_mv[2], side_off_sl[BR_MAX_PASSES];
    void *done;                 /* direct mode: event after the copy into `out` */
    Submission sub;             /* batched mode */
    int in_flight;
} Staging;

typedef struct Bridge {
    int state;                  /* 0 new, 1 active, -1 stepped aside */
    int soft;                   /* stepped aside because of the SEQUENCE's format: the next sequence is looked at again */
    int lazy, direct;
    int device;                 /* the GPU this decoder was dealt to */
    struct Disp *disp;          /* ... and its dispatcher (batched mode) */
    mi355_h264_session *sess;   /* MI355_BRIDGE_SESSION: pictures go through a whole-frame session; pics[i] is surface i */
    int null_submit;            /* MI355_BRIDGE_NULL (developer): pictures are packed and dropped — times the host side alone */
    int keep_field_idc2;        /* MI355_BRIDGE_KEEP_FIELD_IDC2 (tests): field pictures with disable_deblocking_filter_idc 2 are not handed back to the C path */
    int mb_w, mb_h, nmb;
    void *stream;               /* direct mode */
    Staging st[2];
    int cur;                    /* staging set being packed */
    int open;                   /* a picture is being packed */
    DevPic pics[BR_MAX_PICS];
    int c444, npass;            /* 4:4:4: three passes (planes) per picture */
    int mbaff;                  /* sps->mb_aff: frame pictures of this sequence are MBAFF frames (macroblock pairs) */
    int mbaff_frame;            /* the picture being packed is one */
    int bypass;                 /* sps->transform_bypass: macroblocks with qscale 0 are lossless (MI355_MBF_BYPASS) */
    int wide;                   /* the sequence's format goes through the second kernel set (mi355_h264_decode_frames_wide_dev): more than 8 bits or 4:2:2 */
    int bit_depth, idc;         /* sps->bit_depth_luma, sps->chroma_format_idc of the sequence the bridge is set up for */
    int kidc;                   /* the chroma format the kernels see: idc, or 1 for 4:4:4 (planes in the luma role, scratch chroma) */
    int px, csize;              /* bytes per sample (`pixel`) and per coefficient (`dctcoef`) */
    int crows, ncoef;           /* chroma rows of a macroblock (8; 16 in 4:2:2), coefficients per macroblock (384; 512 in 4:2:2) */
    uint8_t *recon[3];          /* unfiltered reconstruction: Y, Cb, Cr (4:4:4: three full-size planes) */
    uint8_t *scratch_c[2];      /* 4:4:4: what the passes use as chroma planes (never looked at) */
    int stride[2];              /* device surfaces: bytes per line, or per macroblock row of tiles */
    size_t plane_bytes[2];      /* device surfaces: luma plane, 4:2:0 chroma plane (tiled: the one plane of Cb + Cr tiles) */
    int tiled;                  /* device pictures are macroblock-tiled (frame-only 4:2:0 sequences) */
    int lin_stride[2];          /* the pinned picture buffer `out`: bytes per line */
    size_t lin_bytes[2];
    /* per picture */
    int nslices, slice_num_of[BR_MAX_SLICES], uses_l1;
    int mbs_packed;             /* macroblocks the decoder delivered for the picture being packed */
    const H264Picture *slot_pic[MI355_H264_MAX_SLOTS];
    int slot_par[MI355_H264_MAX_SLOTS];   /* field pictures: the parity of the reference field (0 top, 1 bottom); -1: a frame */
    int nslots;
    int field, parity;          /* the picture being packed is one field of its frame (PAFF): which */
    int rows, nmb_pic;          /* its macroblock rows and macroblocks */
    unsigned long pictures, waits;
    unsigned long last_set;     /* launch set (1-based) that holds this decoder's latest picture */
    int max_slices;             /* slices of a picture the tables hold (BR_MAX_SLICES) */
    int releasing;              /* br_fail is giving the buffers back: failures met on the way are only reported */
} Bridge;

static __thread Bridge *br_tls;

static void bridge_release(Bridge *b);
/* The decoder leaves the batched path.  A bridge that was set up gives everything back first: pictures in flight come back
 * into their frames (both sets, MI355_BRIDGE_LAZY or not), the dispatcher stops counting this decoder, pinned and device
 * buffers are freed.  When this happens in the MIDDLE of a picture (a slice beyond the tables of the path) the macroblocks
 * packed so far are lost: that picture is damaged, like one the decoder gave up on, and the message says so. */
static void br_fail(Bridge *b, const char *what)
{
    if (b->releasing) { fprintf(stderr, "mi355 bridge: %s (while giving the buffers back)\n", what); return; }
    const int was = b->state;
    if (was >= 0)
        fprintf(stderr, "mi355 bridge: %s — this decoder continues on the reference's C path%s\n", what,
                was > 0 && b->open ? " (the picture being decoded is damaged)" : "");
    if (was > 0) { b->releasing = 1; bridge_release(b); b->releasing = 0; }
    b->state = -1;
}

static int fmt_class(const Bridge *b) { return b->wide ? 10 * b->bit_depth + b->kidc + (b->mbaff_frame ? 1000 : 0) : 0; }
static void *dalloc(size_t n) { return mi355_malloc(n); }
static size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

/* pictures of one class share a launch set: 0 = the 8-bit 4:2:0 / 4:4:4 kernels, otherwise 10 * bit depth + the kernels' chroma format */
static int fmt_class(const struct Bridge *b);
static size_t picture_bytes(const Bridge *b) { return b->c444 ? 3 * b->plane_bytes[0] : b->plane_bytes[0] + (b->tiled ? 1 : 2) * b->plane_bytes[1]; }
static size_t out_bytes(const Bridge *b) { return b->c444 ? 3 * b->lin_bytes[0] : b->lin_bytes[0] + 2 * b->lin_bytes[1]; }
/* the job that turns device picture `pic` into the lines of the pinned buffer `out` */
static void convert_job(const Bridge *b, const DevPic *pic, uint8_t *out, mi355_surface_job *j)
{
    memset(j, 0, sizeof(*j));
    j->lin[0] = out; j->lin[1] = out + b->lin_bytes[0]; j->lin[2] = j->lin[1] + b->lin_bytes[1];
    j->tiled[0] = pic->plane[0]; j->tiled[1] = pic->plane[1];
    j->lin_stride[0] = b->lin_stride[0]; j->lin_stride[1] = b->lin_stride[1];
    j->tiled_stride[0] = b->stride[0]; j->tiled_stride[1] = b->stride[1];
    j->mb_width = b->mb_w; j->mb_height = b->mb_h;
    j->to_tiled = 0;
}

static int staging_alloc(Bridge *b, Staging *s)
{
    /* dependency levels of a picture: at most mb_w + 2 * mb_h along its anti-diagonals; in an MBAFF frame a pair adds two (top, then bottom) */
    const size_t n = (size_t)b->nmb, nlev = (size_t)((b->mbaff ? 2 : 1) * (b->mb_w + 2 * b->mb_h) + 4);
    const int np = b->npass;
    size_t o = 0, o_mb[BR_MAX_PASSES], o_mbd[BR_MAX_PASSES], o_coef[BR_MAX_PASSES], o_sl[BR_MAX_PASSES];
    const size_t o_desc = o;   o = up64(o + 2 * BR_MAX_PASSES * sizeof(mi355_h264_frame));
    for (int p = 0; p < np; p++) {
        o_mb[p] = o;   o = up64(o + n * sizeof(mi355_h264_mb));
        o_mbd[p] = p ? o : o_mb[0];
        if (p) o = up64(o + n * sizeof(mi355_h264_mb));
        o_coef[p] = o; o = up64(o + n * (size_t)b->ncoef * b->csize);
        o_sl[p] = o;   o = up64(o + BR_MAX_SLICES * sizeof(mi355_h264_slice));
    }
    const size_t o_mv0 = o;    o = up64(o + n * 64);
    const size_t o_mv1 = o;    o = up64(o + n * 64);
    const size_t o_is = o;     o = up64(o + nlev * 4);
    const size_t o_il = o;     o = up64(o + n * 4);
    s->size = o;
    s->host = mi355_host_alloc(s->size);
    s->widths = malloc(nlev * 4);
    s->out = mi355_host_alloc(out_bytes(b));
    if (b->direct) { s->done = mi355_event_create(); s->d_desc = dalloc(2 * BR_MAX_PASSES * sizeof(mi355_h264_frame)); s->cvt = mi355_host_alloc(sizeof(mi355_surface_job)); }
    if (b->wide) {
        size_t d = 0;
        for (int p = 0; p < np; p++) { s->side_off_mbd[p] = d; d = up64(d + n * sizeof(mi355_h264_mb)); s->side_off_sl[p] = d; d = up64(d + BR_MAX_SLICES * sizeof(mi355_h264_slice)); }
        for (int l = 0; l < 2; l++) { s->side_off_mv[l] = d; d = up64(d + n * 64); }
        s->d_side = dalloc(d);
        if (!s->d_side) return 0;
    }
    if (!s->host || !s->widths || !s->out || (b->direct && (!s->done || !s->d_desc || !s->cvt))) return 0;
    memset(s->host, 0, s->size);
    s->desc = (mi355_h264_frame *)(s->host + o_desc);
    for (int p = 0; p < np; p++) {
        s->mb[p] = (mi355_h264_mb *)(s->host + o_mb[p]);
        s->mbd[p] = (mi355_h264_mb *)(s->host + o_mbd[p]);
        s->coef[p] = (int16_t *)(s->host + o_coef[p]);
        s->slices[p] = (mi355_h264_slice *)(s->host + o_sl[p]);
    }
    s->mv[0] = (int16_t *)(s->host + o_mv0);
    s->mv[1] = (int16_t *)(s->host + o_mv1);
    s->istart = (int32_t *)(s->host + o_is);
    s->ilist = (uint32_t *)(s->host + o_il);
    return 1;
}

/* ---- the dispatcher: one thread, one HIP stream, the pictures of all streams ---------------------------------------- */
#define DISP_DEPTH 4            /* launch sets in flight, each on its own HIP stream */
#define DISP_MAX_DEVICES 16     /* one dispatcher (thread pair, streams) per GPU of the node that decoders were dealt to */
typedef struct Disp {
    int device;
    pthread_mutex_t mu;
    pthread_cond_t work, filled, finished;
    pthread_t thread, completer;
    int started, broken;
    Submission *head, *tail;
    void *stream[DISP_DEPTH], *ev[DISP_DEPTH];
    mi355_h264_frame *h_desc[DISP_DEPTH], *d_desc[DISP_DEPTH];
    mi355_copy_job *jobs[DISP_DEPTH];       /* device-visible: pictures that come back as they are */
    mi355_surface_job *cvt[DISP_DEPTH];     /* device-visible: tiled pictures that come back as lines */
    Submission *in[DISP_DEPTH][DISP_MAX_BATCH];
    int nin[DISP_DEPTH], rcs[DISP_DEPTH];
    int nbridges, nqueued;                  /* decoders that submit here; pictures waiting in the queue */
    unsigned long issued, completed;        /* launch sets handed to the device / known complete: set q lives in slot q % DISP_DEPTH */
    int32_t widths[DISP_MAX_LEVELS];
    unsigned long batches, pictures;
} Disp;
static Disp disps[DISP_MAX_DEVICES];
static pthread_mutex_t disps_mu = PTHREAD_MUTEX_INITIALIZER;   /* creation of the dispatchers; each has its own lock afterwards */
static int disps_ready, bridge_count;

/* the loop filter's side information of one picture -> HBM, on the set's stream, before its kernels */
static int side_upload(const Bridge *b, const Staging *s, void *st)
{
    if (!s->d_side) return 0;
    const size_t n = (size_t)s->nmb_pic;
    int rc = 0;
    for (int p = 0; p < b->npass; p++) {
        rc |= mi355_memcpy_h2d_async(s->d_side + s->side_off_mbd[p], s->mbd[p], n * sizeof(mi355_h264_mb), st);
        rc |= mi355_memcpy_h2d_async(s->d_side + s->side_off_sl[p], s->slices[p], (size_t)s->nslices * sizeof(mi355_h264_slice), st);
    }
    rc |= mi355_memcpy_h2d_async(s->d_side + s->side_off_mv[0], s->mv[0], n * 64, st);
    if (s->uses_l1) rc |= mi355_memcpy_h2d_async(s->d_side + s->side_off_mv[1], s->mv[1], n * 64, st);
    return rc;
}

/* one launch set: a descriptor copy, the kernels for all its pictures (reconstruction from one descriptor array, loop
 * filter from a second one: they differ for the chroma planes of 4:4:4 pictures), one launch that brings the finished
 * pictures to the streams' pinned buffers; nothing waits here.  The sets of different slots run on different HIP streams:
 * a small set is a chain of launches that each occupy a few compute units for microseconds, and several such chains
 * overlap on the device.  A picture whose predecessor of the same decoder is still in flight (MI355_BRIDGE_LAZY) makes
 * its set wait for that set's event. */
static int disp_enqueue(Disp *D, int slot)
{
    const int n = D->nin[slot];
    void *st = D->stream[slot];
    int mw = 0, mh = 0, maxl = 0, layouts = 0, rc = 0, nd = 0, ncopy = 0, ncvt = 0, cw = 0, chh = 0;
    size_t max_bytes = 0;
    for (int i = 0; i < n; i++) nd += D->in[slot][i]->b->npass;
    mi355_h264_frame *hr = D->h_desc[slot], *hd = D->h_desc[slot] + nd;      /* reconstruction | loop filter */
    for (int i = 0, k = 0; i < n; i++) {
        Submission *sub = D->in[slot][i];
        const Staging *s = sub->s;
        const Bridge *b = sub->b;
        const size_t bytes = picture_bytes(b);
        if (sub->after) rc |= mi355_stream_wait_event(st, D->ev[(sub->after - 1) % DISP_DEPTH]);
        rc |= side_upload(b, s, st);
        if (b->mb_w > mw) mw = b->mb_w;
        if (b->mb_h > mh) mh = b->mb_h;
        for (int l = 0; l < s->maxl; l++)
            if (l >= maxl || s->widths[l] > D->widths[l]) D->widths[l] = s->widths[l];
        if (s->maxl > maxl) maxl = s->maxl;
        layouts |= b->tiled ? MI355_LAYOUTS_TILED : MI355_LAYOUTS_LINEAR;
        for (int p = 0; p < b->npass; p++, k++) { hr[k] = s->desc[p]; hd[k] = s->desc[b->npass + p]; }
        if (b->tiled) {
            convert_job(b, s->pic, s->out, &D->cvt[slot][ncvt++]);
            if (b->mb_w > cw) cw = b->mb_w;
            if (b->mb_h > chh) chh = b->mb_h;
        } else {
            D->jobs[slot][ncopy].src = s->pic->plane[0]; D->jobs[slot][ncopy].dst = s->out; D->jobs[slot][ncopy].bytes = bytes;
            ncopy++;
            if (bytes > max_bytes) max_bytes = bytes;
        }
    }
    mi355_h264_frame *dr = D->d_desc[slot], *dd = D->d_desc[slot] + nd;
    rc |= mi355_memcpy_h2d_async(dr, hr, 2 * (size_t)nd * sizeof(mi355_h264_frame), st);
    const Bridge *b0 = D->in[slot][0]->b;
    int any_inter = 0;          /* a launch set of I pictures: the inter pass would launch a wave per macroblock to find nothing */
    for (int k = 0; k < nd; k++) any_inter |= !(hr[k].flags & MI355_FRAME_NO_INTER);
    if (b0->wide) {      /* High 10 / High 4:2:2 / transform bypass: the second kernel set, reconstruction and loop filter from their descriptor arrays */
        if (!rc && mi355_h264_decode_frames_wide_dev(dr, nd, mw, mh, maxl, D->widths, b0->bit_depth, b0->kidc, any_inter ? 3 : 2, st) != 0) rc = -1;
        if (!rc && mi355_h264_decode_frames_wide_dev(dd, nd, mw, mh, 0, NULL, b0->bit_depth, b0->kidc, D->in[slot][0]->s->cls >= 1000 ? 12 : 4, st) != 0) rc = -1;
    } else {
    if (!rc && any_inter && mi355_h264_recon_inter_sparse_dev(dr, nd, mw, mh, st) != 0) rc = -1;      /* staging in host memory: skip what is not coded */
    if (!rc && mi355_h264_recon_intra_all_dev(dr, nd, mw, mh, maxl, D->widths, st) != 0) rc = -1;
    if (!rc && mi355_h264_deblock_layouts_dev(dd, nd, mw, mh, layouts, st) != 0) rc = -1;      /* the loop filter's kernel(s) for the layouts this batch holds */
    }
    if (!rc && ncopy && mi355_copy_batch_dev(D->jobs[slot], ncopy, max_bytes, st) != 0) rc = -1;
    if (!rc && ncvt && mi355_h264_surface_convert_dev(D->cvt[slot], ncvt, cw, chh, st) != 0) rc = -1;
    rc |= mi355_event_record(D->ev[slot], st);
    return rc;
}

/* takes what the decoder threads have queued into the next free slot and issues it */
static void *disp_main(void *arg)
{
    Disp *D = arg;
    mi355_set_device(D->device);
    pthread_mutex_lock(&D->mu);
    for (;;) {
        /* issue when the device is idle, or when enough pictures wait to make another set worth its launches (a quarter of
         * the decoders: up to four sets of that size are in flight); a remainder goes out when the sets before it are back */
        for (;;) {
            const unsigned long inflight = D->issued - D->completed;
            if (D->head && (inflight == 0 || (inflight < DISP_DEPTH && 4 * D->nqueued >= D->nbridges))) break;
            pthread_cond_wait(&D->work, &D->mu);
        }
        /* what is queued now, at most one picture per stream (a stream's next picture reads this one's output) */
        const int slot = (int)(D->issued % DISP_DEPTH);
        Submission *keep_head = NULL, *keep_tail = NULL, *c = D->head;
        int n = 0, nd = 0;
        while (c) {
            Submission *nx = c->next;
            int later = nd + c->b->npass > DISP_MAX_BATCH;
            if (n && !later) later = D->in[slot][0]->s->cls != c->s->cls;      /* one kernel set per launch set (MBAFF frames: their own loop filter) */
            for (int i = 0; i < n && !later; i++) later = D->in[slot][i]->b == c->b;
            if (later) {
                c->next = NULL;
                if (keep_tail) keep_tail->next = c; else keep_head = c;
                keep_tail = c;
            } else {
                /* the set that holds this decoder's previous picture, if that has not come back yet */
                c->after = c->b->last_set > D->completed ? c->b->last_set : 0;
                c->b->last_set = D->issued + 1;
                D->in[slot][n++] = c; nd += c->b->npass;
                D->nqueued--;
            }
            c = nx;
        }
        D->head = keep_head; D->tail = keep_tail;
        D->nin[slot] = n;
        D->batches++; D->pictures += (unsigned long)n;
        pthread_mutex_unlock(&D->mu);
        const int rc = disp_enqueue(D, slot);
        pthread_mutex_lock(&D->mu);
        D->rcs[slot] = rc;
        D->issued++;
        pthread_cond_signal(&D->filled);
    }
    return NULL;
}

/* waits for the launch sets in the order they were issued and tells their decoder threads */
static void *disp_complete(void *arg)
{
    Disp *D = arg;
    mi355_set_device(D->device);
    pthread_mutex_lock(&D->mu);
    for (;;) {
        while (D->completed == D->issued) pthread_cond_wait(&D->filled, &D->mu);
        const int slot = (int)(D->completed % DISP_DEPTH);
        pthread_mutex_unlock(&D->mu);
        const int rc = mi355_event_sync(D->ev[slot]);
        pthread_mutex_lock(&D->mu);
        for (int i = 0; i < D->nin[slot]; i++) { D->in[slot][i]->rc = rc | D->rcs[slot]; D->in[slot][i]->done = 1; }
        D->completed++;
        pthread_cond_broadcast(&D->finished);
        pthread_cond_signal(&D->work);
    }
    return NULL;
}

/* the dispatcher of `device` (the calling thread is on that device) */
static Disp *disp_start(int device)
{
    if (device < 0 || device >= DISP_MAX_DEVICES) return NULL;
    Disp *D = &disps[device];
    pthread_mutex_lock(&disps_mu);
    if (!disps_ready) {
        for (int i = 0; i < DISP_MAX_DEVICES; i++) {
            pthread_mutex_init(&disps[i].mu, NULL);
            pthread_cond_init(&disps[i].work, NULL); pthread_cond_init(&disps[i].filled, NULL); pthread_cond_init(&disps[i].finished, NULL);
        }
        disps_ready = 1;
    }
    if (!D->started && !D->broken) {
        int ok = 1;
        D->device = device;
        for (int k = 0; k < DISP_DEPTH && ok; k++) {
            D->stream[k] = mi355_stream_create();
            D->ev[k] = mi355_event_create();
            D->h_desc[k] = mi355_host_alloc(2 * DISP_MAX_BATCH * sizeof(mi355_h264_frame));
            D->d_desc[k] = dalloc(2 * DISP_MAX_BATCH * sizeof(mi355_h264_frame));
            D->jobs[k] = mi355_host_alloc(DISP_MAX_BATCH * sizeof(mi355_copy_job));
            D->cvt[k] = mi355_host_alloc(DISP_MAX_BATCH * sizeof(mi355_surface_job));
            ok = D->stream[k] && D->ev[k] && D->h_desc[k] && D->d_desc[k] && D->jobs[k] && D->cvt[k];
        }
        if (ok && pthread_create(&D->thread, NULL, disp_main, D) == 0 && pthread_create(&D->completer, NULL, disp_complete, D) == 0) {
            pthread_detach(D->thread); pthread_detach(D->completer);
            D->started = 1;
        } else D->broken = 1;
    }
    const int ok = D->started;
    pthread_mutex_unlock(&disps_mu);
    return ok ? D : NULL;
}

static int finish_set(Bridge *b, Staging *s);
/* both sets, the older submission first: the two fields of a frame go to the same AVFrame */
static int finish_all(Bridge *b)
{
    const int r0 = finish_set(b, &b->st[b->cur ^ 1]), r1 = finish_set(b, &b->st[b->cur]);
    return r0 ? r0 : r1;
}
static void staging_free(Staging *s)
{
    if (s->host) mi355_host_free(s->host);
    free(s->widths);
    if (s->out) mi355_host_free(s->out);
    if (s->done) mi355_event_destroy(s->done);
    if (s->d_desc) mi355_free(s->d_desc);
    if (s->d_side) mi355_free(s->d_side);
    if (s->cvt) mi355_host_free(s->cvt);
    memset(s, 0, sizeof(*s));
}
/* give back everything that depends on the picture geometry (what is in flight comes back first): the next sequence sets
 * the bridge up again */
static void bridge_release(Bridge *b)
{
    if (b->state > 0) {
        finish_all(b);
        if (!b->direct && b->disp) { pthread_mutex_lock(&b->disp->mu); b->disp->nbridges--; pthread_mutex_unlock(&b->disp->mu); }
    }
    staging_free(&b->st[0]); staging_free(&b->st[1]);
    for (int p = 0; p < 3; p++) { if (b->recon[p]) mi355_free(b->recon[p]); b->recon[p] = NULL; }
    for (int p = 0; p < 2; p++) { if (b->scratch_c[p]) mi355_free(b->scratch_c[p]); b->scratch_c[p] = NULL; }
    for (int i = 0; i < BR_MAX_PICS; i++) if (b->pics[i].plane[0] && !b->sess) mi355_free(b->pics[i].plane[0]);
    memset(b->pics, 0, sizeof(b->pics));
    if (b->sess) { mi355_h264_session_close(b->sess); b->sess = NULL; }
    b->open = 0; b->cur = 0; b->nslots = b->nslices = 0;
}

/* The decoder starts a (new) sequence: h264_init_ps() calls this when the parameters that size its tables change
 * (h264_slice.c:973-985; also for the first sequence of a context).  Pictures still in flight belong to the frames it is
 * about to drop: they come back first.  The bridge keeps its buffers when the new sequence has the geometry it was set up
 * for, gives them back otherwise (bridge_get sets it up again, or steps aside for a format outside this path — and a
 * bridge that had stepped aside for that reason looks at the new sequence again). */
void __wrap_ff_h264_flush_change(H264Context *h)
{
    Bridge *b = br_tls;
    if (b && b->state > 0) {
        finish_all(b);
        b->open = 0;
        const SPS *sps = h->ps.sps;
        const int same = sps && sps->mb_width == b->mb_w && sps->mb_height * (2 - sps->frame_mbs_only_flag) == b->mb_h && sps->bit_depth_luma == b->bit_depth &&
                         sps->chroma_format_idc == b->idc && (!sps->frame_mbs_only_flag && sps->mb_aff) == b->mbaff &&
                         !sps->transform_bypass == !b->bypass && !sps->residual_color_transform_flag &&
                         (!b->tiled || sps->frame_mbs_only_flag);        /* tiled device pictures hold frames only */
        if (!same) { bridge_release(b); b->state = 0; }
    } else if (b && b->state < 0 && b->soft) {
        b->state = 0; b->soft = 0;
    }
    __real_ff_h264_flush_change(h);
}

static pthread_key_t br_key;
static pthread_once_t br_key_once = PTHREAD_ONCE_INIT;
static void br_thread_exit(void *p)
{
    Bridge *b = p;
    if (!b) return;
    b->releasing = 1;                 /* failures on the way out are only reported */
    bridge_release(b);
    if (br_tls == b) br_tls = NULL;
    free(b);
}
static void br_key_make(void) { pthread_key_create(&br_key, br_thread_exit); }

static Bridge *bridge_get(const H264Context *h)
{
    Bridge *b = br_tls;
    if (!b) {
        b = br_tls = calloc(1, sizeof(*b));
        if (!b) return NULL;
        /* a decoder thread that exits gives its pictures, pinned buffers and dispatcher slot back (the key's destructor) */
        pthread_once(&br_key_once, br_key_make);
        pthread_setspecific(br_key, b);
        b->lazy = getenv("MI355_BRIDGE_LAZY") != NULL;
        b->direct = getenv("MI355_BRIDGE_DIRECT") != NULL;
        b->null_submit = getenv("MI355_BRIDGE_NULL") != NULL;
        /* tests: field pictures with disable_deblocking_filter_idc 2 stay on the device (streams on which the reference's inconsistency does not show: see hl_decode_mb below) */
        b->keep_field_idc2 = getenv("MI355_BRIDGE_KEEP_FIELD_IDC2") != NULL;
        /* MI355_BRIDGE_MAX_SLICES: a smaller slice table (tests: a picture with more slices than the path holds) */
        const int ms = getenv("MI355_BRIDGE_MAX_SLICES") ? atoi(getenv("MI355_BRIDGE_MAX_SLICES")) : BR_MAX_SLICES;
        b->max_slices = ms < 1 ? 1 : (ms > BR_MAX_SLICES ? BR_MAX_SLICES : ms);
        if (getenv("MI355_BRIDGE_PLAIN")) b->state = -1;         /* the comparison run: the reference's C path, silently */
    }
    if (b->state) return b;
    const int idc = h->ps.sps->chroma_format_idc;
    /* a sequence that may hold field MACROBLOCKS (mb_adaptive_frame_field_flag) is outside the path as a whole; field PICTURES
     * (PAFF: the choice between a frame and two fields is made per picture) are inside: begin_picture() looks at each one */
    const int seq_mbaff = !h->ps.sps->frame_mbs_only_flag && h->ps.sps->mb_aff;
    if ((seq_mbaff && getenv("MI355_BRIDGE_NO_WIDE")) || (h->mb_height & 1 && !h->ps.sps->frame_mbs_only_flag) || h->ps.sps->bit_depth_luma > 10 || h->ps.sps->bit_depth_luma != h->ps.sps->bit_depth_chroma ||
        (idc != 1 && idc != 2 && idc != 3) || h->ps.sps->residual_color_transform_flag || (getenv("MI355_BRIDGE_NO_WIDE") && (h->pixel_shift || idc == 2 || h->ps.sps->transform_bypass))) {
        br_fail(b, "stream outside the batched path (needs 8- to 10-bit 4:2:0, 4:2:2 or 4:4:4)");
        b->soft = 1;
        return b;
    }
    /* pictures one macroblock wide, and two wide with chroma planes of their own: the reference's decoder is not consistent with ITSELF there (one wide: a P macroblock
     * with 4-wide partitions and zero vectors is not a copy of its reference; two wide: the Cr intermediate of two-reference weighted prediction overwrites the Cb one,
     * h264_mb.c:407-409 — bipred_scratchpad rows are mb_uvlinesize apart, 16 bytes there).  What it outputs for such streams is its business: they stay on its path */
    if (h->mb_width == 1 || (h->mb_width == 2 && idc != 3)) {
        br_fail(b, "pictures one or two macroblocks wide are left to the reference's decoder (its bi-prediction scratch rows overlap there: h264_mb.c:407-409)");
        b->soft = 1;
        return b;
    }
    /* one decoder context per thread, no threads inside it (see the header comment) */
    if (h->avctx->active_thread_type || h->nb_slice_ctx > 1) {
        br_fail(b, "decoder opened with frame or slice threads (the bridge needs thread_count = 1 per decoder; run one decoder per stream and thread)");
        b->soft = 1;
        return b;
    }
    /* which GPU: MI355_DEVICE=n names one; MI355_DEVICES=N (or "all") deals the decoders of this process over the first N GPUs of
     * the node, decoder k -> device k mod N — one host process driving the whole node, each decoder's decoded-picture buffer in its
     * own GPU's HBM, one dispatcher per GPU, nothing shared between them */
    {
        const char *dev = getenv("MI355_DEVICE"), *devs = getenv("MI355_DEVICES");
        int device = dev ? atoi(dev) : 0;
        if (!dev && devs) {
            int n = strcmp(devs, "all") ? atoi(devs) : mi355_device_count();
            if (n > mi355_device_count()) n = mi355_device_count();
            if (n > DISP_MAX_DEVICES) n = DISP_MAX_DEVICES;
            pthread_mutex_lock(&disps_mu);
            const int k = bridge_count++;
            pthread_mutex_unlock(&disps_mu);
            device = n > 0 ? k % n : 0;
        }
        if (mi355_get_device() < 0 && mi355_init(device) != 0) { br_fail(b, "no usable MI355X"); return b; }
        if (mi355_set_device(device) != 0) { br_fail(b, "no usable MI355X"); return b; }      /* this decoder thread works on its GPU from now on */
        b->device = device;
    }
    b->mb_w = h->mb_width; b->mb_h = h->mb_height; b->nmb = b->mb_w * b->mb_h;
    b->c444 = idc == 3; b->npass = b->c444 ? 3 : 1;
    b->bit_depth = h->ps.sps->bit_depth_luma; b->idc = idc; b->kidc = idc == 3 ? 1 : idc;
    b->mbaff = seq_mbaff;
    b->wide = b->bit_depth > 8 || idc == 2 || h->ps.sps->transform_bypass || b->mbaff;      /* transform bypass, macroblock pairs: the second kernel set knows them, the first does not */
    b->bypass = h->ps.sps->transform_bypass;
    b->px = b->bit_depth > 8 ? 2 : 1; b->csize = b->bit_depth > 8 ? 4 : 2;
    b->crows = idc == 2 ? 16 : 8; b->ncoef = idc == 2 ? 512 : 384;
    /* frame_num gaps: the decoder fills a lost frame with a host-side copy of the previous one (h264_slice.c:1425-1452) — every
     * picture must be complete in its frame before the next one starts */
    b->lazy = getenv("MI355_BRIDGE_LAZY") != NULL && !h->ps.sps->gaps_in_frame_num_allowed_flag;
    b->lin_stride[0] = (16 * b->mb_w * b->px + 63) & ~63; b->lin_stride[1] = b->lin_stride[0] / 2;
    b->lin_bytes[0] = (size_t)b->lin_stride[0] * 16 * b->mb_h; b->lin_bytes[1] = (size_t)b->lin_stride[1] * b->crows * b->mb_h;
    b->tiled = h->ps.sps->frame_mbs_only_flag && !b->c444 && !b->wide && !getenv("MI355_BRIDGE_LINEAR");
    if (b->tiled) {
        b->stride[0] = MI355_TILE_LUMA_BYTES * b->mb_w; b->stride[1] = MI355_TILE_CHROMA_BYTES * b->mb_w;
        b->plane_bytes[0] = (size_t)b->stride[0] * b->mb_h; b->plane_bytes[1] = (size_t)b->stride[1] * b->mb_h;
    } else {
        b->stride[0] = b->lin_stride[0]; b->stride[1] = b->lin_stride[1];
        b->plane_bytes[0] = b->lin_bytes[0]; b->plane_bytes[1] = b->lin_bytes[1];
    }
    int ok = (b->mbaff ? 2 : 1) * (b->mb_w + 2 * b->mb_h) + 4 <= DISP_MAX_LEVELS;
    if (getenv("MI355_BRIDGE_SESSION") && !b->c444 && !b->wide) {
        /* one surface per H264Picture the decoder may hold; a slice that is not one run of macroblocks goes in run by run */
        const mi355_h264_session_params sp = { b->mb_w, b->mb_h, BR_MAX_PICS, 255, b->tiled ? MI355_SURFACE_TILED : MI355_SURFACE_LINEAR, 0 };
        ok = ok && mi355_h264_session_open(&b->sess, &sp) == 0;
        b->direct = 1; b->lazy = 0;                  /* nothing goes through the dispatcher; every picture is complete at its end */
    }
    if (ok && b->direct && !b->sess) ok = (b->stream = mi355_stream_create()) != NULL;
    if (ok && !b->direct) ok = (b->disp = disp_start(b->device)) != NULL;
    ok = ok && staging_alloc(b, &b->st[0]) && staging_alloc(b, &b->st[1]);
    for (int p = 0; p < 3 && ok; p++) ok = (b->recon[p] = dalloc(b->plane_bytes[b->c444 ? 0 : p > 0])) != NULL;
    for (int p = 0; p < 2 && ok && b->c444; p++) ok = (b->scratch_c[p] = dalloc(b->plane_bytes[1])) != NULL;
    if (!ok) { bridge_release(b); br_fail(b, "device, pinned memory or dispatcher set-up failed"); return b; }
    if (!b->direct) { pthread_mutex_lock(&b->disp->mu); b->disp->nbridges++; pthread_mutex_unlock(&b->disp->mu); }
    b->st[0].sub.b = b->st[1].sub.b = b;
    b->st[0].sub.s = &b->st[0]; b->st[1].sub.s = &b->st[1];
    b->state = 1;
    return b;
}

/* the device picture of a decoder picture.  Slots whose owner is not one of THIS decoder context's pictures belong to a
 * context that was closed (the thread decodes the next stream): they are taken over. */
static DevPic *devpic_of(Bridge *b, const H264Context *h, const H264Picture *p, int create)
{
    DevPic *slot = NULL;
    for (int i = 0; i < BR_MAX_PICS; i++) {
        if (b->pics[i].owner == p) return &b->pics[i];
        if (!slot && !b->pics[i].owner) slot = &b->pics[i];
    }
    if (!create) return NULL;
    for (int i = 0; i < BR_MAX_PICS && !slot; i++)
        if ((uintptr_t)b->pics[i].owner < (uintptr_t)h->DPB || (uintptr_t)b->pics[i].owner >= (uintptr_t)(h->DPB + H264_MAX_PICTURE_COUNT)) slot = &b->pics[i];
    if (!slot) return NULL;
    if (b->sess) slot->plane[0] = (uint8_t *)b->sess;            /* the session owns the surface: pics[i] is surface i */
    if (!slot->plane[0]) {
        uint8_t *base = dalloc(picture_bytes(b));
        if (!base) return NULL;
        const size_t cb = b->c444 ? b->plane_bytes[0] : b->plane_bytes[1];
        slot->plane[0] = base; slot->plane[1] = base + b->plane_bytes[0]; slot->plane[2] = b->tiled ? slot->plane[1] : slot->plane[1] + cb;
    }
    slot->owner = p;
    return slot;
}

/* A reference this bridge never decoded: a frame the decoder made up for a gap in frame_num (h264_field_start fills it with a
 * copy of the previous frame, h264_slice.c), or one decoded before the bridge took over.  Its samples are in the host frame:
 * they go to a new device picture (synchronous copy; nothing on the device uses that picture yet). */
static DevPic *devpic_upload(Bridge *b, const H264Context *h, const H264Picture *p)
{
    if (!p || !p->f || !p->f->data[0]) return NULL;
    DevPic *r = devpic_of(b, h, p, 1);
    if (!r) return NULL;
    if (b->sess) {
        const uint8_t *const src[3] = { p->f->data[0], p->f->data[1], p->f->data[2] };
        const int st[3] = { p->f->linesize[0], p->f->linesize[1], p->f->linesize[2] };
        if (mi355_h264_put_frame(b->sess, (int)(r - b->pics), src, st) != 0) { r->owner = NULL; return NULL; }
        r->frame_num = p->frame_num; r->poc = p->poc; r->data0 = p->f->data[0];
        return r;
    }
    uint8_t *tmp = malloc(picture_bytes(b));
    if (!tmp) { r->owner = NULL; return NULL; }
    uint8_t *dst = tmp;
    if (b->tiled) {
        /* lines -> macroblock tiles (mi355_h264_frame.h): 16 rows of 16 luma samples, then per macroblock 8 rows of 8 Cb, 8 rows of 8 Cr */
        for (int my = 0; my < b->mb_h; my++)
            for (int mx = 0; mx < b->mb_w; mx++) {
                uint8_t *ty = tmp + (size_t)my * b->stride[0] + (size_t)mx * MI355_TILE_LUMA_BYTES;
                uint8_t *tc = tmp + b->plane_bytes[0] + (size_t)my * b->stride[1] + (size_t)mx * MI355_TILE_CHROMA_BYTES;
                for (int y = 0; y < 16; y++) memcpy(ty + 16 * y, p->f->data[0] + (size_t)(16 * my + y) * p->f->linesize[0] + 16 * mx, 16);
                for (int k = 1; k < 3; k++)
                    for (int y = 0; y < 8; y++) memcpy(tc + 64 * (k - 1) + 8 * y, p->f->data[k] + (size_t)(8 * my + y) * p->f->linesize[k] + 8 * mx, 8);
            }
    } else
    for (int k = 0; k < 3; k++) {
        const int half = k && !b->c444;
        const int w = (half ? 8 : 16) * b->mb_w * b->px, hgt = (half ? b->crows : 16) * b->mb_h, st = b->stride[half];
        for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * st, p->f->data[k] + (size_t)y * p->f->linesize[k], (size_t)w);
        dst += b->plane_bytes[half];
    }
    const int rc = mi355_memcpy_h2d(r->plane[0], tmp, picture_bytes(b));
    free(tmp);
    if (rc != 0) { r->owner = NULL; return NULL; }
    r->frame_num = p->frame_num; r->poc = p->poc; r->data0 = p->f->data[0];
    return r;
}

static int slot_of(Bridge *b, const H264Picture *p, int par)
{
    for (int i = 0; i < b->nslots; i++)
        if (b->slot_pic[i] == p && b->slot_par[i] == par) return i;
    if (b->nslots >= MI355_H264_MAX_SLOTS) return -1;
    b->slot_pic[b->nslots] = p;
    b->slot_par[b->nslots] = par;
    return b->nslots++;
}

/* the picture submitted from this staging set is complete on the device and in `out`: wait for that, then put it where
 * the decoder will look for it */
static int finish_set(Bridge *b, Staging *s)
{
    if (!s->in_flight) return 0;
    int rc;
    if (b->direct) rc = mi355_event_sync(s->done);
    else {
        Disp *D = b->disp;
        pthread_mutex_lock(&D->mu);
        while (!s->sub.done) pthread_cond_wait(&D->finished, &D->mu);
        rc = s->sub.rc;
        pthread_mutex_unlock(&D->mu);
    }
    s->in_flight = 0;
    if (rc) return rc;
    /* a field picture brings its own lines only: the frame's other field may have come back (or will come back) from another set */
    const uint8_t *src = s->out;
    const int y0 = s->field ? s->parity : 0, dy = s->field ? 2 : 1;
    for (int k = 0; k < 3; k++) {
        const int half = k && !b->c444;
        const int w = (half ? 8 : 16) * b->mb_w * b->px, hgt = (half ? b->crows : 16) * b->mb_h, st = b->lin_stride[half];
        for (int y = y0; y < hgt; y += dy) memcpy(s->frame_data[k] + (size_t)y * s->frame_linesize[k], src + (size_t)y * st, (size_t)w);
        src += b->lin_bytes[half];
    }
    return 0;
}

static void begin_picture(Bridge *b, const H264Context *h)
{
    /* the staging set must be free again: what was submitted from it two pictures ago has come back */
    b->cur ^= 1;
    Staging *s = &b->st[b->cur];
    if (s->in_flight) {
        b->waits++;
        if (finish_set(b, s) != 0) { br_fail(b, "a picture did not come back from the device"); return; }     /* the caller looks at b->state */
    }
    for (int p = 0; p < b->npass; p++) {
        memset(s->mb[p], 0, (size_t)b->nmb * sizeof(mi355_h264_mb));
        if (p) memset(s->mbd[p], 0, (size_t)b->nmb * sizeof(mi355_h264_mb));
        memset(s->slices[p], 0, BR_MAX_SLICES * sizeof(mi355_h264_slice));
    }
    memset(s->mv[0], 0, (size_t)b->nmb * 64);
    memset(s->mv[1], 0, (size_t)b->nmb * 64);
    b->nslices = b->nslots = b->uses_l1 = b->mbs_packed = 0;
    b->field = h->picture_structure != PICT_FRAME;
    b->mbaff_frame = FRAME_MBAFF(h) != 0;
    b->parity = h->picture_structure == PICT_BOTTOM_FIELD;
    b->rows = b->field ? b->mb_h / 2 : b->mb_h;
    b->nmb_pic = b->mb_w * b->rows;
    b->open = 1;
}

static int slice_index(Bridge *b, const H264Context *h, const H264SliceContext *sl)
{
    for (int i = 0; i < b->nslices; i++)
        if (b->slice_num_of[i] == sl->slice_num) return i;
    if (b->nslices >= b->max_slices) return -1;
    mi355_h264_slice *s = &b->st[b->cur].slices[0][b->nslices];
    b->slice_num_of[b->nslices] = sl->slice_num;
    s->use_weight = sl->pwt.use_weight;
    s->use_weight_chroma = sl->pwt.use_weight_chroma;
    s->luma_log2_weight_denom = sl->pwt.luma_log2_weight_denom;
    s->chroma_log2_weight_denom = sl->pwt.chroma_log2_weight_denom;
    s->list_count = sl->list_count;
    for (unsigned list = 0; list < sl->list_count; list++) {
        if (sl->ref_count[list] > MI355_H264_MAX_REFS) return -1;        /* 17..32 fields per list: more than the slice table holds */
        for (unsigned i = 0; i < sl->ref_count[list]; i++) {
            /* a field picture's list entries are fields: (frame, parity) names the reference */
            const int slot = slot_of(b, sl->ref_list[list][i].parent, b->field ? (sl->ref_list[list][i].reference & 3) - 1 : -1);
            if (slot < 0) return -1;
            s->ref_slot[list][i] = (uint8_t)slot;
        }
    }
    for (int r = 0; r < MI355_H264_MAX_REFS; r++) {
        for (int l = 0; l < 2; l++)
            for (int k = 0; k < 2; k++) {
                s->luma_weight[r][l][k] = (int16_t)sl->pwt.luma_weight[r][l][k];
                for (int c = 0; c < 2; c++) s->chroma_weight[r][l][c][k] = (int16_t)sl->pwt.chroma_weight[r][l][c][k];
            }
        for (int r1 = 0; r1 < MI355_H264_MAX_REFS; r1++) s->implicit_weight[r][r1] = (int16_t)sl->pwt.implicit_weight[r][r1][0];
    }
    if (b->mbaff_frame && sl->pwt.use_weight == 2)        /* field macroblocks: the tables implicit_weight_table(h, sl, 0 / 1) filled (h264_slice.c:1786-1790) */
        for (int p = 0; p < 2; p++)
            for (unsigned r0 = 0; r0 < 2 * sl->ref_count[0]; r0++)
                for (unsigned r1 = 0; r1 < 2 * sl->ref_count[1]; r1++)
                    s->implicit_weight_field[p][r0][r1] = (int16_t)sl->pwt.implicit_weight[(16 + r0) ^ p][(16 + r1) ^ p][p];
    for (int t = 0; t < 2; t++)
        for (int q = 0; q < 52; q++) s->chroma_qp_table[t][q] = h->ps.pps->chroma_qp_table[t][q];
    /* 4:4:4: the plane's own weights in the luma role (mc_part_weighted, h264_mb.c:386-470: chroma_weight_op = luma_weight_op
     * with sl->pwt.chroma_weight and chroma_log2_weight_denom; tables hold the identity where a flag was not sent) */
    for (int p = 1; p < b->npass; p++) {
        mi355_h264_slice *c = &b->st[b->cur].slices[p][b->nslices];
        *c = *s;
        c->luma_log2_weight_denom = s->chroma_log2_weight_denom;
        c->use_weight_chroma = 0;
        for (int r = 0; r < MI355_H264_MAX_REFS; r++)
            for (int l = 0; l < 2; l++)
                for (int k = 0; k < 2; k++) c->luma_weight[r][l][k] = s->chroma_weight[r][l][p - 1][k];
    }
    return b->nslices++;
}

/* where level k of sl->mb_luma_dc goes in a transform-bypass Intra16x16 macroblock: dc_mapping[] of hl_decode_mb_predict_luma (h264_mb.c:712-718) */
static const uint16_t br_dc_mapping[16] = { 0 * 16, 1 * 16, 4 * 16, 5 * 16, 2 * 16, 3 * 16, 6 * 16, 7 * 16, 8 * 16, 9 * 16, 12 * 16, 13 * 16, 10 * 16, 11 * 16, 14 * 16, 15 * 16 };

/* I_PCM samples for the second kernel set, one per coefficient slot: samples `first` .. `first + n - 1` of the macroblock's PCM payload —
 * bytes at 8 bits, bit_depth-wide big-endian fields otherwise (h264_mb_template.c:99-153) */
static void pcm_unpack(const Bridge *b, const uint8_t *src, int first, int n, uint8_t *cf)
{
    for (int i = 0; i < n; i++) {
        unsigned v = 0;
        if (b->bit_depth == 8) v = src[first + i];
        else {
            const size_t bit = (size_t)(first + i) * b->bit_depth;
            for (int k = 0; k < b->bit_depth; k++) v = (v << 1) | ((src[(bit + k) >> 3] >> (7 - ((bit + k) & 7))) & 1);
        }
        if (b->csize == 4) ((int32_t *)cf)[i] = (int32_t)v; else ((int16_t *)cf)[i] = (int16_t)v;
    }
}

/* 4:4:4: the records, coefficients and filter records of planes 1 and 2 (hl_decode_mb_predict_luma / _idct_luma with
 * p = 1, 2: h264_mb_template.c:323-343; coefficient flags at scan8[16 * p + i], DC levels in sl->mb_luma_dc[p], the DC
 * multiplier of dequant4_coeff[p] at the plane's QP, h264_mb.c:617-640) */
static void pack_planes_444(const H264Context *h, H264SliceContext *sl, Staging *st, int idx, int mb_type, int luma_coded)
{
    mi355_h264_mb *m0 = &st->mb[0][idx];
    m0->chroma_pred_mode = 6;                        /* DC_128_PRED8x8: the scratch chroma planes are predicted from nothing */
    for (int p = 1; p < 3; p++) {
        mi355_h264_mb *m = &st->mb[p][idx];
        const Bridge *b = br_tls;
        const int cs = b->csize;
        uint8_t *cf = (uint8_t *)st->coef[p] + (size_t)idx * 384 * cs;
        *m = *m0;
        m->nnz_mask = 0;
        m->qp = (int8_t)h->ps.pps->chroma_qp_table[p - 1][m0->qp & 0xff];
        m->dc_qmul[0] = h->ps.pps->dequant4_coeff[p][sl->chroma_qp[p - 1]][0];
        if (IS_INTRA(mb_type) || luma_coded) memset(cf, 0, 384 * (size_t)cs);
        if (IS_INTRA_PCM(mb_type)) {
            if (b->wide) pcm_unpack(b, sl->intra_pcm_ptr, 256 * p, 256, cf);
            else memcpy(cf, sl->intra_pcm_ptr + 256 * p, 256);
            m->nnz_mask = 0xFFFFFF;
        } else {
            if (luma_coded) {
                for (int i = 0; i < 16; i++) {
                    const int src = IS_8x8DCT(mb_type) ? (i & ~3) : i;
                    if (sl->non_zero_count_cache[scan8[16 * p + src]]) m->nnz_mask |= 1u << i;
                }
                memcpy(cf, (const uint8_t *)sl->mb + 256 * (size_t)p * cs, 256 * (size_t)cs);
            }
            if (IS_INTRA16x16(mb_type) && sl->non_zero_count_cache[scan8[LUMA_DC_BLOCK_INDEX + p]]) {
                m->nnz_mask |= 1u << MI355_NNZ_LUMA_DC;
                for (int k = 0; k < 16; k++)
                    memcpy(cf + (size_t)((m0->flags & MI355_MBF_BYPASS) ? br_dc_mapping[k] : mi355_luma_dc_slot(k)) * cs, (const uint8_t *)sl->mb_luma_dc[p] + (size_t)k * cs, (size_t)cs);
            }
        }
        /* the loop filter derives the boundary strengths of every plane from the luma coefficient flags */
        st->mbd[p][idx] = *m;
        st->mbd[p][idx].nnz_mask = m0->nnz_mask;
    }
}

void __wrap_ff_h264_hl_decode_mb(const H264Context *h, H264SliceContext *sl)
{
    Bridge *b = bridge_get(h);
    if (!b || b->state < 0) { __real_ff_h264_hl_decode_mb(h, sl); return; }
    if (!b->open && h->picture_structure != PICT_FRAME && sl->deblocking_filter == 2 && !b->keep_field_idc2) {
        /* a FIELD picture whose slices filter only their own edges: whether the reference predicts an intra macroblock from the unfiltered or from the filtered above-left
         * sample it decides from slice_table[mb_xy - 1 - mb_stride] (h264_mb.c:525-527) — in a field picture the OTHER field's row, whose entries an earlier picture
         * left.  The device predicts from unfiltered samples throughout (the standard's rule) and has no filtered ones at that point: such pictures are the reference's.
         * Seen at the picture's first macroblock: nothing of it has been packed, the pictures in flight come back, the decoder continues on its own path.  (A picture
         * that reaches this combination only in a later slice stays here.) */
        br_fail(b, "field picture with disable_deblocking_filter_idc 2 (the reference decides its intra border from the other field's slice table: h264_mb.c:525-527)");
        b->soft = 1;
        __real_ff_h264_hl_decode_mb(h, sl);
        return;
    }
    if (!b->open) begin_picture(b, h);
    if (b->state < 0) { __real_ff_h264_hl_decode_mb(h, sl); return; }
    Staging *st = &b->st[b->cur];
    /* in a field picture sl->mb_y counts FRAME macroblock rows (2 * field row + parity, h264_slice.c:2324-2329, 2456-2460) */
    const int mb_xy = sl->mb_xy, mb_row = sl->mb_y >> b->field, idx = sl->mb_x + mb_row * b->mb_w;
    const int mb_type = h->cur_pic.mb_type[mb_xy];
    mi355_h264_mb *m = &st->mb[0][idx];
    b->mbs_packed++;
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
        if (sl->deblocking_filter == 2) m->flags |= MI355_MBF_FILTER_OWN_SLICE;
        if (sl->mb_x > 0 && (sl->deblocking_filter != 2 || h->slice_table[mb_xy - 1] == sl->slice_num)) m->flags |= MI355_MBF_LEFT_EDGE;
        if (mb_row > 0 && (sl->deblocking_filter != 2 || h->slice_table[mb_xy - (h->mb_stride << b->field)] == sl->slice_num)) m->flags |= MI355_MBF_TOP_EDGE;
    }
    if (sl->pwt.use_weight) m->flags |= MI355_MBF_WEIGHTED;
    const int bypass = b->bypass && sl->qscale == 0;                  /* h264_mb_template.c:51 */
    if (bypass) m->flags |= MI355_MBF_BYPASS | (h->ps.sps->profile_idc == 244 ? MI355_MBF_BYPASS_PRED : 0) | (h->x264_build < 151U ? MI355_MBF_BYPASS_X264OLD : 0);
    m->intra16x16_pred_mode = (uint8_t)sl->intra16x16_pred_mode;
    m->chroma_pred_mode = (uint8_t)sl->chroma_pred_mode;
    m->topleft_samples_available = (uint16_t)sl->topleft_samples_available;
    m->topright_samples_available = (uint16_t)sl->topright_samples_available;
    m->dc_qmul[0] = h->ps.pps->dequant4_coeff[0][sl->qscale][0];
    const int cq3 = b->idc == 2 ? 3 : 0;          /* chroma422_dc_dequant_idct takes the multiplier of QPc + 3 (h264_mb_template.c:232-236) */
    m->dc_qmul[1] = h->ps.pps->dequant4_coeff[intra ? 1 : 4][sl->chroma_qp[0] + cq3][0];
    m->dc_qmul[2] = h->ps.pps->dequant4_coeff[intra ? 2 : 5][sl->chroma_qp[1] + cq3][0];
    memset(m->ref_idx, -1, sizeof(m->ref_idx));

    /* `dctcoef` is int32_t when the samples have more than 8 bits: sl->mb, sl->mb_luma_dc and the staging block are addressed in bytes */
    const size_t cs = (size_t)b->csize, ncc = b->idc == 2 ? 128 : 64;
    uint8_t *cf = (uint8_t *)st->coef[0] + (size_t)idx * b->ncoef * cs;
    uint8_t *mbp = (uint8_t *)sl->mb;
    /* an inter macroblock without coefficients (cbp 0) never has its block fetched (mi355_h264_recon_inter_sparse_dev):
     * no need to clear it either */
    const int reads_coefs = intra || (cbp & 0x3F);
    if (IS_INTRA_PCM(mb_type)) {
        if (b->wide) pcm_unpack(b, sl->intra_pcm_ptr, 0, b->c444 ? 256 : 256 + 2 * 8 * b->crows, cf);
        else { memcpy(cf, sl->intra_pcm_ptr, 384); memset(cf + 384, 0, 384); }
        m->nnz_mask = 0xFFFFFF;
        memset(m->u.intra4x4_pred_mode, 0, 16);
        if (b->c444) pack_planes_444(h, sl, st, idx, mb_type, 0);
    } else {
        /* coefficient masks: the count caches are only meaningful where cbp says something was coded */
        const int luma_coded = IS_INTRA16x16(mb_type) || (cbp & 15);
        if (luma_coded) {
            for (int i = 0; i < 16; i++) {
                const int src = IS_8x8DCT(mb_type) ? (i & ~3) : i;
                if (sl->non_zero_count_cache[scan8[src]]) m->nnz_mask |= 1u << i;
            }
            memcpy(cf, mbp, 256 * cs);
        } else if (reads_coefs) memset(cf, 0, 256 * cs);
        if (IS_INTRA16x16(mb_type) && sl->non_zero_count_cache[scan8[LUMA_DC_BLOCK_INDEX]]) {
            m->nnz_mask |= 1u << MI355_NNZ_LUMA_DC;
            for (int k = 0; k < 16; k++) memcpy(cf + (size_t)(bypass ? br_dc_mapping[k] : mi355_luma_dc_slot(k)) * cs, (const uint8_t *)sl->mb_luma_dc[0] + k * cs, cs);
        }
        if (cbp & 0x30) {
            memcpy(cf + 256 * cs, mbp + 256 * cs, ncc * cs);
            memcpy(cf + (256 + ncc) * cs, mbp + 512 * cs, ncc * cs);
            if (cbp & 0x20)
                for (int j = 0; j < 4; j++) {
                    if (sl->non_zero_count_cache[scan8[16 + j]]) m->nnz_mask |= 1u << MI355_NNZ_CB(j);
                    if (sl->non_zero_count_cache[scan8[32 + j]]) m->nnz_mask |= 1u << MI355_NNZ_CR(j);
                }
            if (sl->non_zero_count_cache[scan8[CHROMA_DC_BLOCK_INDEX + 0]]) m->nnz_mask |= 1u << MI355_NNZ_CB_DC;
            if (sl->non_zero_count_cache[scan8[CHROMA_DC_BLOCK_INDEX + 1]]) m->nnz_mask |= 1u << MI355_NNZ_CR_DC;
        } else if (reads_coefs) memset(cf + 256 * cs, 0, 2 * ncc * cs);
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
                    if (r >= 0 && b->mbaff_frame && IS_INTERLACED(mb_type)) {
                        /* a field macroblock of an MBAFF frame: reference index r counts FIELDS, same parity first — what hl_decode_mb turns into
                         * ref_list[list][(16 + r) ^ (mb_y & 1)] (h264_mb_template.c:77-98; [16 + 2i] / [16 + 2i + 1] = the top / bottom field of frame i,
                         * h264_refs.c) — a slot of its own per (frame, parity), and the chroma vector's parity correction of h264_mb.c:287-291 */
                        const int par = (r & 1) ^ (sl->mb_y & 1);
                        const int slot = slot_of(b, sl->ref_list[list][r >> 1].parent, par);
                        if (slot < 0) { br_fail(b, "more reference pictures than the batched path holds"); __real_ff_h264_hl_decode_mb(h, sl); return; }
                        m->u.inter.ref_pic[list][q] = (uint8_t)slot;
                        m->u.inter.chroma_dy[list][q] = (int8_t)(2 * ((sl->mb_y & 1) - par));
                    } else
                    if (r >= 0) m->u.inter.ref_pic[list][q] = st->slices[0][si].ref_slot[list][r];
                    if (r >= 0 && b->field) m->u.inter.chroma_dy[list][q] = (int8_t)(2 * (b->parity - ((sl->ref_list[list][r].reference & 3) - 1)));
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
        if (b->c444) pack_planes_444(h, sl, st, idx, mb_type, luma_coded);
        if (b->c444) {
            if (luma_coded || (cbp & 0x30)) memset(mbp, 0, 16 * 48 * cs);
        } else {
            if (luma_coded) memset(mbp, 0, 256 * cs);
            if (cbp & 0x30) { memset(mbp + 256 * cs, 0, ncc * cs); memset(mbp + 512 * cs, 0, ncc * cs); }
        }
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

/* MI355_BRIDGE_SESSION: the packed picture through the whole-frame session (the AVHWAccel-shaped calls) */
static int submit_session(Bridge *b, H264Context *h, Staging *s, DevPic *cur)
{
    mi355_h264_picture_params pp;
    memset(&pp, 0, sizeof(pp));
    pp.surface = (int)(cur - b->pics);
    pp.nslots = b->nslots;
    pp.two_lists = b->uses_l1;
    pp.field = b->field ? 1 + b->parity : 0;
    for (int i = 0; i < b->nslots; i++) {
        const H264Picture *rp = b->slot_pic[i];
        DevPic *r = devpic_of(b, h, rp, 0);
        if (r && rp != h->cur_pic_ptr && (r->frame_num != rp->frame_num || r->poc != rp->poc || r->data0 != rp->f->data[0])) r = NULL;
        if (!r && !(r = devpic_upload(b, h, rp))) return -2;
        pp.ref_surface[i] = (int)(r - b->pics);
        pp.ref_parity[i] = b->slot_par[i] > 0;
    }
    if (mi355_h264_start_frame(b->sess, &pp) != 0) return -3;                        /* AVHWAccel.start_frame */
    /* AVHWAccel.decode_slice, once per run of consecutive macroblocks of a slice */
    for (int first = 0; first < b->nmb_pic; ) {
        const int si = s->mb[0][first].slice_id;
        int n = 1;
        while (first + n < b->nmb_pic && s->mb[0][first + n].slice_id == si) n++;
        if (mi355_h264_decode_slice(b->sess, &s->slices[0][si], first, n, NULL, s->mb[0] + first, s->mv[0] + (size_t)first * 32,
                                    b->uses_l1 ? s->mv[1] + (size_t)first * 32 : NULL, s->coef[0] + (size_t)first * 384) != 0) return -4;
        first += n;
    }
    if (mi355_h264_end_frame(b->sess) != 0) return -5;                               /* AVHWAccel.end_frame */
    const AVFrame *fr = h->cur_pic_ptr->f;
    uint8_t *const dst[3] = { fr->data[0], fr->data[1], fr->data[2] };
    const int st[3] = { fr->linesize[0], fr->linesize[1], fr->linesize[2] };
    if (mi355_h264_get_frame(b->sess, pp.surface, dst, st) != 0) return -6;
    b->pictures++;
    return 0;
}

static int submit_picture(Bridge *b, H264Context *h)
{
    Staging *s = &b->st[b->cur];
    DevPic *cur = devpic_of(b, h, h->cur_pic_ptr, 1);
    if (!cur) return -1;
    cur->frame_num = h->cur_pic_ptr->frame_num; cur->poc = h->cur_pic_ptr->poc; cur->data0 = h->cur_pic_ptr->f->data[0];
    if (b->sess) { s->pic = cur; return submit_session(b, h, s, cur); }
    int lw = 0;
    const int maxl = (b->mbaff_frame ? mi355_h264_intra_schedule_mbaff : mi355_h264_intra_schedule)(s->mb[0], b->mb_w, b->rows, s->ilist, s->istart, &lw);
    if (maxl < 0) return -1;
    for (int l = 0; l < maxl; l++) s->widths[l] = s->istart[l + 1] - s->istart[l];
    s->maxl = maxl;
    const int np = b->npass;
    for (int p = 0; p < np; p++) {
        /* pass p: 4:2:0 = the picture; 4:4:4 = plane p in the luma role, scratch surfaces in the chroma roles.  The kernels
         * read the staging block in place (device-visible host memory) */
        mi355_h264_frame *f = &s->desc[p];
        memset(f, 0, sizeof(*f));
        /* a field picture: every other line of the frame's planes — first line at the field's parity, strides doubled; its
         * references are fields addressed the same way (the unfiltered reconstruction is a surface of its own: plain rows) */
        const int fs = b->field ? 2 : 1;
        f->mb_width = b->mb_w; f->mb_height = b->rows;
        f->field_picture = b->field;
        f->surface_layout = b->tiled ? MI355_SURFACE_TILED : MI355_SURFACE_LINEAR;
        f->dst_stride[0] = fs * b->stride[0]; f->recon_stride[0] = b->stride[0];
        f->dst_stride[1] = fs * b->stride[1]; f->recon_stride[1] = b->stride[1];
        const size_t fo[2] = { b->field && b->parity ? (size_t)b->stride[0] : 0, b->field && b->parity ? (size_t)b->stride[1] : 0 };
        if (b->c444) {
            f->dst[0] = cur->plane[p] + fo[0]; f->recon[0] = b->recon[p];
            for (int k = 1; k < 3; k++) f->dst[k] = f->recon[k] = b->scratch_c[k - 1];
            f->dst_stride[1] = f->recon_stride[1] = b->stride[1];
        } else
            for (int k = 0; k < 3; k++) { f->dst[k] = cur->plane[k] + fo[k > 0]; f->recon[k] = b->recon[k]; }
        for (int i = 0; i < b->nslots; i++) {
            const H264Picture *rp = b->slot_pic[i];
            DevPic *r = devpic_of(b, h, rp, 0);
            /* a copy of what that entry held before?  (not asked of the frame being decoded: its second field predicts from its first) */
            if (r && rp != h->cur_pic_ptr && (r->frame_num != rp->frame_num || r->poc != rp->poc || r->data0 != rp->f->data[0])) r = NULL;
            if (!r && !(r = devpic_upload(b, h, rp))) return -2;
            const size_t ro[2] = { b->slot_par[i] > 0 ? (size_t)b->stride[0] : 0, b->slot_par[i] > 0 ? (size_t)b->stride[1] : 0 };
            if (b->c444) { f->ref[i][0] = r->plane[p] + ro[0]; f->ref[i][1] = b->scratch_c[0]; f->ref[i][2] = b->scratch_c[1]; }
            else for (int k = 0; k < 3; k++) f->ref[i][k] = r->plane[k] + ro[k > 0];
        }
        f->mb = s->mb[p]; f->mv[0] = s->mv[0]; f->mv[1] = b->uses_l1 ? s->mv[1] : NULL; f->coef = s->coef[p];
        f->slices = s->slices[p]; f->nslices = b->nslices;
        f->max_intra_level = maxl; f->intra_list = s->ilist; f->intra_level_start = s->istart; f->max_level_width = lw;
        f->flags = (maxl > 0 && s->istart[maxl] == b->nmb_pic ? MI355_FRAME_NO_INTER : 0) |      /* an I picture: the inter pass has nothing to do */
                   (b->mbaff_frame ? MI355_FRAME_MBAFF : 0);
        s->desc[np + p] = *f;                        /* the loop filter's view */
        s->desc[np + p].mb = s->mbd[p];
        if (s->d_side) {                             /* the second kernel set: side information from HBM (side_upload) */
            mi355_h264_frame *fd = &s->desc[np + p];
            fd->mb = (const mi355_h264_mb *)(s->d_side + s->side_off_mbd[p]);
            fd->slices = (const mi355_h264_slice *)(s->d_side + s->side_off_sl[p]);
            fd->mv[0] = (const int16_t *)(s->d_side + s->side_off_mv[0]);
            fd->mv[1] = b->uses_l1 ? (const int16_t *)(s->d_side + s->side_off_mv[1]) : NULL;
        }
    }
    s->pic = cur;
    s->field = b->field; s->parity = b->parity;
    s->nmb_pic = b->nmb_pic; s->nslices = b->nslices; s->uses_l1 = b->uses_l1;
    s->cls = fmt_class(b);
    /* the finished picture goes to the frame the decoder hands out (coded size; the reference crops on output) */
    const AVFrame *fr = h->cur_pic_ptr->f;
    for (int k = 0; k < 3; k++) { s->frame_data[k] = fr->data[k]; s->frame_linesize[k] = fr->linesize[k]; }
    if (getenv("MI355_BRIDGE_DEBUG")) {
        fprintf(stderr, "picture %lu mbaff %d maxl %d\n", b->pictures, b->mbaff_frame, maxl);
        for (int y = 0; y < b->rows; y++) { for (int x = 0; x < b->mb_w; x++) { const mi355_h264_mb *m = &s->mb[0][y * b->mb_w + x]; fprintf(stderr, " %c%c%d", (m->mb_type & 0x80) ? 'F' : 'p', (m->mb_type & 7) ? ((m->mb_type & 1) ? ((m->mb_type & 0x01000000) ? '8' : '4') : ((m->mb_type & 2) ? 'I' : 'P')) : 'i', m->intra_level); } fprintf(stderr, "\n"); }
    }
    if (b->direct) {
        if (mi355_memcpy_h2d_async(s->d_desc, s->desc, 2 * (size_t)np * sizeof(*s->desc), b->stream)) return -3;
        if (b->wide) {
            if (side_upload(b, s, b->stream)) return -3;
            if (mi355_h264_decode_frames_wide_dev(s->d_desc, np, b->mb_w, b->mb_h, maxl, s->widths, b->bit_depth, b->kidc, 3, b->stream) != 0 ||
                mi355_h264_decode_frames_wide_dev(s->d_desc + np, np, b->mb_w, b->mb_h, 0, NULL, b->bit_depth, b->kidc, b->mbaff_frame ? 12 : 4, b->stream) != 0) return -4;
        } else
        if (mi355_h264_recon_inter_sparse_dev(s->d_desc, np, b->mb_w, b->mb_h, b->stream) != 0 ||
            mi355_h264_recon_intra_all_dev(s->d_desc, np, b->mb_w, b->mb_h, maxl, s->widths, b->stream) != 0 ||
            mi355_h264_deblock_layouts_dev(s->d_desc + np, np, b->mb_w, b->mb_h, b->tiled ? MI355_LAYOUTS_TILED : MI355_LAYOUTS_LINEAR, b->stream) != 0) return -4;
        if (b->tiled) {
            convert_job(b, cur, s->out, s->cvt);
            if (mi355_h264_surface_convert_dev(s->cvt, 1, b->mb_w, b->mb_h, b->stream) != 0) return -5;
        } else if (mi355_memcpy_d2h_async(s->out, cur->plane[0], picture_bytes(b), b->stream)) return -5;
        if (mi355_event_record(s->done, b->stream)) return -5;
    } else {
        Disp *D = b->disp;
        pthread_mutex_lock(&D->mu);
        s->sub.done = 0; s->sub.rc = 0; s->sub.next = NULL;
        if (D->tail) D->tail->next = &s->sub; else D->head = &s->sub;
        D->tail = &s->sub;
        D->nqueued++;
        pthread_cond_signal(&D->work);
        pthread_mutex_unlock(&D->mu);
    }
    s->in_flight = 1;
    b->pictures++;
    return 0;
}

int __wrap_ff_h264_field_end(H264Context *h, H264SliceContext *sl, int in_setup)
{
    Bridge *b = br_tls;
    if (b && b->state > 0 && b->open) {
        b->open = 0;
        /* a damaged stream: the decoder gave up on a slice and macroblocks are missing.  What was delivered is reconstructed and
         * brought back (the missing ones hold whatever the device made of empty records, as they hold stale data in the
         * reference's frame); then this decoder continues on the reference's C path — its error concealment, where built in,
         * rewrites the frame on the host (ff_er_frame_end below), and the device copy of the picture would no longer be what
         * later pictures must predict from. */
        const int incomplete = b->mbs_packed != b->nmb_pic;
        if (b->null_submit) {
            b->pictures++;
        } else if (submit_picture(b, h) != 0) {
            /* the picture is lost for this path; what was enqueued must drain before the host touches the frames again */
            finish_all(b);
            br_fail(b, "submitting a picture to the device failed");
        } else if (incomplete) {
            finish_all(b);
            br_fail(b, "incomplete picture (damaged stream)");
        } else {
            /* wait only for what the decoder is about to hand out: h->output_frame was chosen when the picture started
             * (h264_select_output_frame, h264_slice.c:1173-1290, called from h264_field_start :1528) and shares its buffers
             * with the H264Picture it refers to; without MI355_BRIDGE_LAZY every picture is complete before this returns */
            const uint8_t *out0 = h->output_frame && h->output_frame->buf[0] ? h->output_frame->data[0] : NULL;
            for (int k = 0; k < 2; k++) {
                Staging *s = &b->st[b->cur ^ 1 ^ k];         /* the older submission first */
                if (s->in_flight && (!b->lazy || (out0 && s->frame_data[0] == out0)))
                    if (finish_set(b, s) != 0) br_fail(b, "a picture did not come back from the device");
            }
        }
    }
    /* reference marking runs inside the real function: memory_management_control_operation 5 rewrites the picture's frame_num
     * and POC (h264_refs.c) — the device copy is labelled with what the picture holds afterwards */
    DevPic *dp = b && b->state > 0 ? b->st[b->cur].pic : NULL;
    const H264Picture *hp = h->cur_pic_ptr;
    const int ret = __real_ff_h264_field_end(h, sl, in_setup);
    if (dp && hp && dp->owner == hp) { dp->frame_num = hp->frame_num; dp->poc = hp->poc; }
    return ret;
}

/* for hosts that want the numbers (the throughput harness prints them) */
void mi355_h264_bridge_stats(unsigned long *pictures, unsigned long *staging_waits, int *active)
{
    Bridge *b = br_tls;
    if (pictures) *pictures = b ? b->pictures : 0;
    if (staging_waits) *staging_waits = b ? b->waits : 0;
    if (active) *active = b ? b->state : 0;
}
/* launch sets the dispatcher issued and the pictures they held (process-wide) */
void mi355_h264_bridge_batch_stats(unsigned long *batches, unsigned long *pictures)
{
    unsigned long nb = 0, np = 0;
    pthread_mutex_lock(&disps_mu);
    for (int i = 0; i < DISP_MAX_DEVICES && disps_ready; i++) {
        pthread_mutex_lock(&disps[i].mu);
        nb += disps[i].batches; np += disps[i].pictures;
        pthread_mutex_unlock(&disps[i].mu);
    }
    pthread_mutex_unlock(&disps_mu);
    if (batches) *batches = nb;
    if (pictures) *pictures = np;
}
/* a decoder thread that ends (or flushes with MI355_BRIDGE_LAZY) calls this: everything it submitted is complete and in
 * its frames afterwards */
void mi355_h264_bridge_drain(void)
{
    Bridge *b = br_tls;
    if (b && b->state > 0) finish_all(b);
}
