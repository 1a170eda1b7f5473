/*
 * mi355_hevc_lf_bridge.c — the reference's HEVC decoder with its in-loop DEBLOCKING done per PICTURE on the MI355X.
 *
 * The reference deblocks CTB by CTB as the slice decoder advances (hls_slice_data -> ff_hevc_hls_filters ->
 * ff_hevc_hls_filter -> deblocking_filter_CTB, hevcdec.c:2334-2339, hevc_filter.c:728-745).  Linked into the decoder with
 *     -Wl,--wrap=ff_hevc_hls_filters,--wrap=ff_hevc_hls_filter,--wrap=ff_hevc_deblocking_boundary_strengths
 * this file turns that into one call per picture: the per-CTB calls are dropped, and when the slice decoder reports the
 * picture's last CTB (hevcdec.c:2337-2339) the frame-level arrays it left behind — vertical_bs / horizontal_bs, qp_y_tab,
 * is_pcm, the per-CTB DBParams — go to mi355_hevc_deblock_pictures_dev() (include/mi355_hevc_batch.h) exactly as they lie
 * in HEVCContext, with the unfiltered picture; the filtered picture comes back into s->frame.  Compiled against the
 * reference's headers (oracle/Makefile, _ref/hevc_lf_emu / _ref/hevc_lf_gpu).
 *
 * Sample adaptive offset follows on the device (sequences with sps->sao_enabled): the reference filters a CTB's samples in
 * up to four pieces, each when the deblocking of the CTBs around it is final (sao_filter_CTB, hevc_filter.c:188-313: the
 * CTB itself minus the strips its right / lower neighbours will still deblock, plus those strips of its left / upper /
 * upper-left neighbours, each with the OWNER's parameters).  On a fully deblocked picture the pieces are independent:
 * this file lists them all (one mi355_hevc_sao_ctb_job per CTB component: the up to four calls that filter the CTB's own samples,
 * with the edge flags the reference derives from slice addresses and slice_loop_filter_across_slices_enabled_flag)
 * and mi355_hevc_sao_ctbs_dev() runs them in one
 * launch, reading the deblocked picture and writing the picture the decoder outputs and predicts from (s->sao_frame).
 *
 * Scope of this binding: 4:2:0, decoders without frame threads (progress is reported once per picture); tiles included (the
 * edge rules of loop_filter_across_tiles_enabled_flag are applied where the edges are marked and where the SAO pieces are listed).
 * Everything else keeps the reference's path — MI355_HEVC_LF_PLAIN=1 keeps it for every picture.
 * One decoder = one stream of pictures here; a host with many decoders batches pictures of all of them into ONE call
 * (`npics`), as contrib/libav/mi355_h264_bridge.c does for H.264.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/hevcdec.h"
#include "mi355_hevc_batch.h"
#include "mi355dsp.h"             /* mi355_init (the table structs are skipped: the reference's headers came first) */
#include "mi355_h264_frame.h"      /* the runtime entry points: mi355_malloc / mi355_memcpy_* / mi355_sync */

void __real_ff_hevc_hls_filters(HEVCContext *s, int x_ctb, int y_ctb, int ctb_size);
void __real_ff_hevc_hls_filter(HEVCContext *s, int x, int y);
void __real_ff_hevc_deblocking_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size);

/* one decoder per thread (as the reconstruction bridge): the state is the thread's */
static unsigned long g_lf_pictures;                    /* pictures filtered on the device, all decoders (threads) of the process */
static __thread struct {
    size_t plane_bytes[3], bs_bytes, qp_bytes, pcm_bytes, db_bytes;
    uint8_t *plane[3], *vbs, *hbs, *qp, *pcm, *db;
    uint8_t *out[3], *jobs;                /* SAO: the output picture, the job list */
    size_t out_bytes[3], jobs_bytes;
    mi355_hevc_sao_ctb_job *host_jobs;
    size_t host_jobs_n;
    /* boundary strengths on the device: what the walk over the coding tree knows about every 4x4 cell's left / top side */
    uint8_t *edge_flags, *d_edge, *d_mvf, *d_cbf;
    size_t edge_cells, d_edge_bytes, d_mvf_bytes, d_cbf_bytes;
    mi355_hevc_bs_picture *d_bs_desc;
    const void *bs_ref;                    /* the picture the flags belong to */
    int bs_lists_set, bs_uniform, bs_slice;
    /* the calls of ff_hevc_deblocking_boundary_strengths the picture has made so far, not yet passed on (see bs_replay) */
    struct BsCall { uint16_t x0, y0; uint8_t log2, boundary_flags, across; } *bs_calls;
    size_t bs_ncalls, bs_ccalls;
    int bs_deferred;
    unsigned long bs_calls_dropped;
    uint8_t *h_out; size_t h_out_bytes;             /* pinned bounce buffer of the picture coming back */
    int32_t ref_poc[2][16];
    unsigned long bs_pictures;
    mi355_hevc_lf_picture *desc;
    void *stream;
    unsigned long pictures;
    int plain, failed;
    long min_pixels;                                /* pictures smaller than this stay with the reference's C path (MI355_HEVC_BRIDGE_MIN_PIXELS) */
} lf;

static void fail(const char *what);
static int active(const HEVCContext *s)
{
    static __thread int init;
    if (!init) {
        init = 1;
        lf.plain = getenv("MI355_HEVC_LF_PLAIN") != NULL;
        /* A scheduling policy: one decoder's picture is a launch set per dependency level, a few tens of microseconds each whatever the picture's
         * size, while the C functions' time falls with the area — measured (bench.py hevc_bridge_*, one decoder): 1920x1080 39 pictures/s against
         * 34, 832x480 120 against 172, 136x72 560 against 5400.  Below this many luma samples the bridges step aside (a sequence is all or
         * nothing: the reference pictures must live where the decoder's path expects them).  0 = always on the device (the tests). */
        lf.min_pixels = getenv("MI355_HEVC_BRIDGE_MIN_PIXELS") ? atol(getenv("MI355_HEVC_BRIDGE_MIN_PIXELS")) : 1500000L;
    }
    if (lf.plain || lf.failed || (s->avctx->active_thread_type & FF_THREAD_FRAME) || s->ps.sps->chroma_format_idc != 1 ||
        (long)s->ps.sps->width * s->ps.sps->height < lf.min_pixels) return 0;
    /* the device is set up when the first picture that goes there arrives: a process that only ever decodes small pictures never pays for a
     * HIP context (~0.1 s — with the default policy that was the whole difference to the C decoder on the small streams of bench.py: r04o) */
    static __thread int dev_init;
    if (!dev_init) {
        dev_init = 1;
        /* a decoder waits for its picture's chain every picture, and a host runs many decoders: waiting threads sleep (16 decoders on 16 cores: 183 -> 226
         * pictures/s, 32: 214 -> 348 — the spinning had cost more processor time than the parsing; MI355_BLOCKING_SYNC=0 spins) */
        mi355_prefer_blocking_sync(1);
        if (mi355_init(getenv("MI355_DEVICE") ? atoi(getenv("MI355_DEVICE")) : 0) != 0) fail("no MI355X");
        /* this decoder's own stream for its passes: with several decoders in the process a synchronisation waits for this decoder's work, not for
         * every other decoder's launch chain (the copies are host-synchronous hipMemcpy calls: done when they return).  MI355_HEVC_BRIDGE_DEFAULT_STREAM=1:
         * everything on the default stream, as before round 4's last session */
        else if (!getenv("MI355_HEVC_BRIDGE_DEFAULT_STREAM")) lf.stream = mi355_stream_create();
    }
    return !lf.failed;
}

static int ensure(uint8_t **p, size_t *have, size_t want)
{
    if (*p && *have >= want) return 0;
    if (*p) mi355_free(*p);
    *p = mi355_malloc(want);
    *have = *p ? want : 0;
    return *p ? 0 : -1;
}

/* the reconstruction bridge (contrib/libav/mi355_hevc_bridge.c, when linked in) asks per picture: the two must agree */
int mi355_hevc_lf_bridge_active(const HEVCContext *s) { return active(s); }
/* ... and hands over the picture it reconstructed on the device: 1 = cur[] holds the unfiltered picture (device planes), the
 * finished one goes to fin[]; 0 = not its picture (the host frame is uploaded, as without it); < 0 = failed */
int mi355_hevc_recon_finish(HEVCContext *s, uint8_t *cur[3], uint8_t *fin[3]) __attribute__((weak));
void *mi355_hevc_recon_done_event(void) __attribute__((weak));

static void fail(const char *what)
{
    fprintf(stderr, "mi355 hevc lf bridge: %s; the reference's path takes over\n", what);
    lf.failed = 1;
}


/* ---- boundary strengths.  The reference derives them block by block while it walks the coding tree
 * (ff_hevc_deblocking_boundary_strengths, hevc_filter.c:585-725, called for transform-tree leaves, PCM and residual-free
 * coding units).  Here the walk only MARKS the cell sides it would examine — block edges on the 8x8 grid that are
 * filtered (slice edges with filtering across them off are left unmarked) and the grid lines inside blocks that are not
 * intra — and mi355_hevc_boundary_strengths_dev() derives every strength of the picture in one pass from tab_mvf /
 * cbf_luma as they lie in the decoder.  The device pass compares reference pictures through ONE table per picture: a
 * picture whose slices use different lists keeps the host's strengths (the reference's function still runs; a binding that
 * knows its streams may drop it).  MI355_HEVC_BS_HOST=1 keeps the host's strengths for every picture. */
static void bs_begin(const HEVCContext *s)
{
    const size_t cells = (size_t)(s->ps.sps->width >> 2) * (s->ps.sps->height >> 2);
    if (lf.edge_cells < cells) {
        free(lf.edge_flags);
        lf.edge_flags = malloc(cells);
        lf.edge_cells = lf.edge_flags ? cells : 0;
    }
    if (lf.edge_flags) memset(lf.edge_flags, 0, cells);
    lf.bs_ref = s->ref;
    lf.bs_lists_set = 0;
    lf.bs_uniform = lf.edge_flags != NULL && !getenv("MI355_HEVC_BS_HOST");
    lf.bs_slice = -1;
    lf.bs_ncalls = 0;
    lf.bs_deferred = lf.bs_uniform;
    memset(lf.ref_poc, 0, sizeof(lf.ref_poc));
}

/* While a picture's strengths are expected from the device, the reference's own function is NOT run (it is a tenth of the decoder's
 * host time next to the bridge): its calls are noted — position, size, and the three things it reads that change from block to
 * block (lc->boundary_flags, the slice's filter-across flag, the slice's list through s->ref->refPicList) — and run only if the
 * picture turns out to need the host's strengths (a slice with other lists than the first).  Everything else the function reads
 * (tab_mvf, cbf_luma, the per-CTB list table behind ff_hevc_get_ref_list) stays as it is until the picture ends. */
static void bs_replay(HEVCContext *s)
{
    HEVCLocalContext *lc = &s->HEVClc;
    const int flags = lc->boundary_flags, across = s->sh.slice_loop_filter_across_slices_enabled_flag;
    RefPicList *const rpl = s->ref->refPicList;
    for (size_t i = 0; i < lf.bs_ncalls; i++) {
        const struct BsCall *c = &lf.bs_calls[i];
        lc->boundary_flags = c->boundary_flags;
        s->sh.slice_loop_filter_across_slices_enabled_flag = c->across;
        s->ref->refPicList = ff_hevc_get_ref_list(s, s->ref, c->x0, c->y0);
        __real_ff_hevc_deblocking_boundary_strengths(s, c->x0, c->y0, c->log2);
    }
    lc->boundary_flags = flags; s->sh.slice_loop_filter_across_slices_enabled_flag = across; s->ref->refPicList = rpl;
    lf.bs_ncalls = 0;
    lf.bs_deferred = 0;
}

/* a new slice: its reference lists must be the picture's for the device to rate motion edges with one list table */
static void bs_slice_lists(HEVCContext *s)
{
    if (!lf.bs_uniform || lf.bs_slice == (int)s->sh.slice_addr) return;
    lf.bs_slice = (int)s->sh.slice_addr;
    if (s->sh.slice_type != HEVC_SLICE_I && s->ref->refPicList) {
        int32_t now[2][16];
        memset(now, 0, sizeof(now));
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < s->ref->refPicList[l].nb_refs && i < 16; i++) now[l][i] = s->ref->refPicList[l].list[i];
        if (!lf.bs_lists_set) { memcpy(lf.ref_poc, now, sizeof(now)); lf.bs_lists_set = 1; }
        else if (memcmp(lf.ref_poc, now, sizeof(now))) { lf.bs_uniform = 0; if (lf.bs_deferred) bs_replay(s); }
    }
}

void __wrap_ff_hevc_deblocking_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_trafo_size)
{
    if (!active(s)) { __real_ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size); return; }
    if (lf.bs_ref != s->ref) bs_begin(s);
    if (lf.bs_uniform) bs_slice_lists(s);                            /* may find other lists: runs the calls noted so far */
    if (!lf.bs_uniform) { __real_ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size); return; }
    const HEVCSPS *sps = s->ps.sps;
    const HEVCLocalContext *lc = &s->HEVClc;
    if (lf.bs_ncalls == lf.bs_ccalls) {
        const size_t c = lf.bs_ccalls ? 2 * lf.bs_ccalls : 4096;
        struct BsCall *q = realloc(lf.bs_calls, c * sizeof(*q));
        if (!q) { lf.bs_uniform = 0; bs_replay(s); __real_ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_trafo_size); return; }
        lf.bs_calls = q; lf.bs_ccalls = c;
    }
    lf.bs_calls[lf.bs_ncalls++] = (struct BsCall){ (uint16_t)x0, (uint16_t)y0, (uint8_t)log2_trafo_size, (uint8_t)lc->boundary_flags,
                                                   (uint8_t)s->sh.slice_loop_filter_across_slices_enabled_flag };
    const int size = 1 << log2_trafo_size, cw = sps->width >> 2, ctb_mask = (1 << sps->log2_ctb_size) - 1;
    const int inner = log2_trafo_size > sps->log2_min_pu_size &&
                      !s->ref->tab_mvf[(y0 >> sps->log2_min_pu_size) * sps->min_pu_width + (x0 >> sps->log2_min_pu_size)].is_intra;
    /* a slice or tile edge with filtering across it switched off is not an edge (hevc_filter.c:599-607, :661-669) */
    const int no_tile = !s->ps.pps->loop_filter_across_tiles_enabled_flag;
    const int top = y0 > 0 && !(y0 & 7) &&
                    !(((!s->sh.slice_loop_filter_across_slices_enabled_flag && (lc->boundary_flags & BOUNDARY_UPPER_SLICE)) ||
                       (no_tile && (lc->boundary_flags & BOUNDARY_UPPER_TILE))) && !(y0 & ctb_mask));
    const int left = x0 > 0 && !(x0 & 7) &&
                     !(((!s->sh.slice_loop_filter_across_slices_enabled_flag && (lc->boundary_flags & BOUNDARY_LEFT_SLICE)) ||
                        (no_tile && (lc->boundary_flags & BOUNDARY_LEFT_TILE))) && !(x0 & ctb_mask));
    for (int j = 0; j < size; j += 4)
        for (int i = 0; i < size; i += 4) {
            uint8_t f = 0;
            if (i == 0 && left) f |= MI355_HEVC_EDGE_L_BLOCK;
            if (j == 0 && top) f |= MI355_HEVC_EDGE_T_BLOCK;
            if (inner && i && !(i & 7)) f |= MI355_HEVC_EDGE_L_INNER;
            if (inner && j && !(j & 7)) f |= MI355_HEVC_EDGE_T_INNER;
            if (f) lf.edge_flags[((y0 + j) >> 2) * cw + ((x0 + i) >> 2)] |= f;
        }
}

/* -> 1: vertical_bs / horizontal_bs of the picture were derived on the device (lf.vbs / lf.hbs hold them) */
static int bs_on_device_try(HEVCContext *s, size_t bs_bytes)
{
    const HEVCSPS *sps = s->ps.sps;
    if (lf.bs_ref != s->ref) bs_begin(s);                           /* no slice of the picture is deblocked: no edge marked */
    if (!lf.bs_uniform || (sps->width & 7) || (sps->height & 7)) return 0;
    const size_t cells = (size_t)(sps->width >> 2) * (sps->height >> 2);
    const size_t mvf = (size_t)sps->min_pu_width * sps->min_pu_height * sizeof(mi355_hevc_mvfield);
    const size_t cbf = (size_t)sps->min_tb_width * sps->min_tb_height;
    if (sizeof(MvField) != sizeof(mi355_hevc_mvfield)) return 0;
    if (ensure(&lf.d_edge, &lf.d_edge_bytes, cells) || ensure(&lf.d_mvf, &lf.d_mvf_bytes, mvf) || ensure(&lf.d_cbf, &lf.d_cbf_bytes, cbf)) return 0;
    if (!lf.d_bs_desc && !(lf.d_bs_desc = mi355_malloc(sizeof(*lf.d_bs_desc)))) return 0;
    mi355_hevc_bs_picture d;
    memset(&d, 0, sizeof(d));
    d.width = sps->width; d.height = sps->height;
    d.log2_min_pu_size = sps->log2_min_pu_size; d.log2_min_tb_size = sps->log2_min_tb_size;
    d.min_pu_width = sps->min_pu_width; d.min_tb_width = sps->min_tb_width;
    d.bs_width = s->bs_width;
    d.tab_mvf = (const mi355_hevc_mvfield *)lf.d_mvf; d.cbf_luma = lf.d_cbf; d.edge_flags = lf.d_edge;
    memcpy(d.ref_poc, lf.ref_poc, sizeof(d.ref_poc));
    d.vertical_bs = lf.vbs; d.horizontal_bs = lf.hbs;
    (void)bs_bytes;
    int rc = mi355_memcpy_h2d(lf.d_edge, lf.edge_flags, cells) | mi355_memcpy_h2d(lf.d_mvf, s->ref->tab_mvf, mvf) |
             mi355_memcpy_h2d(lf.d_cbf, s->cbf_luma, cbf) | mi355_memcpy_h2d(lf.d_bs_desc, &d, sizeof(d));
    if (rc) return 0;
    if (mi355_hevc_boundary_strengths_dev(lf.d_bs_desc, 1, sps->width, sps->height, lf.stream) != 0) return 0;
    lf.bs_pictures++;
    return 1;
}

static int bs_on_device(HEVCContext *s, size_t bs_bytes)
{
    const int done = bs_on_device_try(s, bs_bytes);
    if (!done && lf.bs_deferred) bs_replay(s);                      /* the host's strengths after all: the calls noted, now */
    else if (done) lf.bs_calls_dropped += lf.bs_ncalls;
    return done;
}

/* The pieces of sao_filter_CTB for every CTB of the picture.  Piece k of CTB (cx, cy) belongs to the CTB at
 * (cx - (k >> 1), cy - (k & 1)) — k = the reference's class number: 0 the CTB itself, 1 the strip of the CTB above,
 * 2 of the CTB to the left, 3 of the one above-left — and is filtered with that CTB's parameters. */
static int sao_jobs(const HEVCContext *s, uint8_t *const dst[3], uint8_t *const src[3], mi355_hevc_sao_ctb_job *jobs)
{
    const HEVCSPS *sps = s->ps.sps;
    const int cw = sps->ctb_width, chn = sps->ctb_height;
    const int no_tile = s->ps.pps->tiles_enabled_flag && !s->ps.pps->loop_filter_across_tiles_enabled_flag;
    const int *tid = s->ps.pps->tile_id, *ts = s->ps.pps->ctb_addr_rs_to_ts;
    int n = 0;
    /* one job per CTB component = the OWNER's samples: what the reference filters with this CTB's parameters while this CTB
     * (class 0), the CTB to its right (class 2), the CTB below (class 1) and the one below-right (class 3) pass through */
    for (int oy = 0; oy < chn; oy++)
        for (int ox = 0; ox < cw; ox++) {
            const SAOParams *p = &s->sao[oy * cw + ox];
            for (int c = 0; c < 3; c++) {
                const int sh = c ? 1 : 0;
                const int size = (1 << sps->log2_ctb_size) >> sh;
                mi355_hevc_sao_ctb_job *j = &jobs[n++];
                memset(j, 0, sizeof(*j));
                const size_t off = (size_t)(oy * size) * s->frame->linesize[c] + ((size_t)(ox * size) << sps->pixel_shift);
                j->dst = dst[c] + off; j->src = src[c] + off;
                j->stride = s->frame->linesize[c];
                j->c_idx = (uint8_t)c;
                for (int k = 0; k < 4; k++) {
                    /* the CTB the reference calls class k for */
                    const int cx = ox + (k >> 1), cy = oy + (k & 1);
                    if (cx >= cw || cy >= chn) continue;
                    const int here = cy * cw + cx;
                    const int has_l = cx > 0, has_u = cy > 0;
                    /* slice address and "filter across slice edges" of the four CTBs around that CTB's top-left corner */
                    const int a_c = s->tab_slice_address[here], a_l = has_l ? s->tab_slice_address[here - 1] : a_c;
                    const int a_u = has_u ? s->tab_slice_address[here - cw] : a_c, a_ul = has_l && has_u ? s->tab_slice_address[here - cw - 1] : a_c;
                    const int f_c = s->filter_slice_edges[here], f_l = has_l ? s->filter_slice_edges[here - 1] : 1, f_u = has_u ? s->filter_slice_edges[here - cw] : 1;
                    /* tile edges with filtering across them off (hevc_filter.c:208-266): a tile edge runs through the whole picture,
                     * so the column / row of THIS CTB decides for the pieces above / to the left too */
                    const int lt = no_tile && has_l && tid[ts[here]] != tid[ts[here - 1]];
                    const int ut = no_tile && has_u && tid[ts[here]] != tid[ts[here - cw]];
                    uint8_t vert[4] = { 0 }, horiz[4] = { 0 }, diag[4] = { 0 };
                    if (has_l) vert[0] = vert[2] = (!f_c && a_c != a_l) || lt;
                    if (has_u) horiz[0] = horiz[1] = (!f_c && a_c != a_u) || ut;
                    if (has_l && has_u) {
                        vert[1] = vert[3] = (!f_u && a_u != a_ul) || lt;
                        horiz[2] = horiz[3] = (!f_l && a_l != a_ul) || ut;
                        diag[0] = diag[3] = (!f_c && a_c != a_ul) || lt || ut;
                        /* the anti-diagonal joins the left and the upper CTB: the later of the two decides */
                        diag[1] = diag[2] = (a_l > a_u ? !f_l : a_l < a_u ? !f_u : 0) || lt || ut;
                    }
                    const int x0 = cx * size, y0 = cy * size;
                    mi355_hevc_sao_piece *q = &j->piece[j->npieces++];
                    for (int e = 0; e < 5; e++) q->offset_val[e] = p->offset_val[c][e];
                    q->cls = (uint8_t)k;
                    q->type = p->type_idx[c] == SAO_EDGE ? 2 : p->type_idx[c] == SAO_BAND ? 1 : 0;
                    q->eo_class = (uint8_t)p->eo_class[c]; q->band_position = p->band_position[c];
                    q->vert_edge = vert[k]; q->horiz_edge = horiz[k]; q->diag_edge = diag[k];
                    q->borders = (uint8_t)((cx == 0) | ((cy == 0) << 1) | ((cx == cw - 1) << 2) | ((cy == chn - 1) << 3));
                    q->dx = (int16_t)((k >> 1) * size); q->dy = (int16_t)((k & 1) * size);
                    q->width = (int16_t)FFMIN(size, (sps->width >> sh) - x0); q->height = (int16_t)FFMIN(size, (sps->height >> sh) - y0);
                }
            }
        }
    return n;
}

/* a finished plane comes back through a pinned buffer: the copy over the link runs at the link's rate (into the decoder's pageable
 * frame it is staged by the runtime in small pieces), the copy into the frame at memory speed */
static int picture_d2h(uint8_t *dst, const uint8_t *dev, size_t n)
{
    if (lf.h_out_bytes < n) {
        if (lf.h_out) mi355_host_free(lf.h_out);
        lf.h_out = mi355_host_alloc(n);
        lf.h_out_bytes = lf.h_out ? n : 0;
    }
    if (!lf.h_out) return mi355_memcpy_d2h(dst, dev, n);
    if (mi355_memcpy_d2h(lf.h_out, dev, n) != 0) return -1;
    memcpy(dst, lf.h_out, n);
    return 0;
}

static int filter_picture(HEVCContext *s)
{
    const HEVCSPS *sps = s->ps.sps;
    const int h[3] = { sps->height, sps->height >> 1, sps->height >> 1 };
    size_t sz[3];
    /* lf.stream is this decoder's own stream; the plain copies below are host-synchronous (hipMemcpy): what a pass reads is in device memory
     * before the pass is launched, and a buffer is only written again after the synchronisation that ended the picture before */
    uint8_t *plane[3], *out[3], *rcur[3], *rfin[3];
    const int on_dev = mi355_hevc_recon_finish ? mi355_hevc_recon_finish(s, rcur, rfin) : 0;
    if (on_dev < 0) return -3;                                   /* the picture exists nowhere: see __wrap_ff_hevc_hls_filter */
    if (on_dev > 0 && mi355_hevc_recon_done_event) {             /* the reconstruction's launches are in the bridge's stream: this stream follows them */
        void *ev = mi355_hevc_recon_done_event();
        if (ev && mi355_stream_wait_event(lf.stream, ev) != 0) return -2;
    }
    for (int i = 0; i < 3; i++) {
        sz[i] = (size_t)s->frame->linesize[i] * h[i];
        if (s->frame->linesize[i] <= 0) return -1;
        if (on_dev) { plane[i] = rcur[i]; continue; }
        if (ensure(&lf.plane[i], &lf.plane_bytes[i], sz[i])) return -1;
        plane[i] = lf.plane[i];
    }
    const size_t bs = 2 * (size_t)s->bs_width * (s->bs_height + 1);
    const size_t qp = (size_t)((sps->width >> sps->log2_min_cb_size) + 1) * ((sps->height >> sps->log2_min_cb_size) + 1);
    const size_t pcm = (size_t)sps->min_pu_width * sps->min_pu_height;
    const size_t db = (size_t)sps->ctb_width * sps->ctb_height * sizeof(*s->deblock);
    size_t have;
    have = lf.bs_bytes; if (ensure(&lf.vbs, &have, bs)) return -1;
    if (ensure(&lf.hbs, &lf.bs_bytes, bs)) return -1;
    if (ensure(&lf.qp, &lf.qp_bytes, qp) || ensure(&lf.pcm, &lf.pcm_bytes, pcm) || ensure(&lf.db, &lf.db_bytes, db)) return -1;
    if (!lf.desc && !(lf.desc = mi355_malloc(sizeof(*lf.desc)))) return -1;

    mi355_hevc_lf_picture d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < 3; i++) { d.data[i] = plane[i]; d.linesize[i] = s->frame->linesize[i]; }
    d.width = sps->width; d.height = sps->height;
    d.log2_ctb_size = sps->log2_ctb_size;
    d.log2_min_cb_size = sps->log2_min_cb_size;
    d.log2_min_pu_size = sps->log2_min_pu_size;
    d.min_cb_width = sps->min_cb_width;
    d.min_pu_width = sps->min_pu_width; d.min_pu_height = sps->min_pu_height;
    d.ctb_width = sps->ctb_width;
    d.bs_width = s->bs_width;
    d.vertical_bs = lf.vbs; d.horizontal_bs = lf.hbs;
    d.qp_y_tab = (const int8_t *)lf.qp;
    d.is_pcm = lf.pcm;
    d.deblock = (const mi355_hevc_db_params *)lf.db;
    d.pcmf = (sps->pcm_enabled_flag && sps->pcm.loop_filter_disable_flag) || s->ps.pps->transquant_bypass_enable_flag;
    d.cb_qp_offset = s->ps.pps->cb_qp_offset; d.cr_qp_offset = s->ps.pps->cr_qp_offset;

    int rc = 0;
    for (int i = 0; i < 3 && !on_dev; i++) rc |= mi355_memcpy_h2d(plane[i], s->frame->data[i], sz[i]);
    if (!bs_on_device(s, bs)) rc |= mi355_memcpy_h2d(lf.vbs, s->vertical_bs, bs) | mi355_memcpy_h2d(lf.hbs, s->horizontal_bs, bs);
    lf.bs_ref = NULL;
    rc |= mi355_memcpy_h2d(lf.qp, s->qp_y_tab, qp) | mi355_memcpy_h2d(lf.pcm, s->is_pcm, pcm) | mi355_memcpy_h2d(lf.db, s->deblock, db);
    rc |= mi355_memcpy_h2d(lf.desc, &d, sizeof(d));
    if (rc) return -2;
    if (mi355_hevc_deblock_pictures_dev(lf.desc, 1, sps->width, sps->height, sps->bit_depth, lf.stream) != 0) return -2;
    if (!sps->sao_enabled) {
        if (mi355_sync(lf.stream) != 0) return -2;
        for (int i = 0; i < 3; i++) rc |= picture_d2h(s->frame->data[i], plane[i], sz[i]);
        if (rc) return -2;
        lf.pictures++; __atomic_fetch_add(&g_lf_pictures, 1ul, __ATOMIC_RELAXED);
        return 0;
    }
    /* SAO: deblocked picture -> the picture the decoder keeps, one job per CTB component (the pieces that make up its own samples) */
    const size_t max_jobs = (size_t)sps->ctb_width * sps->ctb_height * 3;
    if (lf.host_jobs_n < max_jobs) {
        free(lf.host_jobs);
        lf.host_jobs = malloc(max_jobs * sizeof(*lf.host_jobs));
        lf.host_jobs_n = lf.host_jobs ? max_jobs : 0;
        if (!lf.host_jobs) return -1;
    }
    if (ensure(&lf.jobs, &lf.jobs_bytes, max_jobs * sizeof(*lf.host_jobs))) return -1;
    for (int i = 0; i < 3; i++) {
        if (on_dev) { out[i] = rfin[i]; continue; }
        if (ensure(&lf.out[i], &lf.out_bytes[i], sz[i])) return -1;
        out[i] = lf.out[i];
    }
    const int n = sao_jobs(s, out, plane, lf.host_jobs);
    if (mi355_sync(lf.stream) != 0) return -2;
    if (n) rc |= mi355_memcpy_h2d(lf.jobs, lf.host_jobs, (size_t)n * sizeof(*lf.host_jobs));
    if (rc) return -2;
    if (n && mi355_hevc_sao_ctbs_dev((const mi355_hevc_sao_ctb_job *)lf.jobs, n, sps->bit_depth, lf.stream) != 0) return -2;
    /* cu_transquant_bypass / pcm_loop_filter_disabled blocks keep their deblocked samples: the decoder copies them from s->frame into
     * s->sao_frame on the host afterwards (restore_tqb_pixels, hevcdec.c:2343-2370, called :2596-2600).  A picture reconstructed on the
     * device has them only there: the same copy for the device picture (window jobs: plain block copies), and the deblocked picture goes
     * to the host's s->frame too, for the decoder's own copy */
    const int restore = on_dev && (s->ps.pps->transquant_bypass_enable_flag || (sps->pcm.loop_filter_disable_flag && sps->pcm_enabled_flag));
    if (restore) {
        const int mps = 1 << sps->log2_min_pu_size;
        size_t nj = 0;
        for (size_t i = 0; i < pcm; i++) nj += s->is_pcm[i] != 0;
        if (nj) {
            mi355_edge_emu_job *hj = malloc(3 * nj * sizeof(*hj));
            uint8_t *dj = mi355_malloc(3 * nj * sizeof(*hj));
            if (!hj || !dj) { free(hj); if (dj) mi355_free(dj); return -1; }
            size_t k = 0;
            for (int y = 0; y < sps->min_pu_height; y++)
                for (int x = 0; x < sps->min_pu_width; x++) {
                    if (!s->is_pcm[y * sps->min_pu_width + x]) continue;
                    for (int c = 0; c < 3; c++) {
                        const int hs = sps->hshift[c], vs = sps->vshift[c];
                        const size_t off = (size_t)((y * mps) >> vs) * s->frame->linesize[c] + ((size_t)((x * mps) >> hs) << sps->pixel_shift);
                        mi355_edge_emu_job *j = &hj[k++];
                        memset(j, 0, sizeof(*j));
                        j->dst = out[c] + off; j->src = plane[c] + off;
                        j->dst_stride = j->src_stride = s->frame->linesize[c];
                        /* the reference copies `min_pu_size >> hshift` BYTES per row (hevcdec.c:2357-2363): half the block's samples above 8 bit */
                        j->block_w = (mps >> hs) >> sps->pixel_shift; j->block_h = mps >> vs;
                        j->src_x = (x * mps) >> hs; j->src_y = (y * mps) >> vs; j->w = sps->width >> hs; j->h = sps->height >> vs;
                    }
                }
            rc = mi355_memcpy_h2d(dj, hj, k * sizeof(*hj));
            if (!rc && mi355_edge_emu_batch_dev((const mi355_edge_emu_job *)dj, (int)k, sps->bit_depth, lf.stream) != 0) rc = -1;
            rc |= mi355_sync(lf.stream);
            free(hj); mi355_free(dj);
            if (rc) return -2;
        }
    }
    if (mi355_sync(lf.stream) != 0) return -2;
    for (int i = 0; i < 3; i++) rc |= picture_d2h(s->sao_frame->data[i], out[i], sz[i]);
    if (restore) for (int i = 0; i < 3; i++) rc |= picture_d2h(s->frame->data[i], plane[i], sz[i]);
    if (rc) return -2;
    lf.pictures++; __atomic_fetch_add(&g_lf_pictures, 1ul, __ATOMIC_RELAXED);
    return 0;
}

void __wrap_ff_hevc_hls_filters(HEVCContext *s, int x_ctb, int y_ctb, int ctb_size)
{
    if (!active(s)) { __real_ff_hevc_hls_filters(s, x_ctb, y_ctb, ctb_size); return; }
    /* nothing per CTB — the picture is filtered when its last CTB arrives.  One thing is noted here, because every CTB of every
     * slice passes: a slice with deblocking switched off never calls ff_hevc_deblocking_boundary_strengths (hevcdec.c, hls_transform_tree / hls_coding_unit: called under !disable_deblocking_filter_flag),
     * so its reference lists would not be compared with the picture's — a deblocked neighbour rating the edge between them on the
     * device could use the wrong list table.  The lists of such a slice are compared here. */
    if (lf.bs_ref != s->ref) bs_begin(s);
    if (s->sh.disable_deblocking_filter_flag) bs_slice_lists(s);
}

/* called by the slice decoder itself only for the picture's last CTB (hevcdec.c:2337-2339); the per-CTB calls come
 * through ff_hevc_hls_filters inside hevc_filter.c and never pass here */
void __wrap_ff_hevc_hls_filter(HEVCContext *s, int x, int y)
{
    if (!active(s)) { __real_ff_hevc_hls_filter(s, x, y); return; }
    if (getenv("MI355_HEVC_LF_TRACE")) fprintf(stderr, "lf: picture poc %d, last CTB at %d, %d (slice from CTB %d)\n", s->poc, x, y, s->sh.slice_ctb_addr_rs);
    const int frc = filter_picture(s);
    if (frc == -3) { fail("the device reconstruction of a picture failed: that picture is lost"); return; }
    if (frc != 0) {
        /* nothing of this picture has been filtered yet: the reference's own loop over all CTBs does it */
        fail("the device pass failed");
        const int ctb = 1 << s->ps.sps->log2_ctb_size;
        for (int yy = 0; yy < s->ps.sps->height; yy += ctb)
            for (int xx = 0; xx < s->ps.sps->width; xx += ctb) __real_ff_hevc_hls_filter(s, xx, yy);
    }
}

unsigned long mi355_hevc_lf_bridge_pictures(void) { return __atomic_load_n(&g_lf_pictures, __ATOMIC_RELAXED); }      /* of every decoder of the process */
unsigned long mi355_hevc_lf_bridge_bs_pictures(void) { return lf.bs_pictures; }
