/*
 * mi355_hevc_bridge.c — the HEVC Tier-2 bridge: PRODUCT glue that lives beside the reference's HEVC decoder and turns it into
 * host = parsing and entropy decoding only, MI355X = everything the per-block DSP did.  With contrib/libav/mi355_hevc_lf_bridge.c
 * (the in-loop filters of a whole picture on the device) it keeps a picture in HBM from its first predicted sample to the
 * sample-adaptive-offset output the next pictures predict from.
 *
 * The decoder's reconstruction is static code (hls_prediction_unit hevcdec.c:1695, luma_mc / chroma_mc :1528 / :1582,
 * hls_transform_unit / hls_residual_coding :1263 / :902, intra_prediction through hls_transform_unit :1270-1285, hls_pcm_sample
 * :1461) that reaches the samples ONLY through three pointer tables: HEVCDSPContext (hevcdsp.h:41-114), HEVCPredContext.intra_pred
 * (hevcdec.h:399-409) and VideoDSPContext.emulated_edge_mc (videodsp.h:52).  Linked with
 *     -Wl,--wrap=ff_hevc_dsp_init,--wrap=ff_hevc_pred_init,--wrap=ff_videodsp_init,--wrap=ff_hevc_frame_rps
 * (plus the filter bridge's wraps) this file fills those entries with functions that RECORD what the decoder asked for instead
 * of doing it:
 *     emulated_edge_mc                      -> a window job (mi355_edge_emu_batch_dev) whose output a prediction job will name
 *     put_hevc_qpel / put_hevc_epel         -> remembered under the intermediate buffer they were to fill
 *     put_unweighted_pred[_avg] / weighted_pred[_avg] (+ _chroma)
 *                                           -> one fused prediction job (mi355_hevc_mcpred_batch_dev) from the remembered call(s)
 *     idct / idct_dc / transform_4x4_luma / dequant, then add_residual
 *                                           -> one transform-unit job (mi355_hevc_residual_batch_dev) with a copy of the coefficients
 *     put_pcm                               -> a transform-unit job of kind PCM (the samples, read from the bitstream on the host)
 *     intra_pred[]                          -> an intra block (mi355_hevc_intra_pred_blocks_dev) with lc->na and the mode as they are now
 * Every job gets a dependency LEVEL from a map of which level last wrote each 4x4 (chroma 2x2) cell: a prediction block goes one
 * level above whatever wrote its area, a residual above its prediction, an intra block above the neighbours it reads.  When the
 * slice decoder reports the picture's last CTB, the filter bridge calls mi355_hevc_recon_finish(): the jobs go out level by level
 * (inter pictures: two or three levels; intra pictures: as deep as their prediction chains), then boundary strengths, deblocking
 * and SAO run on the same device picture, and the finished picture is copied to the host frame the decoder outputs.  Reference
 * samples never cross PCIe: the decoded picture buffer lives in HBM, one device surface per host frame buffer (found again by
 * the plane a source pointer falls into); a reference this bridge did not decode (a frame the decoder made up for a missing
 * reference) is uploaded once.
 *
 * Many decoders in one process (one per thread): their pictures share launches.  A picture's jobs are sorted by level on its own thread;
 * the launches are issued by a thread that finds a free SLOT (a stream with its own merged arrays; four by default,
 * MI355_HEVC_BRIDGE_SETS_IN_FLIGHT), for EVERY picture that is waiting when the slot's last set has finished — level l of all of them is ONE
 * launch (mi355_hevc_recon_level_dev: the level's prediction blocks, transform units and intra blocks side by side; the job arrays of the set
 * are merged level-major, jobs name their samples by device pointer and an intra block names its picture's descriptor by index).  Sets of
 * different slots run side by side on the device; a decoder's own copies and filter passes run on ITS stream behind the set's event.  A lone
 * decoder is a set of one.  Measured (1080p P / B stream, 16 decoders, pictures/s): 132 with every picture's own launches on the default
 * stream (the form before: MI355_HEVC_BRIDGE_SOLO=1 MI355_HEVC_BRIDGE_DEFAULT_STREAM=1) -> 204; the C decoder 464 — a picture's chain of ~480
 * dependent levels (~18 ms on the device whatever rides along) stays the decoder's cycle, DESIGN.md section 9 item 5.
 *
 * Scope: what the filter bridge takes (4:2:0, no frame threads; tiles, wavefronts and dependent slice segments included), 8 / 9 / 10 bit, one decoder per thread.  A picture
 * outside it is reconstructed by the reference's own functions (the entries forward to the tables the reference filled) and
 * its surface is uploaded when a later picture needs it.  MI355_HEVC_RECON_PLAIN=1 forwards everything.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/avcodec.h"
#include "libavcodec/hevcdec.h"
#include "libavcodec/videodsp.h"
#include "mi355_hevc_batch.h"
#include "mi355dsp.h"
#include "mi355_h264_frame.h"      /* the runtime entry points: mi355_malloc / mi355_memcpy_* / mi355_sync */

void __real_ff_hevc_dsp_init(HEVCDSPContext *c, int bit_depth);
void __real_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth);
void __real_ff_videodsp_init(VideoDSPContext *c, int bpc);
int __real_ff_hevc_frame_rps(HEVCContext *s);

#define MAX_SURF 48
typedef struct Surface {
    const uint8_t *host[3];            /* the host frame's planes: the key */
    int linesize[3], rows[3];
    uint8_t *dev;                      /* one allocation, planes back to back */
    size_t off[3], bytes;
    int valid, poc, seq;               /* the device copy holds the picture with this POC of this sequence (HEVCFrame.sequence) */
    unsigned long used;                /* picture counter at last use (recycling) */
} Surface;
typedef struct Loc { int surf; size_t off; } Loc;          /* surf >= 0: byte `off` of that surface's allocation; -2: of the window scratch */

typedef struct EmuRec { Loc src; size_t dst_off; int src_stride, bw, bh, sx, sy, w, h; } EmuRec;
typedef struct McRec {
    Loc src[2], dst;
    int sstride[2], dstride, w, h, chroma, kind, mx[2], my[2], denom, wt[2], of[2], level;
    Loc srcb[2], dstb;                 /* chroma == 2: the Cr block of the same prediction unit (one job for both planes) */
    int k, x, y;                       /* plane and position of the block (pairing the Cr block with its Cb block) */
} McRec;
typedef struct TuRec { Loc dst; size_t coef_off; int dstride, log2, col_limit, kind, level, fused; } TuRec;     /* fused: runs in its block's prediction launch */
typedef struct IntraRec { mi355_hevc_intra_block b; int level, k, x, y, size, tu; } IntraRec;              /* plane position of the block; tu: its transform unit (R.tu index) or -1 */
typedef struct Pending { const int16_t *tmp; Loc src; int sstride, w, h, mx, my, chroma, live; } Pending;

static __thread struct Recon {
    int init, plain, failed, irap_on_host, split_intra;
    HEVCContext *s;
    int on;                             /* the open picture is reconstructed on the device */
    int bd, px;
    HEVCDSPContext orig_dsp;
    HEVCPredContext orig_pred;
    VideoDSPContext orig_vdsp;
    Surface surf[MAX_SURF];
    int cur;                            /* surface of s->frame */
    unsigned long pictures, on_device, uploads, launches, levels_total;
    /* what the open picture recorded */
    EmuRec *emu; int nemu, cemu; size_t emu_bytes;
    McRec *mc; int nmc, cmc;
    int last_loc;                      /* surface the last locate() found */
    TuRec *tu; int ntu, ctu;
    int16_t *coef; size_t ncoef, ccoef;
    IntraRec *intra; int nintra, cintra;
    uint16_t *lvl[3]; int lw[3], lh[3], lshift[3]; size_t lcells[3];
    int max_level;
    Pending pend[8];
    struct { const uint8_t *buf; size_t dst_off; int stride, rows; } last_emu;
    int transform_kind, transform_col_limit; const int16_t *transform_coeffs;
    /* device staging */
    uint8_t *d_stage; size_t d_stage_bytes; uint8_t *h_stage; size_t h_stage_bytes;
    uint8_t *d_emu; size_t d_emu_bytes;
    uint8_t *d_mvf, *d_zs; size_t d_mvf_bytes, d_zs_bytes;
    void *done_event;                  /* of the launch set that held the last picture */
} R;

/* counters of the whole process (decoders come and go with their threads) */
static unsigned long g_pictures, g_on_device, g_uploads, g_launches, g_levels, g_batches, g_batched_pictures;
#define COUNT(c, n) __atomic_fetch_add(&(c), (unsigned long)(n), __ATOMIC_RELAXED)

static void recon_fail(const char *what)
{
    if (!R.failed) fprintf(stderr, "mi355 hevc bridge: %s; later pictures are reconstructed by the reference's functions\n", what);
    R.failed = 1;
}
static int grow(void **p, int *cap, int need, size_t elem)
{
    if (need <= *cap) return 0;
    int n = *cap ? *cap : 256;
    while (n < need) n *= 2;
    void *q = realloc(*p, (size_t)n * elem);
    if (!q) return -1;
    *p = q; *cap = n;
    return 0;
}
static int dev_ensure(uint8_t **p, size_t *have, size_t want)
{
    if (*p && *have >= want) return 0;
    if (*p) mi355_free(*p);
    *p = mi355_malloc(want + 64);
    *have = *p ? want : 0;
    return *p ? 0 : -1;
}

/* ---- surfaces ---------------------------------------------------------------------------------------------------------- */
static Surface *surface_of_frame(const HEVCContext *s, const AVFrame *f, int create)
{
    Surface *free_slot = NULL, *oldest = NULL;
    for (int i = 0; i < MAX_SURF; i++) {
        Surface *u = &R.surf[i];
        if (u->host[0] == f->data[0] && u->host[0]) {
            /* the same first plane with other planes or another geometry (the buffer pool hands planes out one by one; a new sequence, a new
             * decoder): not the frame this surface mirrors — start over */
            if (u->host[1] != f->data[1] || u->host[2] != f->data[2] || u->linesize[0] != f->linesize[0] || u->linesize[1] != f->linesize[1] ||
                u->rows[0] != s->ps.sps->height) { u->valid = 0; u->host[0] = NULL; free_slot = free_slot ? free_slot : u; continue; }
            return u;
        }
        if (!u->host[0]) { if (!free_slot) free_slot = u; }
        else if (!oldest || u->used < oldest->used) oldest = u;
    }
    if (!create) return NULL;
    Surface *u = free_slot ? free_slot : oldest;
    if (!u) return NULL;
    const int rows[3] = { s->ps.sps->height, s->ps.sps->height >> s->ps.sps->vshift[1], s->ps.sps->height >> s->ps.sps->vshift[2] };
    size_t off = 0;
    for (int i = 0; i < 3; i++) {
        u->host[i] = f->data[i]; u->linesize[i] = f->linesize[i]; u->rows[i] = rows[i];
        u->off[i] = off;
        off += ((size_t)f->linesize[i] * rows[i] + 255) & ~(size_t)255;
    }
    if (!u->dev || u->bytes < off) {
        if (u->dev) mi355_free(u->dev);
        u->dev = mi355_malloc(off + 256);
        u->bytes = u->dev ? off : 0;
        if (!u->dev) { u->host[0] = NULL; return NULL; }
    }
    u->valid = 0;
    u->used = R.pictures;
    return u;
}
/* the surface and byte offset of a pointer into some frame's plane (a frame of the decoded picture buffer this bridge has not
 * seen yet gets its surface now) */
static int locate(const HEVCContext *s, const uint8_t *p, Loc *out)
{
    /* the surface of the previous call first: a prediction unit's calls name the same reference picture (a dozen thousand calls per picture) */
    {
        const Surface *u = &R.surf[R.last_loc];
        if (u->host[0])
            for (int k = 0; k < 3; k++)
                if (p >= u->host[k] && p < u->host[k] + (size_t)u->linesize[k] * u->rows[k]) {
                    out->surf = R.last_loc; out->off = u->off[k] + (size_t)(p - u->host[k]);
                    R.surf[R.last_loc].used = R.pictures;
                    return 0;
                }
    }
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < MAX_SURF; i++) {
            const Surface *u = &R.surf[i];
            if (!u->host[0]) continue;
            for (int k = 0; k < 3; k++)
                if (p >= u->host[k] && p < u->host[k] + (size_t)u->linesize[k] * u->rows[k]) {
                    out->surf = i; out->off = u->off[k] + (size_t)(p - u->host[k]);
                    R.surf[i].used = R.pictures;
                    R.last_loc = i;
                    return 0;
                }
        }
        if (pass) break;
        /* not a frame we know: one of the decoder's frames we have not met (made up for a missing reference, or decoded before the bridge took over) */
        int found = 0;
        for (int i = 0; i < FF_ARRAY_ELEMS(s->DPB) && !found; i++) {
            const AVFrame *f = s->DPB[i].frame;
            if (!f || !f->data[0]) continue;
            for (int k = 0; k < 3 && !found; k++) {
                const int rows = k ? s->ps.sps->height >> s->ps.sps->vshift[k] : s->ps.sps->height;
                if (p >= f->data[k] && p < f->data[k] + (size_t)f->linesize[k] * rows) found = surface_of_frame(s, f, 1) != NULL;
            }
        }
        if (!found) return -1;
    }
    return -1;
}

/* ---- levels ------------------------------------------------------------------------------------------------------------ */
static int plane_of(const Surface *u, size_t off) { return off >= u->off[2] ? 2 : (off >= u->off[1] ? 1 : 0); }
/* cells of plane k covered by the rectangle (x, y, w, h) in samples of that plane, clipped to the plane */
static int level_max(int k, int x, int y, int w, int h)
{
    const int sh = R.lshift[k];
    int x0 = x >> sh, y0 = y >> sh, x1 = (x + w - 1) >> sh, y1 = (y + h - 1) >> sh, m = 0;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 >= R.lw[k]) x1 = R.lw[k] - 1;
    if (y1 >= R.lh[k]) y1 = R.lh[k] - 1;
    for (int yy = y0; yy <= y1; yy++) {
        const uint16_t *row = R.lvl[k] + (size_t)yy * R.lw[k];
        for (int xx = x0; xx <= x1; xx++) if (row[xx] > m) m = row[xx];
    }
    return m;
}
static void level_set(int k, int x, int y, int w, int h, int level)
{
    const int sh = R.lshift[k];
    int x0 = x >> sh, y0 = y >> sh, x1 = (x + w - 1) >> sh, y1 = (y + h - 1) >> sh;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 >= R.lw[k]) x1 = R.lw[k] - 1;
    if (y1 >= R.lh[k]) y1 = R.lh[k] - 1;
    for (int yy = y0; yy <= y1; yy++) {
        uint16_t *row = R.lvl[k] + (size_t)yy * R.lw[k];
        for (int xx = x0; xx <= x1; xx++) row[xx] = (uint16_t)level;
    }
    if (level > R.max_level) R.max_level = level;
}
/* plane and position (samples) of a destination pointer inside the open picture's frame */
static int dst_position(const uint8_t *dst, int *k, int *x, int *y, Loc *loc)
{
    const Surface *u = &R.surf[R.cur];
    for (int i = 0; i < 3; i++)
        if (dst >= u->host[i] && dst < u->host[i] + (size_t)u->linesize[i] * u->rows[i]) {
            const size_t o = (size_t)(dst - u->host[i]);
            *k = i; *y = (int)(o / (size_t)u->linesize[i]); *x = (int)(o % (size_t)u->linesize[i]) / R.px;
            loc->surf = R.cur; loc->off = u->off[i] + o;
            return 0;
        }
    return -1;
}

/* ---- the recording entries --------------------------------------------------------------------------------------------- */
static const uint8_t k_w_luma[8] = { 4, 8, 12, 16, 24, 32, 48, 64 }, k_w_chroma[8] = { 2, 4, 6, 8, 12, 16, 24, 32 };

static void rec_emu(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize, int bw, int bh, int sx, int sy, int w, int h)
{
    if (!R.on) { R.orig_vdsp.emulated_edge_mc(buf, src, buf_linesize, src_linesize, bw, bh, sx, sy, w, h); return; }
    if (grow((void **)&R.emu, &R.cemu, R.nemu + 1, sizeof(*R.emu))) { recon_fail("out of memory"); return; }
    EmuRec *e = &R.emu[R.nemu];
    /* the plane's first sample lies sy rows and sx samples before the window's */
    const uint8_t *origin = src - (ptrdiff_t)sy * src_linesize - (ptrdiff_t)sx * R.px;
    Loc o;
    if (locate(R.s, origin, &o)) { recon_fail("a reference picture the bridge cannot place"); return; }
    /* kept relative to the plane's first sample: the window itself may start outside the plane */
    e->src = o; e->src_stride = (int)src_linesize; e->bw = bw; e->bh = bh; e->sx = sx; e->sy = sy; e->w = w; e->h = h;
    e->dst_off = R.emu_bytes;
    R.emu_bytes += ((size_t)buf_linesize * bh + 63) & ~(size_t)63;
    R.last_emu.buf = buf; R.last_emu.dst_off = e->dst_off; R.last_emu.stride = (int)buf_linesize; R.last_emu.rows = bh;
    R.nemu++;
}

static void rec_mc(int16_t *tmp, uint8_t *src, ptrdiff_t ss, int w, int h, int mx, int my, int chroma, void (*orig)(int16_t *, ptrdiff_t, uint8_t *, ptrdiff_t, int, int, int, int16_t *),
                   ptrdiff_t ds, int16_t *mcbuffer)
{
    if (!R.on) { orig(tmp, ds, src, ss, h, mx, my, mcbuffer); return; }
    Pending *p = NULL;
    for (int i = 0; i < 8; i++) if (R.pend[i].live && R.pend[i].tmp == tmp) p = &R.pend[i];
    for (int i = 0; i < 8 && !p; i++) if (!R.pend[i].live) p = &R.pend[i];
    if (!p) p = &R.pend[0];
    p->tmp = tmp; p->sstride = (int)ss; p->w = w; p->h = h; p->mx = mx; p->my = my; p->chroma = chroma; p->live = 1;
    if (R.last_emu.buf && src >= R.last_emu.buf && src < R.last_emu.buf + (size_t)R.last_emu.stride * R.last_emu.rows) {
        p->src.surf = -2; p->src.off = R.last_emu.dst_off + (size_t)(src - R.last_emu.buf);
    } else if (locate(R.s, src, &p->src)) { recon_fail("a reference picture the bridge cannot place"); p->live = 0; }
}
#define MC_FN(name, tab, chroma, i) \
    static void name##_##i(int16_t *dst, ptrdiff_t ds, uint8_t *src, ptrdiff_t ss, int height, int mx, int my, int16_t *mcbuffer) \
    { rec_mc(dst, src, ss, tab[i], height, mx, my, chroma, R.orig_dsp.put_hevc_##name[!!my][!!mx][i], ds, mcbuffer); }
#define MC_FNS(name, tab, chroma) MC_FN(name, tab, chroma, 0) MC_FN(name, tab, chroma, 1) MC_FN(name, tab, chroma, 2) MC_FN(name, tab, chroma, 3) \
                                   MC_FN(name, tab, chroma, 4) MC_FN(name, tab, chroma, 5) MC_FN(name, tab, chroma, 6) MC_FN(name, tab, chroma, 7)
MC_FNS(qpel, k_w_luma, 0)
MC_FNS(epel, k_w_chroma, 1)

static Pending *pending_of(const int16_t *tmp)
{
    for (int i = 0; i < 8; i++) if (R.pend[i].live && R.pend[i].tmp == tmp) return &R.pend[i];
    return NULL;
}
/* one prediction block: `kind` MI355_HEVC_PRED_*, sources = the remembered MC calls that filled src1 (and src2) */
static void rec_pred(uint8_t *dst, ptrdiff_t dstride, const int16_t *src1, const int16_t *src2, int w, int h, int kind, int denom, int w0, int w1, int o0, int o1)
{
    Pending *a = pending_of(src1), *b = src2 ? pending_of(src2) : NULL;
    int k, x, y;
    Loc dl;
    if (!a || (src2 && !b) || dst_position(dst, &k, &x, &y, &dl)) { recon_fail("a prediction call the bridge cannot match with its interpolation"); return; }
    if (grow((void **)&R.mc, &R.cmc, R.nmc + 1, sizeof(*R.mc))) { recon_fail("out of memory"); return; }
    McRec *m = &R.mc[R.nmc++];
    memset(m, 0, sizeof(*m));
    m->src[0] = a->src; m->sstride[0] = a->sstride; m->mx[0] = a->mx; m->my[0] = a->my;
    if (b) { m->src[1] = b->src; m->sstride[1] = b->sstride; m->mx[1] = b->mx; m->my[1] = b->my; }
    m->dst = dl; m->dstride = (int)dstride; m->w = w; m->h = h; m->chroma = a->chroma; m->kind = kind;
    m->denom = denom; m->wt[0] = w0; m->wt[1] = w1; m->of[0] = o0; m->of[1] = o1;
    m->k = k; m->x = x; m->y = y;
    m->level = level_max(k, x, y, w, h) + 1;
    level_set(k, x, y, w, h, m->level);
    a->live = 0;                       /* an intermediate is consumed once: its slot is free for the next interpolation */
    if (b) b->live = 0;
    /* chroma_mc runs Cb, then Cr with the same vector (hevcdec.c:1582-1640, called twice per prediction unit): with unweighted prediction
     * the two blocks differ in their planes only and become ONE job (mi355_hevc_batch.h, chroma == 2: windows fetched together, every pass
     * over both planes) on the later of their two levels */
    if (k == 2 && R.nmc >= 2 && (kind == MI355_HEVC_PRED_PUT || kind == MI355_HEVC_PRED_AVG)) {
        McRec *c = m - 1;
        if (c->chroma == 1 && c->k == 1 && c->x == x && c->y == y && c->w == w && c->h == h && c->kind == kind && c->dstride == m->dstride &&
            c->sstride[0] == m->sstride[0] && c->mx[0] == m->mx[0] && c->my[0] == m->my[0] &&
            (!b || (c->sstride[1] == m->sstride[1] && c->mx[1] == m->mx[1] && c->my[1] == m->my[1]))) {
            c->srcb[0] = m->src[0]; c->srcb[1] = m->src[1]; c->dstb = m->dst;
            c->chroma = 2;
            if (m->level > c->level) { c->level = m->level; level_set(1, x, y, w, h, c->level); }
            else if (c->level > m->level) level_set(2, x, y, w, h, c->level);
            R.nmc--;
        }
    }
}
#define PRED_FNS_I(i) \
    static void up_##i(uint8_t *d, ptrdiff_t ds, int16_t *s1, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.put_unweighted_pred[i](d, ds, s1, ss, h); return; } rec_pred(d, ds, s1, NULL, k_w_luma[i], h, MI355_HEVC_PRED_PUT, 0, 0, 0, 0, 0); } \
    static void upc_##i(uint8_t *d, ptrdiff_t ds, int16_t *s1, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.put_unweighted_pred_chroma[i](d, ds, s1, ss, h); return; } rec_pred(d, ds, s1, NULL, k_w_chroma[i], h, MI355_HEVC_PRED_PUT, 0, 0, 0, 0, 0); } \
    static void upa_##i(uint8_t *d, ptrdiff_t ds, int16_t *s1, int16_t *s2, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.put_unweighted_pred_avg[i](d, ds, s1, s2, ss, h); return; } rec_pred(d, ds, s1, s2, k_w_luma[i], h, MI355_HEVC_PRED_AVG, 0, 0, 0, 0, 0); } \
    static void upac_##i(uint8_t *d, ptrdiff_t ds, int16_t *s1, int16_t *s2, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.put_unweighted_pred_avg_chroma[i](d, ds, s1, s2, ss, h); return; } rec_pred(d, ds, s1, s2, k_w_chroma[i], h, MI355_HEVC_PRED_AVG, 0, 0, 0, 0, 0); } \
    static void wp_##i(uint8_t dn, int16_t w, int16_t o, uint8_t *d, ptrdiff_t ds, int16_t *s1, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.weighted_pred[i](dn, w, o, d, ds, s1, ss, h); return; } rec_pred(d, ds, s1, NULL, k_w_luma[i], h, MI355_HEVC_PRED_W, dn, w, 0, o, 0); } \
    static void wpc_##i(uint8_t dn, int16_t w, int16_t o, uint8_t *d, ptrdiff_t ds, int16_t *s1, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.weighted_pred_chroma[i](dn, w, o, d, ds, s1, ss, h); return; } rec_pred(d, ds, s1, NULL, k_w_chroma[i], h, MI355_HEVC_PRED_W, dn, w, 0, o, 0); } \
    static void wpa_##i(uint8_t dn, int16_t w0, int16_t w1, int16_t o0, int16_t o1, uint8_t *d, ptrdiff_t ds, int16_t *s1, int16_t *s2, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.weighted_pred_avg[i](dn, w0, w1, o0, o1, d, ds, s1, s2, ss, h); return; } rec_pred(d, ds, s1, s2, k_w_luma[i], h, MI355_HEVC_PRED_W_AVG, dn, w0, w1, o0, o1); } \
    static void wpac_##i(uint8_t dn, int16_t w0, int16_t w1, int16_t o0, int16_t o1, uint8_t *d, ptrdiff_t ds, int16_t *s1, int16_t *s2, ptrdiff_t ss, int h) { if (!R.on) { R.orig_dsp.weighted_pred_avg_chroma[i](dn, w0, w1, o0, o1, d, ds, s1, s2, ss, h); return; } rec_pred(d, ds, s1, s2, k_w_chroma[i], h, MI355_HEVC_PRED_W_AVG, dn, w0, w1, o0, o1); }
PRED_FNS_I(0) PRED_FNS_I(1) PRED_FNS_I(2) PRED_FNS_I(3) PRED_FNS_I(4) PRED_FNS_I(5) PRED_FNS_I(6) PRED_FNS_I(7)

/* the transform that precedes add_residual (hls_residual_coding, hevcdec.c:1236-1260): remembered until the add arrives */
static void note_transform(const int16_t *coeffs, int kind, int col_limit) { R.transform_coeffs = coeffs; R.transform_kind = kind; R.transform_col_limit = col_limit; }
static void rec_dequant(int16_t *c) { if (!R.on) { R.orig_dsp.dequant(c); return; } note_transform(c, MI355_HEVC_TU_SKIP, 0); }
static void rec_dst4(int16_t *c) { if (!R.on) { R.orig_dsp.transform_4x4_luma(c); return; } note_transform(c, MI355_HEVC_TU_DST4, 0); }
#define IDCT_FNS(i) \
    static void idct_##i(int16_t *c, int col_limit) { if (!R.on) { R.orig_dsp.idct[i](c, col_limit); return; } note_transform(c, MI355_HEVC_TU_IDCT, col_limit); } \
    static void idct_dc_##i(int16_t *c) { if (!R.on) { R.orig_dsp.idct_dc[i](c); return; } note_transform(c, MI355_HEVC_TU_IDCT_DC, 0); }
IDCT_FNS(0) IDCT_FNS(1) IDCT_FNS(2) IDCT_FNS(3)

static void rec_tu(uint8_t *dst, const int16_t *samples, ptrdiff_t stride, int log2, int kind, int col_limit)
{
    int k, x, y;
    Loc dl;
    const int size = 1 << log2, n = size * size;
    if (dst_position(dst, &k, &x, &y, &dl)) { recon_fail("a residual outside the picture"); return; }
    if (grow((void **)&R.tu, &R.ctu, R.ntu + 1, sizeof(*R.tu))) { recon_fail("out of memory"); return; }
    if (R.ncoef + (size_t)n > R.ccoef) {
        size_t c = R.ccoef ? R.ccoef : (1u << 16);
        while (c < R.ncoef + (size_t)n) c *= 2;
        int16_t *q = realloc(R.coef, c * sizeof(int16_t));
        if (!q) { recon_fail("out of memory"); return; }
        R.coef = q; R.ccoef = c;
    }
    TuRec *t = &R.tu[R.ntu++];
    t->dst = dl; t->dstride = (int)stride; t->log2 = log2; t->kind = kind; t->col_limit = col_limit;
    t->coef_off = R.ncoef;
    memcpy(R.coef + R.ncoef, samples, (size_t)n * sizeof(int16_t));
    R.ncoef += (size_t)n;
    t->fused = 0;
    const int here = level_max(k, x, y, size, size);
    /* the unit of the intra block predicted just before (hls_transform_unit, hevcdec.c:1002-1030 then :1238-1260: prediction, then the
     * residual of the same block): it joins the prediction's launch (mi355_hevc_intra_recon_blocks_dev) — one level instead of two */
    if (!R.split_intra && kind != MI355_HEVC_TU_PCM && R.nintra) {
        IntraRec *li = &R.intra[R.nintra - 1];
        if (li->tu < 0 && li->k == k && li->x == x && li->y == y && li->size == size && li->level == here) {
            li->tu = R.ntu - 1;
            t->fused = 1;
            t->level = li->level;
            return;
        }
    }
    t->level = here + 1;
    level_set(k, x, y, size, size, t->level);
}
#define ADD_FN(i) \
    static void add_res_##i(uint8_t *dst, int16_t *res, ptrdiff_t stride) \
    { \
        if (!R.on) { R.orig_dsp.add_residual[i](dst, res, stride); return; } \
        const int have = R.transform_coeffs == res; \
        rec_tu(dst, res, stride, i + 2, have ? R.transform_kind : MI355_HEVC_TU_BYPASS, have ? R.transform_col_limit : 0); \
        R.transform_coeffs = NULL; \
    }
ADD_FN(0) ADD_FN(1) ADD_FN(2) ADD_FN(3)

/* pcm_sample: the samples come from the bitstream — read by the reference's own function into a block of their own, then a job */
static void rec_pcm(uint8_t *dst, ptrdiff_t stride, int size, GetBitContext *gb, int pcm_bit_depth)
{
    if (!R.on) { R.orig_dsp.put_pcm(dst, stride, size, gb, pcm_bit_depth); return; }
    uint8_t blk[32 * 32 * 2];
    int16_t smp[32 * 32];
    int log2 = 2;
    while ((1 << log2) < size) log2++;
    if (size > 32 || (1 << log2) != size) { recon_fail("a PCM block size the bridge does not take"); return; }
    R.orig_dsp.put_pcm(blk, (ptrdiff_t)size * R.px, size, gb, pcm_bit_depth);
    for (int i = 0; i < size * size; i++) smp[i] = R.px == 2 ? (int16_t)((const uint16_t *)blk)[i] : (int16_t)blk[i];
    rec_tu(dst, smp, stride, log2, MI355_HEVC_TU_PCM, 0);
}

static void rec_intra(HEVCContext *s, int x0, int y0, int c_idx, int log2)
{
    if (!R.on) { R.orig_pred.intra_pred[log2 - 2](s, x0, y0, c_idx); return; }
    const HEVCLocalContext *lc = &s->HEVClc;
    if (grow((void **)&R.intra, &R.cintra, R.nintra + 1, sizeof(*R.intra))) { recon_fail("out of memory"); return; }
    IntraRec *r = &R.intra[R.nintra++];
    memset(r, 0, sizeof(*r));
    r->b.pic = 0;
    r->b.x0 = (uint16_t)x0; r->b.y0 = (uint16_t)y0; r->b.log2_size = (uint8_t)log2; r->b.c_idx = (uint8_t)c_idx;
    r->b.mode = (uint8_t)(c_idx ? lc->pu.intra_pred_mode_c : lc->tu.cur_intra_pred_mode);
    r->b.cand = (uint8_t)((lc->na.cand_bottom_left ? MI355_HEVC_CAND_BOTTOM_LEFT : 0) | (lc->na.cand_left ? MI355_HEVC_CAND_LEFT : 0) |
                          (lc->na.cand_up_left ? MI355_HEVC_CAND_UP_LEFT : 0) | (lc->na.cand_up ? MI355_HEVC_CAND_UP : 0) |
                          (lc->na.cand_up_right ? MI355_HEVC_CAND_UP_RIGHT : 0));
    /* the block in its plane, the neighbours it may read: one row above (from the corner, 2 size + 1 samples), one column left */
    const int size = 1 << log2, x = x0 >> (c_idx ? s->ps.sps->hshift[c_idx] : 0), y = y0 >> (c_idx ? s->ps.sps->vshift[c_idx] : 0);
    int m = level_max(c_idx, x, y, size, size);
    const int a = level_max(c_idx, x - 1, y - 1, 2 * size + 1, 1), b = level_max(c_idx, x - 1, y, 1, 2 * size);
    if (a > m) m = a;
    if (b > m) m = b;
    r->level = m + 1;
    r->k = c_idx; r->x = x; r->y = y; r->size = size; r->tu = -1;
    level_set(c_idx, x, y, size, size, r->level);
}
static void intra_2(HEVCContext *s, int x0, int y0, int c) { rec_intra(s, x0, y0, c, 2); }
static void intra_3(HEVCContext *s, int x0, int y0, int c) { rec_intra(s, x0, y0, c, 3); }
static void intra_4(HEVCContext *s, int x0, int y0, int c) { rec_intra(s, x0, y0, c, 4); }
static void intra_5(HEVCContext *s, int x0, int y0, int c) { rec_intra(s, x0, y0, c, 5); }

/* ---- table wraps ------------------------------------------------------------------------------------------------------- */
static void first_use(void)
{
    if (R.init) return;
    R.init = 1;
    R.plain = getenv("MI355_HEVC_RECON_PLAIN") != NULL;
    /* a scheduling policy, off by default: random-access pictures (every slice intra) stay with the reference's functions on the host —
     * intra prediction is a chain of blocks each waiting for its neighbours, nanoseconds apart on a CPU and one launch apart here (a
     * 1920x1080 picture of 4x4 blocks is ~3700 dependency levels) — and come to the device once, finished, when a later picture
     * predicts from them (upload_surface); the filter bridge still filters them on the device */
    R.irap_on_host = getenv("MI355_HEVC_BRIDGE_IRAP_ON_HOST") != NULL;
    R.split_intra = getenv("MI355_HEVC_BRIDGE_SPLIT_INTRA") != NULL;      /* an intra block's prediction and residual as two launches (the form before the fused kernel) */
}
void __wrap_ff_hevc_dsp_init(HEVCDSPContext *c, int bit_depth)
{
    __real_ff_hevc_dsp_init(c, bit_depth);
    first_use();
    R.orig_dsp = *c;
    R.bd = bit_depth; R.px = bit_depth > 8 ? 2 : 1;
    if (R.plain) return;
#define SET8(field, fn) do { c->field[0] = fn##_0; c->field[1] = fn##_1; c->field[2] = fn##_2; c->field[3] = fn##_3; c->field[4] = fn##_4; c->field[5] = fn##_5; c->field[6] = fn##_6; c->field[7] = fn##_7; } while (0)
    for (int v = 0; v < 2; v++)
        for (int h = 0; h < 2; h++) { SET8(put_hevc_qpel[v][h], qpel); SET8(put_hevc_epel[v][h], epel); }
    SET8(put_unweighted_pred, up); SET8(put_unweighted_pred_chroma, upc); SET8(put_unweighted_pred_avg, upa); SET8(put_unweighted_pred_avg_chroma, upac);
    SET8(weighted_pred, wp); SET8(weighted_pred_chroma, wpc); SET8(weighted_pred_avg, wpa); SET8(weighted_pred_avg_chroma, wpac);
#undef SET8
    c->idct[0] = idct_0; c->idct[1] = idct_1; c->idct[2] = idct_2; c->idct[3] = idct_3;
    c->idct_dc[0] = idct_dc_0; c->idct_dc[1] = idct_dc_1; c->idct_dc[2] = idct_dc_2; c->idct_dc[3] = idct_dc_3;
    c->add_residual[0] = add_res_0; c->add_residual[1] = add_res_1; c->add_residual[2] = add_res_2; c->add_residual[3] = add_res_3;
    c->dequant = rec_dequant; c->transform_4x4_luma = rec_dst4; c->put_pcm = rec_pcm;
}
void __wrap_ff_hevc_pred_init(HEVCPredContext *c, int bit_depth)
{
    __real_ff_hevc_pred_init(c, bit_depth);
    first_use();
    R.orig_pred = *c;
    if (R.plain) return;
    c->intra_pred[0] = intra_2; c->intra_pred[1] = intra_3; c->intra_pred[2] = intra_4; c->intra_pred[3] = intra_5;
}
void __wrap_ff_videodsp_init(VideoDSPContext *c, int bpc)
{
    __real_ff_videodsp_init(c, bpc);
    first_use();
    R.orig_vdsp = *c;
    if (R.plain) return;
    c->emulated_edge_mc = rec_emu;
}

/* the filter bridge's test of a picture it takes (mi355_hevc_lf_bridge.c): the two must agree picture by picture */
int mi355_hevc_lf_bridge_active(const HEVCContext *s) __attribute__((weak));

/* a picture starts: hevc_frame_start calls ff_hevc_frame_rps right after ff_hevc_set_new_ref (hevcdec.c:2452-2457) */
int __wrap_ff_hevc_frame_rps(HEVCContext *s)
{
    const int ret = __real_ff_hevc_frame_rps(s);
    first_use();
    if (R.s != s)                       /* another decoder on this thread: nothing on the device is its picture */
        for (int i = 0; i < MAX_SURF; i++) R.surf[i].valid = 0;
    R.s = s;
    /* a surface mirrors a frame the decoder HOLDS (a DPB entry, the picture being decoded, the SAO work frame): anything else is
     * memory that went back to the buffer pool — or to another decoder at the same address — and must not be matched by address
     * ranges (locate) or read (upload_surface) any more */
    for (int i = 0; i < MAX_SURF; i++) {
        Surface *u = &R.surf[i];
        if (!u->host[0]) continue;
        int held = (s->frame && s->frame->data[0] == u->host[0]) || (s->sao_frame && s->sao_frame->data[0] == u->host[0]) ||
                   (s->tmp_frame && s->tmp_frame->data[0] == u->host[0]);
        for (int k = 0; k < FF_ARRAY_ELEMS(s->DPB) && !held; k++)
            held = s->DPB[k].frame && s->DPB[k].frame->data[0] == u->host[0];
        if (!held) { u->valid = 0; u->host[0] = u->host[1] = u->host[2] = NULL; }
    }
    R.on = 0;
    R.pictures++; COUNT(g_pictures, 1);
    if (ret < 0 || R.plain || R.failed || !s->frame || !s->frame->data[0]) return ret;
    if (!mi355_hevc_lf_bridge_active || !mi355_hevc_lf_bridge_active(s)) return ret;
    if (R.irap_on_host && IS_IRAP(s)) return ret;
    const HEVCSPS *sps = s->ps.sps;
    Surface *u = surface_of_frame(s, s->frame, 1);
    if (!u) { recon_fail("no device memory for a picture"); return ret; }
    R.cur = (int)(u - R.surf);
    u->valid = 0;
    /* level maps: luma cells of 4x4, chroma cells of 2x2 samples */
    for (int k = 0; k < 3; k++) {
        R.lshift[k] = k ? 1 : 2;
        R.lw[k] = ((sps->width >> (k ? sps->hshift[k] : 0)) + (1 << R.lshift[k]) - 1) >> R.lshift[k];
        R.lh[k] = ((sps->height >> (k ? sps->vshift[k] : 0)) + (1 << R.lshift[k]) - 1) >> R.lshift[k];
        const size_t cells = (size_t)R.lw[k] * R.lh[k];
        if (R.lcells[k] < cells) {
            free(R.lvl[k]);
            R.lvl[k] = malloc(cells * sizeof(uint16_t));
            R.lcells[k] = R.lvl[k] ? cells : 0;
            if (!R.lvl[k]) { recon_fail("out of memory"); return ret; }
        }
        memset(R.lvl[k], 0, cells * sizeof(uint16_t));
    }
    R.nemu = R.nmc = R.ntu = R.nintra = 0;
    R.ncoef = 0; R.emu_bytes = 0; R.max_level = 0;
    memset(R.pend, 0, sizeof(R.pend));
    memset(&R.last_emu, 0, sizeof(R.last_emu));
    R.transform_coeffs = NULL;
    R.on = 1;
    return ret;
}

/* ---- the picture goes out ----------------------------------------------------------------------------------------------- */
static uint8_t *resolve(const Loc *l) { return l->surf == -2 ? R.d_emu + l->off : R.surf[l->surf].dev + l->off; }
static int upload_surface(const HEVCContext *s, Surface *u)
{
    /* the host frame holds the picture (reconstructed by the reference's functions, or made up by the decoder) */
    for (int k = 0; k < 3; k++)
        if (mi355_memcpy_h2d(u->dev + u->off[k], u->host[k], (size_t)u->linesize[k] * u->rows[k]) != 0) return -1;
    /* it mirrors that frame as the decoder holds it now: later lookups compare the picture's number and sequence */
    for (int i = 0; i < FF_ARRAY_ELEMS(s->DPB); i++) {
        const HEVCFrame *f = &s->DPB[i];
        if (f->frame && f->frame->data[0] == u->host[0]) { u->poc = f->poc; u->seq = f->sequence; }
    }
    u->valid = 1;
    R.uploads++; COUNT(g_uploads, 1);
    return 0;
}
/* is the device copy of surface `i` the picture the decoder's frame holds now? */
static int surface_current(const HEVCContext *s, const Surface *u)
{
    if (!u->valid) return 0;
    for (int i = 0; i < FF_ARRAY_ELEMS(s->DPB); i++) {
        const HEVCFrame *f = &s->DPB[i];
        if (f->frame && f->frame->data[0] == u->host[0]) return f->poc == u->poc && f->sequence == u->seq;
    }
    return 1;      /* the work frame of sequences with SAO (s->tmp_frame) is nobody's reference */
}

/* Called by the filter bridge when the slice decoder reports the picture's last CTB.  Returns 0 when this picture was not
 * recorded (the filter bridge then uploads the host frame as it always did), 1 when the unfiltered reconstruction now lies in
 * cur[] (device planes of s->frame) and the finished picture must go to fin[] (device planes of the frame later pictures
 * predict from: s->sao_frame with SAO, else the same), < 0 on failure. */
/* ---- many decoders, one launch chain -------------------------------------------------------------------------------------------------- */
/* what a thread hands over: its picture's jobs, sorted by level, in its own pinned staging (device pointers inside), and where each level starts */
typedef struct Sub {
    const mi355_edge_emu_job *je; const mi355_hevc_mcpred_job *jm; const mi355_hevc_tu_job *jt, *jf; const mi355_hevc_intra_block *ji;
    const mi355_hevc_intra_picture *desc;
    const int *smc, *stu, *sin;          /* [level] -> first job of that level, levels 1..L ([L + 1]: the end) */
    int L, nemu, bd, split_intra;
    int done, rc;
    unsigned long launches;              /* launches this picture took part in */
    void *done_event;                    /* recorded behind the set's last launch: what this picture's filter passes (on the decoder's own stream) wait for */
} Sub;
#define MAX_WAITING 256
#define MAX_BATCH 32
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static Sub *g_wait[MAX_WAITING];
static int g_nwait, g_solo;
/* merged job arrays: pinned host copy -> device copy on the stream (three of each in turn: the host copy of a batch is free again when the
 * event behind that batch's launches has passed) */
/* A set runs on one of a few SLOTS: a stream of its own (non-blocking: a decoder's copies and filter passes — on ITS stream — neither wait for other
 * decoders' sets nor hold them up; what orders a picture's filter passes behind its set is the set's event, mi355_hevc_recon_done_event) with its
 * own copy of the merged arrays.  Sets of different slots run side by side on the device (they hold pictures of different decoders: nothing connects
 * them) — a chain of hundreds of small dependent launches leaves most of the device idle; a slot takes its next set when its last one has finished,
 * so the pictures that arrive meanwhile ride one chain together. */
#define MAX_SLOTS 8
static struct Merged { uint8_t *h, *d; size_t bytes; void *ev, *stream; int used, busy; } g_slot[MAX_SLOTS];
static int g_nslots = -1, g_default_stream;

static int g_one_launch = -1;             /* MI355_HEVC_BRIDGE_ONE_LAUNCH=1: all levels of a set in ONE launch (mi355_hevc_recon_levels_dev) instead of a launch per level — measured slower for
                                           * sets with wide levels (P / B pictures: a thousand workgroups counting themselves into one word) and level with it for chains of small ones */
static int g_three_launches = -1;         /* MI355_HEVC_BRIDGE_THREE_LAUNCHES=1: a level's job kinds as separate launches (the form before mi355_hevc_recon_level_dev) */

static int launch_batch(struct Merged *m, Sub **b, int K)
{
    if (g_three_launches < 0) { const char *e = getenv("MI355_HEVC_BRIDGE_THREE_LAUNCHES"); g_three_launches = e && *e && *e != '0'; }
    if (g_one_launch < 0) { const char *e = getenv("MI355_HEVC_BRIDGE_ONE_LAUNCH"); g_one_launch = e && *e && *e != '0'; }
    if (!m->stream && !g_default_stream) m->stream = mi355_stream_create();
    void *const st = m->stream;
    int maxL = 0, nemu = 0, nmc = 0, ntu = 0, nin = 0;
    for (int k = 0; k < K; k++) {
        if (b[k]->L > maxL) maxL = b[k]->L;
        nemu += b[k]->nemu; nmc += b[k]->smc[b[k]->L + 1]; ntu += b[k]->stu[b[k]->L + 1]; nin += b[k]->sin[b[k]->L + 1];
    }
    size_t o = 0;
    const size_t o_desc = o; o += ((size_t)K * sizeof(mi355_hevc_intra_picture) + 63) & ~(size_t)63;
    const size_t o_emu = o;  o += ((size_t)nemu * sizeof(mi355_edge_emu_job) + 63) & ~(size_t)63;
    const size_t o_mc = o;   o += ((size_t)nmc * sizeof(mi355_hevc_mcpred_job) + 63) & ~(size_t)63;
    const size_t o_tu = o;   o += ((size_t)ntu * sizeof(mi355_hevc_tu_job) + 63) & ~(size_t)63;
    const size_t o_in = o;   o += ((size_t)nin * sizeof(mi355_hevc_intra_block) + 63) & ~(size_t)63;
    const size_t o_fu = o;   o += ((size_t)nin * sizeof(mi355_hevc_tu_job) + 63) & ~(size_t)63;
    const size_t o_lv = o;   o += ((size_t)(maxL + 1) * sizeof(mi355_hevc_level) + 63) & ~(size_t)63;
    if (m->used && mi355_event_sync(m->ev) != 0) return -1;       /* (the caller has waited already: the arrays are free) */
    m->used = 0;
    if (m->bytes < o) {
        if (m->h) mi355_host_free(m->h);
        if (m->d) { mi355_sync(st); mi355_free(m->d); }          /* launches of an earlier batch may still read it */
        m->bytes = o + o / 2;
        m->h = mi355_host_alloc(m->bytes);
        m->d = mi355_malloc(m->bytes);
        if (!m->h || !m->d) { m->bytes = 0; return -1; }
    }
    if (!m->ev && !(m->ev = mi355_event_create())) return -1;
    int *first = malloc((size_t)(maxL + 2) * 3 * sizeof(int));     /* merged [level] -> first job, per kind */
    if (!first) return -1;
    int *fmc = first, *ftu = first + (maxL + 2), *fin = first + 2 * (maxL + 2);
    mi355_hevc_intra_picture *md = (mi355_hevc_intra_picture *)(m->h + o_desc);
    mi355_edge_emu_job *me = (mi355_edge_emu_job *)(m->h + o_emu);
    mi355_hevc_mcpred_job *mm = (mi355_hevc_mcpred_job *)(m->h + o_mc);
    mi355_hevc_tu_job *mt = (mi355_hevc_tu_job *)(m->h + o_tu), *mf = (mi355_hevc_tu_job *)(m->h + o_fu);
    mi355_hevc_intra_block *mi = (mi355_hevc_intra_block *)(m->h + o_in);
    int ae = 0, am = 0, at = 0, ai = 0;
    for (int k = 0; k < K; k++) {
        md[k] = *b[k]->desc;
        if (b[k]->nemu) memcpy(me + ae, b[k]->je, (size_t)b[k]->nemu * sizeof(*me));
        ae += b[k]->nemu;
    }
    for (int l = 1; l <= maxL; l++) {
        fmc[l] = am; ftu[l] = at; fin[l] = ai;
        for (int k = 0; k < K; k++) {
            const Sub *u = b[k];
            if (l > u->L) continue;
            const int nm = u->smc[l + 1] - u->smc[l], nt = u->stu[l + 1] - u->stu[l], ni = u->sin[l + 1] - u->sin[l];
            if (nm) memcpy(mm + am, u->jm + u->smc[l], (size_t)nm * sizeof(*mm));
            if (nt) memcpy(mt + at, u->jt + u->stu[l], (size_t)nt * sizeof(*mt));
            if (ni) {
                memcpy(mi + ai, u->ji + u->sin[l], (size_t)ni * sizeof(*mi));
                memcpy(mf + ai, u->jf + u->sin[l], (size_t)ni * sizeof(*mf));
                for (int i = 0; i < ni; i++) mi[ai + i].pic = k;
            }
            am += nm; at += nt; ai += ni;
        }
    }
    fmc[maxL + 1] = am; ftu[maxL + 1] = at; fin[maxL + 1] = ai;
    const int bd = b[0]->bd;
    /* every level of the set in ONE launch: the level table beside the job arrays */
    const int one_launch = g_one_launch && !b[0]->split_intra && !g_three_launches && maxL > 1;
    int nlv = 0;
    unsigned total_wg = 0;
    if (one_launch) {
        mi355_hevc_level *lv = (mi355_hevc_level *)(m->h + o_lv);
        for (int l = 1; l <= maxL; l++) {
            const int nm = fmc[l + 1] - fmc[l], nt = ftu[l + 1] - ftu[l], ni = fin[l + 1] - fin[l];
            if (!(nm + nt + ni)) continue;
            lv[nlv++] = (mi355_hevc_level){ total_wg, (uint32_t)fmc[l], (uint32_t)nm, (uint32_t)ftu[l], (uint32_t)nt, (uint32_t)fin[l], (uint32_t)ni, 0 };
            total_wg += (unsigned)(nm + (nt + 1) / 2 + ni);
        }
    }
    int rc = mi355_memcpy_h2d_async(m->d, m->h, o, st);
    unsigned long launches = 0;
    if (!rc && nemu && mi355_edge_emu_batch_dev((const mi355_edge_emu_job *)(m->d + o_emu), nemu, bd, st) != 0) rc = -1;
    if (one_launch && nlv && !rc) {
        if (mi355_hevc_recon_levels_dev((const mi355_hevc_level *)(m->d + o_lv), nlv, (int)total_wg, (const mi355_hevc_mcpred_job *)(m->d + o_mc),
                                        (const mi355_hevc_tu_job *)(m->d + o_tu), (const mi355_hevc_intra_picture *)(m->d + o_desc),
                                        (const mi355_hevc_intra_block *)(m->d + o_in), (const mi355_hevc_tu_job *)(m->d + o_fu), bd, st) != 0) rc = -1;
        launches = 1;
    }
    for (int l = 1; l <= maxL && !rc && !one_launch; l++) {
        const int nm = fmc[l + 1] - fmc[l], nt = ftu[l + 1] - ftu[l], ni = fin[l + 1] - fin[l];
        if (!b[0]->split_intra && !g_three_launches) {
            /* the level's three job kinds in one launch */
            if (nm + nt + ni && mi355_hevc_recon_level_dev((const mi355_hevc_mcpred_job *)(m->d + o_mc) + fmc[l], nm, (const mi355_hevc_tu_job *)(m->d + o_tu) + ftu[l], nt,
                                                          (const mi355_hevc_intra_picture *)(m->d + o_desc), (const mi355_hevc_intra_block *)(m->d + o_in) + fin[l],
                                                          (const mi355_hevc_tu_job *)(m->d + o_fu) + fin[l], ni, bd, st) != 0) rc = -1;
            launches += (unsigned long)(nm + nt + ni != 0);
            continue;
        }
        if (nm && mi355_hevc_mcpred_batch_dev((const mi355_hevc_mcpred_job *)(m->d + o_mc) + fmc[l], nm, bd, st) != 0) rc = -1;
        if (nt && mi355_hevc_residual_batch_dev((const mi355_hevc_tu_job *)(m->d + o_tu) + ftu[l], nt, bd, st) != 0) rc = -1;
        if (ni && !b[0]->split_intra &&
            mi355_hevc_intra_recon_blocks_dev((const mi355_hevc_intra_picture *)(m->d + o_desc), (const mi355_hevc_intra_block *)(m->d + o_in) + fin[l],
                                              (const mi355_hevc_tu_job *)(m->d + o_fu) + fin[l], ni, bd, st) != 0) rc = -1;
        if (ni && b[0]->split_intra &&
            mi355_hevc_intra_pred_blocks_dev((const mi355_hevc_intra_picture *)(m->d + o_desc), (const mi355_hevc_intra_block *)(m->d + o_in) + fin[l], ni, bd, st) != 0) rc = -1;
        launches += (unsigned long)((nm != 0) + (nt != 0) + (ni != 0));
    }
    free(first);
    if (mi355_event_record(m->ev, st) == 0) m->used = 1;
    else if (mi355_sync(st) != 0) rc = -1;
    COUNT(g_launches, launches); COUNT(g_batches, 1); COUNT(g_batched_pictures, K);
    for (int k = 0; k < K; k++) { b[k]->launches = launches; b[k]->done_event = m->used ? m->ev : NULL; }
    return rc;
}

/* The calling thread's picture is launched — by this thread, together with every picture of the same sample format that is waiting, or by
 * the thread that is at it already.  Returns when the launches are in the stream (the filter passes this thread queues next follow them). */
static int commit_launches(Sub *me)
{
    pthread_mutex_lock(&g_mu);
    if (g_nslots < 0) {
        const char *e = getenv("MI355_HEVC_BRIDGE_SETS_IN_FLIGHT"), *solo = getenv("MI355_HEVC_BRIDGE_SOLO");
        g_default_stream = getenv("MI355_HEVC_BRIDGE_DEFAULT_STREAM") != NULL;
        g_nslots = e && *e ? atoi(e) : 4;
        if (g_nslots < 1 || g_default_stream) g_nslots = 1;
        if (g_nslots > MAX_SLOTS) g_nslots = MAX_SLOTS;
        g_solo = solo && *solo && *solo != '0';
    }
    while (g_nwait == MAX_WAITING) pthread_cond_wait(&g_cv, &g_mu);
    g_wait[g_nwait++] = me;
    while (!me->done) {
        struct Merged *m = NULL;
        for (int i = 0; i < g_nslots && !m; i++) if (!g_slot[i].busy) m = &g_slot[i];
        if (!m) { pthread_cond_wait(&g_cv, &g_mu); continue; }
        m->busy = 1;
        if (m->used) {                                /* the slot's last set: while it runs, pictures gather for the next one */
            pthread_mutex_unlock(&g_mu);
            mi355_event_sync(m->ev);
            pthread_mutex_lock(&g_mu);
        }
        /* the oldest waiting picture and all that can share its launches (same bit depth, same intra form); MI355_HEVC_BRIDGE_SOLO=1: it alone */
        Sub *b[MAX_BATCH];
        int K = 0, keep = 0;
        for (int i = 0; i < g_nwait; i++) {
            Sub *u = g_wait[i];
            if (K < (g_solo ? 1 : MAX_BATCH) && (K == 0 || (u->bd == b[0]->bd && u->split_intra == b[0]->split_intra))) b[K++] = u;
            else g_wait[keep++] = u;
        }
        g_nwait = keep;
        if (!K) { m->busy = 0; pthread_cond_broadcast(&g_cv); continue; }       /* another thread took this one's picture along meanwhile */
        pthread_mutex_unlock(&g_mu);
        const int rc = launch_batch(m, b, K);
        pthread_mutex_lock(&g_mu);
        for (int k = 0; k < K; k++) { b[k]->rc = rc; b[k]->done = 1; }
        m->busy = 0;
        pthread_cond_broadcast(&g_cv);
    }
    pthread_mutex_unlock(&g_mu);
    return me->rc;
}

int mi355_hevc_recon_finish(HEVCContext *s, uint8_t *cur[3], uint8_t *fin[3])
{
    if (!R.on || R.s != s) return 0;
    R.on = 0;
    if (R.failed) return -1;
    const HEVCSPS *sps = s->ps.sps;
    Surface *uc = &R.surf[R.cur];
    /* references: whatever the recorded jobs name must hold the decoder's current picture */
    for (int i = 0; i < R.nmc + R.nemu; i++) {
        const Loc *ls[2]; int nl = 0;
        if (i < R.nmc) { ls[nl++] = &R.mc[i].src[0]; if (R.mc[i].kind & 1) ls[nl++] = &R.mc[i].src[1]; }
        else ls[nl++] = &R.emu[i - R.nmc].src;
        for (int q = 0; q < nl; q++) {
            if (ls[q]->surf < 0) continue;
            Surface *u = &R.surf[ls[q]->surf];
            if (u == uc) return -1;                              /* a picture cannot predict from itself */
            if (!surface_current(s, u) && upload_surface(s, u)) return -1;
        }
    }
    /* staging: jobs of every kind sorted by level, coefficients, the intra descriptor */
    const size_t mvf_bytes = (size_t)sps->min_pu_width * sps->min_pu_height * sizeof(MvField);
    const size_t zs_bytes = (size_t)sps->min_tb_width * sps->min_tb_height * sizeof(int);
    size_t o = 0;
    const size_t o_emu = o;   o += ((size_t)R.nemu * sizeof(mi355_edge_emu_job) + 63) & ~(size_t)63;
    const size_t o_mc = o;    o += ((size_t)R.nmc * sizeof(mi355_hevc_mcpred_job) + 63) & ~(size_t)63;
    const size_t o_tu = o;    o += ((size_t)R.ntu * sizeof(mi355_hevc_tu_job) + 63) & ~(size_t)63;
    const size_t o_in = o;    o += ((size_t)R.nintra * sizeof(mi355_hevc_intra_block) + 63) & ~(size_t)63;
    const size_t o_fu = o;    o += ((size_t)R.nintra * sizeof(mi355_hevc_tu_job) + 63) & ~(size_t)63;       /* the unit of every intra block (coeffs NULL: none) */
    const size_t o_desc = o;  o += (sizeof(mi355_hevc_intra_picture) + 63) & ~(size_t)63;
    const size_t o_coef = o;  o += (R.ncoef * sizeof(int16_t) + 63) & ~(size_t)63;
    if (R.h_stage_bytes < o) {                      /* pinned: the one copy of a picture's jobs and coefficients runs at the link's rate */
        if (R.h_stage) mi355_host_free(R.h_stage);
        R.h_stage = mi355_host_alloc(o + o / 4);
        R.h_stage_bytes = R.h_stage ? o + o / 4 : 0;
        if (!R.h_stage) return -1;
    }
    if (dev_ensure(&R.d_stage, &R.d_stage_bytes, o) || dev_ensure(&R.d_emu, &R.d_emu_bytes, R.emu_bytes + 64)) return -1;
    if (R.nintra && (dev_ensure(&R.d_mvf, &R.d_mvf_bytes, mvf_bytes) || dev_ensure(&R.d_zs, &R.d_zs_bytes, zs_bytes))) return -1;
    const int L = R.max_level;
    if (getenv("MI355_HEVC_RECON_TRACE")) {
        int lm = 0, lt = 0, li = 0;
        for (int i = 0; i < R.nmc; i++) if (R.mc[i].level > lm) lm = R.mc[i].level;
        for (int i = 0; i < R.ntu; i++) if (R.tu[i].level > lt) lt = R.tu[i].level;
        for (int i = 0; i < R.nintra; i++) if (R.intra[i].level > li) li = R.intra[i].level;
        fprintf(stderr, "recon: poc %d: %d prediction blocks (levels <= %d), %d transform units (<= %d), %d intra blocks (<= %d), %d windows\n", s->poc, R.nmc, lm, R.ntu, lt, R.nintra, li, R.nemu);
    }
    /* counting sort by level of the three job kinds */
    int *start = calloc((size_t)(L + 2) * 3, sizeof(int));
    if (!start) return -1;
    int *smc = start, *stu = start + (L + 2), *sin = start + 2 * (L + 2);
    for (int i = 0; i < R.nmc; i++) smc[R.mc[i].level + 1]++;
    for (int i = 0; i < R.ntu; i++) if (!R.tu[i].fused) stu[R.tu[i].level + 1]++;
    for (int i = 0; i < R.nintra; i++) sin[R.intra[i].level + 1]++;
    for (int l = 1; l <= L + 1; l++) { smc[l] += smc[l - 1]; stu[l] += stu[l - 1]; sin[l] += sin[l - 1]; }
    mi355_edge_emu_job *je = (mi355_edge_emu_job *)(R.h_stage + o_emu);
    mi355_hevc_mcpred_job *jm = (mi355_hevc_mcpred_job *)(R.h_stage + o_mc);
    mi355_hevc_tu_job *jt = (mi355_hevc_tu_job *)(R.h_stage + o_tu);
    mi355_hevc_intra_block *ji = (mi355_hevc_intra_block *)(R.h_stage + o_in);
    for (int i = 0; i < R.nemu; i++) {
        const EmuRec *e = &R.emu[i];
        mi355_edge_emu_job *j = &je[i];
        memset(j, 0, sizeof(*j));
        j->dst = R.d_emu + e->dst_off;
        /* the job's `src` is the window's first sample: the plane's first sample + sy rows + sx samples */
        j->src = resolve(&e->src) + (ptrdiff_t)e->sy * e->src_stride + (ptrdiff_t)e->sx * R.px;
        j->dst_stride = EDGE_EMU_BUFFER_STRIDE * R.px; j->src_stride = e->src_stride;
        j->block_w = e->bw; j->block_h = e->bh; j->src_x = e->sx; j->src_y = e->sy; j->w = e->w; j->h = e->h;
    }
    {
        int *fill = malloc((size_t)(L + 2) * sizeof(int));
        if (!fill) { free(start); return -1; }
        memcpy(fill, smc, (size_t)(L + 2) * sizeof(int));
        for (int i = 0; i < R.nmc; i++) {
            const McRec *m = &R.mc[i];
            mi355_hevc_mcpred_job *j = &jm[fill[m->level]++];
            memset(j, 0, sizeof(*j));
            j->src0 = resolve(&m->src[0]); j->src0_stride = m->sstride[0];
            if (m->kind & 1) { j->src1 = resolve(&m->src[1]); j->src1_stride = m->sstride[1]; }
            j->dst = resolve(&m->dst); j->dst_stride = m->dstride;
            j->width = (uint8_t)m->w; j->height = (uint8_t)m->h; j->chroma = (uint8_t)m->chroma; j->kind = (uint8_t)m->kind;
            if (m->chroma == 2) { j->src0_b = resolve(&m->srcb[0]); if (m->kind & 1) j->src1_b = resolve(&m->srcb[1]); j->dst_b = resolve(&m->dstb); }
            j->mx0 = (uint8_t)m->mx[0]; j->my0 = (uint8_t)m->my[0]; j->mx1 = (uint8_t)m->mx[1]; j->my1 = (uint8_t)m->my[1];
            j->denom = (uint8_t)m->denom; j->w0 = (int16_t)m->wt[0]; j->w1 = (int16_t)m->wt[1]; j->o0 = (int16_t)m->of[0]; j->o1 = (int16_t)m->of[1];
        }
        memcpy(fill, stu, (size_t)(L + 2) * sizeof(int));
        for (int i = 0; i < R.ntu; i++) {
            const TuRec *t = &R.tu[i];
            if (t->fused) continue;
            mi355_hevc_tu_job *j = &jt[fill[t->level]++];
            memset(j, 0, sizeof(*j));
            j->coeffs = (int16_t *)(R.d_stage + o_coef) + t->coef_off;
            j->dst = resolve(&t->dst); j->dst_stride = t->dstride;
            j->log2_size = (uint8_t)t->log2; j->col_limit = (uint8_t)t->col_limit; j->kind = (uint8_t)t->kind;
        }
        memcpy(fill, sin, (size_t)(L + 2) * sizeof(int));
        mi355_hevc_tu_job *jf = (mi355_hevc_tu_job *)(R.h_stage + o_fu);
        for (int i = 0; i < R.nintra; i++) {
            const int at = fill[R.intra[i].level]++;
            ji[at] = R.intra[i].b;
            mi355_hevc_tu_job *j = &jf[at];
            memset(j, 0, sizeof(*j));
            if (R.intra[i].tu >= 0) {
                const TuRec *t = &R.tu[R.intra[i].tu];
                j->coeffs = (int16_t *)(R.d_stage + o_coef) + t->coef_off;
                j->dst = resolve(&t->dst); j->dst_stride = t->dstride;
                j->log2_size = (uint8_t)t->log2; j->col_limit = (uint8_t)t->col_limit; j->kind = (uint8_t)t->kind;
            }
        }
        free(fill);
    }
    mi355_hevc_intra_picture *d = (mi355_hevc_intra_picture *)(R.h_stage + o_desc);
    memset(d, 0, sizeof(*d));
    for (int k = 0; k < 3; k++) { d->data[k] = uc->dev + uc->off[k]; d->linesize[k] = uc->linesize[k]; }
    d->width = sps->width; d->height = sps->height; d->hshift = sps->hshift[1]; d->vshift = sps->vshift[1];
    d->log2_min_pu_size = sps->log2_min_pu_size; d->log2_min_tb_size = sps->log2_min_tb_size;
    d->min_pu_width = sps->min_pu_width; d->min_pu_height = sps->min_pu_height; d->min_tb_width = sps->min_tb_width;
    d->constrained_intra_pred = s->ps.pps->constrained_intra_pred_flag;
    d->strong_intra_smoothing = sps->sps_strong_intra_smoothing_enable_flag;
    d->tab_mvf = (const mi355_hevc_mvfield *)R.d_mvf; d->min_tb_addr_zs = (const int32_t *)R.d_zs;
    memcpy(R.h_stage + o_coef, R.coef, R.ncoef * sizeof(int16_t));
    /* the coefficients go to this decoder's device staging (the jobs name them there); the job arrays stay here, for the merge */
    int rc = o > o_coef ? mi355_memcpy_h2d(R.d_stage + o_coef, R.h_stage + o_coef, o - o_coef) : 0;
    if (R.nintra) rc |= mi355_memcpy_h2d(R.d_mvf, s->ref->tab_mvf, mvf_bytes) | mi355_memcpy_h2d(R.d_zs, s->ps.pps->min_tb_addr_zs, zs_bytes);
    if (rc) { free(start); return -1; }
    /* windows first (they read reference pictures only), then level by level: the three kinds of one level touch disjoint samples —
     * together with whatever other decoders of this process have waiting (commit_launches) */
    {
        Sub me;
        memset(&me, 0, sizeof(me));
        me.je = je; me.jm = jm; me.jt = jt; me.ji = ji; me.jf = (const mi355_hevc_tu_job *)(R.h_stage + o_fu); me.desc = d;
        me.smc = smc; me.stu = stu; me.sin = sin; me.L = L; me.nemu = R.nemu; me.bd = R.bd; me.split_intra = R.split_intra;
        rc = commit_launches(&me);
        R.launches += me.launches;
        R.done_event = me.done_event;
    }
    free(start);
    if (rc) return -1;
    R.levels_total += (unsigned long)L; COUNT(g_levels, L);
    /* where the finished picture goes */
    Surface *uf = uc;
    if (sps->sao_enabled) {
        uf = surface_of_frame(s, s->sao_frame, 1);
        if (!uf) return -1;
    }
    for (int k = 0; k < 3; k++) { cur[k] = uc->dev + uc->off[k]; fin[k] = uf->dev + uf->off[k]; }
    uf->valid = 1; uf->poc = s->poc; uf->seq = s->seq_decode;         /* true once the filter bridge's passes (queued behind these launches) have run */
    if (uf != uc) uc->valid = 0;
    R.on_device++; COUNT(g_on_device, 1);
    return 1;
}

/* for hosts that want the numbers */
void mi355_hevc_bridge_stats(unsigned long *pictures, unsigned long *on_device, unsigned long *uploads, unsigned long *launches, unsigned long *levels)
{
    if (pictures) *pictures = __atomic_load_n(&g_pictures, __ATOMIC_RELAXED);
    if (on_device) *on_device = __atomic_load_n(&g_on_device, __ATOMIC_RELAXED);
    if (uploads) *uploads = __atomic_load_n(&g_uploads, __ATOMIC_RELAXED);
    if (launches) *launches = __atomic_load_n(&g_launches, __ATOMIC_RELAXED);
    if (levels) *levels = __atomic_load_n(&g_levels, __ATOMIC_RELAXED);
}
/* launch sets issued and the pictures they held (one decoder: equal) */
void mi355_hevc_bridge_batch_stats(unsigned long *sets, unsigned long *pictures)
{
    if (sets) *sets = __atomic_load_n(&g_batches, __ATOMIC_RELAXED);
    if (pictures) *pictures = __atomic_load_n(&g_batched_pictures, __ATOMIC_RELAXED);
}
/* the event behind the launches of the calling decoder's last reconstructed picture (NULL: they ran on the default stream, which orders by itself):
 * work on another stream that reads the picture waits for it (mi355_stream_wait_event) */
void *mi355_hevc_recon_done_event(void) { return g_default_stream ? NULL : R.done_event; }

