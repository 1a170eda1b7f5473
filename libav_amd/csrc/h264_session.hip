/* h264_session.hip — whole-frame decoding sessions (include/mi355_h264_session.h): the AVHWAccel-shaped boundary
 * (start_frame / decode_slice / end_frame, libavcodec/avcodec.h:3062-3086) over the Tier-2 passes.  Host code only: it
 * uses nothing but the library's public C ABI (mi355_h264_frame.h), so what it does is what any caller of Tier 2 has to do —
 * surfaces, staging, intra schedule, descriptor, one launch set per picture. */
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/mi355_h264_session.h"
#include "../../include/mi355dsp.h"

namespace {

constexpr int NSETS = 2;                 /* staging sets: one is filled by the host while the other's copy is in flight */
constexpr unsigned NJOBS = 64;           /* conversion jobs in flight per session */
constexpr size_t up64(size_t v) { return (v + 63) & ~(size_t)63; }

/* the calling thread works on a context's device for the duration of an entry point (public C ABI only, like the rest of this file) */
struct OnDevice {
    int prev;                            /* the thread's own setting (-1: it follows the process default), or -2: nothing to put back */
    explicit OnDevice(int device) : prev(mi355_get_thread_device())
    {
        if (device >= 0 && device != mi355_get_device()) (void)mi355_set_device(device);
        else prev = -2;
    }
    ~OnDevice() { if (prev != -2) (void)mi355_set_device(prev); }
    OnDevice(const OnDevice &) = delete;
    OnDevice &operator=(const OnDevice &) = delete;
};

struct Layout {                          /* byte offsets inside a staging block (host and device blocks share it) */
    size_t desc, slices, mb, mv0, mv1, coef, ilist, istart, total;
};

struct Set {
    uint8_t *host = nullptr;             /* pinned */
    uint8_t *dev = nullptr;
    void *copied = nullptr;              /* event: the block's last host -> device copy has finished */
    bool used = false;
};

}  // namespace

struct mi355_h264_group {
    int device = -1;
    void *stream = nullptr;
    std::vector<mi355_h264_session *> pending;      /* sessions whose latest picture waits for the next flush */
    mi355_h264_frame *h_desc = nullptr, *d_desc = nullptr;   /* pinned / device: the flush's descriptor array */
    int cap = 0;
    void *copied = nullptr;                          /* the last flush's descriptor copy has left h_desc */
    bool used = false;
    std::vector<int32_t> widths;
    int members = 0;
};

struct mi355_h264_session {
    int device = -1;
    mi355_h264_group *group = nullptr;
    bool pending = false;                 /* grouped: the picture of staging set `cur` has not been launched yet */
    int pend_levels = 0;
    int mb_w = 0, mb_h = 0, nmb = 0, nsurf = 0, max_slices = 0;
    int stride[2] = { 0, 0 };             /* bytes per line, or per macroblock row of tiles */
    bool tiled = false;                   /* surfaces are macroblock-tiled (mi355_h264_frame.h) */
    int lin_stride[2] = { 0, 0 };         /* tiled sessions: the line strides of `lin` */
    uint8_t *lin = nullptr;               /* tiled sessions: one picture as planes with line strides (device), between the tiles and the host */
    size_t lin_off[3] = { 0, 0, 0 }, lin_bytes = 0;
    mi355_surface_job *jobs = nullptr;    /* tiled sessions: pinned conversion jobs (ring of NJOBS: a job is read when its launch runs) */
    void *job_done[NJOBS] = {};           /* per ring slot: event behind the launch that reads it (a slot is reused only after it) */
    bool job_used[NJOBS] = {};
    unsigned njob = 0;
    size_t plane_off[3] = { 0, 0, 0 }, surf_bytes = 0;
    uint8_t *surfaces = nullptr;         /* nsurf decoded pictures, then NSETS unfiltered reconstructions */
    void **surf_done = nullptr;          /* per surface: event after its picture's loop filter */
    bool *surf_valid = nullptr;
    Layout lay{};
    Set set[NSETS];
    void *stream = nullptr, *copy_stream = nullptr;
    uint8_t *covered = nullptr;          /* per macroblock of the open picture: some slice delivered it */
    int32_t *level_widths = nullptr;
    /* the open picture */
    bool open = false;
    int cur = 0, nslices = 0, ncovered = 0;
    int rows = 0, nmb_pic = 0;            /* macroblock rows / macroblocks of the open picture (a field has half the frame's) */
    unsigned long frames = 0;
    mi355_h264_picture_params pp{};

    uint8_t *plane(int surface, int p) const { return surfaces + (size_t)surface * surf_bytes + plane_off[p]; }
};

extern "C" void mi355_h264_session_close(mi355_h264_session *s)
{
    if (!s) return;
    OnDevice on(s->device);
    if (s->group && s->pending) mi355_h264_group_flush(s->group);
    if (s->stream) mi355_sync(s->stream);
    if (s->copy_stream) mi355_sync(s->copy_stream);
    for (int k = 0; k < NSETS; k++) {
        if (s->set[k].host) mi355_host_free(s->set[k].host);
        if (s->set[k].dev) mi355_free(s->set[k].dev);
        if (s->set[k].copied) mi355_event_destroy(s->set[k].copied);
    }
    if (s->surf_done)
        for (int i = 0; i < s->nsurf; i++)
            if (s->surf_done[i]) mi355_event_destroy(s->surf_done[i]);
    std::free(s->surf_done);
    std::free(s->surf_valid);
    std::free(s->covered);
    std::free(s->level_widths);
    if (s->surfaces) mi355_free(s->surfaces);
    if (s->lin) mi355_free(s->lin);
    if (s->jobs) mi355_host_free(s->jobs);
    for (unsigned k = 0; k < NJOBS; k++) if (s->job_done[k]) mi355_event_destroy(s->job_done[k]);
    if (s->stream && !s->group) mi355_stream_destroy(s->stream);
    if (s->copy_stream) mi355_stream_destroy(s->copy_stream);
    if (s->group) s->group->members--;
    delete s;
}

static int session_open_impl(mi355_h264_session **out, const mi355_h264_session_params *p, mi355_h264_group *g)
{
    if (!out || !p || p->mb_width <= 0 || p->mb_height <= 0 || p->num_surfaces < 2 || p->num_surfaces > 64 || p->max_slices < 0 || p->max_slices > 255) return -1;
    if (p->surface_layout != MI355_SURFACE_LINEAR && p->surface_layout != MI355_SURFACE_TILED) return -1;
    if ((long)p->mb_width * p->mb_height >= (1L << 24)) return -1;
    if (p->device < 0 || p->device > mi355_device_count()) return -1;
    const int device = p->device > 0 ? p->device - 1 : mi355_get_device();
    if (device < 0 || (g && g->device != device)) return -1;      /* no device chosen yet / not the group's */
    OnDevice on(device);
    mi355_h264_session *s = new (std::nothrow) mi355_h264_session;
    if (!s) return -3;
    s->device = device;
    s->mb_w = p->mb_width; s->mb_h = p->mb_height; s->nmb = p->mb_width * p->mb_height;
    s->nsurf = p->num_surfaces; s->max_slices = p->max_slices ? p->max_slices : 64;
    /* surfaces: rows of whole 64-byte (luma) / 32-byte (chroma) pieces, the alignment the kernels' 16 / 8-byte paths want */
    s->stride[0] = (int)up64((size_t)16 * s->mb_w); s->stride[1] = s->stride[0] / 2;
    size_t ysz = (size_t)s->stride[0] * 16 * s->mb_h, csz = (size_t)s->stride[1] * 8 * s->mb_h;
    s->plane_off[0] = 0; s->plane_off[1] = ysz; s->plane_off[2] = ysz + csz; s->surf_bytes = up64(ysz + 2 * csz);
    s->tiled = p->surface_layout == MI355_SURFACE_TILED;
    if (s->tiled) {
        /* the line-stride picture between tiles and host; then the surfaces proper: luma tiles, chroma tiles */
        s->lin_stride[0] = s->stride[0]; s->lin_stride[1] = s->stride[1];
        s->lin_off[0] = 0; s->lin_off[1] = ysz; s->lin_off[2] = ysz + csz; s->lin_bytes = s->surf_bytes;
        s->stride[0] = MI355_TILE_LUMA_BYTES * s->mb_w; s->stride[1] = MI355_TILE_CHROMA_BYTES * s->mb_w;
        ysz = (size_t)s->stride[0] * s->mb_h; csz = (size_t)s->stride[1] * s->mb_h;
        s->plane_off[0] = 0; s->plane_off[1] = s->plane_off[2] = ysz; s->surf_bytes = up64(ysz + csz);
    }
    Layout &l = s->lay;
    size_t o = 0;
    l.desc = o;   o += up64(sizeof(mi355_h264_frame));
    l.slices = o; o += up64((size_t)s->max_slices * sizeof(mi355_h264_slice));
    l.mb = o;     o += up64((size_t)s->nmb * sizeof(mi355_h264_mb));
    l.mv0 = o;    o += up64((size_t)s->nmb * 64);
    l.mv1 = o;    o += up64((size_t)s->nmb * 64);
    l.coef = o;   o += up64((size_t)s->nmb * MI355_H264_COEFS_PER_MB * 2);
    l.ilist = o;  o += up64((size_t)s->nmb * 4);
    l.istart = o; o += up64((size_t)(s->mb_w + 2 * s->mb_h + 2) * 4);
    l.total = o;
    bool ok = true;
    s->surfaces = static_cast<uint8_t *>(mi355_malloc((size_t)(s->nsurf + NSETS) * s->surf_bytes));
    ok = ok && s->surfaces;
    if (s->tiled) {
        s->lin = static_cast<uint8_t *>(mi355_malloc(s->lin_bytes));
        s->jobs = static_cast<mi355_surface_job *>(mi355_host_alloc(NJOBS * sizeof(mi355_surface_job)));
        ok = ok && s->lin && s->jobs;
    }
    for (int k = 0; k < NSETS && ok; k++) {
        s->set[k].host = static_cast<uint8_t *>(mi355_host_alloc(l.total));
        s->set[k].dev = static_cast<uint8_t *>(mi355_malloc(l.total));
        s->set[k].copied = mi355_event_create();
        ok = ok && s->set[k].host && s->set[k].dev && s->set[k].copied;
    }
    s->group = g;
    s->stream = g ? g->stream : mi355_stream_create();
    s->copy_stream = mi355_stream_create();
    s->surf_done = static_cast<void **>(std::calloc((size_t)s->nsurf, sizeof(void *)));
    s->surf_valid = static_cast<bool *>(std::calloc((size_t)s->nsurf, sizeof(bool)));
    s->covered = static_cast<uint8_t *>(std::calloc((size_t)s->nmb, 1));
    s->level_widths = static_cast<int32_t *>(std::calloc((size_t)(s->mb_w + 2 * s->mb_h + 2), sizeof(int32_t)));
    ok = ok && s->stream && s->copy_stream && s->surf_done && s->surf_valid && s->covered && s->level_widths;
    for (int i = 0; i < s->nsurf && ok; i++) ok = (s->surf_done[i] = mi355_event_create()) != nullptr;
    if (!ok) { s->group = nullptr; if (g) s->stream = nullptr; mi355_h264_session_close(s); return -3; }
    if (g) g->members++;
    *out = s;
    return 0;
}
extern "C" int mi355_h264_session_open(mi355_h264_session **out, const mi355_h264_session_params *p) { return session_open_impl(out, p, nullptr); }
extern "C" int mi355_h264_session_open_grouped(mi355_h264_session **out, const mi355_h264_session_params *p, mi355_h264_group *g)
{
    return g ? session_open_impl(out, p, g) : -1;
}

extern "C" int mi355_h264_start_frame(mi355_h264_session *s, const mi355_h264_picture_params *pp)
{
    if (!s || !pp || s->open) return -1;
    OnDevice on(s->device);
    /* grouped: this session's previous picture still waits for a launch — it may be a reference of this one, and its staging
     * set comes up for reuse after the next: everything waiting goes out now */
    if (s->group && s->pending) { const int rc = mi355_h264_group_flush(s->group); if (rc) return rc; }
    if (pp->surface < 0 || pp->surface >= s->nsurf || pp->nslots < 0 || pp->nslots > MI355_H264_MAX_SLOTS) return -1;
    if (pp->field < 0 || pp->field > 2 || (pp->field && (s->mb_h & 1))) return -1;
    if (pp->field && s->tiled) return -1;                         /* tiled surfaces hold frame pictures only */
    for (int i = 0; i < pp->nslots; i++) {
        const int r = pp->ref_surface[i];
        /* a picture does not predict from itself — except a field from the OTHER field of its frame */
        if (r < -1 || r >= s->nsurf || (r == pp->surface && !(pp->field && pp->ref_parity[i] == 2 - pp->field))) return -1;
        if (r >= 0 && !s->surf_valid[r]) return -1;           /* a reference nobody decoded */
    }
    s->cur = (int)(s->frames % NSETS);
    Set &st = s->set[s->cur];
    if (st.used && mi355_event_sync(st.copied) != 0) return -2;   /* the host block is free once its copy has left */
    s->pp = *pp;
    s->rows = pp->field ? s->mb_h / 2 : s->mb_h;
    s->nmb_pic = s->mb_w * s->rows;
    s->nslices = 0; s->ncovered = 0;
    std::memset(s->covered, 0, (size_t)s->nmb);
    s->open = true;
    return 0;
}

extern "C" int mi355_h264_decode_slice(mi355_h264_session *s, const mi355_h264_slice *hdr, int first_mb, int nmbs, const int32_t *mb_addr,
                                       const mi355_h264_mb *mb, const int16_t *mv0, const int16_t *mv1, const int16_t *coef)
{
    if (!s || !s->open || !hdr || !mb || !mv0 || !coef || nmbs <= 0) return -1;
    if (s->pp.two_lists && !mv1) return -1;
    if (s->nslices >= s->max_slices) return -1;
    if (!mb_addr && (first_mb < 0 || first_mb + nmbs > s->nmb_pic)) return -1;
    uint8_t *h = s->set[s->cur].host;
    const Layout &l = s->lay;
    const int sid = s->nslices;
    std::memcpy(h + l.slices + (size_t)sid * sizeof(mi355_h264_slice), hdr, sizeof(mi355_h264_slice));
    mi355_h264_mb *dmb = reinterpret_cast<mi355_h264_mb *>(h + l.mb);
    if (!mb_addr) {
        /* a run of consecutive macroblocks: the four arrays are four copies */
        std::memcpy(dmb + first_mb, mb, (size_t)nmbs * sizeof(mi355_h264_mb));
        std::memcpy(h + l.mv0 + (size_t)first_mb * 64, mv0, (size_t)nmbs * 64);
        if (s->pp.two_lists) std::memcpy(h + l.mv1 + (size_t)first_mb * 64, mv1, (size_t)nmbs * 64);
        std::memcpy(h + l.coef + (size_t)first_mb * MI355_H264_COEFS_PER_MB * 2, coef, (size_t)nmbs * MI355_H264_COEFS_PER_MB * 2);
    }
    for (int i = 0; i < nmbs; i++) {
        const int a = mb_addr ? mb_addr[i] : first_mb + i;
        if (a < 0 || a >= s->nmb_pic) return -1;
        if (mb_addr) {
            dmb[a] = mb[i];
            std::memcpy(h + l.mv0 + (size_t)a * 64, mv0 + (size_t)i * 32, 64);
            if (s->pp.two_lists) std::memcpy(h + l.mv1 + (size_t)a * 64, mv1 + (size_t)i * 32, 64);
            std::memcpy(h + l.coef + (size_t)a * MI355_H264_COEFS_PER_MB * 2, coef + (size_t)i * MI355_H264_COEFS_PER_MB, MI355_H264_COEFS_PER_MB * 2);
        }
        dmb[a].slice_id = (uint8_t)sid;
        if (!s->covered[a]) { s->covered[a] = 1; s->ncovered++; }
    }
    s->nslices++;
    return 0;
}

extern "C" int mi355_h264_end_frame(mi355_h264_session *s)
{
    if (!s || !s->open) return -1;
    OnDevice on(s->device);
    s->open = false;
    if (s->ncovered != s->nmb_pic || s->nslices == 0) return -4;
    Set &st = s->set[s->cur];
    uint8_t *h = st.host, *d = st.dev;
    const Layout &l = s->lay;
    int width = 0;
    int32_t *istart = reinterpret_cast<int32_t *>(h + l.istart);
    const int levels = mi355_h264_intra_schedule(reinterpret_cast<mi355_h264_mb *>(h + l.mb), s->mb_w, s->rows,
                                                 reinterpret_cast<uint32_t *>(h + l.ilist), istart, &width);
    if (levels < 0) return -1;
    for (int k = 0; k < levels; k++) s->level_widths[k] = istart[k + 1] - istart[k];
    mi355_h264_frame *fr = reinterpret_cast<mi355_h264_frame *>(h + l.desc);
    std::memset(fr, 0, sizeof(*fr));
    /* a field picture: every other line of the surface (first line at the field's parity, strides doubled); its references are
     * fields addressed the same way; the unfiltered reconstruction surface keeps plain rows */
    const int fld = s->pp.field, fs = fld ? 2 : 1;
    fr->mb_width = s->mb_w; fr->mb_height = s->rows;
    fr->field_picture = fld != 0;
    fr->surface_layout = s->tiled ? MI355_SURFACE_TILED : MI355_SURFACE_LINEAR;
    const int recon = s->nsurf + s->cur;
    for (int p = 0; p < 3; p++) { fr->dst[p] = s->plane(s->pp.surface, p) + (fld == 2 ? s->stride[p ? 1 : 0] : 0); fr->recon[p] = s->plane(recon, p); }
    fr->dst_stride[0] = fs * s->stride[0]; fr->recon_stride[0] = s->stride[0];
    fr->dst_stride[1] = fs * s->stride[1]; fr->recon_stride[1] = s->stride[1];
    for (int i = 0; i < MI355_H264_MAX_SLOTS; i++) {
        /* an unused slot points at the first reference (any readable surface: nothing valid names it) */
        const int r = i < s->pp.nslots && s->pp.ref_surface[i] >= 0 ? s->pp.ref_surface[i] : (s->pp.nslots > 0 && s->pp.ref_surface[0] >= 0 ? s->pp.ref_surface[0] : s->pp.surface);
        const int rpar = fld && i < s->pp.nslots ? (s->pp.ref_parity[i] != 0) : 0;
        for (int p = 0; p < 3; p++) fr->ref[i][p] = s->plane(r, p) + (rpar ? s->stride[p ? 1 : 0] : 0);
    }
    fr->mb = reinterpret_cast<const mi355_h264_mb *>(d + l.mb);
    fr->mv[0] = reinterpret_cast<const int16_t *>(d + l.mv0);
    fr->mv[1] = s->pp.two_lists ? reinterpret_cast<const int16_t *>(d + l.mv1) : nullptr;
    fr->coef = reinterpret_cast<const int16_t *>(d + l.coef);
    fr->slices = reinterpret_cast<const mi355_h264_slice *>(d + l.slices);
    fr->nslices = s->nslices;
    fr->max_intra_level = levels;
    fr->intra_list = reinterpret_cast<const uint32_t *>(d + l.ilist);
    fr->intra_level_start = reinterpret_cast<const int32_t *>(d + l.istart);
    fr->max_level_width = width;
    fr->flags = levels > 0 && istart[levels] == s->nmb_pic ? MI355_FRAME_NO_INTER : 0;      /* an I picture: the inter pass has nothing to do */
    if (s->group) {
        /* grouped: the picture waits for mi355_h264_group_flush(), which launches it together with the other sessions' */
        s->pend_levels = levels;
        s->pending = true;
        s->group->pending.push_back(s);
        s->frames++;
        return 0;
    }
    /* one copy for the whole block (P pictures without list 1 still send the unused vector area: it is 6 % of the block) */
    if (mi355_memcpy_h2d_async(d, h, l.total, s->stream) != 0) return -2;
    if (mi355_event_record(st.copied, s->stream) != 0) return -2;
    st.used = true;
    const int rc = mi355_h264_decode_frames_layouts_dev(reinterpret_cast<const mi355_h264_frame *>(d + l.desc), 1, s->mb_w, s->rows, levels,
                                                        s->level_widths, s->tiled ? MI355_LAYOUTS_TILED : MI355_LAYOUTS_LINEAR, s->stream);
    if (rc != 0) return rc == -1 ? -1 : -2;
    if (mi355_event_record(s->surf_done[s->pp.surface], s->stream) != 0) return -2;
    s->surf_valid[s->pp.surface] = true;
    s->frames++;
    return 0;
}

static int flush_if_pending(mi355_h264_session *s) { return s->group && s->pending ? mi355_h264_group_flush(s->group) : 0; }

extern "C" int mi355_h264_surface_wait(mi355_h264_session *s, int surface)
{
    if (!s || surface < 0 || surface >= s->nsurf) return -1;
    OnDevice on(s->device);
    if (flush_if_pending(s) != 0) return -2;
    if (!s->surf_valid[surface]) return -1;
    return mi355_event_sync(s->surf_done[surface]) == 0 ? 0 : -2;
}

/* one conversion between surface `surface` of a tiled session and planes with line strides, on `stream`; the job lives in the
 * session's pinned ring until the launch has read it */
static int convert(mi355_h264_session *s, int surface, uint8_t *const lin[3], const int lin_stride[3], int to_tiled, void *stream)
{
    const unsigned slot = s->njob++ % NJOBS;
    /* the launch that read this slot last must have run (more than NJOBS exports queued on a busy stream overwrote jobs not yet read) */
    if (!s->job_done[slot] && !(s->job_done[slot] = mi355_event_create())) return -4;
    if (s->job_used[slot] && mi355_event_sync(s->job_done[slot]) != 0) return -4;
    mi355_surface_job *j = &s->jobs[slot];
    std::memset(j, 0, sizeof(*j));
    for (int p = 0; p < 3; p++) j->lin[p] = lin[p];
    j->tiled[0] = s->plane(surface, 0); j->tiled[1] = s->plane(surface, 1);
    j->lin_stride[0] = lin_stride[0]; j->lin_stride[1] = lin_stride[1];
    j->tiled_stride[0] = s->stride[0]; j->tiled_stride[1] = s->stride[1];
    j->mb_width = s->mb_w; j->mb_height = s->mb_h; j->to_tiled = to_tiled;
    const int rc = mi355_h264_surface_convert_dev(j, 1, s->mb_w, s->mb_h, stream);
    if (rc == 0 && mi355_event_record(s->job_done[slot], stream) == 0) s->job_used[slot] = true;
    return rc;
}

extern "C" int mi355_h264_get_frame(mi355_h264_session *s, int surface, uint8_t *const dst[3], const int dst_stride[3])
{
    if (!s || !dst || !dst_stride || surface < 0 || surface >= s->nsurf) return -1;
    OnDevice on(s->device);
    if (flush_if_pending(s) != 0) return -2;
    if (!s->surf_valid[surface]) return -1;
    /* on the copy stream, behind the picture's event: later pictures queued on the session's stream are not waited for */
    if (mi355_stream_wait_event(s->copy_stream, s->surf_done[surface]) != 0) return -2;
    if (s->tiled) {
        /* tiles -> lines on the device (s->lin; the copy stream orders its users), then the lines to the host */
        uint8_t *const lin[3] = { s->lin + s->lin_off[0], s->lin + s->lin_off[1], s->lin + s->lin_off[2] };
        const int ls[3] = { s->lin_stride[0], s->lin_stride[1], s->lin_stride[1] };
        if (convert(s, surface, lin, ls, 0, s->copy_stream) != 0) return -2;
    }
    for (int p = 0; p < 3; p++) {
        const size_t w = (size_t)(p ? 8 : 16) * s->mb_w, rows = (size_t)(p ? 8 : 16) * s->mb_h;
        if (!dst[p] || dst_stride[p] < (int)w) return -1;
        const uint8_t *src = s->tiled ? s->lin + s->lin_off[p] : s->plane(surface, p);
        const size_t st = (size_t)(s->tiled ? s->lin_stride[p ? 1 : 0] : s->stride[p ? 1 : 0]);
        if (mi355_memcpy2d_d2h_async(dst[p], (size_t)dst_stride[p], src, st, w, rows, s->copy_stream) != 0) return -2;
    }
    return mi355_sync(s->copy_stream) == 0 ? 0 : -2;
}

extern "C" int mi355_h264_put_frame(mi355_h264_session *s, int surface, const uint8_t *const src[3], const int src_stride[3])
{
    if (!s || !src || !src_stride || s->open || surface < 0 || surface >= s->nsurf) return -1;
    OnDevice on(s->device);

    if (flush_if_pending(s) != 0) return -2;
    const size_t img_bytes = s->tiled ? s->lin_bytes : s->surf_bytes;
    uint8_t *img = static_cast<uint8_t *>(std::malloc(img_bytes));
    if (!img) return -3;
    for (int p = 0; p < 3; p++) {
        const size_t w = (size_t)(p ? 8 : 16) * s->mb_w, rows = (size_t)(p ? 8 : 16) * s->mb_h;
        if (!src[p] || src_stride[p] < (int)w) { std::free(img); return -1; }
        const size_t off = s->tiled ? s->lin_off[p] : s->plane_off[p], st = (size_t)(s->tiled ? s->lin_stride[p ? 1 : 0] : s->stride[p ? 1 : 0]);
        for (size_t r = 0; r < rows; r++) std::memcpy(img + off + r * st, src[p] + r * (size_t)src_stride[p], w);
    }
    /* behind everything queued (pictures that still read the surface's old contents), then wait: `img` goes away */
    int rc;
    if (s->tiled) {
        /* lines -> the session's line picture -> tiles (the copy stream orders the users of s->lin) */
        uint8_t *const lin[3] = { s->lin + s->lin_off[0], s->lin + s->lin_off[1], s->lin + s->lin_off[2] };
        const int ls[3] = { s->lin_stride[0], s->lin_stride[1], s->lin_stride[1] };
        rc = mi355_sync(s->stream) == 0 && mi355_sync(s->copy_stream) == 0 && mi355_memcpy_h2d(s->lin, img, img_bytes) == 0 &&
             convert(s, surface, lin, ls, 1, s->copy_stream) == 0 && mi355_sync(s->copy_stream) == 0 ? 0 : -2;
    } else
        rc = mi355_sync(s->stream) == 0 && mi355_memcpy_h2d(s->surfaces + (size_t)surface * s->surf_bytes, img, s->surf_bytes) == 0 ? 0 : -2;
    std::free(img);
    if (rc == 0) rc = mi355_event_record(s->surf_done[surface], s->stream) == 0 ? 0 : -2;
    if (rc == 0) s->surf_valid[surface] = true;
    return rc;
}

extern "C" const uint8_t *mi355_h264_surface_dev(mi355_h264_session *s, int surface, int plane, int *stride)
{
    if (!s || surface < 0 || surface >= s->nsurf || plane < 0 || plane > 2) return nullptr;
    if (stride) *stride = s->stride[plane ? 1 : 0];
    return s->plane(surface, plane);
}

extern "C" int mi355_h264_export_frame_dev(mi355_h264_session *s, int surface, uint8_t *const dst[3], const int dst_stride[3], void *stream)
{
    if (!s || !dst || !dst_stride || surface < 0 || surface >= s->nsurf) return -1;
    OnDevice on(s->device);
    if (flush_if_pending(s) != 0) return -2;
    if (!s->surf_valid[surface]) return -1;
    for (int p = 0; p < 3; p++)
        if (!dst[p] || dst_stride[p] < (p ? 8 : 16) * s->mb_w || (p == 2 && dst_stride[2] != dst_stride[1])) return -1;
    void *st = stream ? stream : s->stream;
    if (st != s->stream && mi355_stream_wait_event(st, s->surf_done[surface]) != 0) return -2;
    if (s->tiled) return convert(s, surface, dst, dst_stride, 0, st) == 0 ? 0 : -2;
    for (int p = 0; p < 3; p++) {
        const size_t w = (size_t)(p ? 8 : 16) * s->mb_w, rows = (size_t)(p ? 8 : 16) * s->mb_h;
        if (mi355_memcpy2d_d2d_async(dst[p], (size_t)dst_stride[p], s->plane(surface, p), (size_t)s->stride[p ? 1 : 0], w, rows, st) != 0) return -2;
    }
    return 0;
}

extern "C" void *mi355_h264_session_stream(mi355_h264_session *s) { return s ? s->stream : nullptr; }

/* ---- groups: the pictures of many sessions in ONE launch set ------------------------------------------------------- */
extern "C" int mi355_h264_group_create(mi355_h264_group **out)
{
    if (!out) return -1;
    if (mi355_get_device() < 0) return -1;
    mi355_h264_group *g = new (std::nothrow) mi355_h264_group;
    if (!g) return -3;
    g->device = mi355_get_device();
    g->stream = mi355_stream_create();
    g->copied = mi355_event_create();
    if (!g->stream || !g->copied) { mi355_h264_group_destroy(g); return -3; }
    *out = g;
    return 0;
}

extern "C" void mi355_h264_group_destroy(mi355_h264_group *g)
{
    if (!g) return;
    OnDevice on(g->device);
    if (g->stream) mi355_sync(g->stream);
    if (g->h_desc) mi355_host_free(g->h_desc);
    if (g->d_desc) mi355_free(g->d_desc);
    if (g->copied) mi355_event_destroy(g->copied);
    if (g->stream) mi355_stream_destroy(g->stream);
    delete g;
}

extern "C" int mi355_h264_group_flush(mi355_h264_group *g)
{
    if (!g) return -1;
    OnDevice on(g->device);
    const int n = (int)g->pending.size();
    if (!n) return 0;
    if (n > g->cap) {
        if (g->used && mi355_event_sync(g->copied) != 0) return -2;
        if (g->h_desc) mi355_host_free(g->h_desc);
        if (g->d_desc) { mi355_sync(g->stream); mi355_free(g->d_desc); }
        g->cap = n + 16;
        g->h_desc = static_cast<mi355_h264_frame *>(mi355_host_alloc((size_t)g->cap * sizeof(mi355_h264_frame)));
        g->d_desc = static_cast<mi355_h264_frame *>(mi355_malloc((size_t)g->cap * sizeof(mi355_h264_frame)));
        g->used = false;
        if (!g->h_desc || !g->d_desc) { g->cap = 0; return -3; }
    }
    if (g->used && mi355_event_sync(g->copied) != 0) return -2;
    int max_w = 0, max_h = 0, max_l = 0, rc = 0;
    for (mi355_h264_session *s : g->pending) {
        max_w = s->mb_w > max_w ? s->mb_w : max_w; max_h = s->mb_h > max_h ? s->mb_h : max_h;
        max_l = s->pend_levels > max_l ? s->pend_levels : max_l;
    }
    g->widths.assign((size_t)max_l + 1, 0);
    int i = 0, layouts = 0;
    for (mi355_h264_session *s : g->pending) {
        Set &st = s->set[s->cur];
        /* the session's records travel in one copy; its descriptor joins the flush's array */
        if (mi355_memcpy_h2d_async(st.dev, st.host, s->lay.total, g->stream) != 0 || mi355_event_record(st.copied, g->stream) != 0) rc = -2;
        st.used = true;
        g->h_desc[i++] = *reinterpret_cast<const mi355_h264_frame *>(st.host + s->lay.desc);
        layouts |= s->tiled ? MI355_LAYOUTS_TILED : MI355_LAYOUTS_LINEAR;
        for (int l = 0; l < s->pend_levels; l++) if (s->level_widths[l] > g->widths[(size_t)l]) g->widths[(size_t)l] = s->level_widths[l];
    }
    if (rc == 0 && (mi355_memcpy_h2d_async(g->d_desc, g->h_desc, (size_t)n * sizeof(mi355_h264_frame), g->stream) != 0 ||
                    mi355_event_record(g->copied, g->stream) != 0)) rc = -2;
    g->used = true;
    if (rc == 0) {
        const int r = mi355_h264_decode_frames_layouts_dev(g->d_desc, n, max_w, max_h, max_l, g->widths.data(), layouts, g->stream);
        if (r != 0) rc = r == -1 ? -1 : -2;
    }
    for (mi355_h264_session *s : g->pending) {
        if (rc == 0 && mi355_event_record(s->surf_done[s->pp.surface], g->stream) == 0) s->surf_valid[s->pp.surface] = true;
        else if (rc == 0) rc = -2;
        s->pending = false;
    }
    g->pending.clear();
    return rc;
}
