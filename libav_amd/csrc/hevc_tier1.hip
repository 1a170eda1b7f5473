/*
 * hevc_tier1.hip — Tier-1 (per-call, host pointers) HEVC entries: the kernels behind
 * HEVCDSPContext (libavcodec/hevcdsp.h:41-114, filled by hevcdsp.c:135-253) and the pure
 * predictors of HEVCPredContext (hevcdec.h:399-409, hevcpred.c:37-73), bit depths 8/9/10.
 * One call = one staging round trip + one 64-lane workgroup (see mi355_rt.h).  put_pcm and
 * intra_pred[] need the decoder's bit reader / context and stay with the reference's C.
 */
#include "mi355_rt.h"
#include "hevc_dev.h"
#include "../../include/mi355dsp.h"

using namespace mi355;

#define LAUNCH1(kernel, a, ...) hipLaunchKernelGGL(kernel, dim3(1), dim3(64), 0, (a).stream, __VA_ARGS__)

template <int BD> struct Px { static constexpr int bytes = BD > 8 ? 2 : 1; };

/* ---- a13: residual add, transform-skip scaling ------------------------------------------------ */
__global__ void __launch_bounds__(64) k_hevc_add_residual(uint8_t *dst, int st, const int16_t *res, int size, int bd)
{
    for (int i = lane_id(); i < size * size; i += 64) {
        const int y = i / size, x = i - y * size;
        stpx(dst, x + y * st, clip_px(ldpx(dst, x + y * st, bd) + res[i], bd), bd);
    }
}
template <int SIZE, int BD> static void add_residual_shim(uint8_t *dst, int16_t *res, ptrdiff_t stride)
{
    Arena &a = arena();
    Win w = win_pack(a, dst, stride, SIZE * Px<BD>::bytes, SIZE);
    size_t r = a.take(SIZE * SIZE * 2);
    std::memcpy(a.h<int16_t>(r), res, SIZE * SIZE * 2);
    a.upload();
    LAUNCH1(k_hevc_add_residual, a, a.d<uint8_t>(w.off), w.pitch / Px<BD>::bytes, a.d<int16_t>(r), SIZE, BD);
    a.download();
    win_unpack(a, w, dst, stride, 0, 0, SIZE * Px<BD>::bytes, SIZE);
}

/* ---- put_pcm (hevcdsp_template.c:28-41): the raw PCM levels, read from the bitstream with get_bits()
 * (get_bits.h:228-237, big-endian 32-bit window, position clamped to size_in_bits_plus8), are scaled to the
 * sample depth.  Walking the bit reader is host bookkeeping; the sample arithmetic and stores run on the device. */
__global__ void __launch_bounds__(64) k_hevc_put_pcm(uint8_t *dst, int st, const uint16_t *levels, int size, int shift, int bd)
{
    for (int i = lane_id(); i < size * size; i += 64) {
        const int y = i / size, x = i - y * size;
        stpx(dst, x + y * st, (int)((unsigned)levels[i] << shift) & (bd > 8 ? 0xFFFF : 0xFF), bd);
    }
}
template <int BD> static void put_pcm_shim(uint8_t *dst, ptrdiff_t stride, int size, GetBitContext *gb, int pcm_bit_depth)
{
    Arena &a = arena();
    Win w = win_pack(a, nullptr, 0, size * Px<BD>::bytes, size, 0, 0);
    size_t l = a.take((size_t)size * size * 2);
    uint16_t *lv = a.h<uint16_t>(l);
    unsigned index = (unsigned)gb->index;
    const unsigned limit = (unsigned)gb->size_in_bits_plus8;
    for (int i = 0; i < size * size; i++) {
        const uint8_t *p = gb->buffer + (index >> 3);
        const uint32_t cache = (((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]) << (index & 7);
        lv[i] = (uint16_t)(cache >> (32 - pcm_bit_depth));
        index = index + (unsigned)pcm_bit_depth < limit ? index + (unsigned)pcm_bit_depth : limit;
    }
    gb->index = (int)index;
    a.upload();
    LAUNCH1(k_hevc_put_pcm, a, a.d<uint8_t>(w.off), w.pitch / Px<BD>::bytes, a.d<const uint16_t>(l), size, BD - pcm_bit_depth, BD);
    a.download();
    win_unpack(a, w, dst, stride, 0, 0, size * Px<BD>::bytes, size);
}

/* mode 0: dequant (hevcdsp_template.c:84-98), 1: DST 4x4 (:103-136), 2: idct_dc (:238-252, size in arg),
 * 3: full idct with col_limit (:208-236) */
__global__ void __launch_bounds__(64) k_hevc_transform(int16_t *c, int mode, int size, int col_limit, int bd)
{
    __shared__ IdctScratch s;
    const int lane = lane_id(), n = size * size;
    if (mode == 0) {
        const int shift = 13 - bd, off = 1 << (shift - 1);
        if (lane < 16) c[lane] = (int16_t)((c[lane] + off) >> shift);
        return;
    }
    if (mode == 2) {
        const int shift = 14 - bd, add = 1 << (shift - 1);
        const int v = (((c[0] + 1) >> 1) + add) >> shift;
        __syncthreads();
        for (int i = lane; i < n; i += 64) c[i] = (int16_t)v;
        return;
    }
    for (int i = lane; i < n; i += 64) s.c[0][i] = c[i];
    __syncthreads();
    const bool act = lane < 32;           /* one block: the first half-wave works */
    if (mode == 1) hevc_dst4_wave(s.c[0], bd, lane, act);
    else if (size == 4) hevc_idct_half<4>(s.c[0], lane & 31, act, col_limit, bd);
    else if (size == 8) hevc_idct_half<8>(s.c[0], lane & 31, act, col_limit, bd);
    else if (size == 16) hevc_idct_half<16>(s.c[0], lane & 31, act, col_limit, bd);
    else hevc_idct_half<32>(s.c[0], lane & 31, act, col_limit, bd);
    for (int i = lane; i < n; i += 64) c[i] = s.c[0][i];
}
static void transform_run(int16_t *coeffs, int mode, int size, int col_limit, int bd)
{
    Arena &a = arena();
    const size_t n = (size_t)size * size * 2, c = a.take(n);
    std::memcpy(a.h<int16_t>(c), coeffs, n);
    a.upload();
    LAUNCH1(k_hevc_transform, a, a.d<int16_t>(c), mode, size, col_limit, bd);
    a.download();
    std::memcpy(coeffs, a.h<int16_t>(c), n);
}
template <int BD> static void dequant_shim(int16_t *c) { transform_run(c, 0, 4, 0, BD); }
template <int BD> static void dst4_shim(int16_t *c) { transform_run(c, 1, 4, 0, BD); }
template <int SIZE, int BD> static void idct_shim(int16_t *c, int col_limit) { transform_run(c, 3, SIZE, col_limit, BD); }
template <int SIZE, int BD> static void idct_dc_shim(int16_t *c) { transform_run(c, 2, SIZE, 0, BD); }

/* ---- a17: SAO ------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(64)
k_hevc_sao(uint8_t *dbase, int dt, int dox, int doy, const uint8_t *sbase, int st, int sox, int soy, SaoJob j)
{
    const int px = j.bd > 8 ? 2 : 1;
    /* (dox,doy)/(sox,soy): window coordinates of the caller's (0,0) sample */
    __shared__ int tbl[32];
    hevc_sao_wave(dbase + (ptrdiff_t)(doy * dt + dox) * px, dt, sbase + (ptrdiff_t)(soy * st + sox) * px, st, j, tbl);
}
template <int CLS, int BD, int EDGE>
static void sao_run(uint8_t *dst, uint8_t *src, ptrdiff_t stride, SAOParams *sao, int *borders, int width, int height,
                    int c_idx, int vert_edge, int horiz_edge, int diag_edge)
{
    constexpr int px = Px<BD>::bytes;
    SaoJob j{};
    j.width = width; j.height = height; j.c_idx = c_idx; j.cls = CLS; j.bd = BD; j.edge = EDGE;
    for (int k = 0; k < 4; k++) j.borders[k] = borders[k];
    j.vert_edge = vert_edge; j.horiz_edge = horiz_edge; j.diag_edge = diag_edge;
    j.eo_class = sao->eo_class[c_idx]; j.band_position = sao->band_position[c_idx];
    for (int k = 0; k < 5; k++) j.offset_val[k] = sao->offset_val[c_idx][k];
    /* region owned by this class (hevcdsp_template.c:270-718), as in hevc_sao_wave */
    const int chroma = c_idx != 0, cw = (8 >> chroma) + 2, ch = (4 >> chroma) + 2;
    int x0 = 0, y0 = 0, w = width, h = height;
    if (CLS & 1) { y0 = -ch; h = ch; } else if (!borders[3]) h -= ch;
    if (CLS & 2) { x0 = -cw; w = cw; } else if (!borders[2]) w -= cw;
    if (w <= 0 || h <= 0) return;
    /* hull of the samples the filter reads: the region, plus one sample around the part that is
     * compared against its neighbours */
    int rx0 = 0, ry0 = 0, rx1 = w, ry1 = h;
    if (EDGE) {
        const int eo = j.eo_class;
        int ix = 0, iy = 0, mw = w, mh = h;
        if (!(CLS & 2) && eo != 1) { if (borders[0]) ix = 1; if (borders[2]) mw--; }
        if (!(CLS & 1) && eo != 0) { if (borders[1]) iy = 1; if (borders[3]) mh--; }
        if (mw > ix && mh > iy) {
            const int hx = eo != 1, hy = eo != 0;
            rx0 = ix - hx < 0 ? ix - hx : 0; ry0 = iy - hy < 0 ? iy - hy : 0;
            rx1 = mw + hx > w ? mw + hx : w; ry1 = mh + hy > h ? mh + hy : h;
        }
    }
    Arena &a = arena();
    Win s = win_pack(a, src + ((y0 + ry0) * stride) + (ptrdiff_t)(x0 + rx0) * px, stride, (rx1 - rx0) * px, ry1 - ry0);
    Win d = win_pack(a, nullptr, 0, w * px, h, 0, 0);
    a.upload();
    LAUNCH1(k_hevc_sao, a, a.d<uint8_t>(d.off), d.pitch / px, -x0, -y0, a.d<const uint8_t>(s.off), s.pitch / px,
            -(x0 + rx0), -(y0 + ry0), j);
    a.download();
    win_unpack(a, d, dst + y0 * stride + (ptrdiff_t)x0 * px, stride, 0, 0, w * px, h);
}
template <int CLS, int BD>
static void sao_band_shim(uint8_t *dst, uint8_t *src, ptrdiff_t stride, SAOParams *sao, int *borders, int width, int height, int c_idx)
{
    sao_run<CLS, BD, 0>(dst, src, stride, sao, borders, width, height, c_idx, 0, 0, 0);
}
template <int CLS, int BD>
static void sao_edge_shim(uint8_t *dst, uint8_t *src, ptrdiff_t stride, SAOParams *sao, int *borders, int width, int height,
                          int c_idx, uint8_t vert_edge, uint8_t horiz_edge, uint8_t diag_edge)
{
    sao_run<CLS, BD, 1>(dst, src, stride, sao, borders, width, height, c_idx, vert_edge, horiz_edge, diag_edge);
}

/* ---- a14: qpel / epel to the 14-bit intermediate ------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_hevc_mc(int16_t *dst, int ds, const uint8_t *win, int ss, int ox, int oy, int width, int height, int mx, int my, int bd, int taps)
{
    __shared__ HevcMcScratch tmp;
    hevc_mc_wave(dst, ds, win + (ptrdiff_t)(oy * ss + ox) * (bd > 8 ? 2 : 1), ss, width, height, mx, my, bd, taps, tmp);
}
/* V, H: which filters this table slot applies ([v][h] index of put_hevc_qpel/epel) */
template <int W, int V, int H, int BD, int TAPS>
static void mc_shim(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride, int height, int mx, int my, int16_t *mcbuffer)
{
    (void)mcbuffer;
    constexpr int px = Px<BD>::bytes, before = TAPS == 8 ? 3 : 1, after = TAPS == 8 ? 4 : 2;
    /* the slot decides which fractions are looked at (hevcdsp_template.c:729-1089); the reads are
     * exactly the taps' support */
    const int fx = H ? mx : 0, fy = V ? my : 0;
    const int bx = H ? before : 0, ax = H ? after : 0, by = V ? before : 0, ay = V ? after : 0;
    Arena &a = arena();
    Win s = win_pack(a, src - by * srcstride - (ptrdiff_t)bx * px, srcstride, (W + bx + ax) * px, height + by + ay);
    Win d = win_pack(a, nullptr, 0, W * 2, height, 0, 0);
    a.upload();
    LAUNCH1(k_hevc_mc, a, a.d<int16_t>(d.off), d.pitch / 2, a.d<const uint8_t>(s.off), s.pitch / px, bx, by, W, height,
            H ? fx : 0, V ? fy : 0, BD, TAPS);
    a.download();
    win_unpack(a, d, reinterpret_cast<uint8_t *>(dst), dststride, 0, 0, W * 2, height);
}

/* ---- a15: (un)weighted prediction ------------------------------------------------------------------ */
__global__ void __launch_bounds__(64)
k_hevc_pred_out(uint8_t *dst, int dt, const int16_t *s1, const int16_t *s2, int ss, int width, int height, HevcPredParams p, int bd)
{
    for (int i = lane_id(); i < width * height; i += 64) {
        const int y = i / width, x = i - y * width;
        stpx(dst, x + y * dt, hevc_pred_px(p, s1[x + y * ss], s2 ? s2[x + y * ss] : 0, bd), bd);
    }
}
template <int BD>
static void pred_out_run(uint8_t *dst, ptrdiff_t dststride, int16_t *s1, int16_t *s2, ptrdiff_t srcstride, int W, int height, HevcPredParams p)
{
    constexpr int px = Px<BD>::bytes;
    Arena &a = arena();
    Win d = win_pack(a, nullptr, 0, W * px, height, 0, 0);
    Win w1 = win_pack(a, reinterpret_cast<uint8_t *>(s1), srcstride, W * 2, height);
    Win w2 = w1;
    if (s2) w2 = win_pack(a, reinterpret_cast<uint8_t *>(s2), srcstride, W * 2, height);
    a.upload();
    LAUNCH1(k_hevc_pred_out, a, a.d<uint8_t>(d.off), d.pitch / px, a.d<const int16_t>(w1.off),
            s2 ? a.d<const int16_t>(w2.off) : (const int16_t *)nullptr, w1.pitch / 2, W, height, p, BD);
    a.download();
    win_unpack(a, d, dst, dststride, 0, 0, W * px, height);
}
template <int W, int BD> static void unweighted_shim(uint8_t *dst, ptrdiff_t ds, int16_t *src, ptrdiff_t ss, int h)
{
    pred_out_run<BD>(dst, ds, src, nullptr, ss, W, h, HevcPredParams{ 0, 0, 0, 0, 0, 0 });
}
template <int W, int BD> static void unweighted_avg_shim(uint8_t *dst, ptrdiff_t ds, int16_t *s1, int16_t *s2, ptrdiff_t ss, int h)
{
    pred_out_run<BD>(dst, ds, s1, s2, ss, W, h, HevcPredParams{ 1, 0, 0, 0, 0, 0 });
}
template <int W, int BD>
static void weighted_shim(uint8_t denom, int16_t wx, int16_t ox, uint8_t *dst, ptrdiff_t ds, int16_t *src, ptrdiff_t ss, int h)
{
    pred_out_run<BD>(dst, ds, src, nullptr, ss, W, h, HevcPredParams{ 2, denom, wx, 0, ox, 0 });
}
template <int W, int BD>
static void weighted_avg_shim(uint8_t denom, int16_t w0, int16_t w1, int16_t o0, int16_t o1, uint8_t *dst, ptrdiff_t ds,
                              int16_t *s1, int16_t *s2, ptrdiff_t ss, int h)
{
    pred_out_run<BD>(dst, ds, s1, s2, ss, W, h, HevcPredParams{ 3, denom, w0, w1, o0, o1 });
}

/* ---- a16: deblocking ------------------------------------------------------------------------------- */
struct LfArgs {
    int tc[2];
    uint8_t no_p[2], no_q[2];
};
__global__ void __launch_bounds__(64) k_hevc_lf(uint8_t *win, int pitch, int across_rows, int luma, int beta, LfArgs g, int bd)
{
    /* the window holds R samples either side of the edge: rows (horizontal edge) or columns */
    const int R = luma ? 4 : 2, px = bd > 8 ? 2 : 1;
    uint8_t *pix = across_rows ? win + (ptrdiff_t)R * pitch * px : win + R * px;
    const int xs = across_rows ? pitch : 1, ys = across_rows ? 1 : pitch;
    if (luma) hevc_lf_luma_wave(pix, xs, ys, beta, g.tc, g.no_p, g.no_q, bd);
    else      hevc_lf_chroma_wave(pix, xs, ys, g.tc, g.no_p, g.no_q, bd);
}
/* HORIZ: the edge is horizontal (the filter runs across rows: hevc_h_loop_filter_*) */
template <int HORIZ, int LUMA, int BD>
static void lf_run(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q)
{
    constexpr int px = Px<BD>::bytes, R = LUMA ? 4 : 2, WR = LUMA ? 3 : 1; /* samples read / written per side */
    LfArgs g;
    for (int k = 0; k < 2; k++) { g.tc[k] = tc[k]; g.no_p[k] = no_p[k]; g.no_q[k] = no_q[k]; }
    Arena &a = arena();
    Win w = HORIZ ? win_pack(a, pix - R * stride, stride, 8 * px, 2 * R) : win_pack(a, pix - R * px, stride, 2 * R * px, 8);
    a.upload();
    LAUNCH1(k_hevc_lf, a, a.d<uint8_t>(w.off), w.pitch / px, HORIZ, LUMA, beta, g, BD);
    a.download();
    if (HORIZ) win_unpack(a, w, pix - WR * stride, stride, 0, R - WR, 8 * px, 2 * WR);
    else       win_unpack(a, w, pix - WR * px, stride, (R - WR) * px, 0, 2 * WR * px, 8);
}
template <int HORIZ, int BD> static void lf_luma_shim(uint8_t *pix, ptrdiff_t stride, int beta, int *tc, uint8_t *no_p, uint8_t *no_q)
{
    lf_run<HORIZ, 1, BD>(pix, stride, beta, tc, no_p, no_q);
}
template <int HORIZ, int BD> static void lf_chroma_shim(uint8_t *pix, ptrdiff_t stride, int *tc, uint8_t *no_p, uint8_t *no_q)
{
    lf_run<HORIZ, 0, BD>(pix, stride, 0, tc, no_p, no_q);
}

/* ---- a18: pure intra predictors ---------------------------------------------------------------------- */
__global__ void __launch_bounds__(64)
k_hevc_pred(uint8_t *dst, int dt, const uint8_t *top, const uint8_t *left, int nedge, int log2, int kind, int c_idx, int mode, int bd)
{
    __shared__ HevcPredScratch s;
    /* top/left hold elements -1 .. nedge-2 */
    for (int i = lane_id(); i < nedge; i += 64) { s.top[i] = (int16_t)ldpx(top, i, bd); s.left[i] = (int16_t)ldpx(left, i, bd); }
    __syncthreads();
    hevc_pred_wave(s, dst, dt, log2, kind, c_idx, mode, bd);
}
/* stride is in SAMPLES for these three entry points (hevcpred_template.c:31, :349-374) */
template <int BD>
static void pred_run(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int kind, int c_idx, int mode)
{
    constexpr int px = Px<BD>::bytes;
    const int size = 1 << log2, nedge = 2 * size + 1;
    Arena &a = arena();
    const size_t t = a.take(nedge * px), l = a.take(nedge * px);
    std::memcpy(a.h<uint8_t>(t), top - px, nedge * px);
    std::memcpy(a.h<uint8_t>(l), left - px, nedge * px);
    Win d = win_pack(a, nullptr, 0, size * px, size, 0, 0);
    a.upload();
    LAUNCH1(k_hevc_pred, a, a.d<uint8_t>(d.off), d.pitch / px, a.d<const uint8_t>(t), a.d<const uint8_t>(l), nedge, log2, kind, c_idx, mode, BD);
    a.download();
    win_unpack(a, d, src, stride * px, 0, 0, size * px, size);
}
template <int LOG2, int BD> static void planar_shim(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st) { pred_run<BD>(s, t, l, st, LOG2, 0, 0, 0); }
template <int BD> static void dc_shim(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int log2, int c_idx) { pred_run<BD>(s, t, l, st, log2, 1, c_idx, 0); }
template <int LOG2, int BD> static void angular_shim(uint8_t *s, const uint8_t *t, const uint8_t *l, ptrdiff_t st, int c_idx, int mode)
{
    pred_run<BD>(s, t, l, st, LOG2, 2, c_idx, mode);
}

/* ---- table fill ---------------------------------------------------------------------------------------- */
template <int BD> static void fill_dsp(HEVCDSPContext *c)
{
    c->put_pcm = put_pcm_shim<BD>;
    c->add_residual[0] = add_residual_shim<4, BD>;   c->add_residual[1] = add_residual_shim<8, BD>;
    c->add_residual[2] = add_residual_shim<16, BD>;  c->add_residual[3] = add_residual_shim<32, BD>;
    c->dequant = dequant_shim<BD>;
    c->transform_4x4_luma = dst4_shim<BD>;
    c->idct[0] = idct_shim<4, BD>;   c->idct[1] = idct_shim<8, BD>;   c->idct[2] = idct_shim<16, BD>;   c->idct[3] = idct_shim<32, BD>;
    c->idct_dc[0] = idct_dc_shim<4, BD>; c->idct_dc[1] = idct_dc_shim<8, BD>; c->idct_dc[2] = idct_dc_shim<16, BD>; c->idct_dc[3] = idct_dc_shim<32, BD>;
    c->sao_band_filter[0] = sao_band_shim<0, BD>; c->sao_band_filter[1] = sao_band_shim<1, BD>;
    c->sao_band_filter[2] = sao_band_shim<2, BD>; c->sao_band_filter[3] = sao_band_shim<3, BD>;
    c->sao_edge_filter[0] = sao_edge_shim<0, BD>; c->sao_edge_filter[1] = sao_edge_shim<1, BD>;
    c->sao_edge_filter[2] = sao_edge_shim<2, BD>; c->sao_edge_filter[3] = sao_edge_shim<3, BD>;
#define MI355_QPEL(i, W)                                                                                       \
    c->put_hevc_qpel[0][0][i] = mc_shim<W, 0, 0, BD, 8>; c->put_hevc_qpel[0][1][i] = mc_shim<W, 0, 1, BD, 8>;   \
    c->put_hevc_qpel[1][0][i] = mc_shim<W, 1, 0, BD, 8>; c->put_hevc_qpel[1][1][i] = mc_shim<W, 1, 1, BD, 8>;   \
    c->put_unweighted_pred[i] = unweighted_shim<W, BD>; c->put_unweighted_pred_avg[i] = unweighted_avg_shim<W, BD>; \
    c->weighted_pred[i] = weighted_shim<W, BD>;         c->weighted_pred_avg[i] = weighted_avg_shim<W, BD>;
#define MI355_EPEL(i, W)                                                                                       \
    c->put_hevc_epel[0][0][i] = mc_shim<W, 0, 0, BD, 4>; c->put_hevc_epel[0][1][i] = mc_shim<W, 0, 1, BD, 4>;   \
    c->put_hevc_epel[1][0][i] = mc_shim<W, 1, 0, BD, 4>; c->put_hevc_epel[1][1][i] = mc_shim<W, 1, 1, BD, 4>;   \
    c->put_unweighted_pred_chroma[i] = unweighted_shim<W, BD>; c->put_unweighted_pred_avg_chroma[i] = unweighted_avg_shim<W, BD>; \
    c->weighted_pred_chroma[i] = weighted_shim<W, BD>;         c->weighted_pred_avg_chroma[i] = weighted_avg_shim<W, BD>;
    MI355_QPEL(0, 4) MI355_QPEL(1, 8) MI355_QPEL(2, 12) MI355_QPEL(3, 16) MI355_QPEL(4, 24) MI355_QPEL(5, 32) MI355_QPEL(6, 48) MI355_QPEL(7, 64)
    MI355_EPEL(0, 2) MI355_EPEL(1, 4) MI355_EPEL(2, 6) MI355_EPEL(3, 8) MI355_EPEL(4, 12) MI355_EPEL(5, 16) MI355_EPEL(6, 24) MI355_EPEL(7, 32)
#undef MI355_QPEL
#undef MI355_EPEL
    c->hevc_h_loop_filter_luma = c->hevc_h_loop_filter_luma_c = lf_luma_shim<1, BD>;
    c->hevc_v_loop_filter_luma = c->hevc_v_loop_filter_luma_c = lf_luma_shim<0, BD>;
    c->hevc_h_loop_filter_chroma = c->hevc_h_loop_filter_chroma_c = lf_chroma_shim<1, BD>;
    c->hevc_v_loop_filter_chroma = c->hevc_v_loop_filter_chroma_c = lf_chroma_shim<0, BD>;
}
template <int BD> static void fill_pred(HEVCPredContext *h)
{
    h->pred_planar[0] = planar_shim<2, BD>; h->pred_planar[1] = planar_shim<3, BD>;
    h->pred_planar[2] = planar_shim<4, BD>; h->pred_planar[3] = planar_shim<5, BD>;
    h->pred_dc = dc_shim<BD>;
    h->pred_angular[0] = angular_shim<2, BD>; h->pred_angular[1] = angular_shim<3, BD>;
    h->pred_angular[2] = angular_shim<4, BD>; h->pred_angular[3] = angular_shim<5, BD>;
}

extern "C" void ff_hevc_dsp_init_mi355x(HEVCDSPContext *c, const int bit_depth)
{
    if (!ready()) { std::fprintf(stderr, "mi355dsp: ff_hevc_dsp_init_mi355x without mi355_init(); no CPU fallback\n"); std::abort(); }
    switch (bit_depth) {
    case 8:  fill_dsp<8>(c); break;
    case 9:  fill_dsp<9>(c); break;
    case 10: fill_dsp<10>(c); break;
    default: break;   /* the reference only instantiates 8/9/10 (hevcdsp.c:238-247) */
    }
}
extern "C" void ff_hevc_pred_init_mi355x(HEVCPredContext *h, int bit_depth)
{
    if (!ready()) { std::fprintf(stderr, "mi355dsp: ff_hevc_pred_init_mi355x without mi355_init(); no CPU fallback\n"); std::abort(); }
    switch (bit_depth) {
    case 8:  fill_pred<8>(h); break;
    case 9:  fill_pred<9>(h); break;
    case 10: fill_pred<10>(h); break;
    default: break;
    }
}
