/*
 * h264_frame.hip — Tier-2: batched H.264 macroblock reconstruction + deblocking
 * (C ABI in include/mi355_h264_frame.h).
 *
 * Three kernels, all "one 64-lane wavefront = one macroblock", everything the wave
 * needs staged in its own LDS:
 *
 *   k_recon_inter   every inter MB of every picture of the batch in one launch:
 *                   MB record + 384 coefficients -> LDS, quarter-pel luma / eighth-pel
 *                   chroma MC from the reference pictures (clamped addressing replaces
 *                   emulated_edge_mc), weighted prediction, inverse transform
 *                   (4x4 via xor-shuffles, 8x8 via LDS), residual add, store to `recon`.
 *   k_recon_intra   intra MBs, one launch per dependency level (an intra MB needs the
 *                   unfiltered samples of its left/top-left/top/top-right neighbours;
 *                   levels come from mi355_h264_intra_schedule()).
 *   k_deblock       one launch per anti-diagonal d = x + 2y (the reference's raster
 *                   filter order only requires left, top and top-right to be done):
 *                   bS from the MB records, all 8 luma + 4 chroma edges in LDS, then
 *                   the MB plus the 3 neighbour columns/rows it changed go to `dst`.
 *
 * Reference behaviour restated: hl_decode_mb (h264_mb_template.c:41-257), hl_motion
 * (h264_mc_template.c:64-163), mc_part_* / mc_dir_part (h264_mb.c:204-471),
 * hl_decode_mb_predict_luma / _idct_luma (h264_mb.c:612-795), ff_h264_filter_mb +
 * filter_mb_dir + check_mv (h264_loopfilter.c:442-847), fill_filter_caches
 * (h264_slice.c:2056-2196).
 */
#include <cstdlib>
#include <vector>
#include "h264_recon_dev.h"
#include "../../include/mi355dsp.h"      /* the device error word */


namespace {

__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd)
{
    __shared__ MbLds s;
    recon_inter_wave<false>(s, frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}
__attribute__((amdgpu_waves_per_eu(MI355_RECON_WAVES, MI355_RECON_WAVES)))
__global__ void __launch_bounds__(64)
k_recon_inter_sparse(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd)
{
    __shared__ MbLds s;
    recon_inter_wave<true>(s, frames, max_w, max_h, inv_w, inv_h, nblocks, per_xcd);
}

/* ------------------------------------------------------------------------- */
/* intra                                                                        */
/* ------------------------------------------------------------------------- */
constexpr int TP = 32;   /* luma tile pitch: columns -1..23 (8x8 blocks read 16 samples of the row above) */
constexpr int CP = 16;   /* chroma tile pitch: columns -1..7 */
constexpr int TO = 4;    /* column 0 sits on a dword (column -1 at TO - 1): residual add and store move whole dwords */
struct IntraLds {
    MbCore mb;
    __attribute__((aligned(16))) uint8_t tile[17 * TP];
    __attribute__((aligned(16))) uint8_t ctile[2][9 * CP];
    PredScratch ps;
    int dc_t[16];                      /* Intra16x16: the luma DC levels between the two butterfly passes */
};
#define TILE(x, y) s.tile[((y) + 1) * TP + (x) + TO]
#define CTILE(p, x, y) s.ctile[p][((y) + 1) * CP + (x) + TO]

/* AGENT (k_recon_intra_all, a macroblock with intra neighbours of the SAME launch): the neighbours' samples were written by other waves — perhaps of another
 * XCD — a moment ago: they are read, and this macroblock's samples written, at agent scope (past the L2 of the wave's own XCD; the writer's stores
 * write-through), the recipe of k_deblock_tiled's hand-down */
__device__ __forceinline__ uint8_t intra_edge_byte(const uint8_t *p, bool agent)
{
    if (!agent) return *p;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (uint8_t)(agent_load_u32(reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3)) >> (8 * (a & 3)));
}
template <bool AGENT>
__device__ __forceinline__ void intra_store_dword(uint8_t *p, uint32_t v)
{
    if (AGENT) agent_store_u32(reinterpret_cast<uint32_t *>(p), v);
    else *reinterpret_cast<uint32_t *>(p) = v;
}
/* store_mb / store_mb_pitched_tiled (h264_recon_dev.h) with the scope as a parameter */
template <bool TILED, bool AGENT>
__device__ __forceinline__ void intra_store_mb(const uint8_t *y, int ypitch, const uint8_t *cb, const uint8_t *cr, int cpitch, const FrameHot &fr, int mb_x, int mb_y)
{
    if (!AGENT) {
        if (TILED) store_mb_pitched_tiled<true>(y, ypitch, cb, cr, cpitch, fr, mb_x, mb_y);
        else store_mb<true>(y, ypitch, cb, cr, cpitch, fr, mb_x, mb_y);
        return;
    }
    const int lane = lane_id();
    const int row = lane >> 2, seg = lane & 3;
    const int plane = (lane >> 4) & 1, crow = (lane >> 1) & 7, cseg = lane & 1;
    const uint32_t vy = tile_dword<true>(y + row * ypitch + 4 * seg), vc = tile_dword<true>((plane ? cr : cb) + crow * cpitch + 4 * cseg);
    if (TILED) {
        intra_store_dword<true>(fr.recon[0] + tile_y_off(mb_x, mb_y, fr.recon_stride[0]) + 4 * lane, vy);
        if (lane < 32) intra_store_dword<true>(fr.recon[1] + tile_c_off(mb_x, mb_y, fr.recon_stride[1]) + 4 * lane, vc);
    } else {
        intra_store_dword<true>(fr.recon[0] + (uint32_t)(__mul24(mb_y * 16 + row, fr.recon_stride[0]) + mb_x * 16 + 4 * seg), vy);
        if (lane < 32) intra_store_dword<true>((plane ? fr.recon[2] : fr.recon[1]) + (uint32_t)(__mul24(mb_y * 8 + crow, fr.recon_stride[1]) + mb_x * 8 + 4 * cseg), vc);
    }
}

template <bool TILED, bool AGENT = false>
__device__ __forceinline__ void recon_intra_mb(IntraLds &s, const mi355_h264_frame &frd, int mb_xy, bool agent_edges = false)
{
    const int lane = lane_id();
    const FrameHot fr = frame_hot(frd);
    const int mb_x = mb_xy % fr.mb_width, mb_y = mb_xy / fr.mb_width;
    /* Everything this macroblock reads from memory is requested at once — record, vectors, coefficients and the edge
     * samples of the unfiltered neighbours (which depend on the position only) — with loads that no lane skips: a lane
     * without a sample to fetch reads the block's own first sample and drops it.  One memory round trip per wave. */
    const int ys = fr.recon_stride[0], cs = fr.recon_stride[1];
    uint8_t *const ry = fr.recon[0] + (size_t)mb_y * 16 * ys + mb_x * 16;
    const int pic_w = 16 * fr.mb_width;
    const bool top_y = mb_y > 0 && lane < 25 && mb_x * 16 + lane - 1 >= 0 && mb_x * 16 + lane - 1 < pic_w;
    const bool left_y = mb_x > 0 && lane >= 32 && lane < 48;
    const bool top_c = mb_y > 0 && lane < 9 && (mb_x > 0 || lane > 0);
    const bool left_c = mb_x > 0 && lane >= 16 && lane < 24;
    const uint8_t *const rcb = fr.recon[1] + (size_t)mb_y * 8 * cs + mb_x * 8;
    const uint8_t *const rcr = fr.recon[2] + (size_t)mb_y * 8 * cs + mb_x * 8;
    const ptrdiff_t oy = top_y ? (ptrdiff_t)(lane - 1) - ys : (left_y ? (ptrdiff_t)(lane - 32) * ys - 1 : 0);
    const ptrdiff_t oc = top_c ? (ptrdiff_t)(lane - 1) - cs : (left_c ? (ptrdiff_t)(lane - 16) * cs - 1 : 0);
    MbLoad ld;
    load_mb_issue(ld, fr, mb_xy, true, true);
    uint8_t e_y, e_cb, e_cr;
    if (TILED) {
        /* the same samples in tiles: the row above = last row of the tiles above (left, own, right), the column to the left = last
         * column of the left tile; Cb and Cr sit 64 bytes apart in one chroma tile */
        const int xt = mb_x * 16 + lane - 1, xtc = mb_x * 8 + lane - 1;
        const uint32_t ty = top_y ? tile_y_off(xt >> 4, mb_y - 1, ys) + 15 * 16 + (xt & 15)
                                  : (left_y ? tile_y_off(mb_x - 1, mb_y, ys) + (lane - 32) * 16 + 15 : tile_y_off(mb_x, mb_y, ys));
        const uint32_t tc = top_c ? tile_c_off(xtc >> 3, mb_y - 1, cs) + 7 * 8 + (xtc & 7)
                                  : (left_c ? tile_c_off(mb_x - 1, mb_y, cs) + (lane - 16) * 8 + 7 : tile_c_off(mb_x, mb_y, cs));
        e_y = intra_edge_byte(fr.recon[0] + ty, AGENT && agent_edges); e_cb = intra_edge_byte(fr.recon[1] + tc, AGENT && agent_edges);
        e_cr = intra_edge_byte(fr.recon[1] + tc + 64, AGENT && agent_edges);
    } else {
        e_y = intra_edge_byte(ry + oy, AGENT && agent_edges); e_cb = intra_edge_byte(rcb + oc, AGENT && agent_edges); e_cr = intra_edge_byte(rcr + oc, AGENT && agent_edges);
    }
    MI355_ISSUE_FENCE();
    load_mb_commit(s.mb, ld, true, fr.mv[0] != nullptr, fr.mv[1] != nullptr);
    const mi355_h264_mb &h = s.mb.hdr;
    const uint32_t t = h.mb_type;

    if (t & MI355_MB_INTRA_PCM) {        /* h264_mb_template.c:139-153 */
        const uint8_t *src = reinterpret_cast<const uint8_t *>(s.mb.coef);
        intra_store_mb<TILED, AGENT>(src, 16, src + 256, src + 320, 8, fr, mb_x, mb_y);
        return;
    }
    /* edge samples -> tiles */
    if (top_y) TILE(lane - 1, -1) = e_y;
    if (left_y) TILE(-1, lane - 32) = e_y;
    if (top_c) { CTILE(0, lane - 1, -1) = e_cb; CTILE(1, lane - 1, -1) = e_cr; }
    if (left_c) { CTILE(0, -1, lane - 16) = e_cb; CTILE(1, -1, lane - 16) = e_cr; }
    MI355_WAVE_SYNC();

    /* chroma prediction: hpc.pred8x8[chroma_pred_mode], h264_mb_template.c:161-164 */
    for (int p = 0; p < 2; p++) {
        if (lane < 9) s.ps.T[lane] = CTILE(p, lane - 1, -1);
        if (lane >= 16 && lane < 25) s.ps.L[lane - 16] = CTILE(p, -1, lane - 17);
        MI355_WAVE_SYNC();
        chroma8_pred_fast(s.ps, h.chroma_pred_mode, &CTILE(p, 0, 0), CP);
    }

    if (t & MI355_MB_INTRA16x16) {       /* h264_mb.c:701-722 */
        if (lane < 17) s.ps.T[lane] = TILE(lane - 1, -1);
        if (lane >= 32 && lane < 49) s.ps.L[lane - 32] = TILE(-1, lane - 33);
        MI355_WAVE_SYNC();
        intra16_pred_fast(s.ps, h.intra16x16_pred_mode, &TILE(0, 0), TP);
        if ((h.nnz_mask >> MI355_NNZ_LUMA_DC) & 1) {
            /* ff_h264_luma_dc_dequant_idct (h264idct_template.c:242-271) on sixteen lanes: lane 4a + b holds level b of
             * row a; the row butterflies run inside the quads (DPP), the column butterflies after a turn through LDS, and
             * lane 4i + j produces output j of column i — the arithmetic of luma_dc_dequant() (h264_dev.h), which one lane
             * alone needs ~140 instructions for */
            const int b4 = lane & 3;
            const int v = s.mb.coef[luma_dc_slot(lane & 15)];
            const int x1 = quad_xor1(v);
            const int w = (b4 & 1) ? x1 - v : v + x1;               /* b: 0 s, 1 d, 2 u, 3 e of the row */
            const int y = quad_xor2(w);
            const int tv = b4 == 2 ? y - w : (b4 == 1 ? w - y : w + y);   /* s + u, d - e, s - u, d + e */
            if (lane < 16) s.dc_t[(lane & ~3) | (b4 == 1 ? 2 : (b4 == 2 ? 1 : b4))] = tv;
            MI355_WAVE_SYNC();
            if (lane < 16) {
                const int ci = lane >> 2;
                const int t0 = s.dc_t[ci], t1 = s.dc_t[4 + ci], t2 = s.dc_t[8 + ci], t3 = s.dc_t[12 + ci];
                const int ss = t0 + t2, dd = t0 - t2, ee = t1 - t3, uu = t1 + t3;
                const int r = b4 == 0 ? ss + uu : (b4 == 1 ? dd + ee : (b4 == 2 ? dd - ee : ss - uu));
                s.mb.coef[luma_dc_slot(lane)] = (int16_t)((r * (int)h.dc_qmul[0] + 128) >> 8);
            }
            MI355_WAVE_SYNC();
        }
        /* luma (with the DC-only rule) and chroma residual in one pass, two lanes per block */
        residual_tile(s.mb, &TILE(0, 0), TP, &CTILE(0, 0, 0), &CTILE(1, 0, 0), CP, true);
    } else if (t & MI355_MB_8x8DCT) {    /* Intra 8x8: h264_mb.c:626-656 */
        for (int i8 = 0; i8 < 4; i8++) {
            const int x0 = 8 * (i8 & 1), y0 = 8 * (i8 >> 1), i = 4 * i8;
            if (lane < 17) s.ps.T[lane] = TILE(x0 + lane - 1, y0 - 1);
            if (lane >= 32 && lane < 41) s.ps.L[lane - 32] = TILE(x0 - 1, y0 + lane - 33);
            MI355_WAVE_SYNC();
            intra_pred_wave(s.ps, 1, h.u.intra4x4_pred_mode[i], (h.topleft_samples_available << i) & 0x8000,
                            (h.topright_samples_available << i) & 0x4000, &TILE(x0, y0), TP);
            int r[8];
            idct8_lds(s.mb.coef + i8 * 64, lane & 7, lane < 8, r);
            if (lane < 8 && ((h.nnz_mask >> i) & 1)) add_col(&TILE(x0 + lane, y0), TP, r, 8);
            MI355_WAVE_SYNC();
        }
    } else {                              /* Intra 4x4: h264_mb.c:657-700 */
        for (int i = 0; i < 16; i++) {
            const int x0 = 4 * blk_x4(i), y0 = 4 * blk_y4(i);
            const int tr_ok = (h.topright_samples_available << i) & 0x8000;
            if (lane < 5) s.ps.T[lane] = TILE(x0 + lane - 1, y0 - 1);
            else if (lane < 9) s.ps.T[lane] = tr_ok ? TILE(x0 + lane - 1, y0 - 1) : TILE(x0 + 3, y0 - 1);
            if (lane >= 32 && lane < 37) s.ps.L[lane - 32] = TILE(x0 - 1, y0 + lane - 33);
            MI355_WAVE_SYNC();
            intra_pred_wave(s.ps, 0, h.u.intra4x4_pred_mode[i], 0, 0, &TILE(x0, y0), TP);
            const int q = lane & 3;
            int c[4], r[4], row;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) c[k2] = s.mb.coef[i * 16 + q + 4 * k2];
            idct4_quad(c, q, r, row);
            if (lane < 4 && ((h.nnz_mask >> i) & 1)) add_row4<true>(&TILE(x0, y0 + row), r);
            MI355_WAVE_SYNC();
        }
    }
    if (!(t & MI355_MB_INTRA16x16)) residual_chroma<true>(s.mb, &CTILE(0, 0, 0), &CTILE(1, 0, 0), CP);
    intra_store_mb<TILED, AGENT>(&TILE(0, 0), TP, &CTILE(0, 0, 0), &CTILE(1, 0, 0), CP, fr, mb_x, mb_y);
}
#undef TILE

__global__ void __launch_bounds__(64)
k_recon_intra(const mi355_h264_frame *frames, int level, int width)
{
    __shared__ IntraLds s;
    const int f = blockIdx.x / width, k = blockIdx.x - f * width;
    const mi355_h264_frame &frd = frames[f];
    if (level > frd.max_intra_level) return;
    const int first = mi355_global(frd.intra_level_start)[level - 1], count = mi355_global(frd.intra_level_start)[level] - first;
    if (k >= count) return;
    const int mb_xy = (int)mi355_global(frd.intra_list)[first + k];
    if (uniform(frd.surface_layout) == MI355_SURFACE_TILED) recon_intra_mb<true>(s, frd, mb_xy);
    else recon_intra_mb<false>(s, frd, mb_xy);
}

/* Every level of every picture in ONE launch.  Workgroup b takes entry k = b / nframes of the intra list of picture b % nframes.  The list is in level order
 * and a macroblock's level is one more than the highest level among its left, above-left, above and above-right neighbours (mi355_h264_intra_schedule), so the
 * intra macroblocks among those four are earlier entries: the wave reads their type words, waits until each of them has set its byte in `flags` (one byte per
 * macroblock of the launch's grid, zeroed by the host) — set behind write-through stores of the samples, which this wave then reads at agent scope — and sets
 * its own when its stores have reached memory.  No counter anybody shares: two atomics per macroblock on a picture's pair of counters (the first form) took
 * 0.3 - 0.6 us EACH, one after the other.  A launch per level made all pictures wait for the slowest wave of a level, 254 times for an I picture; here a
 * macroblock goes as soon as its own neighbours are done.
 * Progress: workgroups start in the order of their numbers on every XCD, so the lowest-numbered unfinished workgroup is always running, and it waits for
 * nobody (its neighbours have lower numbers).  That order is what devices do, not something the API promises: should one ever start them otherwise, INTRA_NAPS_MAX ends
 * the wait — not a hung device, and not a silently wrong picture either: MI355_ERR_WAIT_EXPIRED is set in the device's error word and the next mi355_sync /
 * mi355_event_sync / mi355_h264_pipelines_sync returns MI355_E_DEVICE_FAULT. */
/* Batches with at least this many levels take the single launch (MI355_INTRA_SINGLE=0 / 1 pins a form).  Measured (profiles/r05s_pass_ms.txt): I pictures 4.17 -> 3.76 ms
 * per 512, 1.58 -> 1.36 per 64 (254 levels); P pictures with 5 % intra macroblocks in four levels 0.68 -> 0.81 ms per 2048 — their level launches are short as they are, and
 * the single launch adds the neighbours' type words (a memory round trip before the macroblock's own loads), the wait for the stores and the zeroing of the flags */
constexpr int INTRA_SINGLE_LEVELS = 16;
constexpr uint32_t INTRA_NAPS_MAX = 1u << 21;          /* ~1 s of naps */
#ifdef MI355_HIP_EMU_H
static inline uint32_t intra_flag_word(const uint8_t *p) { uint32_t v; std::memcpy(&v, reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3), 4); return v; }
static inline void intra_flag_set(uint8_t *p) { *p = 1; }
#else
__device__ __forceinline__ uint32_t intra_flag_word(const uint8_t *p) { return agent_load_u32(reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3)); }
__device__ __forceinline__ void intra_flag_set(uint8_t *p) { __hip_atomic_store(p, (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__global__ void __launch_bounds__(64)
k_recon_intra_all(const mi355_h264_frame *__restrict__ frames, int nframes, int nmb_max, uint8_t *__restrict__ flags, uint32_t *error_word, uint32_t naps_max)
{
    __shared__ IntraLds s;
    const int k = (int)(blockIdx.x / (unsigned)nframes), f = (int)(blockIdx.x - (unsigned)k * (unsigned)nframes);
    const mi355_h264_frame &frd = frames[f];
    const int nlev = uniform(frd.max_intra_level);
    if (nlev <= 0) return;
    if (k >= uniform(mi355_global(frd.intra_level_start)[nlev])) return;
    const int mb_xy = (int)uniform((int)mi355_global(frd.intra_list)[k]);
    const int W = uniform(frd.mb_width), mb_y = mb_xy / W, mb_x = mb_xy - mb_y * W;
    uint8_t *const pic_flags = mi355_global(flags) + (size_t)f * (size_t)nmb_max;
    const int lane = lane_id();
    /* lane i < 4: neighbour i (left, above-left, above, above-right) */
    const int nx = mb_x + (lane == 3 ? 1 : (lane == 2 ? 0 : -1)), ny = mb_y - (lane == 0 ? 0 : 1);
    const bool inside = lane < 4 && nx >= 0 && nx < W && ny >= 0;
    const int nxy = inside ? ny * W + nx : mb_xy;
    const bool dep = inside && (mi355_global_v(reinterpret_cast<const uint32_t *>(mi355_global(frd.mb) + nxy))[0] & MI355_MB_INTRA) != 0;
    static_assert(offsetof(mi355_h264_mb, mb_type) == 0, "the type word of a record");
    const bool waits = __any(dep);
    if (waits) {
        const uint8_t *fp = pic_flags + nxy;
        const int sh = 8 * (int)(reinterpret_cast<uintptr_t>(fp) & 3);
        uint32_t naps = 0;
        while (__any(dep && ((intra_flag_word(fp) >> sh) & 0xFFu) == 0) && naps < naps_max) { wave_nap(); naps++; }
        /* the wait ran out (the device started workgroups in another order than the progress argument above assumes): the macroblock is reconstructed from whatever
         * is there — and the device says so: the wait that follows this launch returns MI355_E_DEVICE_FAULT (include/mi355dsp.h), the caller repeats the batch with
         * mi355_h264_recon_intra_levels_dev (MI355_INTRA_SINGLE=0 pins that form) */
        if (naps >= naps_max && __any(dep && ((intra_flag_word(fp) >> sh) & 0xFFu) == 0) && lane_id() == 0) atomicOr(error_word, (uint32_t)MI355_ERR_WAIT_EXPIRED);
        MI355_ISSUE_FENCE();
    }
    if (uniform(frd.surface_layout) == MI355_SURFACE_TILED) recon_intra_mb<true, true>(s, frd, mb_xy, waits);
    else recon_intra_mb<false, true>(s, frd, mb_xy, waits);
    agent_drain_stores();                                 /* the macroblock's samples have reached memory */
    if (lane == 0 && naps_max) intra_flag_set(pic_flags + mb_xy);       /* (bound 0, the test hook: nobody reports, every wait on a neighbour runs out) */
}

}  // namespace

/* ------------------------------------------------------------------------- */
/* host entry points                                                            */
/* ------------------------------------------------------------------------- */
static int recon_inter_launch(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, bool sparse, void *stream, int layouts = MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED)
{
    if (!mi355::bind() || !d_frames || nframes <= 0) return -1;
    /* div_magic is exact below 2^24 work items: larger batches go out as several launches */
    const int per_frame = max_mb_width * max_mb_height;
    if (per_frame <= 0 || per_frame >= (1 << 24)) return -3;
    const int frames_per_launch = ((1 << 24) - 1) / per_frame;
    const unsigned long long one = 1ull << 40;
    for (int f0 = 0; f0 < nframes; f0 += frames_per_launch) {
        const int nf = nframes - f0 < frames_per_launch ? nframes - f0 : frames_per_launch;
        const int nblocks = nf * per_frame, per_xcd = (nblocks + 7) / 8;
        const unsigned long long iw = (one + max_mb_width - 1) / max_mb_width, ih = (one + max_mb_height - 1) / max_mb_height;
        if (sparse)
            hipLaunchKernelGGL(k_recon_inter_sparse, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, (hipStream_t)stream,
                               d_frames + f0, max_mb_width, max_mb_height, iw, ih, nblocks, per_xcd);
        else if (layouts == MI355_LAYOUTS_TILED) {
            if (!mi355::recon_inter_tiled_launch(d_frames + f0, nf, max_mb_width, max_mb_height, (hipStream_t)stream)) return -4;
        }
        else
            hipLaunchKernelGGL(k_recon_inter, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, (hipStream_t)stream,
                               d_frames + f0, max_mb_width, max_mb_height, iw, ih, nblocks, per_xcd);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_h264_recon_inter_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream)
{
    return recon_inter_launch(d_frames, nframes, max_mb_width, max_mb_height, false, stream);
}
extern "C" int mi355_h264_recon_inter_layouts_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int layouts, void *stream)
{
    if (!(layouts & (MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED))) return -1;
    return recon_inter_launch(d_frames, nframes, max_mb_width, max_mb_height, false, stream, layouts);
}
extern "C" int mi355_h264_recon_inter_sparse_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream)
{
    return recon_inter_launch(d_frames, nframes, max_mb_width, max_mb_height, true, stream);
}

extern "C" int mi355_h264_recon_intra_dev(const mi355_h264_frame *d_frames, int nframes, int max_intra_level, int max_level_width, void *stream)
{
    if (!mi355::bind() || !d_frames || nframes <= 0) return -1;
    if (max_intra_level <= 0 || max_level_width <= 0) return 0;
    for (int level = 1; level <= max_intra_level; level++)
        hipLaunchKernelGGL(k_recon_intra, dim3((unsigned)(nframes * max_level_width)), dim3(64), 0, (hipStream_t)stream,
                           d_frames, level, max_level_width);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

/* the intra pass of a batch whose grid the caller knows (max_mb_width x max_mb_height macroblocks per picture at most): ONE launch, k_recon_intra_all */
extern "C" int mi355_h264_recon_intra_all_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int max_intra_level, const int32_t *level_widths, void *stream)
{
    if (!mi355::bind() || !d_frames || nframes <= 0 || max_mb_width <= 0 || max_mb_height <= 0 || (max_intra_level > 0 && !level_widths)) return -1;
    static const int single = std::getenv("MI355_INTRA_SINGLE") ? std::atoi(std::getenv("MI355_INTRA_SINGLE")) : -1;     /* 0 / 1 pins the form */
    if (single == 0 || (single != 1 && max_intra_level < INTRA_SINGLE_LEVELS)) return mi355_h264_recon_intra_levels_dev(d_frames, nframes, max_intra_level, level_widths, stream);
    long long per_picture = 0;                           /* a picture has at most the sum of the level widths intra macroblocks (the widths are maxima over the batch) */
    for (int level = 1; level <= max_intra_level; level++) per_picture += level_widths[level - 1] > 0 ? level_widths[level - 1] : 0;
    if (per_picture <= 0) return 0;
    const long long nmb = (long long)max_mb_width * max_mb_height;
    if (per_picture > nmb) per_picture = nmb;
    if ((long long)nframes * per_picture > 0x7FFFFFFFLL || (long long)nframes * nmb > 0x7FFFFFFFLL) return -3;
    const size_t bytes = ((size_t)nframes * (size_t)nmb + 3) & ~(size_t)3;
    uint32_t *flags = mi355::sync_words((hipStream_t)stream, bytes / 4);
    if (!flags) return -4;
    MI355_TRY(hipMemsetAsync(flags, 0, bytes, (hipStream_t)stream), -4);
    uint32_t *err = mi355::error_word();
    if (!err) return -4;
    /* test hook: MI355_INTRA_NAPS_MAX=0 — no macroblock reports completion and no wait lasts: every macroblock with an intra neighbour gives up (tests/test_frame_emu.py: the error channel) */
    static const uint32_t naps_max = std::getenv("MI355_INTRA_NAPS_MAX") ? (uint32_t)std::strtoul(std::getenv("MI355_INTRA_NAPS_MAX"), nullptr, 0) : INTRA_NAPS_MAX;
    hipLaunchKernelGGL(k_recon_intra_all, dim3((unsigned)(nframes * per_picture)), dim3(64), 0, (hipStream_t)stream, d_frames, nframes, (int)nmb, reinterpret_cast<uint8_t *>(flags), err, naps_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_h264_recon_intra_levels_dev(const mi355_h264_frame *d_frames, int nframes, int max_intra_level, const int32_t *level_widths, void *stream)
{
    if (!mi355::bind() || !d_frames || nframes <= 0 || (max_intra_level > 0 && !level_widths)) return -1;
    /* one launch per level.  Measured and not kept (round 3, tools/exp_intra.py): ONE launch with a workgroup of eight waves per
     * picture walking the picture's levels (a barrier instead of a launch per level) — all-intra 512 pictures 6.2 ms against 5.5,
     * 64 pictures 5.0 against 1.9, P pictures 1.1 against 0.9: a level of an I picture is up to 60 macroblocks wide, a workgroup
     * works through it in rounds, and the picture's 254 levels become a serial chain of ~20 us each whatever the batch, while a
     * launch runs a level of ALL pictures side by side for ~7.5 us */
    for (int level = 1; level <= max_intra_level; level++) {
        const int width = level_widths[level - 1];
        if (width <= 0) continue;
        if ((long long)nframes * width > 0x7FFFFFFFLL) return -3;
        hipLaunchKernelGGL(k_recon_intra, dim3((unsigned)(nframes * width)), dim3(64), 0, (hipStream_t)stream, d_frames, level, width);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#ifdef MI355_PROF
extern "C" void mi355_debug_rprof(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {};
    MI355_CHECK(hipDeviceSynchronize());
    MI355_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rprof), sizeof(z)));
    if (reset) MI355_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_rprof), z, sizeof(z)));
}
#endif

extern "C" int mi355_h264_decode_frames_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                            int max_intra_level, int max_level_width, void *stream)
{
    int rc = mi355_h264_recon_inter_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
    if (rc) return rc;
    rc = mi355_h264_recon_intra_dev(d_frames, nframes, max_intra_level, max_level_width, stream);
    if (rc) return rc;
    return mi355_h264_deblock_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
}

extern "C" int mi355_h264_decode_frames_layouts_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                                    int max_intra_level, const int32_t *level_widths, int layouts, void *stream)
{
    int rc = mi355_h264_recon_inter_layouts_dev(d_frames, nframes, max_mb_width, max_mb_height, layouts, stream);
    if (rc) return rc;
    rc = mi355_h264_recon_intra_all_dev(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, stream);
    if (rc) return rc;
    return mi355_h264_deblock_layouts_dev(d_frames, nframes, max_mb_width, max_mb_height, layouts, stream);
}

extern "C" int mi355_h264_decode_frames_levels_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                                   int max_intra_level, const int32_t *level_widths, void *stream)
{
    int rc = mi355_h264_recon_inter_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
    if (rc) return rc;
    rc = mi355_h264_recon_intra_all_dev(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, stream);
    if (rc) return rc;
    return mi355_h264_deblock_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
}

/* Descriptors of the host-array entry point travel through a per-thread pinned staging buffer and a per-thread device
 * buffer, both grow-only: no allocation, no synchronisation and no pageable copy per call.  The event marks the last
 * copy out of the staging buffer, so that the next call does not overwrite descriptors still being read. */
namespace {
struct DescStage {
    mi355_h264_frame *host = nullptr, *dev = nullptr;
    size_t cap = 0;
    hipEvent_t copied = nullptr;
    int device = -1;                 /* the device `dev` and `copied` live on: a thread may move between devices (mi355_set_device) */
};
/* one staging set per device this thread has used */
DescStage &desc_stage(int device)
{
    static thread_local std::vector<DescStage> sets;
    for (DescStage &d : sets) if (d.device == device) return d;
    sets.emplace_back();
    sets.back().device = device;
    return sets.back();
}
}
extern "C" int mi355_h264_decode_frames(const mi355_h264_frame *frames, int nframes, void *stream)
{
    if (!mi355::bind() || !frames || nframes <= 0) return -1;
    /* the calling thread's device decides where the descriptors go: its own buffers and event per device (one set for the
     * whole thread recorded an event of GPU 0 on a stream of GPU 1 after mi355_set_device) */
    DescStage &st = desc_stage(mi355::current_device());
    int mw = 0, mh = 0, ml = 0, lw = 0;
    for (int i = 0; i < nframes; i++) {
        if (frames[i].mb_width > mw) mw = frames[i].mb_width;
        if (frames[i].mb_height > mh) mh = frames[i].mb_height;
        if (frames[i].max_intra_level > ml) ml = frames[i].max_intra_level;
        if (frames[i].max_level_width > lw) lw = frames[i].max_level_width;
        /* tiled surfaces hold frame pictures only (a field would be every other row of every tile) */
        if (frames[i].surface_layout != MI355_SURFACE_LINEAR && (frames[i].surface_layout != MI355_SURFACE_TILED || frames[i].field_picture)) return -1;
    }
    if (!st.copied) MI355_TRY(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming), -4);
    else MI355_TRY(hipEventSynchronize(st.copied), -4);
    if ((size_t)nframes > st.cap) {
        size_t ncap = st.cap ? st.cap : 64;
        while (ncap < (size_t)nframes) ncap *= 2;
        if (st.host) (void)hipHostFree(st.host);
        if (st.dev) { MI355_TRY(hipDeviceSynchronize(), -4); (void)hipFree(st.dev); }     /* kernels of earlier calls may still read it */
        st.host = st.dev = nullptr; st.cap = 0;
        MI355_TRY(hipHostMalloc(reinterpret_cast<void **>(&st.host), sizeof(*st.host) * ncap), -4);
        MI355_TRY(hipMalloc(reinterpret_cast<void **>(&st.dev), sizeof(*st.dev) * ncap), -4);
        st.cap = ncap;
    }
    std::memcpy(st.host, frames, sizeof(*frames) * (size_t)nframes);
    MI355_TRY(hipMemcpyAsync(st.dev, st.host, sizeof(*frames) * (size_t)nframes, hipMemcpyHostToDevice, (hipStream_t)stream), -4);
    MI355_TRY(hipEventRecord(st.copied, (hipStream_t)stream), -4);
    /* a picture whose descriptor does not say how wide its widest intra level is (0) gets the safe bound: every
     * macroblock of a level (an anti-diagonal cannot hold more than the smaller picture dimension plus one) */
    if (ml > 0 && lw <= 0) lw = (mw < mh ? mw : mh) + 1;
    return mi355_h264_decode_frames_dev(st.dev, nframes, mw, mh, ml, lw, stream);
}

static int intra_schedule(mi355_h264_mb *mb, int mb_width, int mb_height, uint32_t *list, int32_t *level_start, int *max_level_width, bool pairs);
extern "C" int mi355_h264_intra_schedule(mi355_h264_mb *mb, int mb_width, int mb_height,
                                         uint32_t *list, int32_t *level_start, int *max_level_width)
{
    return intra_schedule(mb, mb_width, mb_height, list, level_start, max_level_width, false);
}
extern "C" int mi355_h264_intra_schedule_mbaff(mi355_h264_mb *mb, int mb_width, int mb_height,
                                               uint32_t *list, int32_t *level_start, int *max_level_width)
{
    if (mb_height & 1) return -1;
    return intra_schedule(mb, mb_width, mb_height, list, level_start, max_level_width, true);
}
static int intra_schedule(mi355_h264_mb *mb, int mb_width, int mb_height, uint32_t *list, int32_t *level_start, int *max_level_width, bool pairs)
{
    /* Levels are kept in a local int array: an all-intra picture reaches level mb_width + 2 * (mb_height - 1)
     * (254 at 1920x1088, 508 at 3840x2160), which the record's 8-bit field cannot hold.  The device reads only
     * `list` / `level_start`; mb[].intra_level receives the level saturated at 255 (informational). */
    const int nmb = mb_width * mb_height;
    if (!mb || !list || !level_start || nmb <= 0) return -1;
    std::vector<int32_t> level((size_t)nmb, 0), count;
    int maxl = 0;
    /* visiting order = decoding order: raster, or pair by pair (top, bottom) along a pair row */
    for (int i = 0; i < nmb; i++) {
            const int y = pairs ? 2 * (i / (2 * mb_width)) + (i & 1) : i / mb_width, x = pairs ? (i % (2 * mb_width)) >> 1 : i % mb_width;
            const int xy = x + y * mb_width;
            mi355_h264_mb &m = mb[xy];
            if (!(m.mb_type & MI355_MB_INTRA)) { m.intra_level = 0; continue; }
            int lv = 0;
            const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
            if (pairs) {
                /* macroblock PAIRS (MBAFF): which macroblock of a neighbouring pair holds a neighbouring sample depends on both pairs' frame / field
                 * coding (6.4.12.2) — both macroblocks of the left, above-left, above and above-right pairs count, and the bottom macroblock of a
                 * pair waits for the top one */
                const int pr = y >> 1;
                for (int k = 0; k < 4; k++)
                    for (int pos = 0; pos < 2; pos++) {
                        const int nx = x + dx[k], ny = 2 * (pr + dy[k]) + pos;
                        if (nx >= 0 && nx < mb_width && ny >= 0 && ny < mb_height && level[(size_t)(nx + ny * mb_width)] > lv)
                            lv = level[(size_t)(nx + ny * mb_width)];
                    }
                if ((y & 1) && level[(size_t)(xy - mb_width)] > lv) lv = level[(size_t)(xy - mb_width)];
            } else
            for (int k = 0; k < 4; k++) {
                const int nx = x + dx[k], ny = y + dy[k];
                if (nx >= 0 && nx < mb_width && ny >= 0 && ny < mb_height && level[(size_t)(nx + ny * mb_width)] > lv)
                    lv = level[(size_t)(nx + ny * mb_width)];
            }
            level[(size_t)xy] = lv + 1;
            m.intra_level = (uint8_t)(lv + 1 > 255 ? 255 : lv + 1);
            if (lv + 1 > maxl) maxl = lv + 1;
        }
    /* counting sort by level (raster order inside a level) */
    count.assign((size_t)maxl + 2, 0);
    for (int i = 0; i < nmb; i++) if (level[(size_t)i]) count[(size_t)level[(size_t)i]]++;
    int width = 0;
    level_start[0] = 0;
    for (int l = 1; l <= maxl; l++) {
        level_start[l] = level_start[l - 1] + count[(size_t)l];
        if (count[(size_t)l] > width) width = count[(size_t)l];
    }
    std::vector<int32_t> fill(level_start, level_start + maxl + 1);
    for (int i = 0; i < nmb; i++) if (level[(size_t)i]) list[fill[(size_t)level[(size_t)i] - 1]++] = (uint32_t)i;
    if (max_level_width) *max_level_width = width;
    return maxl;
}

extern "C" void *mi355_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!mi355::bind() || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mi355_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int mi355_memcpy_h2d(void *dst, const void *src, size_t bytes) { return mi355::bind() && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
extern "C" int mi355_memcpy_d2h(void *dst, const void *src, size_t bytes) { return mi355::bind() && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
extern "C" int mi355_memcpy_d2d(void *dst, const void *src, size_t bytes) { return mi355::bind() && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) == hipSuccess ? 0 : -1; }
extern "C" void *mi355_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (!mi355::bind() || hipHostMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mi355_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int mi355_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream)
{
    return mi355::bind() && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -1;
}
extern "C" int mi355_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream)
{
    return mi355::bind() && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? 0 : -1;
}
extern "C" int mi355_memcpy2d_d2h_async(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream)
{
    return mi355::bind() && hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? 0 : -1;
}
extern "C" int mi355_memcpy2d_d2d_async(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream)
{
    return mi355::bind() && hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -1;
}
namespace {
__global__ void __launch_bounds__(256) k_copy_batch(const mi355_copy_job *jobs, int n)
{
    if ((int)blockIdx.y >= n) return;
    const mi355_copy_job j = mi355_global(jobs)[blockIdx.y];
    typedef uint32_t u32x4 __attribute__((vector_size(16)));
    const u32x4 *src = reinterpret_cast<const u32x4 *>(mi355_global(reinterpret_cast<const uint8_t *>(j.src)));
    u32x4 *dst = reinterpret_cast<u32x4 *>(mi355_global(reinterpret_cast<uint8_t *>(j.dst)));
    const size_t nvec = (size_t)(j.bytes >> 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}
extern "C" int mi355_copy_batch_dev(const mi355_copy_job *jobs, int n, size_t max_bytes, void *stream)
{
    if (!mi355::bind() || !jobs || n <= 0) return -1;
    if (!max_bytes) return 0;
    const size_t per_block = 256 * 16 * 4;
    unsigned gx = (unsigned)((max_bytes + per_block - 1) / per_block);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_copy_batch, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, jobs, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
namespace {
/* linear planes <-> macroblock tiles: 32 lanes per macroblock (16 luma rows of 16 bytes, 16 chroma rows of 8), eight macroblocks side by side
 * per step (128 bytes of every luma row), CONV_STEPS steps per workgroup (2 / 4 / 8 / 16 steps: 1.94 / 2.0 / 1.7 / 1.35 ms per 1024 1080p pictures = 4.8 TB/s, 2.37 ms before: profiles/r06q_detile.txt); the tiled side of a macroblock is one run of 256 + 128 bytes.
 * The job's record is fetched ONCE per wave, a dword per lane (read field by field it was a vector load and a wait per field: the compiler cannot
 * know that the stores do not touch it), and a thread's pieces of all its steps are requested before the first is stored. */
#ifndef MI355_CONV_STEPS
#define MI355_CONV_STEPS 16
#endif
constexpr int CONV_STEPS = MI355_CONV_STEPS;
static_assert(sizeof(mi355_surface_job) == 72, "eighteen dwords per conversion job");
__global__ void __launch_bounds__(256) k_surface_convert(const mi355_surface_job *jobs, int n)
{
    if ((int)blockIdx.y >= n) return;
    mi355_surface_job j;
    {
        const int lane = (int)threadIdx.x & 63;
        const uint32_t rec = mi355_global_v(reinterpret_cast<const uint32_t *>(mi355_global(jobs) + blockIdx.y))[lane < 18 ? lane : 17];
        uint32_t w[18];
#pragma unroll
        for (int k = 0; k < 18; k++) w[k] = (uint32_t)lane_value((int)rec, k);
        __builtin_memcpy(&j, w, sizeof(j));
    }
    const int W = j.mb_width, H = j.mb_height, nmb = W * H;
    const int r = (int)threadIdx.x & 31;
    const bool to_tiled = j.to_tiled != 0, luma = r < 16;
    const int plane = (r - 16) >> 3, row = (r - 16) & 7;
    const bool al16 = ((reinterpret_cast<uintptr_t>(j.lin[0]) | (uintptr_t)j.lin_stride[0]) & 15) == 0;
    const bool al8 = ((reinterpret_cast<uintptr_t>(j.lin[1]) | reinterpret_cast<uintptr_t>(j.lin[2]) | (uintptr_t)j.lin_stride[1]) & 7) == 0;
    uint8_t *lin[CONV_STEPS], *til[CONV_STEPS];
    uint4 v[CONV_STEPS];
#pragma unroll
    for (int q = 0; q < CONV_STEPS; q++) {
        const int mb = ((int)blockIdx.x * CONV_STEPS + q) * 8 + (int)(threadIdx.x >> 5);
        const int mbc = mb < nmb ? mb : nmb - 1;             /* a step past the picture repeats its last macroblock's loads and stores nothing */
        const int mb_y = mbc / W, mb_x = mbc - mb_y * W;
        if (luma) {
            lin[q] = mi355_global_v(j.lin[0]) + (size_t)(mb_y * 16 + r) * j.lin_stride[0] + mb_x * 16;
            til[q] = mi355_global_v(j.tiled[0]) + (size_t)mb_y * j.tiled_stride[0] + mb_x * MI355_TILE_LUMA_BYTES + 16 * r;
            v[q] = to_tiled ? ld16(lin[q], al16) : ld16(til[q], true);
        } else {
            lin[q] = mi355_global_v(plane ? j.lin[2] : j.lin[1]) + (size_t)(mb_y * 8 + row) * j.lin_stride[1] + mb_x * 8;
            til[q] = mi355_global_v(j.tiled[1]) + (size_t)mb_y * j.tiled_stride[1] + mb_x * MI355_TILE_CHROMA_BYTES + 8 * (r - 16);
            const uint2 c = to_tiled ? ld8(lin[q], al8) : ld8(til[q], true);
            v[q] = make_uint4(c.x, c.y, 0u, 0u);
        }
    }
#pragma unroll
    for (int q = 0; q < CONV_STEPS; q++) {
        const int mb = ((int)blockIdx.x * CONV_STEPS + q) * 8 + (int)(threadIdx.x >> 5);
        if (mb >= nmb) break;
        if (luma) { if (to_tiled) st16(til[q], v[q], true); else st16(lin[q], v[q], al16); }
        else { const uint2 c = make_uint2(v[q].x, v[q].y); if (to_tiled) st8(til[q], c, true); else st8(lin[q], c, al8); }
    }
}
}
extern "C" int mi355_h264_surface_convert_dev(const mi355_surface_job *jobs, int n, int max_mb_width, int max_mb_height, void *stream)
{
    if (!mi355::bind() || !jobs || n <= 0 || max_mb_width <= 0 || max_mb_height <= 0) return -1;
    const long long nmb = (long long)max_mb_width * max_mb_height;
    if (nmb > 0x3FFFFFFF || n > 65535) return -3;
    hipLaunchKernelGGL(k_surface_convert, dim3((unsigned)((nmb + 8 * CONV_STEPS - 1) / (8 * CONV_STEPS)), (unsigned)n), dim3(256), 0, (hipStream_t)stream, jobs, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
/* the waits also report the device's error word (include/mi355dsp.h): a kernel that gave up says so there, and the wait that follows returns it */
extern "C" int mi355_event_sync(void *event) { return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? mi355::fault_after_wait() : -1; }
extern "C" int mi355_sync(void *stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? mi355::fault_after_wait() : -1; }

extern "C" void *mi355_stream_create(void)
{
    hipStream_t st = nullptr;
    if (!mi355::bind() || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return st;
}
extern "C" void mi355_stream_destroy(void *st) { mi355::sync_words_release((hipStream_t)st); (void)hipStreamDestroy((hipStream_t)st); }
extern "C" int mi355_stream_wait_event(void *st, void *e) { return hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)e, 0) == hipSuccess ? 0 : -1; }

extern "C" void *mi355_event_create(void)
{
    hipEvent_t e = nullptr;
    if (!mi355::bind() || hipEventCreateWithFlags(&e, mi355::blocking_sync() ? hipEventBlockingSync : hipEventDefault) != hipSuccess) return nullptr;
    return e;
}
extern "C" void mi355_event_destroy(void *e) { (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int mi355_event_record(void *e, void *stream) { return hipEventRecord((hipEvent_t)e, (hipStream_t)stream) == hipSuccess ? 0 : -1; }
/* 1: everything recorded before the event has finished; 0: not yet; -1: error.  Never waits (a caller that polls with a deadline of its own) */
extern "C" int mi355_event_query(void *e)
{
    const hipError_t r = hipEventQuery((hipEvent_t)e);
    if (r == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return r == hipSuccess ? 1 : -1;
}
/* milliseconds between two recorded events; waits for `b` */
extern "C" float mi355_event_elapsed_ms(void *a, void *b)
{
    float ms = 0.f;
    if (hipEventSynchronize((hipEvent_t)b) != hipSuccess || hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f;
    return ms;
}
