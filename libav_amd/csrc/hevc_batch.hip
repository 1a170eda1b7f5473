/*
 * hevc_batch.hip — Tier 2 for the HEVC rows: device-resident batches of the HEVCDSPContext operations
 * (include/mi355_hevc_batch.h).  One wavefront per job (eight edge segments per wavefront in the loop
 * filter); the arithmetic is the wave-level code of hevc_dev.h that the Tier-1 entry points use, so
 * parity with the reference carries over (tests/test_hevc_batch_*.py check it again per batch).
 * Jobs are read with scalar loads (the job index is wave-uniform); samples go HBM <-> LDS/registers
 * once per job.
 */
#include "mi355_rt.h"
#include "hevc_dev.h"
#include "../../include/mi355_hevc_batch.h"
#include "../../include/mi355dsp.h"

using namespace mi355;

namespace {

#include "hevc_batch_dev.h"

__global__ void __launch_bounds__(64) k_hevc_residual_batch(const mi355_hevc_tu_job *jobs, int n, int bd)
{
    __shared__ IdctScratch s;
    const int lane = lane_id(), half = lane >> 5, hl = lane & 31;
    const int idx = 2 * (int)blockIdx.x + half;
    const bool on = idx < n;
    hevc_residual_run(s, jobs[on ? idx : 0], on, half, hl, bd);
}

/* ---- motion compensation ----------------------------------------------------------------------------- */
__global__ void __launch_bounds__(64) k_hevc_mc_batch(const mi355_hevc_mc_job *jobs, int n, int bd)
{
    __shared__ HevcMcScratch tmp;
    if ((int)blockIdx.x >= n) return;
    const mi355_hevc_mc_job j = jobs[blockIdx.x];
    const int px = bd > 8 ? 2 : 1;
    hevc_mc_wave(mi355_global(j.dst), j.dst_stride / 2, mi355_global(j.src), j.src_stride / px, j.width, j.height, j.mx, j.my, bd, j.chroma ? 4 : 8, tmp);
}

__global__ void __launch_bounds__(64) k_hevc_mcpred_batch(const mi355_hevc_mcpred_job *jobs, int n, int bd)
{
    __shared__ HevcMcScratch tmp;
    /* the kept tile of a two-reference prediction lives behind the rows a 16-row tile uses of `tmp` (below): 6.4 KB of LDS per wave
     * instead of 9, 25 waves per CU instead of 18 — the kernel's time follows its occupancy (1.61 -> 1.37 ms on the one-reference chain) */
    int16_t *const keep = tmp.tmp + HEVC_MC_BI_ROWS * HEVC_MC_TPITCH;
    if ((int)blockIdx.x >= n) return;
    const mi355_hevc_mcpred_job j = jobs[blockIdx.x];
    switch ((j.chroma ? 4 : 0) + (j.kind & 3)) {          /* chroma 1 and 2: the 4-tap instances */
    case 0: hevc_mcpred_taps<8, 0>(j, bd, tmp, keep); break;   case 1: hevc_mcpred_taps<8, 1>(j, bd, tmp, keep); break;
    case 2: hevc_mcpred_taps<8, 2>(j, bd, tmp, keep); break;   case 3: hevc_mcpred_taps<8, 3>(j, bd, tmp, keep); break;
    case 4: hevc_mcpred_taps<4, 0>(j, bd, tmp, keep); break;   case 5: hevc_mcpred_taps<4, 1>(j, bd, tmp, keep); break;
    case 6: hevc_mcpred_taps<4, 2>(j, bd, tmp, keep); break;   default: hevc_mcpred_taps<4, 3>(j, bd, tmp, keep); break;
    }
}

__global__ void __launch_bounds__(64) k_hevc_pred_batch(const mi355_hevc_pred_job *jobs, int n, int bd)
{
    if ((int)blockIdx.x >= n) return;
    mi355_hevc_pred_job j = jobs[blockIdx.x];
    j.dst = mi355_global(j.dst); j.src1 = mi355_global(j.src1); j.src2 = mi355_global(j.src2);
    const HevcPredParams p{ j.kind, j.denom, j.w0, j.w1, j.o0, j.o1 };
    const int px = bd > 8 ? 2 : 1, dt = j.dst_stride / px, ss = j.src_stride / 2;
    const bool two = (j.kind & 1) != 0;
    /* two samples per lane and access when the rows allow it (block widths are even) */
    const unsigned al = (unsigned)(uintptr_t)j.src1 | (two ? (unsigned)(uintptr_t)j.src2 : 0u) | (unsigned)j.src_stride |
                        ((unsigned)(uintptr_t)j.dst | (unsigned)j.dst_stride) * (bd > 8 ? 1u : 2u) | ((unsigned)j.width & 1u) * 4u;
    const unsigned al8 = al | ((unsigned)(uintptr_t)j.dst | (unsigned)j.dst_stride) * (bd > 8 ? 0u : 2u) | ((unsigned)j.width & 3u) * 2u;
    if ((al8 & 7) == 0) {      /* four samples per lane: 8-byte loads, one 8- or 4-byte store */
        const int qw = j.width >> 2, n = qw * j.height, inv = mi355_inv20(qw);
        typedef uint32_t u32x2 __attribute__((vector_size(8)));
        /* four segments per lane and round, all loads first (the intermediates and the picture cannot alias, but the
         * compiler does not know) */
        constexpr int U = 4;
        for (int base = 0; base < n; base += 64 * U) {
            u32x2 a[U], b[U];
            int yy[U], xx[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + 64 * u + lane_id(), ic = i < n ? i : n - 1;
                const int y = mi355_div20(ic, inv), x = 4 * (ic - y * qw);
                yy[u] = i < n ? y : -1; xx[u] = x;
                a[u] = *reinterpret_cast<const u32x2 *>(&j.src1[x + y * ss]);
                b[u] = u32x2{ 0u, 0u };
                if (two) b[u] = *reinterpret_cast<const u32x2 *>(&j.src2[x + y * ss]);
            }
            MI355_ISSUE_FENCE();
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (yy[u] < 0) continue;
                int v[4];
#pragma unroll
                for (int k = 0; k < 4; k++)
                    v[k] = hevc_pred_px(p, (int16_t)((a[u][k >> 1] >> (16 * (k & 1))) & 0xFFFF), (int16_t)((b[u][k >> 1] >> (16 * (k & 1))) & 0xFFFF), bd);
                if (bd > 8) *reinterpret_cast<u32x2 *>(j.dst + (size_t)yy[u] * j.dst_stride + 2 * xx[u]) = u32x2{ (uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16) };
                else *reinterpret_cast<uint32_t *>(j.dst + (size_t)yy[u] * j.dst_stride + xx[u]) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
            }
        }
        return;
    }
    if ((al & 3) == 0) {
        const int hw = j.width >> 1, n = hw * j.height, inv = mi355_inv20(hw);
        for (int i = lane_id(); i < n; i += 64) {
            const int y = mi355_div20(i, inv), x = 2 * (i - y * hw);
            const uint32_t a = *reinterpret_cast<const uint32_t *>(&j.src1[x + y * ss]);
            const uint32_t b = two ? *reinterpret_cast<const uint32_t *>(&j.src2[x + y * ss]) : 0u;
            const int v0 = hevc_pred_px(p, (int16_t)(a & 0xFFFF), (int16_t)(b & 0xFFFF), bd), v1 = hevc_pred_px(p, (int16_t)(a >> 16), (int16_t)(b >> 16), bd);
            if (bd > 8) *reinterpret_cast<uint32_t *>(j.dst + (size_t)y * j.dst_stride + 2 * x) = (uint32_t)v0 | ((uint32_t)v1 << 16);
            else *reinterpret_cast<uint16_t *>(j.dst + (size_t)y * j.dst_stride + x) = (uint16_t)(v0 | (v1 << 8));
        }
        return;
    }
    for (int i = lane_id(); i < j.width * j.height; i += 64) {
        const int y = i / j.width, x = i - y * j.width;
        stpx(j.dst, x + y * dt, hevc_pred_px(p, j.src1[x + y * ss], two ? j.src2[x + y * ss] : 0, bd), bd);
    }
}

/* ---- deblocking: eight jobs per wavefront --------------------------------------------------------------- */
__global__ void __launch_bounds__(64) k_hevc_deblock_batch(const mi355_hevc_lf_job *jobs, int n, int bd)
{
    __shared__ mi355_hevc_lf_job sj[8];
    const int lane = lane_id(), slot = lane >> 3;
    const int first = (int)blockIdx.x * 8;
    /* 8 jobs x 8 dwords = one coalesced load */
    if (first + slot < n) reinterpret_cast<uint32_t *>(&sj[slot])[lane & 7] = reinterpret_cast<const uint32_t *>(&jobs[first + slot])[lane & 7];
    __syncthreads();
    const bool on = first + slot < n;
    const mi355_hevc_lf_job &j = sj[slot];
    const int px = bd > 8 ? 2 : 1, st = on ? j.stride / px : 0;
    const int xs = on && j.horizontal_edge ? st : 1, ys = on && j.horizontal_edge ? 1 : st;
    uint8_t *pix = mi355_global_v(on ? j.pix : nullptr);
    /* luma and chroma jobs may share a launch: both filters run, each on its own groups */
    hevc_lf_luma_wave(pix, xs, ys, on ? j.beta : 0, j.tc, j.no_p, j.no_q, bd, true, on && !j.chroma);
    hevc_lf_chroma_wave(pix, xs, ys, j.tc, j.no_p, j.no_q, bd, true, on && j.chroma);
}

/* ---- deblocking of whole pictures from the frame-level arrays (deblocking_filter_CTB, hevc_filter.c:337-505) ------
 * One launch per direction.  A group of eight lanes = one 8-sample edge segment (luma) or one 8-chroma-sample segment
 * (two 8-luma-sample halves); the group derives its own parameters — bS from the reference's arrays, QP average
 * (get_qPy :166), beta / tc through the tables (:35-45, TC_CALC :332, chroma_tc :47-72) with the offsets of the CTB
 * that contains the edge sample, pcm / bypass masks (get_pcm :316) — and filters.  Waves are luma or chroma as a
 * whole.  Segments of one direction never share a sample, so the launch needs no ordering. */
__device__ const uint8_t k_hevc_tctable[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__device__ const uint8_t k_hevc_betatable[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28,
    30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__device__ const uint8_t k_hevc_qp_c[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };

struct LfPic {      /* the descriptor with its pointers in the global address space */
    const mi355_hevc_lf_picture &p;
    __device__ __forceinline__ int qpy(int x, int y) const
    {
        return mi355_global_v(p.qp_y_tab)[(x >> p.log2_min_cb_size) + (y >> p.log2_min_cb_size) * p.min_cb_width];
    }
    __device__ __forceinline__ int pcm(int x, int y) const
    {
        if (x < 0 || y < 0) return 2;
        const int xp = x >> p.log2_min_pu_size, yp = y >> p.log2_min_pu_size;
        if (xp >= p.min_pu_width || yp >= p.min_pu_height) return 2;
        return mi355_global_v(p.is_pcm)[yp * p.min_pu_width + xp];
    }
    __device__ __forceinline__ mi355_hevc_db_params db(int x, int y) const
    {
        return mi355_global_v(p.deblock)[(x >> p.log2_ctb_size) + (y >> p.log2_ctb_size) * p.ctb_width];
    }
    __device__ __forceinline__ int chroma_tc(int qp_y, int c_idx, int tc_offset) const
    {
        const int qp_i = clip3(qp_y + (c_idx == 1 ? p.cb_qp_offset : p.cr_qp_offset), 0, 57);
        const int qp = qp_i < 30 ? qp_i : (qp_i > 43 ? qp_i - 6 : k_hevc_qp_c[qp_i - 30]);
        return k_hevc_tctable[clip3(qp + 2 + tc_offset, 0, 53)];
    }
};
__device__ __forceinline__ int hevc_tc_calc(int qp, int bs, int tc_offset)
{
    return k_hevc_tctable[clip3(qp + 2 * (bs - 1) + (tc_offset >> 1 << 1), 0, 53)];
}

/* A wave looks at 64 consecutive segments of the grid (one per lane), keeps those that have something to filter (bS != 0; chroma: bS 2)
 * and filters them eight at a time, eight lanes per segment: on a picture of 32x32 blocks a quarter of the 8x8 grid's edges carry a
 * strength, and a wave that took eight consecutive segments as they came ran a quarter full (1.55 M waves per 64 2160p pictures, most
 * of their lanes idle: the launch rate of the waves bounded the pass). */
template <int DIR>      /* 0: vertical edges, 1: horizontal edges */
__global__ void __launch_bounds__(64) k_hevc_deblock_pictures(const mi355_hevc_lf_picture *pics, int luma_cols, int luma_rows, int chroma_cols,
                                                              int chroma_rows, int luma_waves, int waves_per_pic, int bd)
{
    const int lane = lane_id(), slot = lane >> 3;
#ifndef MI355_LF_NO_XCD_ORDER
    /* the workgroups of an XCD (they go to the eight in turn) take consecutive waves' worth of segments: XCD k the k-th eighth of the launch */
    const int nb = (int)gridDim.x, bidx = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
#else
    const int bidx = (int)blockIdx.x;
#endif
    const int pic = bidx / waves_per_pic, w = bidx - pic * waves_per_pic;
    /* the picture's record once, a dword per lane, its fields as scalars from there (v_readlane): read field by field through `pics` every use of a field is a
     * vector load and a wait in front of the load that needed it (the strengths' pointer, then the strength) */
    static_assert(sizeof(mi355_hevc_lf_picture) == 136, "the record is read by dword index");
    mi355_hevc_lf_picture p;
    {
        const int rec = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(pics + pic))[lane < 34 ? lane : 33];
        uint32_t wds[34];
#pragma unroll
        for (int k = 0; k < 34; k++) wds[k] = (uint32_t)lane_value(rec, k);
        __builtin_memcpy(&p, wds, sizeof(p));
    }
    const LfPic P{ p };
    const int ps = bd > 8, W = p.width, H = p.height;
    const bool luma = w < luma_waves;
    const int per_plane = chroma_cols * chroma_rows;
    /* strengths of segment `seg` (luma: the 8x8 grid; chroma: plane c on the 16-luma-sample grid; horizontal chroma edges: the reference's
     * pairs start at x = 8 (mod 16), i.e. at -8 (:469-484), a half outside the picture has bS 0) -> does it filter anything */
    auto luma_seg = [&](int seg, int &x, int &y, int &bs0, int &bs1) {
        const int gy = seg / luma_cols, gx = seg - gy * luma_cols;            /* (not mi355_div20: that is exact for seg * luma_cols < 2^20 only) */
        x = 8 * gx; y = 8 * gy;
        bs0 = bs1 = 0;
        if (!(gy < luma_rows && x < W && y < H && (DIR ? y >= 8 : x >= 8))) return false;
        if (DIR) { bs0 = mi355_global_v(p.horizontal_bs)[(x + y * p.bs_width) >> 2]; bs1 = mi355_global_v(p.horizontal_bs)[(x + 4 + y * p.bs_width) >> 2]; }
        else { bs0 = mi355_global_v(p.vertical_bs)[(x >> 3) + (y >> 2) * p.bs_width]; bs1 = mi355_global_v(p.vertical_bs)[(x >> 3) + ((y + 4) >> 2) * p.bs_width]; }
        return (bs0 | bs1) != 0;
    };
    auto chroma_seg = [&](int seg, int &c, int &x, int &y, int &bs0, int &bs1) {
        c = seg >= per_plane ? 2 : 1;
        const int s2 = seg - (c - 1) * per_plane, gy = s2 / chroma_cols, gx = s2 - gy * chroma_cols;
        x = DIR ? 16 * gx - 8 : 16 * gx; y = 16 * gy;
        bs0 = bs1 = 0;
        if (!(seg < 2 * per_plane && y < H && (DIR ? (y >= 16 && x < W) : (x >= 16 && x < W)))) return false;
        if (DIR) {
            bs0 = x < 0 ? 0 : mi355_global_v(p.horizontal_bs)[(x + y * p.bs_width) >> 2];
            bs1 = x + 8 >= W ? 0 : mi355_global_v(p.horizontal_bs)[(x + 8 + y * p.bs_width) >> 2];
        } else {
            bs0 = mi355_global_v(p.vertical_bs)[(x >> 3) + (y >> 2) * p.bs_width];
            bs1 = mi355_global_v(p.vertical_bs)[(x >> 3) + ((y + 8) >> 2) * p.bs_width];
        }
        return bs0 == 2 || bs1 == 2;
    };
    /* ---- the wave's 64 candidates: each lane works out ITS segment's parameters (QP average, beta / tc through the tables, pcm marks) — every live lane at once, one
     * chain of round trips per wave — and lists them in LDS in lane order; the groups of eight lanes then only fetch samples (before: each group derived its
     * segment's parameters itself in front of its samples' loads, a chain per round of eight segments) */
    __shared__ uint2 s_par[64];
    const int seg_base = (luma ? w : w - luma_waves) * 64;
    uint2 par = make_uint2(0u, 0u);
    bool cand;
    {
        int c = 0, x, y, bs0, bs1;
        cand = luma ? luma_seg(seg_base + lane, x, y, bs0, bs1) : chroma_seg(seg_base + lane, c, x, y, bs0, bs1);
        if (cand) {
            int beta = 0, tc0 = 0, tc1 = 0;
            uint32_t nob = 0;
            if (luma) {
                const mi355_hevc_db_params d = P.db(x, y);
                const int qp = (P.qpy(DIR ? x : x - 1, DIR ? y - 1 : y) + P.qpy(x, y) + 1) >> 1;
                beta = k_hevc_betatable[clip3(qp + d.beta_offset, 0, 51)];
                tc0 = bs0 ? hevc_tc_calc(qp, bs0, d.tc_offset) : 0;
                tc1 = bs1 ? hevc_tc_calc(qp, bs1, d.tc_offset) : 0;
                if (p.pcmf) {
                    if (DIR) nob = (P.pcm(x, y - 1) ? 1u : 0u) | (P.pcm(x + 4, y - 1) ? 2u : 0u) | (P.pcm(x, y) ? 4u : 0u) | (P.pcm(x + 4, y) ? 8u : 0u);
                    else nob = (P.pcm(x - 1, y) ? 1u : 0u) | (P.pcm(x - 1, y + 4) ? 2u : 0u) | (P.pcm(x, y) ? 4u : 0u) | (P.pcm(x, y + 4) ? 8u : 0u);
                }
            } else if (DIR) {
                if (bs0 == 2) tc0 = P.chroma_tc((P.qpy(x, y - 1) + P.qpy(x, y) + 1) >> 1, c, P.db(x, y).tc_offset);
                if (bs1 == 2) tc1 = P.chroma_tc((P.qpy(x + 8, y - 1) + P.qpy(x + 8, y) + 1) >> 1, c, P.db(x + 8, y).tc_offset);
                if (p.pcmf) nob = (P.pcm(x, y - 1) ? 1u : 0u) | (P.pcm(x + 8, y - 1) ? 2u : 0u) | (P.pcm(x, y) ? 4u : 0u) | (P.pcm(x + 8, y) ? 8u : 0u);
            } else {
                const int tco = P.db(x, y).tc_offset;
                if (bs0 == 2) tc0 = P.chroma_tc((P.qpy(x - 1, y) + P.qpy(x, y) + 1) >> 1, c, tco);
                if (bs1 == 2) tc1 = P.chroma_tc((P.qpy(x - 1, y + 8) + P.qpy(x, y + 8) + 1) >> 1, c, tco);
                if (p.pcmf) nob = (P.pcm(x - 1, y) ? 1u : 0u) | (P.pcm(x - 1, y + 8) ? 2u : 0u) | (P.pcm(x, y) ? 4u : 0u) | (P.pcm(x, y + 8) ? 8u : 0u);
            }
            /* [0] = (x + 8) | y << 14 | plane << 28, [1] = beta | tc[0] << 8 | tc[1] << 16 | pcm marks << 24 */
            par = make_uint2((uint32_t)(x + 8) | ((uint32_t)y << 14) | ((uint32_t)c << 28), (uint32_t)beta | ((uint32_t)tc0 << 8) | ((uint32_t)tc1 << 16) | (nob << 24));
        }
    }
    const unsigned long long live = __ballot(cand);
    const int count = __popcll(live);
    if (!count) return;
    if (cand) s_par[__popcll(live & ((1ull << lane) - 1ull))] = par;
    __syncthreads();
    for (int it = 0; it * 8 < count; it++) {
        const int k = it * 8 + slot;
        const bool on = k < count;
        const uint2 q = s_par[on ? k : 0];
        const int x = (int)(q.x & 0x3FFF) - 8, y = (int)((q.x >> 14) & 0x3FFF), c = (int)(q.x >> 28);
        const int beta = (int)(q.y & 0xFF);
        int tc[2] = { (int)((q.y >> 8) & 0xFF), (int)((q.y >> 16) & 0xFF) };
        uint8_t no_p[2] = { (uint8_t)((q.y >> 24) & 1), (uint8_t)((q.y >> 25) & 1) }, no_q[2] = { (uint8_t)((q.y >> 26) & 1), (uint8_t)((q.y >> 27) & 1) };
        if (luma) {
            const int st = p.linesize[0] >> ps;
            uint8_t *pix = mi355_global_v(p.data[0]) + (on ? (ptrdiff_t)y * p.linesize[0] + ((ptrdiff_t)x << ps) : 0);
            hevc_lf_luma_wave(pix, DIR ? st : 1, DIR ? 1 : st, beta, tc, no_p, no_q, bd, true, on);
        } else {
            const int cc = c == 2 ? 2 : 1;
            const int st = p.linesize[cc] >> ps;
            uint8_t *pix = mi355_global_v(p.data[cc]) + (on ? (ptrdiff_t)(y / 2) * p.linesize[cc] + (ptrdiff_t)(x / 2) * (1 << ps) : 0);
            hevc_lf_chroma_wave(pix, DIR ? st : 1, DIR ? 1 : st, tc, no_p, no_q, bd, true, on);
        }
    }
}

/* ---- boundary strengths of whole pictures: boundary_strength (hevc_filter.c:507-583) for every cell side on the 8x8 grid,
 * one lane per 4x4 luma cell (see include/mi355_hevc_batch.h for the edge marks) --------------------------------------- */
struct MvF {
    int mvx[2], mvy[2], ref[2], pf[2], intra;
};
__device__ __forceinline__ MvF mvf_load(const mi355_hevc_mvfield *f)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(f);
    const uint32_t a = w[0], b = w[1], c = w[2], d = w[3];
    MvF m;
    m.mvx[0] = (int16_t)(a & 0xFFFF); m.mvy[0] = (int16_t)(a >> 16);
    m.mvx[1] = (int16_t)(b & 0xFFFF); m.mvy[1] = (int16_t)(b >> 16);
    m.ref[0] = (int8_t)(c & 0xFF); m.ref[1] = (int8_t)((c >> 8) & 0xFF);
    m.pf[0] = (int8_t)((c >> 16) & 0xFF); m.pf[1] = (int8_t)(c >> 24);
    m.intra = (int)(d & 0xFF);
    return m;
}
__device__ __forceinline__ bool mv_far4(const MvF &a, int la, const MvF &b, int lb)
{
    return iabs(a.mvx[la] - b.mvx[lb]) >= 4 || iabs(a.mvy[la] - b.mvy[lb]) >= 4;
}
__device__ inline int hevc_bs_pair(const mi355_hevc_bs_picture &p, const MvF &c, int c_cbf, const MvF &n, int n_cbf, bool tu_border)
{
    const int mvs = c.pf[0] + c.pf[1];
    if (tu_border) {
        if (c.intra || n.intra) return 2;
        if (c_cbf || n_cbf) return 1;
    }
    if (mvs != n.pf[0] + n.pf[1]) return 1;
    if (mvs == 2) {
        const int c0 = p.ref_poc[0][c.ref[0] & 15], c1 = p.ref_poc[1][c.ref[1] & 15], n0 = p.ref_poc[0][n.ref[0] & 15], n1 = p.ref_poc[1][n.ref[1] & 15];
        if (c0 == n0 && c0 == c1 && n0 == n1)
            return (mv_far4(n, 0, c, 0) || mv_far4(n, 1, c, 1)) && (mv_far4(n, 1, c, 0) || mv_far4(n, 0, c, 1));
        if (n0 == c0 && n1 == c1) return mv_far4(n, 0, c, 0) || mv_far4(n, 1, c, 1);
        if (n1 == c0 && n0 == c1) return mv_far4(n, 1, c, 0) || mv_far4(n, 0, c, 1);
        return 1;
    }
    const int lc = c.pf[0] ? 0 : 1, ln = n.pf[0] ? 0 : 1;
    if (p.ref_poc[lc][c.ref[lc] & 15] != p.ref_poc[ln][n.ref[ln] & 15]) return 1;
    return mv_far4(c, lc, n, ln);
}
__global__ void __launch_bounds__(256) k_hevc_boundary_strengths(const mi355_hevc_bs_picture *pics, int cells_x, int cells_y)
{
    const int pic = (int)blockIdx.y;
    const mi355_hevc_bs_picture &p = pics[pic];
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int cy = i / cells_x, cx = i - cy * cells_x, x = 4 * cx, y = 4 * cy;
    if (cy >= cells_y || x >= p.width || y >= p.height || (x & y & 4)) return;      /* a cell off both grid lines has no side to rate */
    const int cw = p.width >> 2;
    const int fl = mi355_global_v(p.edge_flags)[cy * cw + cx];
    const mi355_hevc_mvfield *mvf = mi355_global_v(p.tab_mvf);
    const uint8_t *cbf = mi355_global_v(p.cbf_luma);
    const MvF c = mvf_load(&mvf[(y >> p.log2_min_pu_size) * p.min_pu_width + (x >> p.log2_min_pu_size)]);
    const int c_cbf = cbf[(y >> p.log2_min_tb_size) * p.min_tb_width + (x >> p.log2_min_tb_size)];
    if (!(x & 7)) {
        int bs = 0;
        if (x > 0 && (fl & (MI355_HEVC_EDGE_L_BLOCK | MI355_HEVC_EDGE_L_INNER))) {
            const MvF n = mvf_load(&mvf[(y >> p.log2_min_pu_size) * p.min_pu_width + ((x - 1) >> p.log2_min_pu_size)]);
            bs = hevc_bs_pair(p, c, c_cbf, n, cbf[(y >> p.log2_min_tb_size) * p.min_tb_width + ((x - 1) >> p.log2_min_tb_size)], (fl & MI355_HEVC_EDGE_L_BLOCK) != 0);
        }
        mi355_global_v(p.vertical_bs)[(x >> 3) + (y >> 2) * p.bs_width] = (uint8_t)bs;
    }
    if (!(y & 7)) {
        int bs = 0;
        if (y > 0 && (fl & (MI355_HEVC_EDGE_T_BLOCK | MI355_HEVC_EDGE_T_INNER))) {
            const MvF n = mvf_load(&mvf[((y - 1) >> p.log2_min_pu_size) * p.min_pu_width + (x >> p.log2_min_pu_size)]);
            bs = hevc_bs_pair(p, c, c_cbf, n, cbf[((y - 1) >> p.log2_min_tb_size) * p.min_tb_width + (x >> p.log2_min_tb_size)], (fl & MI355_HEVC_EDGE_T_BLOCK) != 0);
        }
        mi355_global_v(p.horizontal_bs)[(x + y * p.bs_width) >> 2] = (uint8_t)bs;
    }
}

/* ---- SAO ------------------------------------------------------------------------------------------------ */
__global__ void __launch_bounds__(64) k_hevc_sao_batch(const mi355_hevc_sao_job *jobs, int n, int bd)
{
    if ((int)blockIdx.x >= n) return;
    const mi355_hevc_sao_job j = jobs[blockIdx.x];
    SaoJob p;
    p.width = j.width; p.height = j.height; p.c_idx = j.c_idx; p.cls = j.cls; p.bd = bd; p.edge = j.edge;
    for (int k = 0; k < 4; k++) p.borders[k] = j.borders[k];
    p.vert_edge = j.vert_edge; p.horiz_edge = j.horiz_edge; p.diag_edge = j.diag_edge;
    p.eo_class = j.eo_class; p.band_position = j.band_position;
    for (int k = 0; k < 5; k++) p.offset_val[k] = j.offset_val[k];
    const int st = j.stride / (bd > 8 ? 2 : 1);
    __shared__ int tbl[32];
    hevc_sao_wave(mi355_global(j.dst), st, mi355_global(j.src), st, p, tbl);
}

/* one component of one CTB (mi355_hevc_batch.h): its up to four pieces, each either filtered (band / edge: the filter functions
 * write every sample of their region) or, where the owning CTB has SAO off, copied — every sample of the output picture is
 * written exactly once, by the job whose pieces partition its neighbourhood */
typedef uint32_t mi355_sao_u32x4a4 __attribute__((vector_size(16), aligned(4)));
__device__ __forceinline__ void sao_copy_region(uint8_t *dst, const uint8_t *src, int stride, int x0, int y0, int w, int h, int px, int tid, int nthreads)
{
    if (w <= 0 || h <= 0) return;
    const int nbytes = w * px;
    const ptrdiff_t o0 = (ptrdiff_t)y0 * stride + (ptrdiff_t)x0 * px;
    if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)stride | (uintptr_t)(x0 * px) | (uintptr_t)nbytes) & 3) == 0) {
        /* rows in 16-byte pieces (dword-aligned vector accesses) and a tail of dwords */
        const int nv = nbytes >> 4, nt = (nbytes & 15) >> 2, per = nv + nt, inv = mi355_inv20(per);
        for (int i = tid; i < per * h; i += nthreads) {
            const int y = mi355_div20(i, inv), k = i - y * per;
            const ptrdiff_t o = o0 + (ptrdiff_t)y * stride;
            if (k < nv) *reinterpret_cast<mi355_sao_u32x4a4 *>(dst + o + 16 * k) = *reinterpret_cast<const mi355_sao_u32x4a4 *>(src + o + 16 * k);
            else *reinterpret_cast<uint32_t *>(dst + o + 16 * nv + 4 * (k - nv)) = *reinterpret_cast<const uint32_t *>(src + o + 16 * nv + 4 * (k - nv));
        }
        return;
    }
    for (int i = tid; i < nbytes * h; i += nthreads) {          /* (the reciprocal division is exact below 2^19 only) */
        const int y = i / nbytes, k = i - y * nbytes;
        dst[o0 + (ptrdiff_t)y * stride + k] = src[o0 + (ptrdiff_t)y * stride + k];
    }
}
/* eight neighbouring samples per lane: whole 16-byte (8-byte at 8 bit) pieces of a row */
typedef uint32_t mi355_sao_u32x4a2 __attribute__((vector_size(16), aligned(2)));
typedef uint32_t mi355_sao_u32x2a1 __attribute__((vector_size(8), aligned(1)));
struct SaoRaw { uint32_t q[4]; };      /* eight samples as loaded: 16 bytes (above 8 bit) or 8 */
__device__ __forceinline__ SaoRaw sao_raw(const uint8_t *p, bool wide)
{
    SaoRaw r;
    if (wide) { const mi355_sao_u32x4a2 q = *reinterpret_cast<const mi355_sao_u32x4a2 *>(p); r.q[0] = q[0]; r.q[1] = q[1]; r.q[2] = q[2]; r.q[3] = q[3]; }
    else { const mi355_sao_u32x2a1 q = *reinterpret_cast<const mi355_sao_u32x2a1 *>(p); r.q[0] = q[0]; r.q[1] = q[1]; r.q[2] = r.q[3] = 0; }
    return r;
}
__device__ __forceinline__ void sao_unpack(const SaoRaw &r, bool wide, int v[8])
{
    if (wide) {
#pragma unroll
        for (int k = 0; k < 4; k++) { v[2 * k] = (int)(r.q[k] & 0xFFFFu); v[2 * k + 1] = (int)(r.q[k] >> 16); }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (int)((r.q[k >> 2] >> (8 * (k & 3))) & 0xFFu);
    }
}
__device__ __forceinline__ void sao_ld8(const uint8_t *p, bool wide, int v[8])
{
    if (wide) {
        const mi355_sao_u32x4a2 q = *reinterpret_cast<const mi355_sao_u32x4a2 *>(p);
#pragma unroll
        for (int k = 0; k < 4; k++) { v[2 * k] = (int)(q[k] & 0xFFFFu); v[2 * k + 1] = (int)(q[k] >> 16); }
    } else {
        const mi355_sao_u32x2a1 q = *reinterpret_cast<const mi355_sao_u32x2a1 *>(p);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (int)((q[k >> 2] >> (8 * (k & 3))) & 0xFFu);
    }
}
/* ---- two 16-bit samples per register for SAO (v_pk_*_u16 / _i16; plain meaning in the emulator) ---- */
#ifdef MI355_HIP_EMU_H
static inline uint32_t sao_pk_sign(uint32_t c, uint32_t a)       /* per half: sign(c - a) as an int16 */
{
    const int l = (int)(c & 0xFFFF) - (int)(a & 0xFFFF), h = (int)(c >> 16) - (int)(a >> 16);
    return pk_make(l < 0 ? -1 : (l > 0 ? 1 : 0), h < 0 ? -1 : (h > 0 ? 1 : 0));
}
static inline uint32_t sao_pk_subs_u(uint32_t a, uint32_t b)     /* unsigned, saturating at 0 */
{
    const int l = (int)(a & 0xFFFF) - (int)(b & 0xFFFF), h = (int)(a >> 16) - (int)(b >> 16);
    return (uint32_t)(l < 0 ? 0 : l) | ((uint32_t)(h < 0 ? 0 : h) << 16);
}
static inline uint32_t sao_pk_min_u(uint32_t a, uint32_t b)
{
    const uint32_t al = a & 0xFFFF, bl = b & 0xFFFF, ah = a >> 16, bh = b >> 16;
    return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16);
}
static inline uint32_t sao_pk_shr_u(uint32_t a, int n) { return ((a & 0xFFFF) >> n) | (((a >> 16) >> n) << 16); }
#else
typedef unsigned short mi355_v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sao_pk_sign(uint32_t c, uint32_t a)
{
    return pk_u(__builtin_elementwise_min(__builtin_elementwise_max(pk_v(c) - pk_v(a), (mi355_v2s)((short)-1)), (mi355_v2s)((short)1)));
}
__device__ __forceinline__ uint32_t sao_pk_subs_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(mi355_v2us, a), __builtin_bit_cast(mi355_v2us, b)));
}
__device__ __forceinline__ uint32_t sao_pk_min_u(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mi355_v2us, a), __builtin_bit_cast(mi355_v2us, b)));
}
__device__ __forceinline__ uint32_t sao_pk_shr_u(uint32_t a, int n)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(mi355_v2us, a) >> (mi355_v2us)((unsigned short)n));
}
#endif

/* eight samples as four registers of 16-bit pairs (8-bit samples widened) and back */
__device__ __forceinline__ void sao_pairs(const SaoRaw &r, bool wide, uint32_t v[4])
{
    if (wide) { v[0] = r.q[0]; v[1] = r.q[1]; v[2] = r.q[2]; v[3] = r.q[3]; }
    else { v[0] = mi355_widen_lo(r.q[0]); v[1] = mi355_widen_hi(r.q[0]); v[2] = mi355_widen_lo(r.q[1]); v[3] = mi355_widen_hi(r.q[1]); }
}
/* the eight samples one position to the left / right of `v`'s (the outermost one: unspecified) */
__device__ __forceinline__ void sao_shift_left(uint32_t v[4])
{
    v[3] = mi355_alignbyte(v[3], v[2], 2); v[2] = mi355_alignbyte(v[2], v[1], 2); v[1] = mi355_alignbyte(v[1], v[0], 2); v[0] = v[0] << 16;
}
__device__ __forceinline__ void sao_shift_right(uint32_t v[4])
{
    v[0] = mi355_alignbyte(v[1], v[0], 2); v[1] = mi355_alignbyte(v[2], v[1], 2); v[2] = mi355_alignbyte(v[3], v[2], 2); v[3] = v[3] >> 16;
}

/* The whole region of an owner CTB in one pass when its pieces differ in nothing that matters: the band filter always (it knows no borders), the
 * edge filter when no piece touches an unfilterable slice / tile edge (nothing is restored) — sao_band_filter / sao_edge_filter
 * (hevcdsp_template.c:270-718) on packed pairs: eight samples per thread and step (a whole 16-byte / 8-byte piece of a row), the loads of a thread's steps
 * issued together, the offsets looked up with v_perm_b32 in a table of bytes held in two registers (offset + 128: |offset| < 128), rows written in whole
 * aligned pieces.  BORDERS: the region touches a picture border (`bo`: bit 0 left, 1 top, 2 right, 3 bottom).  A sample whose neighbour lies outside the
 * picture gets offset_val[0] (the reference's init_x / init_y / width-- / height-- columns and rows, :388-430), and no load leaves the
 * rows and columns the picture has: a neighbour piece that would start left of column 0 / end right of the last column is fetched in place and shifted
 * in registers, a neighbour row above row 0 / below the last row is the row itself (its samples are kept anyway). */
template <bool WIDE, bool EDGE, bool BORDERS>
__device__ __forceinline__ void sao_region_fast(uint8_t *dst, const uint8_t *src, int stride, int W, int H, int eo, int band_position,
                                                const int32_t *offset_val, int bd, int bo, int tid, int nthreads)
{
    constexpr int BIAS = 128;
    constexpr bool wide = WIDE, edge = EDGE;     /* compile-time: a load under a run-time choice of its width is a branch, the load and a wait for it */
    const int px = wide ? 2 : 1, per = W >> 3, inv = mi355_inv20(per), shift = bd - 5;
    const int dx0 = eo == 0 ? -1 : (eo == 1 ? 0 : (eo == 2 ? -1 : 1)), dy0 = eo == 0 ? 0 : -1;
    /* the offsets as bytes at the places the selectors name.  edge: selector = (sign(c - a) + sign(c - b)) & 7 -> 0: 0, 1: 1, 2: 2, 7: -1, 6: -2;
     * edge_idx[] = { 1, 2, 0, 3, 4 } (:310).  band: selector = min((c >> shift) - band_position & 31, 4) -> offsets 1..4, 4: none */
    uint32_t t_lo, t_hi;
    if (edge) {
        t_lo = (uint32_t)(offset_val[0] + BIAS) | ((uint32_t)(offset_val[3] + BIAS) << 8) | ((uint32_t)(offset_val[4] + BIAS) << 16) | ((uint32_t)BIAS << 24);
        t_hi = (uint32_t)BIAS | ((uint32_t)BIAS << 8) | ((uint32_t)(offset_val[1] + BIAS) << 16) | ((uint32_t)(offset_val[2] + BIAS) << 24);
    } else {
        t_lo = (uint32_t)(offset_val[1] + BIAS) | ((uint32_t)(offset_val[2] + BIAS) << 8) | ((uint32_t)(offset_val[3] + BIAS) << 16) | ((uint32_t)(offset_val[4] + BIAS) << 24);
        t_hi = (uint32_t)BIAS * 0x01010101u;
    }
    const uint32_t bias2 = (uint32_t)BIAS * 0x00010001u, max2 = (uint32_t)((1 << bd) - 1) * 0x00010001u, bp2 = (uint32_t)band_position * 0x00010001u;
    /* A step = the eight samples of one piece.  The loads of the next two steps are in flight while a step is worked on and stored (source and
     * destination may be one picture as far as the compiler knows: written in this order, no load waits behind a store): a wave keeps 2-3 KB of
     * requests open all the time instead of waiting out a memory round trip per step — what a wave moves per microsecond it occupies its slot is what
     * bounds this kernel (32 waves per CU x bytes in flight per wave / latency). */
    struct Step { SaoRaw c, a, b; ptrdiff_t o; int fix; };    /* BORDERS: fix bit 0 sample 0 has no neighbour, 1 sample 7, 2 the row; 3 / 4: neighbour a / b fetched in place, to be shifted */
    const int n = per * H;
    auto fetch = [&](int i_) {
        Step s;
        const int i = i_ < n ? i_ : n - 1;       /* a step past the end repeats the last one and is never worked on */
        const int y = mi355_div20(i, inv), x = 8 * (i - y * per);
        s.o = (ptrdiff_t)y * stride + (ptrdiff_t)x * px;
        s.c = sao_raw(src + s.o, wide);
        s.fix = 0;
        s.a = s.c; s.b = s.c;
        if (edge) {
            if (BORDERS) {
                const bool xl = (bo & 1) && x == 0, xr = (bo & 4) && x + 8 == W, yt = (bo & 2) && y == 0, yb = (bo & 8) && y == H - 1;
                const bool keep_row = dy0 && (yt || yb);
                const bool fa = dx0 < 0 ? xl : (dx0 > 0 ? xr : false), fb = dx0 < 0 ? xr : (dx0 > 0 ? xl : false);      /* a lies at dx0, b at -dx0 */
                s.fix = (dx0 && xl ? 1 : 0) | (dx0 && xr ? 2 : 0) | (keep_row ? 4 : 0) | (fa ? 8 : 0) | (fb ? 16 : 0);
                const ptrdiff_t rowa = dy0 && !yt ? -(ptrdiff_t)stride : 0, rowb = dy0 && !yb ? (ptrdiff_t)stride : 0;
                s.a = sao_raw(src + s.o + rowa + (fa ? 0 : dx0 * px), wide);
                s.b = sao_raw(src + s.o + rowb - (fb ? 0 : dx0 * px), wide);
            } else {
                const ptrdiff_t da = (ptrdiff_t)dx0 * px + (ptrdiff_t)dy0 * stride;
#ifdef MI355_EXP_SAO_NONEIGH
                (void)da;
#elif defined(MI355_EXP_SAO_ALIGNED)
                const ptrdiff_t dal = (ptrdiff_t)dy0 * stride;
                s.a = sao_raw(src + s.o + dal, wide); s.b = sao_raw(src + s.o - dal, wide);
#else
                s.a = sao_raw(src + s.o + da, wide); s.b = sao_raw(src + s.o - da, wide);
#endif
            }
        }
        return s;
    };
    auto work = [&](const Step &s) {
        uint32_t c[4], sel[4], v[4];
        sao_pairs(s.c, wide, c);
        if (!edge) {
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = sao_pk_min_u(pk_sub(sao_pk_shr_u(c[k], shift), bp2) & 0x001F001Fu, 0x00040004u);
        } else {
            uint32_t a[4], b[4];
            sao_pairs(s.a, wide, a);
            sao_pairs(s.b, wide, b);
            if (BORDERS) {
                if (s.fix & 8) { if (dx0 < 0) sao_shift_left(a); else sao_shift_right(a); }
                if (s.fix & 16) { if (dx0 < 0) sao_shift_right(b); else sao_shift_left(b); }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = pk_add(sao_pk_sign(c[k], a[k]), sao_pk_sign(c[k], b[k]));
            if (BORDERS) {
                /* a sample without one of its neighbours: selector 0 = offset_val[0], as the reference's border columns and rows (:388-430) */
                const uint32_t row = (s.fix & 4) ? 0xFFFFFFFFu : 0u;
                sel[0] &= ~(row | ((s.fix & 1) ? 0x0000FFFFu : 0u));
                sel[1] &= ~row;
                sel[2] &= ~row;
                sel[3] &= ~(row | ((s.fix & 2) ? 0xFFFF0000u : 0u));
            }
        }
        /* the low bytes of four selectors -> four offset bytes -> two pairs */
        const uint32_t s01 = byte_perm(sel[1], sel[0], 0x06040200u) & 0x07070707u, s23 = byte_perm(sel[3], sel[2], 0x06040200u) & 0x07070707u;
        const uint32_t f01 = byte_perm(t_hi, t_lo, s01), f23 = byte_perm(t_hi, t_lo, s23);
        const uint32_t off[4] = { mi355_widen_lo(f01), mi355_widen_hi(f01), mi355_widen_lo(f23), mi355_widen_hi(f23) };
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = sao_pk_min_u(sao_pk_subs_u(pk_add(c[k], off[k]), bias2), max2);       /* clip(c + offset) */
        if (wide) *reinterpret_cast<mi355_sao_u32x4a2 *>(dst + s.o) = mi355_sao_u32x4a2{ v[0], v[1], v[2], v[3] };
        else *reinterpret_cast<mi355_sao_u32x2a1 *>(dst + s.o) = mi355_sao_u32x2a1{ byte_perm(v[1], v[0], 0x06040200u), byte_perm(v[3], v[2], 0x06040200u) };
    };
    if (tid >= n) return;
    /* three sets of registers taking turns (a set handed on by copying would have to wait for its loads first) */
    Step s0 = fetch(tid), s1 = fetch(tid + nthreads), s2;
    for (int i = tid;;) {
        s2 = fetch(i + 2 * nthreads); MI355_ISSUE_FENCE(); work(s0); if ((i += nthreads) >= n) break;
        s0 = fetch(i + 2 * nthreads); MI355_ISSUE_FENCE(); work(s1); if ((i += nthreads) >= n) break;
        s1 = fetch(i + 2 * nthreads); MI355_ISSUE_FENCE(); work(s2); if ((i += nthreads) >= n) break;
    }
}
constexpr int SAO_CTB_THREADS = 64;
/* The job record (168 bytes = 42 dwords) is fetched ONCE, a dword per lane, and its fields are read out of that register as scalars (v_readlane): the
 * record is not known to be read-only to the compiler (the kernel stores through other pointers), so field-by-field reads are vector loads, each waited
 * for before the branch that depends on it — 20-40 dependent memory round trips per wave before the first sample was requested, which is what this kernel's
 * time consisted of (profiles/r06_experiments.md). */
static_assert(sizeof(mi355_hevc_sao_ctb_job) == 168 && sizeof(mi355_hevc_sao_piece) == 36 && offsetof(mi355_hevc_sao_ctb_job, piece) == 24, "k_hevc_sao_ctbs reads the record by dword index");
/* A wave per job; SAO_CTB_JOBS consecutive jobs share a workgroup, i.e. a CU and its XCD's L2, at the same time: a caller's jobs of one component run along a row of
 * blocks, so a chroma block's 64-byte half of a line is asked for next to its neighbour's other half (fetched once, written back once), and the luma rows of four
 * neighbours are 512 contiguous bytes. */
#ifndef MI355_SAO_CTB_JOBS
#define MI355_SAO_CTB_JOBS 4
#endif
constexpr int SAO_CTB_JOBS = MI355_SAO_CTB_JOBS;
template <bool WIDE>
__global__ void __launch_bounds__(SAO_CTB_THREADS * SAO_CTB_JOBS) k_hevc_sao_ctbs(const mi355_hevc_sao_ctb_job *jobs, int n, int bd)
{
    /* ... and the workgroups of an XCD (they go to the eight in turn) take consecutive groups: XCD k the k-th eighth of the list */
    const int ngroups = (int)gridDim.x, grp = (ngroups & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (ngroups >> 3) + ((int)blockIdx.x >> 3);
    const int job = grp * SAO_CTB_JOBS + uniform((int)(threadIdx.x >> 6));
    if (job >= n) return;
    const mi355_hevc_sao_ctb_job &j = jobs[job];
    const int tid = (int)threadIdx.x & 63;
    const int rec = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(&j))[tid < 42 ? tid : 41];
#define SAO_REC(dw) ((uint32_t)lane_value(rec, (dw)))
    uint8_t *dst0 = mi355_global(reinterpret_cast<uint8_t *>((uintptr_t)SAO_REC(0) | ((uintptr_t)SAO_REC(1) << 32)));
    const uint8_t *src0 = mi355_global(reinterpret_cast<const uint8_t *>((uintptr_t)SAO_REC(2) | ((uintptr_t)SAO_REC(3) << 32)));
    const int stride = (int)SAO_REC(4), chroma = (SAO_REC(5) & 0xFF) != 0, np = (int)((SAO_REC(5) >> 8) & 0xFF);
    const int cw = (8 >> chroma) + 2, ch = (4 >> chroma) + 2, px = bd > 8 ? 2 : 1;
    __shared__ int tbl_all[SAO_CTB_JOBS][32];          /* the piece-by-piece path's table: a wave's own */
    int *const tbl = tbl_all[uniform((int)(threadIdx.x >> 6))];
    const int st = stride / px;
    /* piece k: dwords 6 + 9 k ..: offset_val[5]; cls | type << 8 | eo_class << 16 | band_position << 24; vert | horiz << 8 | diag << 16 | borders << 24;
     * dx | dy << 16; width | height << 16 */
    const uint32_t p0a = SAO_REC(6 + 5), p0b = SAO_REC(6 + 6), p0c = SAO_REC(6 + 7), p0d = SAO_REC(6 + 8);
    /* piece 0 is the owner's own call (class 0): its size is the region's, its parameters are every piece's */
    if (np >= 1 && (p0a & 0xFF) == 0 && p0c == 0) {
        const int type = (int)((p0a >> 8) & 0xFF), W = (int)(int16_t)(p0d & 0xFFFF), H = (int)(int16_t)(p0d >> 16);
        /* every piece of the owner's type, none with a restored edge */
        uint32_t flags = p0b & 0x00FFFFFFu, types = 0;
        if (np > 1) { flags |= SAO_REC(15 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(15 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        if (np > 2) { flags |= SAO_REC(24 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(24 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        if (np > 3) { flags |= SAO_REC(33 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(33 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        const bool unrestored = flags == 0, same = types == 0;
        if (same && type == 0) { sao_copy_region(dst0, src0, stride, 0, 0, W, H, px, tid, SAO_CTB_THREADS); return; }
        const int32_t ov[5] = { (int32_t)SAO_REC(6), (int32_t)SAO_REC(7), (int32_t)SAO_REC(8), (int32_t)SAO_REC(9), (int32_t)SAO_REC(10) };
        bool small = true;
        for (int e = 0; e < 5; e++) small = small && ov[e] > -128 && ov[e] < 128;
        if (same && small && (W & 7) == 0 && (type == 1 || (type == 2 && unrestored))) {
            /* which picture borders the REGION touches: the owner's own left / top; right / bottom when no other CTB's call covers a strip of it */
            const int bo = type == 2 ? (int)(p0b >> 24) : 0, eo = (int)((p0a >> 16) & 0xFF), bp = (int)(p0a >> 24);
            if (type == 1) sao_region_fast<WIDE, false, false>(dst0, src0, stride, W, H, eo, bp, ov, bd, 0, tid, SAO_CTB_THREADS);
            else if (bo) sao_region_fast<WIDE, true, true>(dst0, src0, stride, W, H, eo, bp, ov, bd, bo, tid, SAO_CTB_THREADS);
            else sao_region_fast<WIDE, true, false>(dst0, src0, stride, W, H, eo, bp, ov, bd, 0, tid, SAO_CTB_THREADS);
            return;
        }
    }
#undef SAO_REC
#ifdef MI355_EXP_SAO_NOSLOW
    return;
#endif
    /* piece by piece, as the reference makes its calls (hevc_sao_wave is a wave's function) */
    for (int k = 0; k < np && k < 4; k++) {
        const mi355_hevc_sao_piece &q = j.piece[k];
        const int cls = uniform(q.cls), type = uniform(q.type), W = uniform(q.width), H = uniform(q.height), bo = uniform(q.borders);
        const ptrdiff_t off = (ptrdiff_t)uniform(q.dy) * stride + (ptrdiff_t)uniform(q.dx) * px;
        uint8_t *dst = dst0 + off;
        const uint8_t *src = src0 + off;
        if (type == 0) {
            /* the region hevc_sao_wave would take for this class */
            const int x0 = (cls & 2) ? -cw : 0, y0 = (cls & 1) ? -ch : 0;
            const int w = (cls & 2) ? cw : ((bo & 4) ? W : W - cw), h = (cls & 1) ? ch : ((bo & 8) ? H : H - ch);
            sao_copy_region(dst, src, stride, x0, y0, w, h, px, tid, 64);
            continue;
        }
        SaoJob p;
        p.width = W; p.height = H; p.c_idx = chroma ? 1 : 0; p.cls = cls; p.bd = bd; p.edge = type == 2;
        p.borders[0] = bo & 1; p.borders[1] = (bo >> 1) & 1; p.borders[2] = (bo >> 2) & 1; p.borders[3] = (bo >> 3) & 1;
        p.vert_edge = uniform(q.vert_edge); p.horiz_edge = uniform(q.horiz_edge); p.diag_edge = uniform(q.diag_edge);
        p.eo_class = uniform(q.eo_class); p.band_position = uniform(q.band_position);
        for (int e = 0; e < 5; e++) p.offset_val[e] = uniform(q.offset_val[e]);
        hevc_sao_wave(dst, st, src, st, p, tbl);
        MI355_WAVE_SYNC();
    }
}

/* emulated_edge_mc as a batch: one wave per window */
static_assert(sizeof(mi355_edge_emu_job) == 48 && offsetof(mi355_edge_emu_job, block_w) == 24 && offsetof(mi355_edge_emu_job, w) == 40, "the record is read by dword index");
__global__ void __launch_bounds__(64) k_edge_emu_batch(const mi355_edge_emu_job *jobs, int n, int bd)
{
    if ((int)blockIdx.x >= n) return;
    /* the record once, a dword per lane (field by field it is a chain of vector loads: see k_hevc_sao_ctbs) */
    const int rec = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(jobs + blockIdx.x))[lane_id() < 12 ? lane_id() : 11];
    uint8_t *dst = mi355_global(reinterpret_cast<uint8_t *>((uintptr_t)(uint32_t)lane_value(rec, 0) | ((uintptr_t)(uint32_t)lane_value(rec, 1) << 32)));
    const uint8_t *src = reinterpret_cast<const uint8_t *>((uintptr_t)(uint32_t)lane_value(rec, 2) | ((uintptr_t)(uint32_t)lane_value(rec, 3) << 32));
    const int ds = lane_value(rec, 4), ss = lane_value(rec, 5), bw = lane_value(rec, 6), bh = lane_value(rec, 7);
    const int sx = lane_value(rec, 8), sy = lane_value(rec, 9), w = lane_value(rec, 10), h = lane_value(rec, 11), px = bd > 8 ? 2 : 1;
    /* `src` is the window's first sample (possibly outside the plane): the plane's sample (0, 0) lies sy rows and sx samples before it */
    const uint8_t *plane = mi355_global(src) - (ptrdiff_t)sy * ss - (ptrdiff_t)sx * px;
    if (bw <= 0 || bh <= 0 || w <= 0 || h <= 0) return;
    const int inv = mi355_inv20(bw);
    for (int i = lane_id(); i < bw * bh; i += 64) {
        const int y = mi355_div20(i, inv), x = i - y * bw;
        const int cx = clip3(sx + x, 0, w - 1), cy = clip3(sy + y, 0, h - 1);
        const uint8_t *s = plane + (ptrdiff_t)cy * ss + (ptrdiff_t)cx * px;
        uint8_t *d = dst + (ptrdiff_t)y * ds + (ptrdiff_t)x * px;
        if (bd > 8) *reinterpret_cast<uint16_t *>(d) = *reinterpret_cast<const uint16_t *>(s);
        else *d = *s;
    }
}

/* ---- intra prediction: one wave per transform block -------------------------------------------------------- */
__global__ void __launch_bounds__(64) k_hevc_intra_batch(const mi355_hevc_intra_job *jobs, int n, int bd)
{
    __shared__ HevcPredScratch s;
    if ((int)blockIdx.x >= n) return;
    const mi355_hevc_intra_job j = jobs[blockIdx.x];
    const uint8_t *top = mi355_global(j.top), *left = mi355_global(j.left);
    const int nedge = 2 * (1 << j.log2_size) + 1, px = bd > 8 ? 2 : 1;
    /* elements -1 .. 2 * size - 1 of both neighbour arrays -> LDS */
    for (int i = lane_id(); i < nedge; i += 64) { s.top[i] = (int16_t)ldpx(top - px, i, bd); s.left[i] = (int16_t)ldpx(left - px, i, bd); }
    __syncthreads();
    hevc_pred_wave(s, mi355_global(j.dst), j.stride / px, j.log2_size, j.kind, j.c_idx, j.mode, bd);
}

/* ---- intra prediction at wrapper level: intra_pred (hevcpred_template.c:31-334) — availability, gather, constrained-intra
 * substitution, inference, smoothing, prediction — one wave per transform block.  Lane k owns neighbour k of the left
 * column and of the top row (at most 64 each; lane 0 also owns the corner).  Everything is lane-parallel except the
 * constrained-intra substitution (:172-232), a chain of "take the previous sample unless this one is intra" walks that
 * lane 0 runs over LDS with the is_intra flags fetched beforehand by all lanes: the flag is a rarely used
 * error-resilience tool and its walk is short (at most 4 * size steps). */
struct IntraWrapLds {
    HevcPredScratch pred;                    /* top / left as the prediction reads them; [0] = the corner */
    uint8_t intra_l[66], intra_t[66];        /* [1 + k]: is_intra of the unit that covers left / top neighbour k */
};
static_assert(sizeof(mi355_hevc_intra_picture) == 104 && sizeof(mi355_hevc_intra_block) == 12, "descriptor layout (tests/hevc_intra_cases.py)");

__device__ __forceinline__ void hevc_intra_block_run(IntraWrapLds &s, const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block b, const int bd)
{
    const mi355_hevc_intra_picture p = mi355_global_v(pics)[b.pic];
    const int lane = lane_id();
    const int c = b.c_idx, hs = c ? p.hshift : 0, vs = c ? p.vshift : 0;
    const int log2 = b.log2_size, n = 1 << log2, x0 = b.x0, y0 = b.y0, mode = b.mode;
    const int nl = n << hs;                                  /* size_in_luma: hshift in both directions (:74) */
    const int ntb = nl >> p.log2_min_tb_size, px = bd > 8 ? 2 : 1, st = p.linesize[c] / px;
    uint8_t *org = mi355_global_v(p.data[c]) + ((ptrdiff_t)(y0 >> vs) * p.linesize[c] + (ptrdiff_t)(x0 >> hs) * px);
    const int32_t *zs = mi355_global_v(p.min_tb_addr_zs);
    const int xtb = x0 >> p.log2_min_tb_size, ytb = y0 >> p.log2_min_tb_size, here = zs[ytb * p.min_tb_width + xtb];
    int16_t *L = s.pred.left + 1, *T = s.pred.top + 1;

    /* availability: lc->na, narrowed by decoding order for the two far neighbours (:93-97) */
    bool a_bl = (b.cand & MI355_HEVC_CAND_BOTTOM_LEFT) && here > zs[(ytb + ntb) * p.min_tb_width + xtb - 1];
    bool a_l = b.cand & MI355_HEVC_CAND_LEFT, a_ul = b.cand & MI355_HEVC_CAND_UP_LEFT, a_u = b.cand & MI355_HEVC_CAND_UP;
    bool a_ur = (b.cand & MI355_HEVC_CAND_UP_RIGHT) && here > zs[(ytb - 1) * p.min_tb_width + xtb + ntb];
    const int n_bl = (imin(y0 + 2 * nl, p.height) - (y0 + nl)) >> vs, n_ur = (imin(x0 + 2 * nl, p.width) - (x0 + nl)) >> hs;
    const bool cip = p.constrained_intra_pred == 1;

    if (cip) {                                               /* :104-151: a neighbour counts only if a unit of it is intra */
        const mi355_hevc_mvfield *mvf = mi355_global_v(p.tab_mvf);
        const int l2pu = p.log2_min_pu_size, pu_mask = (1 << l2pu) - 1, pw = p.min_pu_width, ph = p.min_pu_height;
        const int npu = imax(nl >> l2pu, 1);
        const bool on_x = !(x0 & pu_mask), on_y = !(y0 & pu_mask);
        const int xl = (x0 - 1) >> l2pu, yt = (y0 - 1) >> l2pu;
        if (a_bl && on_x) { const int y = (y0 + nl) >> l2pu; a_bl = __any(lane < imin(npu, ph - y) && mvf[xl + (y + lane) * pw].is_intra); }
        if (a_l && on_x)  { const int y = y0 >> l2pu;        a_l  = __any(lane < imin(npu, ph - y) && mvf[xl + (y + lane) * pw].is_intra); }
        if (a_ul)         a_ul = mvf[xl + yt * pw].is_intra != 0;
        if (a_u && on_y)  { const int x = x0 >> l2pu;        a_u  = __any(lane < imin(npu, pw - x) && mvf[x + lane + yt * pw].is_intra); }
        if (a_ur && on_y) { const int x = (x0 + nl) >> l2pu; a_ur = __any(lane < imin(npu, pw - x) && mvf[x + lane + yt * pw].is_intra); }
        /* per-neighbour flags for the substitution walk; coordinates clamped into the field (a consistent caller never
         * makes the walk look outside it) */
        for (int i = lane; i <= 2 * n; i += 64) {
            const int k = i - 1;
            const int xs = clip3((x0 - (1 << hs)) >> l2pu, 0, pw - 1), ys = clip3((y0 + k * (1 << vs)) >> l2pu, 0, ph - 1);
            const int xt = clip3((x0 + k * (1 << hs)) >> l2pu, 0, pw - 1), yu = clip3((y0 - (1 << vs)) >> l2pu, 0, ph - 1);
            s.intra_l[i] = mvf[xs + ys * pw].is_intra;
            s.intra_t[i] = mvf[xt + yu * pw].is_intra;
        }
    }

    /* gather (:152-170): rows / columns past the picture repeat the last one inside; 128 where constrained intra
     * prediction leaves a neighbour out */
    if (lane < 2 * n) {
        const int dflt = cip ? 128 : 0;
        int lv = dflt, tv = dflt;
        if (lane >= n) {
            if (a_bl) lv = ldpx(org, imin(lane, n + n_bl - 1) * st - 1, bd);
            if (a_ur) tv = ldpx(org, imin(lane, n + n_ur - 1) - st, bd);
        } else {
            if (a_l) lv = ldpx(org, lane * st - 1, bd);
            if (a_u) tv = ldpx(org, lane - st, bd);
        }
        L[lane] = (int16_t)lv; T[lane] = (int16_t)tv;
        if (lane == 0) L[-1] = T[-1] = (int16_t)(a_ul ? ldpx(org, -st - 1, bd) : 128);
    }
    MI355_WAVE_SYNC();

    if (cip && (a_bl || a_l || a_ul || a_u || a_ur)) {       /* :172-232 */
        if (lane == 0) {
            const uint8_t *il = s.intra_l + 1, *it = s.intra_t + 1;
            const int ex = a_ur ? 2 * n : n, ey = a_bl ? 2 * n : n;
            const int lim_x = x0 + (ex << hs) < p.width ? ex : (p.width - x0) >> hs;
            const int lim_y = y0 + (ey << vs) < p.height ? ey : (p.height - y0) >> vs;
            int j = 0;
            bool from_top = true;                            /* start the walk at the first intra sample of the top row? */
            if (a_bl || a_l || a_ul) {
                j = n + (a_bl ? n_bl : 0) - 1;               /* lowest intra sample of the left column, corner included */
                while (j > -1 && !il[j]) j--;
                from_top = !il[j];
            }
            if (from_top) {
                const bool ask_corner = (a_bl || a_l || a_ul) || x0 > 0;
                j = 0;
                while (j < lim_x && !it[j]) j++;
                if ((a_bl || a_l || a_ul) || j > 0) {
                    for (int i = j; i > (ask_corner ? -1 : 0); i--) if (!it[i - 1]) T[i - 1] = T[i];
                    if (!ask_corner) T[-1] = T[0];
                }
                L[-1] = T[-1];
                j = 0;
            }
            if (a_bl || a_l) for (int i = j; i < lim_y; i++) if (!il[i]) L[i] = L[i - 1];
            if (!a_l) for (int i = 0; i < n; i++) L[i] = L[-1];
            if (!a_bl) for (int i = n; i < 2 * n; i++) L[i] = L[n - 1];
            const int stop = (x0 != 0 && y0 == 0) ? 0 : -1;  /* no row above the picture to ask about */
            for (int i = lim_y - 1; i > stop; i--) if (x0 == 0 || !il[i - 1]) L[i - 1] = L[i];
            T[-1] = L[-1];
            if (y0 != 0) for (int i = 0; i < lim_x; i++) if (!it[i]) T[i] = T[i - 1];
        }
        MI355_WAVE_SYNC();
    }

    /* unavailable neighbours take the nearest available sample (:233-270); every step: all lanes read the source, then
     * the owners write */
#define FILL_L(from, cnt, v) do { if (lane >= (from) && lane < (from) + (cnt)) L[lane] = (int16_t)(v); } while (0)
#define FILL_T(from, cnt, v) do { if (lane >= (from) && lane < (from) + (cnt)) T[lane] = (int16_t)(v); } while (0)
    if (!a_bl) {
        if (a_l) { const int v = L[n - 1]; MI355_WAVE_SYNC(); FILL_L(n, n, v); }
        else if (a_ul) { const int v = L[-1]; MI355_WAVE_SYNC(); FILL_L(0, 2 * n, v); a_l = true; }
        else if (a_u) { const int v = T[0]; MI355_WAVE_SYNC(); if (lane == 0) L[-1] = (int16_t)v; FILL_L(0, 2 * n, v); a_ul = a_l = true; }
        else if (a_ur) { const int v = T[n]; MI355_WAVE_SYNC(); FILL_T(0, n, v); if (lane == 0) L[-1] = (int16_t)v; FILL_L(0, 2 * n, v); a_u = a_ul = a_l = true; }
        else { const int v = 1 << (bd - 1); if (lane == 0) L[-1] = (int16_t)v; FILL_T(0, 2 * n, v); FILL_L(0, 2 * n, v); }
        MI355_WAVE_SYNC();
    }
    if (!a_l)  { const int v = L[n];     MI355_WAVE_SYNC(); FILL_L(0, n, v); MI355_WAVE_SYNC(); }
    if (!a_ul) { const int v = L[0];     MI355_WAVE_SYNC(); if (lane == 0) L[-1] = (int16_t)v; MI355_WAVE_SYNC(); }
    if (!a_u)  { const int v = L[-1];    MI355_WAVE_SYNC(); FILL_T(0, n, v); MI355_WAVE_SYNC(); }
    if (!a_ur) { const int v = T[n - 1]; MI355_WAVE_SYNC(); FILL_T(n, n, v); MI355_WAVE_SYNC(); }
    { const int v = L[-1]; MI355_WAVE_SYNC(); if (lane == 0) T[-1] = (int16_t)v; MI355_WAVE_SYNC(); }
#undef FILL_L
#undef FILL_T

    /* smoothing of the neighbours (:272-318) */
    if (c == 0 && mode != 1 && n != 4) {
        const int limit = log2 == 3 ? 7 : log2 == 4 ? 1 : 0;
        if (imin(iabs(mode - 26), iabs(mode - 10)) > limit) {
            const int corner = L[-1], thr = 1 << (bd - 5);
            const bool strong = p.strong_intra_smoothing && log2 == 5 && iabs(corner + T[63] - 2 * T[31]) < thr && iabs(corner + L[63] - 2 * L[31]) < thr;
            int fl = 0, ft = 0, fc = corner;
            if (lane < 2 * n) {
                fl = L[lane]; ft = T[lane];
                if (strong) {
                    if (lane < 63) { fl = ((63 - lane) * corner + (lane + 1) * L[63] + 32) >> 6; ft = ((63 - lane) * corner + (lane + 1) * T[63] + 32) >> 6; }
                } else if (lane < 2 * n - 1) {
                    fl = (L[lane + 1] + 2 * fl + L[lane - 1] + 2) >> 2;
                    ft = (T[lane + 1] + 2 * ft + T[lane - 1] + 2) >> 2;
                }
            }
            if (!strong) fc = (L[0] + 2 * corner + T[0] + 2) >> 2;
            MI355_WAVE_SYNC();
            if (lane < 2 * n) { L[lane] = (int16_t)fl; T[lane] = (int16_t)ft; }
            if (lane == 0) L[-1] = T[-1] = (int16_t)fc;
            MI355_WAVE_SYNC();
        }
    }
    hevc_pred_wave(s.pred, org, st, log2, mode == 0 ? 0 : mode == 1 ? 1 : 2, c, mode, bd);
}

__global__ void __launch_bounds__(64) k_hevc_intra_blocks(const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks, int n_blocks, int bd)
{
    __shared__ IntraWrapLds s;
    if ((int)blockIdx.x >= n_blocks) return;
    hevc_intra_block_run(s, pics, mi355_global_v(blocks)[blockIdx.x], bd);
}

/* An intra transform block as the reference's hls_transform_unit runs it (hevcdec.c:1002-1030, :1238-1260): the prediction of the
 * block, then its residual on the samples just predicted — one launch per dependency level instead of two.  tus[i] belongs to
 * blocks[i] (coeffs NULL: a block without residual); the unit runs on the wave's first half. */
__global__ void __launch_bounds__(64) k_hevc_intra_recon_blocks(const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks,
                                                                const mi355_hevc_tu_job *tus, int n_blocks, int bd)
{
    __shared__ IntraWrapLds s;
    __shared__ IdctScratch t;
    if ((int)blockIdx.x >= n_blocks) return;
    hevc_intra_block_run(s, pics, mi355_global_v(blocks)[blockIdx.x], bd);
    const mi355_hevc_tu_job j = mi355_global_v(tus)[blockIdx.x];
    if (!j.coeffs) return;
    __syncthreads();            /* workgroup-scope release / acquire: the prediction's stores are what the residual's loads of the same samples see */
    const int lane = lane_id();
    hevc_residual_run(t, j, lane < 32, lane >> 5, lane & 31, bd);
}

/* One dependency LEVEL of a batch of pictures in one launch: its prediction blocks, its transform units and its intra blocks (each with its
 * residual) touch disjoint samples, so the three job kinds run side by side — workgroup b takes prediction job b, then pairs of transform
 * units, then intra blocks.  A caller that walks levels (contrib/libav/mi355_hevc_bridge.c: hundreds per picture, most of them a handful of
 * jobs) issues one launch per level instead of up to three. */
union LevelLds {
    HevcMcScratch mc;
    IdctScratch tu;
    struct { IntraWrapLds s; IdctScratch t; } in;
};
__global__ void __launch_bounds__(64) k_hevc_recon_level(const mi355_hevc_mcpred_job *mc, int nm, const mi355_hevc_tu_job *tu, int nt,
                                                         const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks,
                                                         const mi355_hevc_tu_job *btus, int ni, int bd)
{
    __shared__ LevelLds u;
    int b = (int)blockIdx.x;
    const int lane = lane_id();
    if (b < nm) {
        int16_t *const keep = u.mc.tmp + HEVC_MC_BI_ROWS * HEVC_MC_TPITCH;
        const mi355_hevc_mcpred_job j = mc[b];
        switch ((j.chroma ? 4 : 0) + (j.kind & 3)) {
        case 0: hevc_mcpred_taps<8, 0>(j, bd, u.mc, keep); break;   case 1: hevc_mcpred_taps<8, 1>(j, bd, u.mc, keep); break;
        case 2: hevc_mcpred_taps<8, 2>(j, bd, u.mc, keep); break;   case 3: hevc_mcpred_taps<8, 3>(j, bd, u.mc, keep); break;
        case 4: hevc_mcpred_taps<4, 0>(j, bd, u.mc, keep); break;   case 5: hevc_mcpred_taps<4, 1>(j, bd, u.mc, keep); break;
        case 6: hevc_mcpred_taps<4, 2>(j, bd, u.mc, keep); break;   default: hevc_mcpred_taps<4, 3>(j, bd, u.mc, keep); break;
        }
        return;
    }
    b -= nm;
    if (b < (nt + 1) / 2) {
        const int half = lane >> 5, idx = 2 * b + half;
        const bool on = idx < nt;
        hevc_residual_run(u.tu, tu[on ? idx : 0], on, half, lane & 31, bd);
        return;
    }
    b -= (nt + 1) / 2;
    if (b >= ni) return;
    hevc_intra_block_run(u.in.s, pics, mi355_global_v(blocks)[b], bd);
    const mi355_hevc_tu_job j = mi355_global_v(btus)[b];
    if (!j.coeffs) return;
    __syncthreads();            /* as in k_hevc_intra_recon_blocks: the prediction's stores are what the residual's loads see */
    hevc_residual_run(u.in.t, j, lane < 32, lane >> 5, lane & 31, bd);
}

/* EVERY level of a batch of pictures in ONE launch (mi355_hevc_recon_levels_dev): the workgroups of all levels in level order, each doing what its
 * workgroup of k_hevc_recon_level does.  A workgroup takes a ticket (sync[0]: tickets go out in the order workgroups START, whatever numbers the device gave
 * them), finds the level its ticket lies in, and waits until every workgroup of the levels before has counted itself into sync[1] — behind an agent-scope
 * release of its stores; the waiter's loads follow an agent-scope acquire.  The same order between levels as a launch per level gives, for one launch: an
 * all-intra 1080p picture is a chain of ~2800 levels of a handful of blocks each.
 * MEASURED (profiles/r06_experiments.md 12, tools/exp_levels.py): the hop inside the launch is 3.0 us per level for levels of two workgroups — against 3.5 us for back-to-back
 * launches of mi355_hevc_recon_level_dev, whose host side keeps the queue ahead of the device — and grows with the level's size (every workgroup counts itself into ONE word:
 * 7 us at 32 workgroups per level, 147 us at 1000), where a launch boundary stays at 3.5 - 4 us.  The entry point is for chains of SMALL levels whose caller cannot keep a
 * queue full; the reference-side bridge keeps its launch per level (MI355_HEVC_BRIDGE_ONE_LAUNCH=1 switches it over).
 * Progress: the unfinished workgroup with the lowest ticket waits for workgroups with lower tickets only — all finished.  A wait that runs out all the same
 * (LV_NAPS_MAX) sets MI355_ERR_WAIT_EXPIRED in the device's error word: the next mi355_sync / mi355_event_sync returns MI355_E_DEVICE_FAULT.
 * A waiter far from its turn sleeps longer between looks (every resident wave looking at one word every half microsecond would be that word's whole traffic). */
constexpr uint32_t LV_NAPS_MAX = 1u << 21;
#ifdef MI355_HIP_EMU_H
static inline uint32_t lv_load(const uint32_t *p) { return *p; }
static inline void lv_release() {}
static inline void lv_acquire() {}
static inline void lv_nap(uint32_t) { std::fprintf(stderr, "k_hevc_recon_levels: a level waits for one that has not run (emulator: workgroups run in order)\n"); std::abort(); }
#else
__device__ __forceinline__ uint32_t lv_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lv_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lv_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void lv_nap(uint32_t far)
{
    __builtin_amdgcn_s_sleep(8);
    for (uint32_t k = 0; k < far; k++) __builtin_amdgcn_s_sleep(32);
}
#endif
static_assert(sizeof(mi355_hevc_level) == 32, "eight dwords per level record (fetched a dword per lane)");
__global__ void __launch_bounds__(64) k_hevc_recon_levels(const mi355_hevc_level *levels, int n_levels, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tu,
                                                          const mi355_hevc_intra_picture *pics, const mi355_hevc_intra_block *blocks, const mi355_hevc_tu_job *btus,
                                                          int bd, uint32_t *sync, uint32_t *error_word, uint32_t naps_max)
{
    __shared__ LevelLds u;
    const int lane = lane_id();
    uint32_t tk = 0;
    if (lane == 0) tk = atomicAdd(mi355_global(sync), 1u);
    tk = (uint32_t)lane_value((int)tk, 0);
    /* the level of ticket tk: the last one whose first workgroup is <= tk — 64 probes per round trip */
    const uint32_t *lw = reinterpret_cast<const uint32_t *>(mi355_global(levels));
    int lo = 0, cnt = n_levels;                        /* the answer lies in [lo, lo + cnt) */
    while (cnt > 1) {
        const int stride = (cnt + 63) >> 6, idx = lo + lane * stride;
        const bool in = lane * stride < cnt;
        const uint32_t first = in ? mi355_global_v(lw)[8 * (size_t)idx] : 0xFFFFFFFFu;
        const int below = __popcll(__ballot(in && first <= tk));          /* >= 1: level lo starts at or before tk */
        const int base = lo + (below - 1) * stride;
        cnt = imin(stride, lo + cnt - base);
        lo = base;
    }
    const uint32_t rec = mi355_global_v(lw)[8 * (size_t)lo + (lane < 8 ? lane : 7)];
    const uint32_t first_wg = (uint32_t)lane_value((int)rec, 0);
    const int mc0 = lane_value((int)rec, 1), nm = lane_value((int)rec, 2), tu0 = lane_value((int)rec, 3), nt = lane_value((int)rec, 4);
    const int in0 = lane_value((int)rec, 5), ni = lane_value((int)rec, 6);
    uint32_t *const done = mi355_global(sync) + 1;
    if (first_wg) {
        uint32_t naps = 0, seen;
        for (;;) {
            seen = (uint32_t)lane_value((int)lv_load(mi355_global_v(done)), 0);
            if (seen >= first_wg || naps >= naps_max) break;
            lv_nap(imin((int)((first_wg - seen) >> 3), 24));
            naps++;
        }
        if (seen < first_wg && lane == 0) atomicOr(error_word, (uint32_t)MI355_ERR_WAIT_EXPIRED);
        lv_acquire();
    }
    int b = (int)(tk - first_wg);
    if (b < nm) {
        int16_t *const keep = u.mc.tmp + HEVC_MC_BI_ROWS * HEVC_MC_TPITCH;
        const mi355_hevc_mcpred_job j = mc[mc0 + b];
        switch ((j.chroma ? 4 : 0) + (j.kind & 3)) {
        case 0: hevc_mcpred_taps<8, 0>(j, bd, u.mc, keep); break;   case 1: hevc_mcpred_taps<8, 1>(j, bd, u.mc, keep); break;
        case 2: hevc_mcpred_taps<8, 2>(j, bd, u.mc, keep); break;   case 3: hevc_mcpred_taps<8, 3>(j, bd, u.mc, keep); break;
        case 4: hevc_mcpred_taps<4, 0>(j, bd, u.mc, keep); break;   case 5: hevc_mcpred_taps<4, 1>(j, bd, u.mc, keep); break;
        case 6: hevc_mcpred_taps<4, 2>(j, bd, u.mc, keep); break;   default: hevc_mcpred_taps<4, 3>(j, bd, u.mc, keep); break;
        }
    } else if ((b -= nm) < (nt + 1) / 2) {
        const int half = lane >> 5, idx = 2 * b + half;
        const bool on = idx < nt;
        hevc_residual_run(u.tu, tu[tu0 + (on ? idx : 0)], on, half, lane & 31, bd);
    } else if ((b -= (nt + 1) / 2) < ni) {
        hevc_intra_block_run(u.in.s, pics, mi355_global_v(blocks)[in0 + b], bd);
        const mi355_hevc_tu_job j = mi355_global_v(btus)[in0 + b];
        if (j.coeffs) {
            __syncthreads();        /* as in k_hevc_intra_recon_blocks: the prediction's stores are what the residual's loads see */
            hevc_residual_run(u.in.t, j, lane < 32, lane >> 5, lane & 31, bd);
        }
    }
    lv_release();                   /* this workgroup's samples are where every other workgroup's loads (behind their acquire) find them */
    if (lane == 0 && naps_max) atomicAdd(done, 1u);       /* (bound 0, the test hook: nobody counts, every wait runs out) */
}

/* ---- a16 + a17 fused: deblocking and SAO of a coding tree block in ONE workgroup (mi355_hevc_filter_ctbs_dev) --------------------------------------------
 * What deblocking_filter_CTB (hevc_filter.c:337-505) and sao_filter_CTB (:188-314) do to a block's own samples, from the UNFILTERED reconstruction to the output
 * picture, without the deblocked picture ever leaving the chip.  The block and 8 samples around it are fetched into LDS (rows and columns -4 .. size + 3 are
 * used: a sample of the block's one-sample ring — SAO's neighbours — is changed by edges up to 3 samples away, whose decisions read 4 samples on either side);
 * vertical edges are filtered there, then horizontal ones on the result (the order of the reference's two passes over a picture: every vertical edge reads
 * unfiltered samples only, every horizontal one reads what the vertical pass left), then SAO reads the tile and writes the output picture in whole pieces.
 * Edges on the block's border and in its ring are filtered by this workgroup AND by the neighbour's for its own ring: both start from the same unfiltered
 * samples and get the same values.  Traffic: the reconstruction read 1.27 times (the ring), the output written once — against read + write for each of the two
 * deblocking launches' edges and read + write for SAO. */
constexpr int FT_THREADS = 256, FT_Y = 80, FT_C = 48;      /* tile sides in samples for a 64x64 block: the block + 8 samples each way */
template <bool WIDE> struct __attribute__((aligned(16))) FtTile {
    uint8_t y[FT_Y * FT_Y * (WIDE ? 2 : 1)];
    uint8_t c[2][FT_C * FT_C * (WIDE ? 2 : 1)];
};
/* one line of an edge in the tile: the eight samples across it one by one (LDS), the decisions and the arithmetic of hevc_lf_luma_wave */
__device__ __forceinline__ void ft_lf_luma(uint8_t *pix, int xs, int ys, int beta, const int *tc_, const uint8_t *no_p_, const uint8_t *no_q_, int bd, bool on)
{
    const int l = lane_id() & 7;
    int p[4], q[4], np[3], nq[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p[k] = on ? ldpx(pix, -(k + 1) * xs + l * ys, bd) : 0;
        q[k] = on ? ldpx(pix, k * xs + l * ys, bd) : 0;
    }
    if (!hevc_lf_luma_core(p, q, beta, tc_, no_p_, no_q_, bd, on, np, nq)) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (np[k] != p[k]) stpx(pix, -(k + 1) * xs + l * ys, np[k], bd);
        if (nq[k] != q[k]) stpx(pix, k * xs + l * ys, nq[k], bd);
    }
}
/* candidates of a pass: the luma edge segments (8 samples along the edge: two halves of 4 with a strength each) that touch rows / columns -4 .. size + 3 of the
 * block, then those of the two chroma planes (8 chroma samples: two halves of 4 = 8 luma samples each; ring -2 .. size / 2 + 1).  Candidate `cand` of pass DIR ->
 * everything the filter needs, as two dwords: [0] = tile x | tile y << 8 | plane << 16 | no_p / no_q bits << 24 (the segment's first q-side sample; tile sample
 * (0, 0) = picture sample (x0 - 8, y0 - 8), chroma likewise in its plane), [1] = beta | tc[0] << 8 | tc[1] << 16 | live << 24 (live: a strength that filters —
 * luma != 0, chroma 2).  Worked out ONCE per candidate, by one thread, while the tile's samples are on their way (k_hevc_deblock_pictures' groups of eight lanes each
 * derive their segment's parameters themselves, in front of the samples' loads: three dependent round trips per group of segments). */
template <int DIR>
__device__ __forceinline__ uint2 ft_params(const mi355_hevc_lf_picture &p, int x0, int y0, int S, int cand)
{
    const LfPic P{ p };
    const int W = p.width, H = p.height;
    const int nUl = (S >> 3) + 2, nl = ((S >> 3) + 1) * nUl;
    const int nUc = DIR ? (S >> 4) + 1 : (S >> 4) + 2, nc = ((S >> 4) + 1) * nUc;
    const uint8_t *vbs = mi355_global_v(p.vertical_bs), *hbs = mi355_global_v(p.horizontal_bs);
    int plane = 0, x = 0, y = 0, bs0 = 0, bs1 = 0;
    bool live = false;
    if (cand < nl) {
        const int e = mi355_div20(cand, mi355_inv20(nUl)), u = cand - e * nUl;
        if (DIR == 0) { x = x0 + 8 * e; y = y0 - 8 + 8 * u; } else { y = y0 + 8 * e; x = x0 - 8 + 8 * u; }
        if (x >= 0 && y >= 0 && x < W && y < H && (DIR ? y >= 8 : x >= 8)) {
            if (DIR) { bs0 = hbs[(x + y * p.bs_width) >> 2]; bs1 = hbs[(x + 4 + y * p.bs_width) >> 2]; }
            else { bs0 = vbs[(x >> 3) + (y >> 2) * p.bs_width]; bs1 = vbs[(x >> 3) + ((y + 4) >> 2) * p.bs_width]; }
            if (u == 0) bs0 = 0;                       /* the halves beyond the four rows / columns around the block: not this block's business, not in the tile */
            if (u == nUl - 1) bs1 = 0;
            live = (bs0 | bs1) != 0;
        }
    } else if (cand - nl < 2 * nc) {
        const int c1 = cand - nl, pl = mi355_div20(c1, mi355_inv20(nc)), c2 = c1 - pl * nc, e = mi355_div20(c2, mi355_inv20(nUc)), u = c2 - e * nUc;
        plane = 1 + pl;
        if (DIR == 0) {
            x = x0 + 16 * e; y = y0 - 16 + 16 * u;
            if (x >= 16 && x < W && y >= 0 && y < H) {
                bs0 = vbs[(x >> 3) + (y >> 2) * p.bs_width]; bs1 = vbs[(x >> 3) + ((y + 8) >> 2) * p.bs_width];
                if (u == 0) bs0 = 0;
                if (u == nUc - 1) bs1 = 0;
            }
        } else {
            /* the reference's pairs of horizontal chroma segments start at x = 8 (mod 16), i.e. at -8 (hevc_filter.c:469-484); a half outside the picture has bS 0 */
            y = y0 + 16 * e; x = x0 - 8 + 16 * u;
            if (y >= 16 && y < H && x < W) {
                bs0 = x < 0 ? 0 : hbs[(x + y * p.bs_width) >> 2];
                bs1 = x + 8 >= W ? 0 : hbs[(x + 8 + y * p.bs_width) >> 2];
            }
        }
        live = bs0 == 2 || bs1 == 2;
    }
    if (!live) return make_uint2(0u, 0u);
    int beta = 0, tc[2] = { 0, 0 }, no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
    if (plane == 0) {
        const mi355_hevc_db_params d = P.db(x, y);
        const int qp = (P.qpy(DIR ? x : x - 1, DIR ? y - 1 : y) + P.qpy(x, y) + 1) >> 1;
        beta = k_hevc_betatable[clip3(qp + d.beta_offset, 0, 51)];
        tc[0] = bs0 ? hevc_tc_calc(qp, bs0, d.tc_offset) : 0;
        tc[1] = bs1 ? hevc_tc_calc(qp, bs1, d.tc_offset) : 0;
        if (p.pcmf) {
            if (DIR) { no_p[0] = P.pcm(x, y - 1); no_p[1] = P.pcm(x + 4, y - 1); no_q[0] = P.pcm(x, y); no_q[1] = P.pcm(x + 4, y); }
            else { no_p[0] = P.pcm(x - 1, y); no_p[1] = P.pcm(x - 1, y + 4); no_q[0] = P.pcm(x, y); no_q[1] = P.pcm(x, y + 4); }
        }
    } else if (DIR) {
        if (bs0 == 2) tc[0] = P.chroma_tc((P.qpy(x, y - 1) + P.qpy(x, y) + 1) >> 1, plane, P.db(x, y).tc_offset);
        if (bs1 == 2) tc[1] = P.chroma_tc((P.qpy(x + 8, y - 1) + P.qpy(x + 8, y) + 1) >> 1, plane, P.db(x + 8, y).tc_offset);
        if (p.pcmf) { no_p[0] = P.pcm(x, y - 1); no_p[1] = P.pcm(x + 8, y - 1); no_q[0] = P.pcm(x, y); no_q[1] = P.pcm(x + 8, y); }
    } else {
        const int tco = P.db(x, y).tc_offset;
        if (bs0 == 2) tc[0] = P.chroma_tc((P.qpy(x - 1, y) + P.qpy(x, y) + 1) >> 1, plane, tco);
        if (bs1 == 2) tc[1] = P.chroma_tc((P.qpy(x - 1, y + 8) + P.qpy(x, y + 8) + 1) >> 1, plane, tco);
        if (p.pcmf) { no_p[0] = P.pcm(x - 1, y); no_p[1] = P.pcm(x - 1, y + 8); no_q[0] = P.pcm(x, y); no_q[1] = P.pcm(x, y + 8); }
    }
    const int tx = plane == 0 ? x - x0 + 8 : (x >> 1) - (x0 >> 1) + 8, ty = plane == 0 ? y - y0 + 8 : (y >> 1) - (y0 >> 1) + 8;
    /* the pcm / bypass marks as the filters test them: != 0 */
    const uint32_t nob = (no_p[0] ? 1u : 0u) | (no_p[1] ? 2u : 0u) | (no_q[0] ? 4u : 0u) | (no_q[1] ? 8u : 0u);
    return make_uint2((uint32_t)tx | ((uint32_t)ty << 8) | ((uint32_t)plane << 16) | (nob << 24), (uint32_t)beta | ((uint32_t)tc[0] << 8) | ((uint32_t)tc[1] << 16) | (1u << 24));
}
/* a pass over the tile: a wave lists the live ones among its candidates (every fourth: a share of the luma and of the chroma segments each) and works through them
 * eight at a time, eight lanes per segment; nothing but LDS is touched */
template <int DIR, bool WIDE>
__device__ __forceinline__ void ft_deblock_pass(FtTile<WIDE> &t, const uint2 *par, int bd, int tid, uint8_t (*s_act)[64])
{
    const int wave = tid >> 6, lane = tid & 63, slot = lane >> 3, ps = WIDE ? 1 : 0;
    const bool cand = (par[lane * 4 + wave].y >> 24) != 0;
    const unsigned long long live = __ballot(cand);
    const int count = __popcll(live);
    if (cand) s_act[wave][__popcll(live & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    MI355_WAVE_SYNC();
    for (int it = 0; it * 8 < count; it++) {
        const int k = it * 8 + slot;
        const bool on = k < count;
        const uint2 q = par[(int)s_act[wave][on ? k : 0] * 4 + wave];
        const int tx = (int)(q.x & 0xFF), ty = (int)((q.x >> 8) & 0xFF), plane = (int)((q.x >> 16) & 0xFF);
        const int beta = (int)(q.y & 0xFF);
        int tc[2] = { (int)((q.y >> 8) & 0xFF), (int)((q.y >> 16) & 0xFF) };
        uint8_t no_p[2] = { (uint8_t)((q.x >> 24) & 1), (uint8_t)((q.x >> 25) & 1) }, no_q[2] = { (uint8_t)((q.x >> 26) & 1), (uint8_t)((q.x >> 27) & 1) };
        const bool luma = plane == 0;
        const int pitch = luma ? FT_Y : FT_C;
        uint8_t *pix = (luma ? t.y : (plane == 1 ? t.c[0] : t.c[1])) + ((ty * pitch + tx) << ps);
        ft_lf_luma(pix, DIR ? pitch : 1, DIR ? 1 : pitch, beta, tc, no_p, no_q, bd, on && luma);
        hevc_lf_chroma_wave(pix, DIR ? pitch : 1, DIR ? 1 : pitch, tc, no_p, no_q, bd, true, on && !luma);
    }
}

/* SAO of a region whose deblocked samples (and their one-sample ring) lie in LDS: sao_region_fast's arithmetic, neighbours from the tile (the pieces before / behind
 * a piece by whole dwords and a funnel shift), the output picture written in whole 16-byte / 8-byte pieces.  `tile`: the region's sample (0, 0), tp bytes per row. */
template <bool WIDE, bool EDGE, bool BORDERS>
__device__ __forceinline__ void ft_sao_region(const uint8_t *tile, int tp, uint8_t *dst, int stride, int W, int H, int eo, int band_position, const int32_t *offset_val,
                                              int bd, int bo, int first, int nthreads)
{
    constexpr int BIAS = 128, PX = WIDE ? 2 : 1;
    const int per = W >> 3, inv = mi355_inv20(per), shift = bd - 5;
    const int dx0 = eo == 0 ? -1 : (eo == 1 ? 0 : (eo == 2 ? -1 : 1)), dy0 = eo == 0 ? 0 : -1;
    uint32_t t_lo, t_hi;
    if (EDGE) {
        t_lo = (uint32_t)(offset_val[0] + BIAS) | ((uint32_t)(offset_val[3] + BIAS) << 8) | ((uint32_t)(offset_val[4] + BIAS) << 16) | ((uint32_t)BIAS << 24);
        t_hi = (uint32_t)BIAS | ((uint32_t)BIAS << 8) | ((uint32_t)(offset_val[1] + BIAS) << 16) | ((uint32_t)(offset_val[2] + BIAS) << 24);
    } else {
        t_lo = (uint32_t)(offset_val[1] + BIAS) | ((uint32_t)(offset_val[2] + BIAS) << 8) | ((uint32_t)(offset_val[3] + BIAS) << 16) | ((uint32_t)(offset_val[4] + BIAS) << 24);
        t_hi = (uint32_t)BIAS * 0x01010101u;
    }
    const uint32_t bias2 = (uint32_t)BIAS * 0x00010001u, max2 = (uint32_t)((1 << bd) - 1) * 0x00010001u, bp2 = (uint32_t)band_position * 0x00010001u;
    /* eight samples at (x + dx, y + dy), dx in -1 .. 1, as pairs: the aligned piece, and for dx != 0 the dword before / behind it shifted in */
    auto fetch = [&](int x, int y, int dx, uint32_t v[4]) {
        const uint8_t *q = tile + y * tp + x * PX;
        SaoRaw r;
        uint32_t side = 0u;
        if (WIDE) {
            const uint4 w = *reinterpret_cast<const uint4 *>(q);
            r.q[0] = w.x; r.q[1] = w.y; r.q[2] = w.z; r.q[3] = w.w;
            if (dx) side = *reinterpret_cast<const uint32_t *>(q + (dx < 0 ? -4 : 16));
        } else {
            const uint2 w = *reinterpret_cast<const uint2 *>(q);
            r.q[0] = w.x; r.q[1] = w.y; r.q[2] = r.q[3] = 0u;
            if (dx) side = *reinterpret_cast<const uint32_t *>(q + (dx < 0 ? -4 : 8));
        }
        sao_pairs(r, WIDE, v);
        if (dx < 0) {
            const uint32_t prev = WIDE ? side >> 16 : side >> 24;                   /* the sample before the piece */
            sao_shift_left(v); v[0] |= prev;
        } else if (dx > 0) {
            const uint32_t next = WIDE ? side & 0xFFFFu : side & 0xFFu;          /* the sample behind it */
            sao_shift_right(v); v[3] |= next << 16;
        }
    };
    const int n = per * H;
    for (int i = first; i < n; i += nthreads) {
        const int y = mi355_div20(i, inv), x = 8 * (i - y * per);
        uint32_t c[4], sel[4], v[4];
        fetch(x, y, 0, c);
        if (!EDGE) {
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = sao_pk_min_u(pk_sub(sao_pk_shr_u(c[k], shift), bp2) & 0x001F001Fu, 0x00040004u);
        } else {
            uint32_t a[4], b[4];
            fetch(x, y + dy0, dx0, a);
            fetch(x, y - dy0, -dx0, b);
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = pk_add(sao_pk_sign(c[k], a[k]), sao_pk_sign(c[k], b[k]));
            if (BORDERS) {
                /* a sample without one of its neighbours (the picture ends there): selector 0 = offset_val[0] (hevcdsp_template.c:388-430) */
                const bool xl = (bo & 1) && x == 0, xr = (bo & 4) && x + 8 == W, yt = (bo & 2) && y == 0, yb = (bo & 8) && y == H - 1;
                const uint32_t row = dy0 && (yt || yb) ? 0xFFFFFFFFu : 0u;
                sel[0] &= ~(row | (dx0 && xl ? 0x0000FFFFu : 0u));
                sel[1] &= ~row;
                sel[2] &= ~row;
                sel[3] &= ~(row | (dx0 && xr ? 0xFFFF0000u : 0u));
            }
        }
        const uint32_t s01 = byte_perm(sel[1], sel[0], 0x06040200u) & 0x07070707u, s23 = byte_perm(sel[3], sel[2], 0x06040200u) & 0x07070707u;
        const uint32_t f01 = byte_perm(t_hi, t_lo, s01), f23 = byte_perm(t_hi, t_lo, s23);
        const uint32_t off[4] = { mi355_widen_lo(f01), mi355_widen_hi(f01), mi355_widen_lo(f23), mi355_widen_hi(f23) };
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = sao_pk_min_u(sao_pk_subs_u(pk_add(c[k], off[k]), bias2), max2);
        uint8_t *d = dst + (ptrdiff_t)y * stride + x * PX;
        if (WIDE) *reinterpret_cast<mi355_sao_u32x4a2 *>(d) = mi355_sao_u32x4a2{ v[0], v[1], v[2], v[3] };
        else *reinterpret_cast<mi355_sao_u32x2a1 *>(d) = mi355_sao_u32x2a1{ byte_perm(v[1], v[0], 0x06040200u), byte_perm(v[3], v[2], 0x06040200u) };
    }
}
/* the region leaves as it is (SAO off for the block's component) */
template <bool WIDE>
__device__ __forceinline__ void ft_copy_region(const uint8_t *tile, int tp, uint8_t *dst, int stride, int W, int H, int first, int nthreads)
{
    constexpr int PX = WIDE ? 2 : 1;
    const int per = W >> 2, inv = mi355_inv20(per);          /* pieces of four samples: a region's width is a multiple of four (chroma of an 8-multiple) */
    for (int i = first; i < per * H; i += nthreads) {
        const int y = mi355_div20(i, inv), x = 4 * (i - y * per);
        const uint8_t *q = tile + y * tp + x * PX;
        uint8_t *d = dst + (ptrdiff_t)y * stride + x * PX;
        if (WIDE) { const uint2 w = *reinterpret_cast<const uint2 *>(q); *reinterpret_cast<mi355_sao_u32x2a1 *>(d) = mi355_sao_u32x2a1{ w.x, w.y }; }
        else { const uint32_t w = *reinterpret_cast<const uint32_t *>(q); __builtin_memcpy(d, &w, 4); }
    }
}

static_assert(sizeof(mi355_hevc_filter_ctb_job) == 20, "the record is read by dword index");
template <bool WIDE, bool VPASS>
__global__ void __launch_bounds__(FT_THREADS) k_hevc_filter_ctbs(const mi355_hevc_lf_picture *pics, const mi355_hevc_filter_ctb_job *ctbs, int n, const mi355_hevc_sao_ctb_job *sao,
                                                                 int log2_ctb, int bd, uint32_t *error_word)
{
    constexpr int PX = WIDE ? 2 : 1;
    __shared__ FtTile<WIDE> tile;
    __shared__ uint8_t s_act[FT_THREADS / 64][64];
    __shared__ uint2 s_par[2][FT_THREADS];
    if ((int)blockIdx.x >= n) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int S = 1 << log2_ctb;
    /* the block's record (5 dwords), then — a dword per lane each — the picture's and the three SAO jobs' */
    const int brec = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(ctbs + blockIdx.x))[lane < 5 ? lane : 4];
    const int pic = lane_value(brec, 0), x0 = lane_value(brec, 1) & 0xFFFF, y0 = (int)((uint32_t)lane_value(brec, 1) >> 16);
    const int prec = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(pics + pic))[lane < 34 ? lane : 33];
    /* the SAO job of the component this wave takes in the last phase: luma for every wave's first two rounds; the chroma planes: waves 0, 1 Cb, waves 2, 3 Cr */
    const mi355_hevc_sao_ctb_job *job_y = sao + (uint32_t)lane_value(brec, 2), *job_c = sao + (uint32_t)lane_value(brec, wave < 2 ? 3 : 4);
    const int srec_y = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(job_y))[lane < 42 ? lane : 41];
    const int srec_c = (int)mi355_global_v(reinterpret_cast<const uint32_t *>(job_c))[lane < 42 ? lane : 41];
    mi355_hevc_lf_picture p;
    {
        uint32_t wds[34];
#pragma unroll
        for (int k = 0; k < 34; k++) wds[k] = (uint32_t)lane_value(prec, k);
        __builtin_memcpy(&p, wds, sizeof(p));
    }
    const int W = p.width, H = p.height;
    /* ---- the tile: luma rows / columns -8 .. S + 7 in pieces of eight samples (rows -4 .. S + 3 are used), chroma -8 .. S / 2 + 7 (rows -2 .. S / 2 + 1) */
    {
        const int npr = (S >> 3) + 2, nrow = S + 8, inv = mi355_inv20(npr);
        const uint8_t *src = mi355_global(p.data[0]);
        uint32_t v[3][4];
        int at[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int i = tid + FT_THREADS * u, r = mi355_div20(i, inv), pc = i - r * npr;
            const int ty = 4 + r, y = y0 - 8 + ty, x = x0 - 8 + 8 * pc;
            const bool ok = i < npr * nrow && y >= 0 && y < H && x >= 0 && x < W;
            at[u] = ok ? (ty * FT_Y + 8 * pc) * PX : -1;
            if (ok) {
                const uint8_t *q = src + (ptrdiff_t)y * p.linesize[0] + (ptrdiff_t)x * PX;
                if (WIDE) __builtin_memcpy(v[u], q, 16); else { __builtin_memcpy(v[u], q, 8); v[u][2] = v[u][3] = 0u; }
            }
        }
        const int nprc = (S >> 4) + 2, nrowc = (S >> 1) + 4, invc = mi355_inv20(nprc), Wc = W >> 1, Hc = H >> 1;
        uint32_t vc[2][4];
        int atc[2], fullc[2];
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const int i = tid, r = mi355_div20(i, invc), pc = i - r * nprc;
            const int ty = 6 + r, y = (y0 >> 1) - 8 + ty, x = (x0 >> 1) - 8 + 8 * pc;
            const bool ok = i < nprc * nrowc && y >= 0 && y < Hc && x >= 0 && x + 4 <= Wc;
            atc[pl] = ok ? (ty * FT_C + 8 * pc) * PX : -1;
            fullc[pl] = x + 8 <= Wc;                    /* a plane's width is a multiple of four: the last piece of a row may be half a piece */
            vc[pl][0] = vc[pl][1] = vc[pl][2] = vc[pl][3] = 0u;
            if (ok) {
                const uint8_t *q = mi355_global(p.data[1 + pl]) + (ptrdiff_t)y * p.linesize[1 + pl] + (ptrdiff_t)x * PX;
                if (fullc[pl]) { if (WIDE) __builtin_memcpy(vc[pl], q, 16); else __builtin_memcpy(vc[pl], q, 8); }
                else { if (WIDE) __builtin_memcpy(vc[pl], q, 8); else __builtin_memcpy(vc[pl], q, 4); }
            }
        }
        MI355_ISSUE_FENCE();
        /* the segments' parameters while the samples are on their way: this thread's candidate of the vertical and of the horizontal pass */
        if (VPASS) s_par[0][tid] = ft_params<0>(p, x0, y0, S, tid);
        s_par[1][tid] = ft_params<1>(p, x0, y0, S, tid);
#pragma unroll
        for (int u = 0; u < 3; u++)
            if (at[u] >= 0) {
                if (WIDE) *reinterpret_cast<uint4 *>(tile.y + at[u]) = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
                else *reinterpret_cast<uint2 *>(tile.y + at[u]) = make_uint2(v[u][0], v[u][1]);
            }
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
            if (atc[pl] >= 0) {
                if (WIDE) *reinterpret_cast<uint4 *>(tile.c[pl] + atc[pl]) = make_uint4(vc[pl][0], vc[pl][1], vc[pl][2], vc[pl][3]);
                else *reinterpret_cast<uint2 *>(tile.c[pl] + atc[pl]) = make_uint2(vc[pl][0], vc[pl][1]);
            }
    }
    __syncthreads();
    if (VPASS) {
        ft_deblock_pass<0, WIDE>(tile, s_par[0], bd, tid, s_act);
        __syncthreads();
    }
    ft_deblock_pass<1, WIDE>(tile, s_par[1], bd, tid, s_act);
    __syncthreads();
    /* ---- SAO: rounds 0 and 1 the luma region (every thread), round 2 the chroma regions (threads 0..127 Cb, 128..255 Cr) */
#pragma unroll 1
    for (int round = 0; round < 2; round++) {
        const bool chroma = round == 1;
        const int rec = chroma ? srec_c : srec_y;
#define SAO_REC(dw) ((uint32_t)lane_value(rec, (dw)))
        uint8_t *dst0 = mi355_global(reinterpret_cast<uint8_t *>((uintptr_t)SAO_REC(0) | ((uintptr_t)SAO_REC(1) << 32)));
        const int stride = (int)SAO_REC(4), np = (int)((SAO_REC(5) >> 8) & 0xFF);
        const uint32_t p0a = SAO_REC(6 + 5), p0b = SAO_REC(6 + 6), p0c = SAO_REC(6 + 7), p0d = SAO_REC(6 + 8);
        const int type = (int)((p0a >> 8) & 0xFF), RW = (int)(int16_t)(p0d & 0xFFFF), RH = (int)(int16_t)(p0d >> 16);
        uint32_t flags = p0b & 0x00FFFFFFu, types = 0;
        if (np > 1) { flags |= SAO_REC(15 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(15 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        if (np > 2) { flags |= SAO_REC(24 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(24 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        if (np > 3) { flags |= SAO_REC(33 + 6) & 0x00FFFFFFu; types |= ((SAO_REC(33 + 5) >> 8) & 0xFF) ^ (uint32_t)type; }
        const int32_t ov[5] = { (int32_t)SAO_REC(6), (int32_t)SAO_REC(7), (int32_t)SAO_REC(8), (int32_t)SAO_REC(9), (int32_t)SAO_REC(10) };
        bool fits = np >= 1 && (p0a & 0xFF) == 0 && p0c == 0 && types == 0 && (RW & 3) == 0;
        for (int e = 0; e < 5; e++) fits = fits && ov[e] > -128 && ov[e] < 128;
        if (type == 2) fits = fits && flags == 0 && (RW & 7) == 0;
        if (type == 1) fits = fits && (RW & 7) == 0;
        /* the region in the tile, this thread's first step and the step count's stride */
        const uint8_t *region = chroma ? tile.c[wave < 2 ? 0 : 1] + (8 * FT_C + 8) * PX : tile.y + (8 * FT_Y + 8) * PX;
        const int tp = (chroma ? FT_C : FT_Y) * PX, first = chroma ? (tid & 127) : tid, nth = chroma ? 128 : FT_THREADS;
        if (!fits) {
            /* a job outside the whole-region forms (pieces of different kinds, an edge that is restored, offsets beyond a byte): this entry point does not take it —
             * the block's component is left unwritten and the device says so (include/mi355_hevc_batch.h) */
            if (error_word && lane == 0) atomicOr(error_word, (uint32_t)MI355_ERR_FILTER_CTB_FORM);
            continue;
        }
        const int eo = (int)((p0a >> 16) & 0xFF), bp = (int)(p0a >> 24), bo = type == 2 ? (int)(p0b >> 24) : 0;
        if (type == 0) ft_copy_region<WIDE>(region, tp, dst0, stride, RW, RH, first, nth);
        else if (type == 1) ft_sao_region<WIDE, false, false>(region, tp, dst0, stride, RW, RH, eo, bp, ov, bd, 0, first, nth);
        else if (bo) ft_sao_region<WIDE, true, true>(region, tp, dst0, stride, RW, RH, eo, bp, ov, bd, bo, first, nth);
        else ft_sao_region<WIDE, true, false>(region, tp, dst0, stride, RW, RH, eo, bp, ov, bd, 0, first, nth);
#undef SAO_REC
    }
}

bool check(int bit_depth, const void *jobs, int n)
{
    /* bind(): the calling thread's device (mi355_set_device, else mi355_init's) — not whatever device the thread last used */
    if (!bind()) { std::fprintf(stderr, "mi355dsp: HEVC batch entry point without mi355_init(); no CPU fallback\n"); std::abort(); }
    return jobs && n > 0 && (bit_depth == 8 || bit_depth == 9 || bit_depth == 10);
}

}  // namespace

static_assert(sizeof(mi355_hevc_lf_job) == 32, "eight dwords per deblocking job (k_hevc_deblock_batch loads them as such)");

extern "C" int mi355_hevc_residual_batch_dev(const mi355_hevc_tu_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_residual_batch, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_intra_recon_blocks_dev(const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks,
                                                 const mi355_hevc_tu_job *d_tus, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_blocks, n) || !d_pics || !d_tus) return -1;
    hipLaunchKernelGGL(k_hevc_intra_recon_blocks, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_pics, d_blocks, d_tus, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_recon_level_dev(const mi355_hevc_mcpred_job *d_mc, int n_mc, const mi355_hevc_tu_job *d_tus, int n_tus,
                                          const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks,
                                          const mi355_hevc_tu_job *d_block_tus, int n_blocks, int bit_depth, void *stream)
{
    if (n_mc < 0 || n_tus < 0 || n_blocks < 0 || (n_mc && !d_mc) || (n_tus && !d_tus) || (n_blocks && (!d_pics || !d_blocks || !d_block_tus))) return -1;
    const int wgs = n_mc + (n_tus + 1) / 2 + n_blocks;
    if (!check(bit_depth, &wgs, wgs)) return -1;
    hipLaunchKernelGGL(k_hevc_recon_level, dim3((unsigned)wgs), dim3(64), 0, (hipStream_t)stream, d_mc, n_mc, d_tus, n_tus, d_pics, d_blocks, d_block_tus,
                       n_blocks, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_hevc_recon_levels_dev(const mi355_hevc_level *d_levels, int n_levels, int n_workgroups, const mi355_hevc_mcpred_job *d_mc, const mi355_hevc_tu_job *d_tus,
                                           const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks, const mi355_hevc_tu_job *d_block_tus,
                                           int bit_depth, void *stream)
{
    if (n_levels <= 0 || !d_levels) return -1;
    if (!check(bit_depth, d_levels, n_workgroups)) return -1;
    hipStream_t st = (hipStream_t)stream;
    uint32_t *sync = mi355::sync_words(st, 16);
    uint32_t *err = mi355::error_word();
    if (!sync || !err) return -4;
    if (hipMemsetAsync(sync, 0, 16 * sizeof(uint32_t), st) != hipSuccess) return -4;
    /* MI355_LEVELS_NAPS_MAX (developer / test hook): the bound of a wait; 0: nobody counts itself done, every wait runs out at once (tests/test_hevc_batch_*: the error word) */
    static const char *e = std::getenv("MI355_LEVELS_NAPS_MAX");
    const uint32_t naps_max = e && *e ? (uint32_t)std::strtoul(e, nullptr, 0) : LV_NAPS_MAX;
    hipLaunchKernelGGL(k_hevc_recon_levels, dim3((unsigned)n_workgroups), dim3(64), 0, st, d_levels, n_levels, d_mc, d_tus, d_pics, d_blocks, d_block_tus, bit_depth,
                       sync, err, naps_max);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_hevc_mc_batch_dev(const mi355_hevc_mc_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_mc_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_mcpred_batch_dev(const mi355_hevc_mcpred_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_mcpred_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_pred_batch_dev(const mi355_hevc_pred_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_pred_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_boundary_strengths_dev(const mi355_hevc_bs_picture *d_pics, int npics, int max_width, int max_height, void *stream)
{
    if (!ready() || !d_pics || npics <= 0 || max_width <= 0 || max_height <= 0) return -1;
    const int cx = (max_width + 3) / 4, cy = (max_height + 3) / 4;
    hipLaunchKernelGGL(k_hevc_boundary_strengths, dim3((unsigned)((cx * cy + 255) / 256), (unsigned)npics), dim3(256), 0, (hipStream_t)stream, d_pics, cx, cy);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_deblock_pictures_dev(const mi355_hevc_lf_picture *d_pics, int npics, int max_width, int max_height, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_pics, npics) || max_width <= 0 || max_height <= 0) return -1;
    const int lc = (max_width + 7) / 8, lr = (max_height + 7) / 8;
    static const int dirs = std::getenv("MI355_DEBLOCK_DIRS") ? std::atoi(std::getenv("MI355_DEBLOCK_DIRS")) : 3;      /* developer experiment: bit 0 vertical edges, bit 1 horizontal */
    for (int dir = 0; dir < 2; dir++) {
        if (!(dirs & (1 << dir))) continue;
        /* horizontal chroma segments start at x = -8: one more column may be needed */
        const int cc = dir ? (max_width + 8 + 15) / 16 : (max_width + 15) / 16, cr = (max_height + 15) / 16;
        const int lw = (lc * lr + 63) / 64, cw = (2 * cc * cr + 63) / 64;      /* a wave takes 64 candidate segments */
        const dim3 grid((unsigned)((lw + cw) * npics));
        if (dir == 0) hipLaunchKernelGGL(k_hevc_deblock_pictures<0>, grid, dim3(64), 0, (hipStream_t)stream, d_pics, lc, lr, cc, cr, lw, lw + cw, bit_depth);
        else hipLaunchKernelGGL(k_hevc_deblock_pictures<1>, grid, dim3(64), 0, (hipStream_t)stream, d_pics, lc, lr, cc, cr, lw, lw + cw, bit_depth);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_deblock_batch_dev(const mi355_hevc_lf_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_deblock_batch, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_intra_batch_dev(const mi355_hevc_intra_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_intra_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_intra_pred_blocks_dev(const mi355_hevc_intra_picture *d_pics, const mi355_hevc_intra_block *d_blocks, int n,
                                                int bit_depth, void *stream)
{
    if (!check(bit_depth, d_blocks, n) || !d_pics) return -1;
    hipLaunchKernelGGL(k_hevc_intra_blocks, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_pics, d_blocks, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_sao_ctbs_dev(const mi355_hevc_sao_ctb_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    const dim3 grid((unsigned)((n + SAO_CTB_JOBS - 1) / SAO_CTB_JOBS)), block(SAO_CTB_THREADS * SAO_CTB_JOBS);
    if (bit_depth > 8) hipLaunchKernelGGL(k_hevc_sao_ctbs<true>, grid, block, 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    else hipLaunchKernelGGL(k_hevc_sao_ctbs<false>, grid, block, 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_filter_ctbs_dev(const mi355_hevc_lf_picture *d_pics, const mi355_hevc_filter_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_sao_ctb_job *d_sao,
                                          int log2_ctb_size, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_ctbs, n_ctbs) || !d_pics || !d_sao || log2_ctb_size < 4 || log2_ctb_size > 6) return -1;
    uint32_t *err = mi355::error_word();
    if (!err) return -4;
    static const bool skip_v = std::getenv("MI355_FT_SKIP_V") != nullptr;      /* developer experiment: the vertical edges were filtered in the picture before (profiles/r06_experiments.md) */
    if (skip_v) {
        if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_filter_ctbs<true, false>), dim3((unsigned)n_ctbs), dim3(FT_THREADS), 0, (hipStream_t)stream, d_pics, d_ctbs, n_ctbs, d_sao, log2_ctb_size, bit_depth, err);
        else hipLaunchKernelGGL((k_hevc_filter_ctbs<false, false>), dim3((unsigned)n_ctbs), dim3(FT_THREADS), 0, (hipStream_t)stream, d_pics, d_ctbs, n_ctbs, d_sao, log2_ctb_size, bit_depth, err);
    } else if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_filter_ctbs<true, true>), dim3((unsigned)n_ctbs), dim3(FT_THREADS), 0, (hipStream_t)stream, d_pics, d_ctbs, n_ctbs, d_sao, log2_ctb_size, bit_depth, err);
    else hipLaunchKernelGGL((k_hevc_filter_ctbs<false, true>), dim3((unsigned)n_ctbs), dim3(FT_THREADS), 0, (hipStream_t)stream, d_pics, d_ctbs, n_ctbs, d_sao, log2_ctb_size, bit_depth, err);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_edge_emu_batch_dev(const mi355_edge_emu_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_edge_emu_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mi355_hevc_sao_batch_dev(const mi355_hevc_sao_job *d_jobs, int n, int bit_depth, void *stream)
{
    if (!check(bit_depth, d_jobs, n)) return -1;
    hipLaunchKernelGGL(k_hevc_sao_batch, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, d_jobs, n, bit_depth);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
