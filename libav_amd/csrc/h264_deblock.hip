/*
 * h264_deblock.hip — Tier-2: the in-loop deblocking filter of a batch of pictures (C ABI: mi355_h264_deblock_dev,
 * include/mi355_h264_frame.h).  Reference behaviour restated: ff_h264_filter_mb + filter_mb_dir + check_mv
 * (h264_loopfilter.c:442-847), fill_filter_caches (h264_slice.c:2056-2196), the edge filters of h264dsp_template.c:104-328,
 * in the reference's raster order (loop_filter, h264_slice.c:2198).
 */
#include <cstdlib>
#include <vector>
#include "h264_frame_dev.h"

using namespace mi355;

namespace {

#ifdef MI355_PROF   /* developer instrumentation (tools/prof_deblock.sh): per-phase shader-clock totals of the first blocks */
__device__ unsigned long long g_prof[16];
#define PROF_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define PROF_MARK(i) do { } while (0)
#endif

/* ------------------------------------------------------------------------- */
/* deblocking                                                                   */
/* ------------------------------------------------------------------------- */
/* Tables 8-16 / 8-17 of the standard (alpha', beta', tC0 for bS 1..3); indices clamp to 0..51,
 * which is what the reference's 52-entry guard bands implement (h264_loopfilter.c:41-101) */
__device__ const uint8_t k_alpha[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
    32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
__device__ const uint8_t k_beta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
    9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
__device__ const uint8_t k_tc0[52][3] = {
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},
    {1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},
    {3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},{9,12,18},{10,13,20},
    {11,15,23},{13,17,25} };

/* motion of one 4x4 block as the loop filter sees it: picture identity per list (-1 = unused, what
 * ref_cache holds after the ref2frm mapping, h264_slice.c:2023-2029) and the packed mv words */
struct BlkMotion {
    int ref[2];
    uint32_t mv[2];
};
__device__ __forceinline__ bool mv_far(uint32_t a, uint32_t b)
{
    return iabs((int16_t)(a & 0xFFFF) - (int16_t)(b & 0xFFFF)) >= 4 || iabs((int16_t)(a >> 16) - (int16_t)(b >> 16)) >= 4;
}
/* check_mv, h264_loopfilter.c:442-470, frame macroblocks (mvy_limit 4) */
__device__ inline int check_mv(const BlkMotion &p, const BlkMotion &q, int list_count)
{
    bool v = p.ref[0] != q.ref[0];
    if (!v && p.ref[0] != -1) v = mv_far(p.mv[0], q.mv[0]);
    if (list_count == 2) {
        if (!v) v = p.ref[1] != q.ref[1] || mv_far(p.mv[1], q.mv[1]);
        if (v) {
            if (p.ref[0] != q.ref[1] || p.ref[1] != q.ref[0]) return 1;
            return mv_far(p.mv[0], q.mv[1]) || mv_far(p.mv[1], q.mv[0]);
        }
    }
    return v;
}
__device__ __forceinline__ int ref_identity(const mi355_h264_mb &m, int list, int x4, int y4)
{
    if (m.mb_type & MI355_MB_INTRA) return -1;
    const int r = m.u.inter.ref_pic[list][(x4 >> 1) + 2 * (y4 >> 1)];
    return r == 0xFF ? -1 : r;
}

/* One wavefront deblocks a band of four macroblock rows of one picture, walking left to right:
 * lanes 16g..16g+15 own row 4*band+g and at step t work on macroblock x = t - 2g, so the
 * reference's raster dependencies (left, top, top-right: h264_slice.c:2198) are met by lock-step
 * execution inside the wave — no flags, no per-diagonal launches.
 *
 * Samples move between HBM and LDS in CHUNKS of DCH macroblocks per row (16*DCH-byte luma / 8*DCH-byte
 * chroma row pieces, adjacent lanes on adjacent addresses): group g's chunk c covers macroblocks
 * DCH*c-2g .. DCH*c-2g+DCH-1, so all four groups change chunk at the same step.  A chunk is loaded from `recon`
 * when its first macroblock comes up, filtered in place in the LDS tile, and written to `dst` one
 * step after its last macroblock (by then the next macroblock's left edge has patched its last
 * columns).  Rows above a group's macroblock come from the tile of the group above (same wave, two
 * steps ahead) or, for the band's first row, from `dst` as the previous band left it; a group does
 * not write the bottom three rows that the group below will still filter and write itself.
 *
 * Side information never passes through LDS: every lane fetches the record fields and the four motion vectors its
 * boundary-strength role needs straight from memory one step ahead (unpredicated loads from clamped addresses, so the
 * compiler's wait counts stay exact), computes ONE strength per direction — role (segment = l >> 2, edge = l & 3) —
 * and the four strengths of a line meet through quad broadcasts (DPP).  alpha / beta / tc0 (tables 8-16 / 8-17 in LDS)
 * are looked up once per (component, edge kind) by nine lanes of the group and shared through LDS; the per-edge tc0 is a
 * byte select (v_perm_b32) on the packed strengths.
 *
 * Inside a step a lane holds one luma ROW (4 samples of the left neighbour + 16) and one chroma row
 * in registers for the vertical edges; then the tile is read by COLUMN for the horizontal edges.  Each edge evaluates
 * the filterSamplesFlag conditions first and leaves when no line of the wave passes (the reference's own per-line
 * `continue`, h264dsp_template.c:117-121, taken at wave granularity); results are bit-identical either way. */
#ifndef MI355_DCH_LOG
#define MI355_DCH_LOG 2
#endif
constexpr int DCH_LOG = MI355_DCH_LOG, DCH = 1 << DCH_LOG;    /* macroblocks per chunk */
#ifndef MI355_DPAD
#define MI355_DPAD 8
#endif
constexpr int DY_PITCH = 16 * DCH + MI355_DPAD;   /* + 8: the sixteen rows of a 16-lane, 8-byte access fall on 32 different banks */
constexpr int DC_PITCH = 8 * DCH + MI355_DPAD;
constexpr int DIO_ROWS = 16 / DCH;                /* rows one 16-lane chunk access covers */
constexpr int DCH_ISSUE = DCH >= 4 ? 1 : DCH - 1;  /* position in a chunk at which the next chunk's loads are issued */
/* Small-batch form (k_deblock_bands): waves per workgroup, and by how many steps the wave of band b trails the wave of band
 * b - 1.  Band b's group 0 requests the rows above its chunk c (macroblocks 4c .. 4c + 3) at its step 4(c - 1) + DCH_ISSUE;
 * the wave above has them final and written once its group 3 (macroblock x at step x + 6) has filtered macroblock 4c + 4
 * — whose left edge still changes columns 13..15 of macroblock 4c + 3 — and flushed that chunk (c + 2 of that group, at step
 * 4(c + 2) + 5 = 4c + 13); the barrier at the end of that step publishes the stores.  So the wave above must be at least
 * (4c + 14) - (4c - 4 + DCH_ISSUE) = 18 - DCH_ISSUE steps ahead.  The tiles of KW bands share the CU's 160 KB of LDS with
 * the other workgroups on it: 8 / KW workgroups (pictures) per CU. */
/* in general: the last macroblock of the lower band's chunk c lies in chunk c + (DCH + 5) / DCH of the upper wave's group 3 */
constexpr int DEBLOCK_LAG_MIN = DCH * ((DCH + 5) / DCH + 2) + 2 - DCH_ISSUE;
constexpr int DEBLOCK_LAG = DEBLOCK_LAG_MIN + (DCH == 4 ? 3 : 1);
static_assert((DCH == 4 && DEBLOCK_LAG_MIN == 17) || (DCH == 2 && DEBLOCK_LAG_MIN == 11), "lag of the small-batch deblocking form");
constexpr int DEBLOCK_WAVES_PER_CU = DCH == 4 ? 8 : 12;     /* band tiles (waves) a CU holds: LDS with chunks of four, registers (168) with chunks of two */
struct DeblockLds {
    uint8_t y[4][2][20][DY_PITCH];      /* [group][chunk parity]: rows -4..15 of DCH macroblocks */
    uint8_t c[4][2][2][10][DC_PITCH];   /* [group][chunk parity][plane]: rows -2..7 */
    /* [group][component * 3 + kind]: component luma / Cb / Cr, kind inner / left / top edge:
     * word 0 = alpha | beta << 8, word 1 = tc0 by (bS & 3): bytes (0, tc0[bS 1], tc0[bS 2], tc0[bS 3]), chroma + 1 */
    uint32_t parm[4][9][2];
    uint8_t t_alpha[52], t_beta[52];
    uint32_t t_tc0[52];                 /* (0, tc0[0], tc0[1], tc0[2]) per indexA */
};

/* The fields of a record the filter looks at */
struct MbInfo {
    uint32_t type, nnz, w2, w3, w11, ref0, ref1;
    __device__ __forceinline__ int qp() const { return (int8_t)((w2 >> 16) & 0xFF); }
    __device__ __forceinline__ int flags() const { return (int)(w2 >> 24); }
    __device__ __forceinline__ int alpha_off() const { return (int8_t)(w3 & 0xFF); }
    __device__ __forceinline__ int beta_off() const { return (int8_t)((w3 >> 8) & 0xFF); }
    __device__ __forceinline__ int slice_id() const { return (int)(w11 & 0xFF); }
    __device__ __forceinline__ int qpc(int p) const { return (int)((w11 >> (16 + 8 * p)) & 0xFF); }
};
/* words 0-3 (0-2 for a neighbour, whose slice offsets are not looked at) and 11-13 of the record at byte offset `off` of
 * `base`: two loads, no predicate.  No loaded word may be dead: the register of a dead word is handed to something else,
 * and the first write to it then waits for this load (a memory latency per step when that something is set per step). */
template <bool W3>
__device__ __forceinline__ MbInfo mb_info_load(const uint8_t *base, uint32_t off)
{
#ifdef MI355_HIP_EMU_H
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + off);
    return MbInfo{ w[0], w[1], w[2], W3 ? w[3] : 0u, w[11], w[12], w[13] };
#else
    typedef uint32_t u32x3 __attribute__((vector_size(12)));
    const u32x3 b = *reinterpret_cast<const u32x3 *>(base + off + 44);
    if (W3) {
        const mi355_u32x4 a = *reinterpret_cast<const mi355_u32x4 *>(base + off);
        return MbInfo{ a[0], a[1], a[2], a[3], b[0], b[1], b[2] };
    }
    const u32x3 a = *reinterpret_cast<const u32x3 *>(base + off);
    return MbInfo{ a[0], a[1], a[2], 0u, b[0], b[1], b[2] };
#endif
}

/* check_mv (h264_loopfilter.c:442-470, frame macroblocks, mvy_limit 4) on raw reference bytes (0xFF = unused; the
 * comparisons are equalities, and intra macroblocks never get here) */
__device__ __forceinline__ bool check_mv_raw(uint32_t rp0, uint32_t rq0, uint32_t rp1, uint32_t rq1, const uint32_t mp[2], const uint32_t mq[2], bool two_lists, uint32_t far)
{
    bool v = rp0 != rq0 || (rp0 != 0xFF && pk_absdiff_far(mp[0], mq[0], far));
    if (two_lists) {
        v = v || rp1 != rq1 || pk_absdiff_far(mp[1], mq[1], far);
        const bool cross = rp0 != rq1 || rp1 != rq0 || pk_absdiff_far(mp[0], mq[1], far) || pk_absdiff_far(mp[1], mq[0], far);
        v = v && cross;
    }
    return v;
}
/* lane constants of a boundary-strength role in one direction: nnz bits and reference-byte shifts of the block on this
 * side of the edge (p) and across it (q: in the neighbour macroblock when the edge is the macroblock edge) */
struct BsRole {
    uint32_t pbit, qbit, psh, qsh;
};
/* one boundary strength, filter_mb_dir h264_loopfilter.c:472-713.  q*: the macroblock across the edge (the
 * neighbour for edge 0, this one otherwise) */
__device__ __forceinline__ uint32_t bs_role(const MbInfo &h, const MbInfo &nb, bool outer, bool odd, bool enabled, const BsRole &r,
                                            const uint32_t mp[2], const uint32_t mq[2], bool two_lists, uint32_t far, uint32_t intra_edge)
{
    const uint32_t q_type = outer ? nb.type : h.type, q_nnz = outer ? nb.nnz : h.nnz;
    const uint32_t q_ref0 = outer ? nb.ref0 : h.ref0, q_ref1 = outer ? nb.ref1 : h.ref1;
    const bool any_intra = ((h.type | q_type) & MI355_MB_INTRA) != 0;
    const bool coded = ((h.nnz & r.pbit) | (q_nnz & r.qbit)) != 0;
    const uint32_t rp0 = (h.ref0 >> r.psh) & 0xFF, rq0 = (q_ref0 >> r.qsh) & 0xFF;
    const uint32_t rp1 = (h.ref1 >> r.psh) & 0xFF, rq1 = (q_ref1 >> r.qsh) & 0xFF;
    const uint32_t mvbs = check_mv_raw(rp0, rq0, rp1, rq1, mp, mq, two_lists, far) ? 1u : 0u;
    uint32_t bs = any_intra ? (outer ? intra_edge : 3u) : (coded ? 2u : mvbs);      /* intra_edge: 4, or 3 on the horizontal macroblock edges of a field picture */
    if (!enabled || (!outer && odd && (h.type & MI355_MB_8x8DCT))) bs = 0;
    return bs;
}

/* Luma edge, one line: p3 p2 p1 p0 | q0 q1 q2 q3.  bS 1..3: h264_loop_filter_luma (h264dsp_template.c:104-150) with the
 * line's conditions folded into the clipping bounds (tc = 0 leaves a sample as it is); bS 4
 * (h264_loop_filter_luma_intra :165-210) only where a lane of the wave has it (MAY_INTRA: macroblock edges only).
 * Returns 0 when no line of the WAVE passes the conditions (nothing changed), else 1, or 2 when the bS 4 filter ran. */
template <bool MAY_INTRA>
__device__ __forceinline__ int luma_line(int p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int q3,
                                         int bs, int alpha, int beta, int tc0)
{
    const bool f = bs != 0 && absdiff8(p0, q0) < alpha && absdiff8(p1, p0) < beta && absdiff8(q1, q0) < beta;
    if (!__any(f)) return 0;
    const bool ap = absdiff8(p2, p0) < beta, aq = absdiff8(q2, q0) < beta;
    const bool fn = f && bs < 4;
    const int avg = (p0 + q0 + 1) >> 1;
    const int tp = fn && ap ? tc0 : 0, tq = fn && aq ? tc0 : 0, tc = fn ? tc0 + (int)ap + (int)aq : 0;
    const int np1 = p1 + med3i(((p2 + avg) >> 1) - p1, -tp, tp), nq1 = q1 + med3i(((q2 + avg) >> 1) - q1, -tq, tq);
    const int delta = med3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    const int P0 = p0, P1 = p1, P2 = p2, Q0 = q0, Q1 = q1, Q2 = q2;
    p1 = np1; q1 = nq1;
    p0 = clip_u8(P0 + delta);
    q0 = clip_u8(Q0 - delta);
    if (MAY_INTRA) {
        const bool fi = f && bs == 4;
        if (__any(fi)) {
            const bool strong = absdiff8(P0, Q0) < ((alpha >> 2) + 2), sp = strong && ap, sq = strong && aq;
            const int wp0 = (2 * P1 + P0 + Q1 + 2) >> 2, wq0 = (2 * Q1 + Q0 + P1 + 2) >> 2;
            const int s4 = P0 + Q0 + 4;
            const int ip0 = sp ? (P2 + 2 * P1 + P0 + Q0 + Q1 + s4) >> 3 : wp0;
            const int ip1 = sp ? (P2 + P1 + P0 + Q0 + 2) >> 2 : P1;
            const int ip2 = sp ? (2 * p3 + 3 * P2 + P1 + s4) >> 3 : P2;
            const int iq0 = sq ? (P1 + P0 + Q0 + 2 * Q1 + Q2 + s4) >> 3 : wq0;
            const int iq1 = sq ? (P0 + Q0 + Q1 + Q2 + 2) >> 2 : Q1;
            const int iq2 = sq ? (2 * q3 + 3 * Q2 + Q1 + s4) >> 3 : Q2;
            p0 = fi ? ip0 : p0; p1 = fi ? ip1 : p1; p2 = fi ? ip2 : p2;
            q0 = fi ? iq0 : q0; q1 = fi ? iq1 : q1; q2 = fi ? iq2 : q2;
            return 2;
        }
    }
    return 1;
}
/* Chroma edge, one line: p1 p0 | q0 q1: h264_loop_filter_chroma / _intra (h264dsp_template.c:212-265); `tc1` is the
 * caller's tc0 + 1 (h264_loopfilter.c:126-129).  Returns whether any line of the wave changed. */
__device__ __forceinline__ bool chroma_line(int p1, int &p0, int &q0, int q1, int bs, int alpha, int beta, int tc1)
{
    const bool f = bs != 0 && absdiff8(p0, q0) < alpha && absdiff8(p1, p0) < beta && absdiff8(q1, q0) < beta;
    if (!__any(f)) return false;
    const int tc = f && bs < 4 ? tc1 : 0;
    const int delta = med3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    const bool fi = f && bs == 4;
    const int np0 = fi ? (2 * p1 + p0 + q1 + 2) >> 2 : clip_u8(p0 + delta);
    const int nq0 = fi ? (2 * q1 + q0 + p1 + 2) >> 2 : clip_u8(q0 - delta);
    p0 = np0; q0 = nq0;
    return true;
}
/* the same on a row held as dwords: P = samples -4..-1 of the edge, Q = samples 0..3 (memory order) */
template <bool MAY_INTRA>
__device__ __forceinline__ bool luma_row_edge(uint32_t &P, uint32_t &Q, int bs, int alpha, int beta, int tc0)
{
    int p3 = P & 0xFF, p2 = (P >> 8) & 0xFF, p1 = (P >> 16) & 0xFF, p0 = P >> 24;
    int q0 = Q & 0xFF, q1 = (Q >> 8) & 0xFF, q2 = (Q >> 16) & 0xFF, q3 = Q >> 24;
    const int r = luma_line<MAY_INTRA>(p3, p2, p1, p0, q0, q1, q2, q3, bs, alpha, beta, tc0);
    if (!r) return false;
    P = (uint32_t)p3 | ((uint32_t)p2 << 8) | ((uint32_t)p1 << 16) | ((uint32_t)p0 << 24);
    Q = (uint32_t)q0 | ((uint32_t)q1 << 8) | ((uint32_t)q2 << 16) | ((uint32_t)q3 << 24);
    return true;
}
__device__ __forceinline__ bool chroma_row_edge(uint32_t &P, uint32_t &Q, int bs, int alpha, int beta, int tc1)
{
    int p0 = P >> 24, q0 = Q & 0xFF;
    if (!chroma_line((P >> 16) & 0xFF, p0, q0, (Q >> 8) & 0xFF, bs, alpha, beta, tc1)) return false;
    P = (P & 0x00FFFFFFu) | ((uint32_t)p0 << 24);
    Q = (Q & 0xFFFFFF00u) | (uint32_t)q0;
    return true;
}

/* KW: waves per workgroup.  1 = a workgroup is one wave and one band (the throughput form: one launch per band, thousands of
 * pictures per launch).  KW > 1 = the small-batch form: KW waves of a workgroup walk KW consecutive bands of one picture
 * in the same launch, wave w starting DEBLOCK_LAG steps after wave w - 1, all waves meeting at a workgroup barrier after
 * every step: a band reads the rows above it (written by the wave above) only after that wave has written them and the
 * barrier's fence has made them visible — see DEBLOCK_LAG. */
/* TILED: recon and dst are macroblock-tiled surfaces (mi355_h264_frame.h): a lane's share of a chunk is then four consecutive
 * rows of ONE macroblock (64 contiguous bytes per four lanes, whole cache lines per macroblock) instead of one row piece of
 * each of four macroblocks; the LDS tiles and everything between load and store are the same. */
template <bool TWO_LISTS, int KW, bool TILED>     /* TWO_LISTS: list-1 vectors exist (B pictures): a compile-time switch, so that a P picture carries no list-1 state at all */
__device__ __forceinline__ void deblock_band(DeblockLds &s, const mi355_h264_frame &fr, int band, int wave)
{
    const int lane = lane_id(), g = lane >> 4, l = lane & 15;
    const int mb_y = 4 * band + g, W = fr.mb_width;
    const bool row_ok = mb_y < fr.mb_height;
    /* a field picture (PAFF): vertical vector limit 2 instead of 4 (h264_loopfilter.c:723), strength 3 on horizontal intra macroblock edges */
    const bool field = uniform(fr.field_picture) != 0;
    const uint32_t mv_far = field ? 0xFFFEFFFCu : 0xFFFCFFFCu;
    const int rs = fr.recon_stride[0], rcs = fr.recon_stride[1], ds = fr.dst_stride[0], dcs = fr.dst_stride[1];
    const int cp = l >> 3, cr = l & 7;                       /* this lane's chroma plane and row / column */
    constexpr bool two_lists = TWO_LISTS;                    /* sl->list_count == 2 exactly when list-1 vectors exist */
    const int nsteps = W + 6;
    const bool has_t = row_ok && mb_y > 0;
    /* the group below (same wave) filters and writes this row's bottom three luma rows / last chroma row */
    const bool below = g < 3 && mb_y + 1 < fr.mb_height;
    /* 16-byte (luma) / 8-byte (chroma) pieces can move as one access when pointers and strides allow */
    const bool al16 = TILED || ((reinterpret_cast<uintptr_t>(mi355_global(fr.recon[0])) | reinterpret_cast<uintptr_t>(mi355_global(fr.dst[0])) | (uintptr_t)rs | (uintptr_t)ds) & 15) == 0;
    const bool al8 = TILED || ((reinterpret_cast<uintptr_t>(mi355_global(fr.recon[1])) | reinterpret_cast<uintptr_t>(mi355_global(fr.recon[2])) | reinterpret_cast<uintptr_t>(mi355_global(fr.dst[1])) |
                       reinterpret_cast<uintptr_t>(mi355_global(fr.dst[2])) | (uintptr_t)rcs | (uintptr_t)dcs) & 7) == 0;
    if (lane < 52) {
        s.t_alpha[lane] = k_alpha[lane]; s.t_beta[lane] = k_beta[lane];
        s.t_tc0[lane] = ((uint32_t)k_tc0[lane][0] << 8) | ((uint32_t)k_tc0[lane][1] << 16) | ((uint32_t)k_tc0[lane][2] << 24);
    }
    /* ---- boundary-strength role of this lane: segment l >> 2 of edge l & 3, in both directions ---------------- */
    const int seg = l >> 2, edge = l & 3;
    const bool outer = edge == 0, odd = (edge & 1) != 0;
    /* dir 0 (vertical edges): p = block (edge, seg), q = (edge - 1, seg) or the left neighbour's (3, seg);
     * dir 1 (horizontal):     p = block (seg, edge), q = (seg, edge - 1) or the top neighbour's (seg, 3) */
    const int qe = outer ? 3 : edge - 1;
    const BsRole r0{ 1u << blk_index(edge, seg), 1u << blk_index(qe, seg), 8u * ((edge >> 1) + 2 * (seg >> 1)), 8u * ((qe >> 1) + 2 * (seg >> 1)) };
    const BsRole r1{ 1u << blk_index(seg, edge), 1u << blk_index(seg, qe), 8u * ((seg >> 1) + 2 * (edge >> 1)), 8u * ((seg >> 1) + 2 * (qe >> 1)) };
    /* vector offsets (bytes) from the macroblock's first vector; the neighbours' are the 16 vectors before / W * 16 before */
    const int o_p0 = 4 * (edge + 4 * seg), o_q0 = outer ? 4 * (3 + 4 * seg) - 64 : 4 * (edge - 1 + 4 * seg);
    const int o_p1 = 4 * (seg + 4 * edge), o_q1 = outer ? 4 * (seg + 12) - 64 * W : 4 * (seg + 4 * (edge - 1));
    const uint8_t *const rec_base = reinterpret_cast<const uint8_t *>(mi355_global(fr.mb));
    const uint8_t *const mv_base0 = reinterpret_cast<const uint8_t *>(mi355_global(fr.mv[0]));
    const uint8_t *const mv_base1 = two_lists ? reinterpret_cast<const uint8_t *>(mi355_global(fr.mv[1])) : mv_base0;
    /* what a lane fetches for one macroblock ahead of time (records and vectors: never written by the filter) */
    struct Pre {
        MbInfo h, ht;
        uint32_t p0[2], q0[2], p1[2], q1[2];
    };
    auto prefetch = [&](Pre &p, int x) {
        const bool ok = row_ok && x >= 0 && x < W;
        const uint32_t xy = ok ? (uint32_t)(mb_y * W + x) : 0u;
        const uint32_t roff = xy * 64u, toff = ok && has_t ? roff - 64u * (uint32_t)W : roff;
        p.h = mb_info_load<true>(rec_base, roff);
        p.ht = mb_info_load<false>(rec_base, toff);
        /* clamped addresses: a neighbour that does not exist reads this macroblock's own vector (its strength is masked) */
        const uint32_t a_p0 = roff + (uint32_t)o_p0, a_p1 = roff + (uint32_t)o_p1;
        const uint32_t a_q0 = (ok && x > 0) || !outer ? roff + (uint32_t)o_q0 : a_p0;
        const uint32_t a_q1 = (ok && has_t) || !outer ? roff + (uint32_t)o_q1 : a_p1;
        p.p0[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_p0); p.q0[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_q0);
        p.p1[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_p1); p.q1[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_q1);
        p.p0[1] = p.q0[1] = p.p1[1] = p.q1[1] = 0;           /* constants without list 1 (no register, no write) */
        if (two_lists) {
            p.p0[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_p0); p.q0[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_q0);
            p.p1[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_p1); p.q1[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_q1);
        }
    };
    /* chunk I/O roles of a lane: piece p of a row pair */
    const int io_p = TILED ? l >> (4 - DCH_LOG) : l & (DCH - 1), io_r = TILED ? l & (DIO_ROWS - 1) : l >> DCH_LOG;
    /* plane pointers once, in scalar registers: a per-lane fetch from the descriptor inside the chunk functions would
     * put a dependent load (and a wait for everything in flight) in front of every access */
    const uint8_t *const recon_cb = mi355_global(fr.recon[1]), *const recon_cr = mi355_global(fr.recon[2]);
    uint8_t *const dst_cb = mi355_global(fr.dst[1]), *const dst_cr = mi355_global(fr.dst[2]);
    uint8_t *const dst_y = mi355_global(fr.dst[0]) + (ptrdiff_t)(row_ok ? mb_y : 0) * 16 * ds;

    /* Chunk c (macroblocks DCH*c - 2g .. + DCH-1) of this group's row: the loads are issued a few steps before the
     * chunk is needed and land in registers (`issue_chunk`); `commit_chunk` moves them into tile parity c & 1 when the
     * walk reaches the chunk, so the memory latency hides behind the steps in between. */
    constexpr int NY = 16 / DIO_ROWS, NTOP = (4 + DIO_ROWS - 1) / DIO_ROWS;   /* accesses for 16 rows / for the 4 rows above */
    uint4 vy[NY], ty[NTOP];
    uint2 vc[NY], tc[NTOP];
    /* All loads are unconditional, from clamped positions: a macroblock outside the row reads one that exists (its tile
     * is never filtered and never written back), and the lanes of groups 1-3 repeat group 0's addresses for the rows
     * above the band (same cache lines, no extra traffic; only group 0 keeps them).  A predicated load costs a mask
     * sequence and a zero fill per access and makes the compiler's wait counts inexact. */
    const int mb_yc = row_ok ? mb_y : fr.mb_height - 1;
    const uint8_t *const recon_yc = mi355_global(fr.recon[0]) + (ptrdiff_t)mb_yc * 16 * rs;
    /* first row above the band (row 0 when there is none).  A band below the picture (a smaller picture in a batch of mixed
     * sizes, or the tail of a multi-band workgroup) reads where the picture's last band would: in bounds, never used */
    const int band_c = imin(band, (fr.mb_height - 1) >> 2);
    const int top_y0 = band_c > 0 ? 4 * band_c * 16 - 4 : 0, top_c0 = band_c > 0 ? 4 * band_c * 8 - 2 : 0;
    uint8_t *const dst_y0 = mi355_global(fr.dst[0]);
    const uint8_t *const recon_y0 = mi355_global(fr.recon[0]);
    const int top_mb = band_c > 0 ? 4 * band_c - 1 : 0;
    auto issue_chunk = [&](int c) {
        const int xr = DCH * c - 2 * g + io_p, x = xr < 0 ? 0 : (xr < W ? xr : W - 1);
        const int xt0 = DCH * c + io_p, xt = xt0 < W ? xt0 : W - 1;          /* group 0's macroblock */
#pragma unroll
        for (int it = 0; it < NY; it++) {
            const int row = DIO_ROWS * it + io_r;            /* luma row; as chroma: plane = row >> 3, row & 7 */
            if (TILED) {
                vy[it] = ld16(recon_y0 + tile_y_off(x, mb_yc, rs) + 16 * row, true);
                vc[it] = ld8(recon_cb + tile_c_off(x, mb_yc, rcs) + 8 * row, true);         /* Cb rows 0..7, Cr rows 8..15 of the chroma tile */
                continue;
            }
            vy[it] = ld16(recon_yc + (uint32_t)(__mul24(row, rs) + x * 16), al16);
            vc[it] = ld8(((row >> 3) ? recon_cr : recon_cb) + (uint32_t)(__mul24(mb_yc * 8 + (row & 7), rcs) + x * 8), al8);
        }
#pragma unroll
        for (int it = 0; it < NTOP; it++) {
            const int row = (DIO_ROWS * it + io_r) & 3;      /* 0..3: luma rows -4..-1; chroma: plane = row >> 1, row -2 + (row & 1) */
            if (TILED) {
                /* the last four luma rows / last two rows of each chroma plane of the tile above the band (tile 0 when there is none) */
                ty[it] = ld16(dst_y0 + tile_y_off(xt, top_mb, ds) + 16 * (12 + row), true);
                tc[it] = ld8(dst_cb + tile_c_off(xt, top_mb, dcs) + 64 * (row >> 1) + 8 * (6 + (row & 1)), true);
                continue;
            }
            ty[it] = ld16(dst_y0 + (uint32_t)(__mul24(top_y0 + row, ds) + xt * 16), al16);
            tc[it] = ld8(((row >> 1) ? dst_cr : dst_cb) + (uint32_t)(__mul24(top_c0 + (row & 1), dcs) + xt * 8), al8);
        }
    };
    auto commit_chunk = [&](int c) {
        const int b = c & 1;
#pragma unroll
        for (int it = 0; it < NY; it++) {
            const int row = DIO_ROWS * it + io_r;
            lds16(&s.y[g][b][4 + row][16 * io_p], vy[it]);
            *reinterpret_cast<uint2 *>(&s.c[g][b][row >> 3][2 + (row & 7)][8 * io_p]) = vc[it];
        }
        if (g == 0) {
#pragma unroll
            for (int it = 0; it < NTOP; it++) {
                const int row = DIO_ROWS * it + io_r;
                if (row < 4) {
                    lds16(&s.y[g][b][row][16 * io_p], ty[it]);
                    *reinterpret_cast<uint2 *>(&s.c[g][b][row >> 1][row & 1][8 * io_p]) = tc[it];
                }
            }
        }
    };
    /* write chunk c back to `dst` */
    auto flush_chunk = [&](int c) {
        const int x = DCH * c - 2 * g + io_p, b = c & 1;
        const bool ok = row_ok && x >= 0 && x < W;
        const int y_first = has_t ? 1 : 4, y_last = below ? 16 : 19;      /* tile rows: -3.. / 0..  up to 12 / 15 */
        const int c_first = has_t ? 1 : 2, c_last = below ? 8 : 9;
        constexpr int NF = (20 + DIO_ROWS - 1) / DIO_ROWS;                   /* 20 luma tile rows; 2 x 10 chroma tile rows */
#pragma unroll
        for (int it = 0; it < NF; it++) {
            const int row = DIO_ROWS * it + io_r;
            const int plane = row >= 10, crow = row - 10 * plane;
            if (TILED) {
                /* tile rows -4..-1 (chroma -2, -1) are the last rows of the tile above */
                if (ok && row >= y_first && row <= y_last)
                    st16(dst_y0 + (row >= 4 ? tile_y_off(x, mb_y, ds) + 16 * (row - 4) : tile_y_off(x, mb_y - 1, ds) + 16 * (12 + row)), lds16(&s.y[g][b][row][16 * io_p]), true);
                if (ok && crow >= c_first && crow <= c_last)
                    st8(dst_cb + (crow >= 2 ? tile_c_off(x, mb_y, dcs) + 8 * (crow - 2) : tile_c_off(x, mb_y - 1, dcs) + 8 * (6 + crow)) + 64 * plane,
                        *reinterpret_cast<const uint2 *>(&s.c[g][b][plane][crow][8 * io_p]), true);
                continue;
            }
            if (ok && row >= y_first && row <= y_last)
                st16(dst_y + (ptrdiff_t)(row - 4) * ds + x * 16, lds16(&s.y[g][b][row][16 * io_p]), al16);
            if (ok && crow >= c_first && crow <= c_last)
                st8((plane ? dst_cr : dst_cb) + (ptrdiff_t)(mb_y * 8 + crow - 2) * dcs + x * 8, *reinterpret_cast<const uint2 *>(&s.c[g][b][plane][crow][8 * io_p]), al8);
        }
    };

#ifdef MI355_PROF
    unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_readcyclecounter();
#endif
    Pre pre = {};
    MbInfo hl = {};                                          /* the left neighbour's fields: last step's macroblock */
    int flushed = 0;                                         /* chunks already written back */
    auto start = [&]() __attribute__((always_inline)) {
        issue_chunk(0);
        prefetch(pre, -2 * g);
        hl = pre.h;
    };
    auto step = [&](int t) __attribute__((always_inline)) {
        const int mb_x = t - 2 * g;
        const int ck = t >> DCH_LOG, j = t & (DCH - 1), b = ck & 1;   /* chunk, position in it, tile parity */
        const bool valid = row_ok && mb_x >= 0 && mb_x < W;
        const Pre cur = pre;
        PROF_MARK(0);
        /* ---- chunk turnover ---------------------------------------------------------------------- */
        if (j == 1 && ck >= 1) {                             /* the previous chunk got its last left-edge patch in step t-1 */
            flush_chunk(ck - 1);
            flushed = ck;
        }
        PROF_MARK(7);
        if (j == 0) commit_chunk(ck);
        PROF_MARK(6);
        if (j == DCH_ISSUE) issue_chunk(ck + 1);             /* after the flush above: the stores go first */
        PROF_MARK(5);
        /* ---- next step's records and vectors ------------------------------------------------------- */
        prefetch(pre, mb_x + 1);
        PROF_MARK(1);
        MI355_WAVE_SYNC();                                   /* the chunk committed above is visible */
        /* ---- rows above from the group above ------------------------------------------------------- */
        if (g > 0 && valid) {
            /* macroblock x of the row above sits at position (t-2) % DCH of that group's chunk (t-2) / DCH */
            const int jb = (t - 2) & (DCH - 1), bb = ((t - 2) >> DCH_LOG) & 1;
            *reinterpret_cast<uint32_t *>(&s.y[g][b][l >> 2][16 * j + 4 * (l & 3)]) =
                *reinterpret_cast<const uint32_t *>(&s.y[g - 1][bb][16 + (l >> 2)][16 * jb + 4 * (l & 3)]);
            if (l < 8)
                *reinterpret_cast<uint32_t *>(&s.c[g][b][l >> 2][(l >> 1) & 1][8 * j + 4 * (l & 1)]) =
                    *reinterpret_cast<const uint32_t *>(&s.c[g - 1][bb][l >> 2][8 + ((l >> 1) & 1)][8 * jb + 4 * (l & 1)]);
        }
        PROF_MARK(2);
        /* ---- boundary strengths, in registers ------------------------------------------------------- */
        const MbInfo &h = cur.h, &ht = cur.ht;
        const bool filter = valid && !(h.flags() & MI355_MBF_NO_DEBLOCK);
        const bool have_left = filter && mb_x > 0 && (h.flags() & MI355_MBF_LEFT_EDGE), have_top = filter && has_t && (h.flags() & MI355_MBF_TOP_EDGE);
        const uint32_t b0 = bs_role(h, hl, outer, odd, filter && (!outer || have_left), r0, cur.p0, cur.q0, two_lists, mv_far, 4u);
        const uint32_t b1 = bs_role(h, ht, outer, odd, filter && (!outer || have_top), r1, cur.p1, cur.q1, two_lists, mv_far, field ? 3u : 4u);
        /* the four strengths of a line: edge e of this lane's segment sits in lane e of its group of four */
        const uint32_t bsw0 = (uint32_t)quad_bcast<0>((int)b0) | ((uint32_t)quad_bcast<1>((int)b0) << 8) | ((uint32_t)quad_bcast<2>((int)b0) << 16) | ((uint32_t)quad_bcast<3>((int)b0) << 24);
        const uint32_t bsw1 = (uint32_t)quad_bcast<0>((int)b1) | ((uint32_t)quad_bcast<1>((int)b1) << 8) | ((uint32_t)quad_bcast<2>((int)b1) << 16) | ((uint32_t)quad_bcast<3>((int)b1) << 24);
        /* a chroma line (row / column cr of plane cp) lies in luma segment cr >> 1 */
        const int csrc = (lane & ~15) | ((cr >> 1) << 2);
        const uint32_t bsc0 = (uint32_t)__shfl((int)bsw0, csrc), bsc1 = (uint32_t)__shfl((int)bsw1, csrc);
        PROF_MARK(3);
        /* ---- alpha / beta / tc0: lane k < 9 of a group looks up (component k / 3, edge kind k % 3) ------------ */
        {
            const int comp = l < 3 ? 0 : (l < 6 ? 1 : 2), kind = l - 3 * comp;          /* lanes >= 9 repeat a valid role */
            const int kindc = kind > 2 ? 2 : kind;
            const MbInfo &nb = kindc == 1 ? hl : ht;
            int qa = comp ? h.qpc(comp - 1) : h.qp();
            int qb = comp ? nb.qpc(comp - 1) : nb.qp();
            /* chroma QP of a neighbour as the CURRENT slice's table sees it (h264_loopfilter.c:628-629); the table fetch
             * (neighbour in another slice) is consumed inside its branch */
            if (comp && kindc && nb.slice_id() != h.slice_id() && (kindc == 1 ? have_left : have_top)) {
                int v = mi355_global(fr.slices)[h.slice_id()].chroma_qp_table[comp - 1][nb.qp()];
                MI355_PIN(v);
                qb = v;
            }
            const int qp = kindc ? (qa + qb + 1) >> 1 : qa;
            const int ia = clip3(qp + h.alpha_off(), 0, 51), ib = clip3(qp + h.beta_off(), 0, 51);
            const uint32_t w0 = (uint32_t)s.t_alpha[ia] | ((uint32_t)s.t_beta[ib] << 8);
            const uint32_t w1 = s.t_tc0[ia] + (comp ? 0x01010100u : 0u);
            if (l < 9) { s.parm[g][l][0] = w0; s.parm[g][l][1] = w1; }
        }
        MI355_WAVE_SYNC();
        const uint32_t *pl = s.parm[g][0], *pc = s.parm[g][3 + 3 * cp];
        const uint32_t ab_i = pl[0], tr_i = pl[1], ab_l = pl[2], tr_l = pl[3], ab_t = pl[4], tr_t = pl[5];
        const uint32_t cab_i = pc[0], ctr_i = pc[1], cab_l = pc[2], ctr_l = pc[3], cab_t = pc[4], ctr_t = pc[5];
        /* tc0 of the four edges of a line at once: byte e = row[bS_e & 3] */
        const uint32_t tci0 = byte_perm(0, tr_i, bsw0 & 0x03030303u), tcl0 = byte_perm(0, tr_l, bsw0 & 3u);
        const uint32_t tci1 = byte_perm(0, tr_i, bsw1 & 0x03030303u), tct1 = byte_perm(0, tr_t, bsw1 & 3u);
        const uint32_t cci0 = byte_perm(0, ctr_i, bsc0 & 0x03030303u), ccl0 = byte_perm(0, ctr_l, bsc0 & 3u);
        const uint32_t cci1 = byte_perm(0, ctr_i, bsc1 & 0x03030303u), cct1 = byte_perm(0, ctr_t, bsc1 & 3u);
#define AB_A(w) ((int)((w) & 0xFF))
#define AB_B(w) ((int)(((w) >> 8) & 0xFF))
#define BYTE(w, e) ((int)(((w) >> (8 * (e))) & 0xFF))

        PROF_MARK(4);
        /* ---- vertical edges, one luma row + one chroma row per lane, in registers.  The four
         * samples left of the MB are the previous MB's last columns (previous chunk when j == 0). ----- */
        {
            uint8_t *rowp = &s.y[g][b][4 + l][16 * j];
            uint8_t *leftp = j ? rowp - 4 : &s.y[g][b ^ 1][4 + l][16 * (DCH - 1) + 12];
            const uint4 own = lds16(rowp);
            uint32_t wl = *reinterpret_cast<const uint32_t *>(leftp), w0 = own.x, w1 = own.y, w2 = own.z, w3 = own.w;
            uint8_t *crowp = &s.c[g][b][cp][2 + cr][8 * j];
            uint8_t *cleftp = j ? crowp - 4 : &s.c[g][b ^ 1][cp][2 + cr][8 * (DCH - 1) + 4];
            const uint2 cown = *reinterpret_cast<const uint2 *>(crowp);
            uint32_t cl = *reinterpret_cast<const uint32_t *>(cleftp), cw0 = cown.x, cw1 = cown.y;      /* columns -4..-1, 0..3, 4..7 */
            const bool c0 = luma_row_edge<true>(wl, w0, BYTE(bsw0, 0), AB_A(ab_l), AB_B(ab_l), BYTE(tcl0, 0));
            const bool c1 = luma_row_edge<false>(w0, w1, BYTE(bsw0, 1), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 1));
            const bool c2 = luma_row_edge<false>(w1, w2, BYTE(bsw0, 2), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 2));
            const bool c3 = luma_row_edge<false>(w2, w3, BYTE(bsw0, 3), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 3));
            if (c0) *reinterpret_cast<uint32_t *>(leftp) = wl;
            if (c0 || c1 || c2 || c3) lds16(rowp, make_uint4(w0, w1, w2, w3));
            const bool d0 = chroma_row_edge(cl, cw0, BYTE(bsc0, 0), AB_A(cab_l), AB_B(cab_l), BYTE(ccl0, 0));
            const bool d1 = chroma_row_edge(cw0, cw1, BYTE(bsc0, 2), AB_A(cab_i), AB_B(cab_i), BYTE(cci0, 2));
            if (d0) *reinterpret_cast<uint32_t *>(cleftp) = cl;
            if (d0 || d1) *reinterpret_cast<mi355_u32x2 *>(crowp) = mi355_u32x2{ cw0, cw1 };
        }
        PROF_MARK(5);
        MI355_WAVE_SYNC();
        /* ---- horizontal edges, one luma column + one chroma column per lane ------------------------------ */
        {
            uint8_t *colp = &s.y[g][b][0][16 * j + l];
            uint8_t *ccolp = &s.c[g][b][cp][0][8 * j + cr];
            int y0 = colp[0 * DY_PITCH], y1 = colp[1 * DY_PITCH], y2 = colp[2 * DY_PITCH], y3 = colp[3 * DY_PITCH];
            int y4 = colp[4 * DY_PITCH], y5 = colp[5 * DY_PITCH], y6 = colp[6 * DY_PITCH], y7 = colp[7 * DY_PITCH];
            int y8 = colp[8 * DY_PITCH], y9 = colp[9 * DY_PITCH], y10 = colp[10 * DY_PITCH], y11 = colp[11 * DY_PITCH];
            int y12 = colp[12 * DY_PITCH], y13 = colp[13 * DY_PITCH], y14 = colp[14 * DY_PITCH], y15 = colp[15 * DY_PITCH];
            int y16 = colp[16 * DY_PITCH], y17 = colp[17 * DY_PITCH], y18 = colp[18 * DY_PITCH];
            int u0 = ccolp[0 * DC_PITCH], u1 = ccolp[1 * DC_PITCH], u2 = ccolp[2 * DC_PITCH], u3 = ccolp[3 * DC_PITCH];
            int u4 = ccolp[4 * DC_PITCH], u5 = ccolp[5 * DC_PITCH], u6 = ccolp[6 * DC_PITCH], u7 = ccolp[7 * DC_PITCH];
            const int e0 = luma_line<true>(y0, y1, y2, y3, y4, y5, y6, y7, BYTE(bsw1, 0), AB_A(ab_t), AB_B(ab_t), BYTE(tct1, 0));
            if (e0) {
                if (e0 == 2) { colp[1 * DY_PITCH] = (uint8_t)y1; colp[6 * DY_PITCH] = (uint8_t)y6; }
                colp[2 * DY_PITCH] = (uint8_t)y2; colp[3 * DY_PITCH] = (uint8_t)y3; colp[4 * DY_PITCH] = (uint8_t)y4; colp[5 * DY_PITCH] = (uint8_t)y5;
            }
            if (luma_line<false>(y4, y5, y6, y7, y8, y9, y10, y11, BYTE(bsw1, 1), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 1))) {
                colp[6 * DY_PITCH] = (uint8_t)y6; colp[7 * DY_PITCH] = (uint8_t)y7; colp[8 * DY_PITCH] = (uint8_t)y8; colp[9 * DY_PITCH] = (uint8_t)y9;
            }
            if (luma_line<false>(y8, y9, y10, y11, y12, y13, y14, y15, BYTE(bsw1, 2), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 2))) {
                colp[10 * DY_PITCH] = (uint8_t)y10; colp[11 * DY_PITCH] = (uint8_t)y11; colp[12 * DY_PITCH] = (uint8_t)y12; colp[13 * DY_PITCH] = (uint8_t)y13;
            }
            int y19 = 0;
            if (luma_line<false>(y12, y13, y14, y15, y16, y17, y18, y19, BYTE(bsw1, 3), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 3))) {
                colp[14 * DY_PITCH] = (uint8_t)y14; colp[15 * DY_PITCH] = (uint8_t)y15; colp[16 * DY_PITCH] = (uint8_t)y16; colp[17 * DY_PITCH] = (uint8_t)y17;
            }
            if (chroma_line(u0, u1, u2, u3, BYTE(bsc1, 0), AB_A(cab_t), AB_B(cab_t), BYTE(cct1, 0))) {
                ccolp[1 * DC_PITCH] = (uint8_t)u1; ccolp[2 * DC_PITCH] = (uint8_t)u2;
            }
            if (chroma_line(u4, u5, u6, u7, BYTE(bsc1, 2), AB_A(cab_i), AB_B(cab_i), BYTE(cci1, 2))) {
                ccolp[5 * DC_PITCH] = (uint8_t)u5; ccolp[6 * DC_PITCH] = (uint8_t)u6;
            }
        }
#undef AB_A
#undef AB_B
#undef BYTE
        PROF_MARK(6);
        MI355_WAVE_SYNC();   /* the tile is final for this macroblock: the group below and the next step may read it */
        hl = h;
    };
    /* chunks still in LDS */
    auto finish = [&]() __attribute__((always_inline)) {
        for (int c = flushed; c <= (nsteps - 1) >> DCH_LOG; c++) flush_chunk(c);
    };
    if constexpr (KW == 1) {
        (void)wave;
        start();
        for (int t = 0; t < nsteps; t++) step(t);
        finish();
    } else {
        /* every wave runs the same number of global steps and meets the others at the barrier after each one (a wave
         * whose band lies below the picture only attends).  __syncthreads() = workgroup-scope release + acquire: the
         * stores a wave issued in this step (flush_chunk) are visible to the waves of this workgroup after it. */
        const int t0 = DEBLOCK_LAG * wave, t_end = nsteps + DEBLOCK_LAG * (KW - 1);
        for (int tt = 0; tt < t_end; tt++) {
            const int t = tt - t0;
            if (t == 0) start();
            if (t >= 0 && t < nsteps) step(t);
            if (t == nsteps - 1) finish();
            __syncthreads();
        }
    }
#ifdef MI355_PROF
    PROF_MARK(7);
    if (blockIdx.x < 64 && lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_prof[i], prof_acc[i]);
#endif
}

#ifdef MI355_DEBLOCK_WAVES
__attribute__((amdgpu_waves_per_eu(MI355_DEBLOCK_WAVES, MI355_DEBLOCK_WAVES)))
#endif
__global__ void __launch_bounds__(64)
k_deblock(const mi355_h264_frame *__restrict__ frames, int band, int skip_tiled)      /* skip_tiled: k_deblock_tiled takes the tiled pictures of the batch */
{
    __shared__ DeblockLds s;
    const mi355_h264_frame &fr = frames[blockIdx.x];
    if (uniform(fr.surface_layout) == MI355_SURFACE_TILED) {
        if (skip_tiled) return;
        if (mi355_global(fr.mv[1]) != nullptr) deblock_band<true, 1, true>(s, fr, band, 0);
        else deblock_band<false, 1, true>(s, fr, band, 0);
        return;
    }
    if (mi355_global(fr.mv[1]) != nullptr) deblock_band<true, 1, false>(s, fr, band, 0);
    else deblock_band<false, 1, false>(s, fr, band, 0);
}

/* the small-batch form: KW consecutive bands of a picture per workgroup, one wave each (see deblock_band) */
template <int KW>
__global__ void __launch_bounds__(64 * KW)
k_deblock_bands(const mi355_h264_frame *__restrict__ frames, int band0, int skip_tiled)
{
    __shared__ DeblockLds s[KW];
    const mi355_h264_frame &fr = frames[blockIdx.x];
    const int wave = (int)(threadIdx.x >> 6);
    if (uniform(fr.surface_layout) == MI355_SURFACE_TILED) {
        if (skip_tiled) return;
        if (mi355_global(fr.mv[1]) != nullptr) deblock_band<true, KW, true>(s[wave], fr, band0 + wave, wave);
        else deblock_band<false, KW, true>(s[wave], fr, band0 + wave, wave);
        return;
    }
    if (mi355_global(fr.mv[1]) != nullptr) deblock_band<true, KW, false>(s[wave], fr, band0 + wave, wave);
    else deblock_band<false, KW, false>(s[wave], fr, band0 + wave, wave);
}



/* ===================================================================================================================== */
/* The loop filter on macroblock-tiled surfaces (round 4): every band of every picture in ONE launch, tiles by LDS-DMA     */
/* ===================================================================================================================== */
/* What stays of deblock_band above: a wave = a band of four macroblock rows of one picture, lanes 16g..16g+15 = row 4 * band + g,
 * group g at macroblock x = t - 2g in step t (the reference's raster dependencies by lock-step execution), boundary strengths by
 * lane role, the edge filters themselves.  What changes:
 *
 *  - bands do not wait for launches.  All bands of all pictures are workgroups of one launch, taken in band-major order from a
 *    ticket counter (a wave only ever waits for a lower ticket, which is running or done: no dependence on the order the
 *    hardware dispatches workgroups in), and band b + 1 follows band b a dozen macroblocks behind: band b hands the second tile
 *    lines and the chroma tiles of its last row down through `dst` with agent-scope write-through stores and publishes how
 *    many macroblocks are out in a progress counter; band b + 1 polls the counter and fetches them with agent-scope loads
 *    (/opt/skills/guides/cdna_hip_programming.md section 6, guideline 16, form R1: sc1 payload, drained, flag; sc1 loads on the
 *    consumer).  With 2048 pictures the launch keeps every SIMD at its register-bound occupancy (the per-band launches were
 *    2048 waves = two per SIMD whatever the kernel needed); with 64 pictures a picture's seventeen bands overlap;
 *  - the unit that moves is ONE MACROBLOCK per group and step: a tile is two + one whole cache lines whoever reads it.  The
 *    tiles of the four groups' next macroblocks go straight from memory into an LDS ring (global_load_lds_dwordx4: no register
 *    holds a sample in flight; the first version of this form kept them in registers: 145 of them, three waves per SIMD), a
 *    whole step before they are used:
 *      ring[step & 3][group]: a 16-byte piece lands at base + 16 * lane, so the ring is lane-linear and unpadded; the four
 *      groups' column accesses of the horizontal edges are kept off each other's banks by WHICH piece a lane fetches: row r of
 *      group g's tile lives at 16 * (r ^ g) (chroma: 16-byte piece k at 16 * (k ^ g));
 *      the second tile line + chroma tile of the macroblock above a band arrive the same way (16 lanes, agent scope) into
 *      above[step & 1], swizzled with 3 (= the group "above group 0");
 *  - the rows above a macroblock are not copied: the horizontal edges read and patch rows 12..15 of the tile in the ring of the
 *    group above, and that tile's SECOND cache line (rows 8..15) and its chroma tile are written to `dst` by the group BELOW
 *    once they are final — every store is a whole 128-byte line of a tile;
 *  - what a step makes final is written out at the START of the next step, right after the wait that begins it: that wait
 *    finds stores a whole step old, and the progress counter is published behind it without a drain of its own;
 *  - the boundary-strength roles sit in a 768-byte LDS table, the address arithmetic in a dozen per-lane constants.
 * The kernel runs at the VALU's issue rate (profiles/r04d_pmc_deblock_tiled_f2048.txt: 555 VALU per step, 84 % of the issue slots
 * at four waves per SIMD). */
struct __attribute__((aligned(128))) Deblock3Lds {
    uint8_t y[4][4][256];
    uint8_t c[4][4][128];
    uint8_t above[2][256];
    uint32_t parm[2][4][9][2];  /* [step & 1 in the two-wave form, 0 otherwise] */
    uint32_t role[16][12];      /* per lane of a group: BsRole r0, BsRole r1, o_p0, o_q0, o_p1, o_q1 */
    uint8_t t_alpha[52], t_beta[52];
    uint32_t t_tc0[52];
};
/* the two-wave form's mailbox: the strengths of the four edges of this lane's luma / chroma lines, both directions, for the step after this one */
struct __attribute__((aligned(16))) Deblock3Mail { uint32_t bs[2][64][4]; };
static_assert(sizeof(Deblock3Lds) <= 8192 + 288, "nineteen or twenty waves per CU");

/* NW = 1: the wave does everything (the throughput form).  NW = 2 (few pictures: SIMDs stand idle and a lone wave's step is a serial chain
 * of ~1100 instructions): the band's workgroup is TWO waves — wave 0 runs only the edge phases of step t, wave 1 meanwhile writes out what
 * step t - 1 finished, issues the loads of step t + 1 and derives the strengths and parameters of step t + 1, which reach wave 0 through LDS
 * (`mail`, parm[step & 1]); two workgroup barriers per step (LDS only: the loads in flight stay in flight). */
#ifdef MI355_HIP_EMU_H
#define MI355_WG_BARRIER_LDS() __syncthreads()
#else
#define MI355_WG_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
template <bool TWO_LISTS, int NW>
__device__ __forceinline__ void deblock3_band(Deblock3Lds &s, Deblock3Mail *mail, const mi355_h264_frame &fr, int band, uint32_t *prog, int wave)
{
    const bool edge_wave = NW == 1 || wave == 0, srv_wave = NW == 1 || wave == 1;
    const int W = uniform(fr.mb_width), H = uniform(fr.mb_height);
    const bool field = uniform(fr.field_picture) != 0;
    const uint32_t mv_far = field ? 0xFFFEFFFCu : 0xFFFCFFFCu;
    const int rs = uniform(fr.recon_stride[0]), rcs = uniform(fr.recon_stride[1]), ds = uniform(fr.dst_stride[0]), dcs = uniform(fr.dst_stride[1]);
    constexpr bool two_lists = TWO_LISTS;
    const int nsteps = W + 7;
    const bool hand = 4 * band + 4 < H;                      /* a band follows: group 3's row is handed to it */
    const bool top_band = band > 0;
    const int ya = top_band ? 4 * band - 1 : 0;
    const uint8_t *const rec_base = reinterpret_cast<const uint8_t *>(mi355_global(fr.mb));
    const uint8_t *const mv_base0 = reinterpret_cast<const uint8_t *>(mi355_global(fr.mv[0]));
    const uint8_t *const mv_base1 = two_lists ? reinterpret_cast<const uint8_t *>(mi355_global(fr.mv[1])) : mv_base0;
    const uint8_t *const recon_y0 = mi355_global(fr.recon[0]), *const recon_c0 = mi355_global(fr.recon[1]);
    uint8_t *const dst_y0 = mi355_global(fr.dst[0]), *const dst_c0 = mi355_global(fr.dst[1]);
    uint32_t *const prog_above = prog + (top_band ? band - 1 : 0), *const prog_self = prog + band;
    if (NW == 1 || wave == 0) {
        const int lane = lane_id(), l = lane & 15;
        if (lane < 52) {
            s.t_alpha[lane] = k_alpha[lane]; s.t_beta[lane] = k_beta[lane];
            s.t_tc0[lane] = ((uint32_t)k_tc0[lane][0] << 8) | ((uint32_t)k_tc0[lane][1] << 16) | ((uint32_t)k_tc0[lane][2] << 24);
        }
        if (lane < 16) {
            /* role of lane l of a group: segment l >> 2 of edge l & 3, in both directions (as deblock_band) */
            const int seg = l >> 2, edge = l & 3, qe = edge == 0 ? 3 : edge - 1;
            const bool outer = edge == 0;
            uint32_t *r = s.role[l];
            r[0] = 1u << blk_index(edge, seg); r[1] = 1u << blk_index(qe, seg); r[2] = 8u * ((edge >> 1) + 2 * (seg >> 1)); r[3] = 8u * ((qe >> 1) + 2 * (seg >> 1));
            r[4] = 1u << blk_index(seg, edge); r[5] = 1u << blk_index(seg, qe); r[6] = 8u * ((seg >> 1) + 2 * (edge >> 1)); r[7] = 8u * ((seg >> 1) + 2 * (qe >> 1));
            r[8] = (uint32_t)(4 * (edge + 4 * seg));
            r[9] = (uint32_t)(outer ? 4 * (3 + 4 * seg) - 64 : 4 * (edge - 1 + 4 * seg));
            r[10] = (uint32_t)(4 * (seg + 4 * edge));
            r[11] = (uint32_t)(outer ? 4 * (seg + 12) - 64 * W : 4 * (seg + 4 * (edge - 1)));
        }
    }
    if (NW == 1) MI355_WAVE_SYNC(); else MI355_WG_BARRIER_LDS();

    struct Pre {
        MbInfo h, ht;
        uint32_t p0[2], q0[2], p1[2], q1[2];
    };
    /* Per-lane constants of the address arithmetic, computed once: a dozen registers against ~150 instructions per step (the first
     * version recomputed rows, tile offsets and swizzles from the lane number in every step: 637 VALU per step, the kernel bound by them).
     * LDS offsets are byte offsets from `s`, without the term of the ring slot; memory offsets are those of step 0, a step adds a tile. */
    uint8_t *const lds = reinterpret_cast<uint8_t *>(&s);
    constexpr uint32_t OY = (uint32_t)offsetof(Deblock3Lds, y), OC = (uint32_t)offsetof(Deblock3Lds, c), OA = (uint32_t)offsetof(Deblock3Lds, above);
    uint32_t k_g2, k_rec, k_topd, k_dy, k_dc, k_gc2, k_da, k_sab, k_sc, k_sd, k_se, k_la, k_lc, k_ld, k_rv, k_cv, k_flags;
    enum { KF_ROW = 1, KF_TOP = 2, KF_BOTTOM = 4, KF_LO = 8, KF_G0 = 16, KF_ROW0 = 32 };
    {
        const int lane = lane_id(), g = lane >> 4, l = lane & 15, ga = (g - 1) & 3;
        const int mb_y = 4 * band + g, mb_yc = mb_y < H ? mb_y : H - 1;
        const bool row_ok = mb_y < H, has_t = row_ok && mb_y > 0, bottom = row_ok && (g == 3 || mb_y == H - 1), lo = l < 8;
        k_flags = (row_ok ? KF_ROW : 0) | (has_t ? KF_TOP : 0) | (bottom ? KF_BOTTOM : 0) | (lo ? KF_LO : 0) | (g == 0 ? KF_G0 : 0) | (mb_yc == 0 ? KF_ROW0 : 0);
        k_g2 = (uint32_t)(2 * g);
        k_rec = (uint32_t)(64 * mb_yc * W);
        k_topd = mb_yc > 0 ? 64u * (uint32_t)W : 0u;
        k_dy = (uint32_t)(mb_yc * rs + 16 * (l ^ g));
        {   /* chroma tiles: lanes 0..31, lane -> (group lane >> 3, piece lane & 7) */
            const int gc = (lane >> 3) & 3, q = lane & 7, yc = 4 * band + gc < H ? 4 * band + gc : H - 1;
            k_dc = (uint32_t)(yc * rcs + 16 * (q ^ gc));
            k_gc2 = (uint32_t)(2 * gc);
        }
        /* the macroblock above group 0's: lanes 0..7 its second luma line, lanes 8..15 its chroma tile (offsets into dst[0] / dst[1]) */
        k_da = (lane & 8) ? (uint32_t)(ya * dcs + 16 * ((lane & 7) ^ 3)) : (uint32_t)(ya * ds + 128 + 16 * ((lane & 7) ^ 3));
        k_sab = lo ? (uint32_t)(mb_y * ds - (2 * g + 1) * 256 + 16 * l) : (uint32_t)((mb_y - 1) * ds - 2 * g * 256 + 16 * l);
        k_sc = (uint32_t)((mb_y - 1) * dcs - 2 * g * 128 + 8 * l);
        k_sd = (uint32_t)(mb_y * ds - (2 * g + 1) * 256 + 128 + 8 * l);
        k_se = (uint32_t)(mb_y * dcs - (2 * g + 1) * 128 + 8 * l);
        k_rv = (uint32_t)(256 * g + 16 * (l ^ g));                                         /* luma row l of the own tile */
        k_cv = (uint32_t)(128 * g + 16 * ((l >> 1) ^ g) + 8 * (l & 1));                    /* chroma row (plane l >> 3, row l & 7) of the own tile */
        k_la = lo ? k_rv : (uint32_t)(g ? 256 * ga + 128 + 16 * ((l - 8) ^ ga) : 16 * ((l - 8) ^ 3));       /* (a): row l of the left tile / (b): row l of the tile above */
        k_lc = (uint32_t)(g ? 128 * ga + 16 * ((l >> 1) ^ ga) + 8 * (l & 1) : 128 + 16 * ((l >> 1) ^ 3) + 8 * (l & 1));   /* chroma row of the tile above */
        k_ld = (uint32_t)(256 * g + 16 * ((8 + (l >> 1)) ^ g) + 8 * (l & 1));              /* half (l & 1) of row 8 + (l >> 1) of the own tile */
    }
    auto kf = [&](uint32_t bit) { return (k_flags & bit) != 0; };
    /* records and vectors of group g's macroblock t1 - 2g: into registers */
    auto prefetch = [&](Pre &p, int t1) {
        const int l = lane_id() & 15;
        const bool outer = (l & 3) == 0;
        const int xc = med3i(t1 - (int)k_g2, 0, W - 1);
        const uint32_t roff = k_rec + ((uint32_t)xc << 6), toff = roff - k_topd;
        p.h = mb_info_load<true>(rec_base, roff);
        p.ht = mb_info_load<false>(rec_base, toff);
        const mi355_u32x4 o = *reinterpret_cast<const mi355_u32x4 *>(&s.role[l][8]);
        const uint32_t a_p0 = roff + o[0], a_p1 = roff + o[2];
        const uint32_t a_q0 = xc > 0 || !outer ? roff + o[1] : a_p0;
        const uint32_t a_q1 = !kf(KF_ROW0) || !outer ? roff + o[3] : a_p1;
        p.p0[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_p0); p.q0[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_q0);
        p.p1[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_p1); p.q1[0] = *reinterpret_cast<const uint32_t *>(mv_base0 + a_q1);
        p.p0[1] = p.q0[1] = p.p1[1] = p.q1[1] = 0;
        if (two_lists) {
            p.p0[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_p0); p.q0[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_q0);
            p.p1[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_p1); p.q1[1] = *reinterpret_cast<const uint32_t *>(mv_base1 + a_q1);
        }
    };
    /* the tiles of step t1 (group g: macroblock t1 - 2g, clamped into its row) into ring slot t1 & 3; group 0's macroblock above into above[t1 & 1] */
    auto dma_issue = [&](int t1) {
        const int lane = lane_id();
        lds_dma16<false>(recon_y0 + (k_dy + ((uint32_t)med3i(t1 - (int)k_g2, 0, W - 1) << 8)), &s.y[t1 & 3][0][0]);
        if (lane < 32) lds_dma16<false>(recon_c0 + (k_dc + ((uint32_t)med3i(t1 - (int)k_gc2, 0, W - 1) << 7)), &s.c[t1 & 3][0][0]);
        if (top_band && lane < 16) {
            const int xa = t1 < W ? t1 : W - 1;
            const uint8_t *src = (lane & 8) ? dst_c0 + (k_da + ((uint32_t)xa << 7)) : dst_y0 + (k_da + ((uint32_t)xa << 8));
            lds_dma16<true>(src, &s.above[t1 & 1][0]);
        }
        MI355_ISSUE_FENCE();
    };
    uint32_t seen = 0;
    auto await_above = [&](int x0) {
        if (!top_band) return;
        const uint32_t need = (uint32_t)(x0 + 1 < W ? x0 + 1 : W);
        if (seen >= need) return;
        for (;;) {
            seen = (uint32_t)uniform((int)agent_load_u32(mi355_global_v(prog_above)));
            if (seen >= need) break;
            wave_nap();
        }
        MI355_ISSUE_FENCE();
    };
    /* what step ts made final goes out (at the start of step ts + 1): a whole tile line per eight lanes.
     * lanes 0..7 of a group: rows 0..7 of the previous macroblock of its row (its last columns got macroblock x's left edge);
     * lanes 8..15: rows 8..15 of the macroblock above (its last rows got macroblock x's top edge), and that one's chroma tile;
     * a row nobody of this wave works below: the previous macroblock's own second line and chroma tile as well */
    auto stores = [&](int ts) {
        const uint32_t u1y = OY + 1024u * (uint32_t)((ts - 1) & 3), u2y = OY + 1024u * (uint32_t)((ts - 2) & 3), ua = OA + 256u * (uint32_t)(ts & 1);
        const uint32_t u1c = OC + 512u * (uint32_t)((ts - 1) & 3), u2c = OC + 512u * (uint32_t)((ts - 2) & 3);
        const uint32_t xhi = (uint32_t)ts - k_g2, xlo = xhi - 1u;                        /* macroblock x of this step, and the one before it */
        const bool in_lo = xlo < (uint32_t)W, in_hi = xhi < (uint32_t)W;
        const bool lo = kf(KF_LO), g0 = kf(KF_G0);
        const bool ok = lo ? (kf(KF_ROW) && in_lo) : (kf(KF_TOP) && in_hi);
        const uint4 v = lds16(lds + (k_la + (lo ? u1y : (g0 ? ua : u2y))));
        if (ok) st16(dst_y0 + (k_sab + ((uint32_t)ts << 8)), v, true);
        const uint2 vc = *reinterpret_cast<const uint2 *>(lds + (k_lc + (g0 ? ua : u2c)));
        if (kf(KF_TOP) && in_hi) st8(dst_c0 + (k_sc + ((uint32_t)ts << 7)), vc, true);
        const uint2 by = *reinterpret_cast<const uint2 *>(lds + (k_ld + u1y)), bc = *reinterpret_cast<const uint2 *>(lds + (k_cv + u1c));
        if (!hand && kf(KF_BOTTOM) && in_lo) {              /* with a band below, group 3's row went out in hand_down(), a step earlier */
            st8(dst_y0 + (k_sd + ((uint32_t)ts << 8)), by, true);
            st8(dst_c0 + (k_se + ((uint32_t)ts << 7)), bc, true);
        }
        MI355_WAVE_SYNC();                                   /* every lane has read its pieces: the slots are free for the DMA issued next (program order
                                                                on the device; a rendezvous of the fibers in the emulator) */
    };

    /* Group 3's row of a band with a band below: the second tile line and the chroma tile of the PREVIOUS macroblock are what that band
     * waits for, and they are ready as soon as this step's vertical edges have patched its last columns — they leave right then, written
     * through, and are announced behind the wait that begins the next step (the band below finishes and rewrites them) */
    auto hand_down = [&](int t) {
        const uint32_t u1y = OY + 1024u * (uint32_t)((t - 1) & 3), u1c = OC + 512u * (uint32_t)((t - 1) & 3);
        const uint32_t xlo = (uint32_t)t - k_g2 - 1u;
        const uint2 by = *reinterpret_cast<const uint2 *>(lds + (k_ld + u1y)), bc = *reinterpret_cast<const uint2 *>(lds + (k_cv + u1c));
        if (kf(KF_BOTTOM) && xlo < (uint32_t)W) {
            agent_store8(dst_y0 + (k_sd + ((uint32_t)t << 8)), by, true);
            agent_store8(dst_c0 + (k_se + ((uint32_t)t << 7)), bc, true);
        }
    };
    Pre pre = {}, pre_next = {};
    MbInfo hl = {};
#define AB_A(w) ((int)((w) & 0xFF))
#define AB_B(w) ((int)(((w) >> 8) & 0xFF))
#define BYTE(w, e) ((int)(((w) >> (8 * (e))) & 0xFF))
    /* boundary strengths of step t's macroblocks from `pre` (their records and vectors) and `hl` (the previous macroblock's), packed per line:
     * byte e of bsw0 / bsw1 = the strength of vertical / horizontal edge e of this lane's luma row / column, bsc0 / bsc1 the same for its chroma
     * line; alpha / beta / tc0 of the step into parm[pb] */
    auto strengths = [&](int t, int pb, uint32_t &bsw0, uint32_t &bsw1, uint32_t &bsc0, uint32_t &bsc1) {
        const int lane = lane_id(), g = lane >> 4, l = lane & 15;
        const int mb_y = 4 * band + g, mb_x = t - 2 * g;
        const bool row_ok = mb_y < H, has_t = row_ok && mb_y > 0, valid = row_ok && mb_x >= 0 && mb_x < W;
        const bool outer = (l & 3) == 0, odd = (l & 1) != 0;
        const MbInfo &h = pre.h, &ht = pre.ht;
        const mi355_u32x4 ra = *reinterpret_cast<const mi355_u32x4 *>(&s.role[l][0]), rb = *reinterpret_cast<const mi355_u32x4 *>(&s.role[l][4]);
        const BsRole r0{ ra[0], ra[1], ra[2], ra[3] }, r1{ rb[0], rb[1], rb[2], rb[3] };
        const bool filter = valid && !(h.flags() & MI355_MBF_NO_DEBLOCK);
        const bool have_left = filter && mb_x > 0 && (h.flags() & MI355_MBF_LEFT_EDGE), have_top = filter && has_t && (h.flags() & MI355_MBF_TOP_EDGE);
        const uint32_t b0 = bs_role(h, hl, outer, odd, filter && (!outer || have_left), r0, pre.p0, pre.q0, two_lists, mv_far, 4u);
        const uint32_t b1 = bs_role(h, ht, outer, odd, filter && (!outer || have_top), r1, pre.p1, pre.q1, two_lists, mv_far, field ? 3u : 4u);
        bsw0 = (uint32_t)quad_bcast<0>((int)b0) | ((uint32_t)quad_bcast<1>((int)b0) << 8) | ((uint32_t)quad_bcast<2>((int)b0) << 16) | ((uint32_t)quad_bcast<3>((int)b0) << 24);
        bsw1 = (uint32_t)quad_bcast<0>((int)b1) | ((uint32_t)quad_bcast<1>((int)b1) << 8) | ((uint32_t)quad_bcast<2>((int)b1) << 16) | ((uint32_t)quad_bcast<3>((int)b1) << 24);
        const int csrc = (lane & ~15) | (((l & 7) >> 1) << 2);
        bsc0 = (uint32_t)__shfl((int)bsw0, csrc); bsc1 = (uint32_t)__shfl((int)bsw1, csrc);
        /* alpha / beta / tc0: lane k < 9 of a group looks up (component k / 3, edge kind k % 3) */
        const int comp = l < 3 ? 0 : (l < 6 ? 1 : 2), kind = l - 3 * comp;
        const int kindc = kind > 2 ? 2 : kind;
        const MbInfo &nb = kindc == 1 ? hl : ht;
        int qa = comp ? h.qpc(comp - 1) : h.qp();
        int qb = comp ? nb.qpc(comp - 1) : nb.qp();
        if (comp && kindc && nb.slice_id() != h.slice_id() && (kindc == 1 ? have_left : have_top)) {
            int v = mi355_global(fr.slices)[h.slice_id()].chroma_qp_table[comp - 1][nb.qp()];
            MI355_PIN(v);
            qb = v;
        }
        const int qp = kindc ? (qa + qb + 1) >> 1 : qa;
        const int ia = clip3(qp + h.alpha_off(), 0, 51), ib = clip3(qp + h.beta_off(), 0, 51);
        const uint32_t w0 = (uint32_t)s.t_alpha[ia] | ((uint32_t)s.t_beta[ib] << 8);
        const uint32_t w1 = s.t_tc0[ia] + (comp ? 0x01010100u : 0u);
        if (l < 9) { s.parm[pb][g][l][0] = w0; s.parm[pb][g][l][1] = w1; }
        hl = h;
    };
    auto edges_v = [&](int t, int pb, uint32_t bsw0, uint32_t bsc0) {
        /* ---- vertical edges: lane l of a group = luma row l and chroma row (plane l >> 3, row l & 7), from and to the ring ---------------- */
        {
            const int lane = lane_id(), g = lane >> 4, l = lane & 15;
            const uint32_t *pl = s.parm[pb][g][0], *pc = s.parm[pb][g][3 + 3 * (l >> 3)];
            const uint32_t ab_i = pl[0], tr_i = pl[1], ab_l = pl[2], tr_l = pl[3];
            const uint32_t cab_i = pc[0], ctr_i = pc[1], cab_l = pc[2], ctr_l = pc[3];
            const uint32_t tci0 = byte_perm(0, tr_i, bsw0 & 0x03030303u), tcl0 = byte_perm(0, tr_l, bsw0 & 3u);
            const uint32_t cci0 = byte_perm(0, ctr_i, bsc0 & 0x03030303u), ccl0 = byte_perm(0, ctr_l, bsc0 & 3u);
            uint8_t *rowp = lds + (k_rv + OY + 1024u * (uint32_t)(t & 3)), *leftp = lds + (k_rv + OY + 12u + 1024u * (uint32_t)((t - 1) & 3));
            uint8_t *crowp = lds + (k_cv + OC + 512u * (uint32_t)(t & 3)), *cleftp = lds + (k_cv + OC + 4u + 512u * (uint32_t)((t - 1) & 3));
            const uint4 own = lds16(rowp);
            const uint2 cown = *reinterpret_cast<const uint2 *>(crowp);
            uint32_t wl = *reinterpret_cast<const uint32_t *>(leftp), w0 = own.x, w1 = own.y, w2 = own.z, w3 = own.w;
            uint32_t cl = *reinterpret_cast<const uint32_t *>(cleftp), cw0 = cown.x, cw1 = cown.y;
            const bool c0 = luma_row_edge<true>(wl, w0, BYTE(bsw0, 0), AB_A(ab_l), AB_B(ab_l), BYTE(tcl0, 0));
            const bool c1 = luma_row_edge<false>(w0, w1, BYTE(bsw0, 1), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 1));
            const bool c2 = luma_row_edge<false>(w1, w2, BYTE(bsw0, 2), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 2));
            const bool c3 = luma_row_edge<false>(w2, w3, BYTE(bsw0, 3), AB_A(ab_i), AB_B(ab_i), BYTE(tci0, 3));
            if (c0) *reinterpret_cast<uint32_t *>(leftp) = wl;
            if (c0 || c1 || c2 || c3) lds16(rowp, make_uint4(w0, w1, w2, w3));
            const bool d0 = chroma_row_edge(cl, cw0, BYTE(bsc0, 0), AB_A(cab_l), AB_B(cab_l), BYTE(ccl0, 0));
            const bool d1 = chroma_row_edge(cw0, cw1, BYTE(bsc0, 2), AB_A(cab_i), AB_B(cab_i), BYTE(cci0, 2));
            if (d0) *reinterpret_cast<uint32_t *>(cleftp) = cl;
            if (d0 || d1) *reinterpret_cast<mi355_u32x2 *>(crowp) = mi355_u32x2{ cw0, cw1 };
        }
    };
    auto edges_h = [&](int t, int pb, uint32_t bsw1, uint32_t bsc1) {
        /* ---- horizontal edges: lane l of a group = luma column l and chroma column (plane l >> 3, column l & 7); rows -4..-1 (chroma -2, -1)
         * are rows 12..15 (6, 7) of the tile above — in the ring of the group above, or in above[] — read and patched where they lie -------- */
        {
            const int lane = lane_id(), g = lane >> 4, l = lane & 15, ga = (g - 1) & 3, cp = l >> 3, cr = l & 7;
            const uint32_t *pl = s.parm[pb][g][0], *pc = s.parm[pb][g][3 + 3 * cp];
            const uint32_t ab_i = pl[0], tr_i = pl[1], ab_t = pl[4], tr_t = pl[5];
            const uint32_t cab_i = pc[0], ctr_i = pc[1], cab_t = pc[4], ctr_t = pc[5];
            const uint32_t tci1 = byte_perm(0, tr_i, bsw1 & 0x03030303u), tct1 = byte_perm(0, tr_t, bsw1 & 3u);
            const uint32_t cci1 = byte_perm(0, ctr_i, bsc1 & 0x03030303u), cct1 = byte_perm(0, ctr_t, bsc1 & 3u);
            uint8_t *const Y = &s.y[t & 3][g][l], *const C = &s.c[t & 3][g][64 * cp + cr];
            uint8_t *const A = (g ? &s.y[(t - 2) & 3][ga][128] : &s.above[t & 1][0]) + l;          /* row 8 + m at A[16 * (m ^ ga)] */
            uint8_t *const CA = (g ? &s.c[(t - 2) & 3][ga][0] : &s.above[t & 1][128]) + 64 * cp + cr;
            /* row r of the own tile at Y[16 * (r ^ g)]: r = 4k + j -> 64k + 16 * (j ^ g): one base per j (the swizzle is an exclusive or: not an
             * address offset the instruction could carry), the rest immediate offsets */
            const int g16 = 16 * g, a16 = 16 * ga;
            uint8_t *const Y0 = Y + g16, *const Y1 = Y + (g16 ^ 16), *const Y2 = Y + (g16 ^ 32), *const Y3 = Y + (g16 ^ 48);
            uint8_t *const A0 = A + 64 + a16, *const A1 = A + 64 + (a16 ^ 16), *const A2 = A + 64 + (a16 ^ 32), *const A3 = A + 64 + (a16 ^ 48);
            uint8_t *const C0 = C + g16, *const C1 = C + (g16 ^ 16), *const C2 = C + (g16 ^ 32), *const CA3 = CA + (a16 ^ 48);
#define YJ(j) ((j) == 0 ? Y0 : ((j) == 1 ? Y1 : ((j) == 2 ? Y2 : Y3)))
#define AJ(j) ((j) == 0 ? A0 : ((j) == 1 ? A1 : ((j) == 2 ? A2 : A3)))
#define YR(r) YJ((r) & 3)[64 * ((r) >> 2)]
#define AR(r) AJ((r) & 3)[0]
#define CR(r) (((r) >> 1) == 0 ? C0 : (((r) >> 1) == 1 ? C1 : C2))[8 * ((r) & 1)]
#define CAR(r) CA3[8 * ((r) & 1)]
            int y0 = AR(12), y1 = AR(13), y2 = AR(14), y3 = AR(15);
            int y4 = YR(0), y5 = YR(1), y6 = YR(2), y7 = YR(3), y8 = YR(4), y9 = YR(5), y10 = YR(6), y11 = YR(7);
            int y12 = YR(8), y13 = YR(9), y14 = YR(10), y15 = YR(11), y16 = YR(12), y17 = YR(13), y18 = YR(14);
            int u0 = CAR(6), u1 = CAR(7), u2 = CR(0), u3 = CR(1), u4 = CR(2), u5 = CR(3), u6 = CR(4), u7 = CR(5);
            const int e0 = luma_line<true>(y0, y1, y2, y3, y4, y5, y6, y7, BYTE(bsw1, 0), AB_A(ab_t), AB_B(ab_t), BYTE(tct1, 0));
            if (e0) {
                if (e0 == 2) { AR(13) = (uint8_t)y1; YR(2) = (uint8_t)y6; }
                AR(14) = (uint8_t)y2; AR(15) = (uint8_t)y3; YR(0) = (uint8_t)y4; YR(1) = (uint8_t)y5;
            }
            if (luma_line<false>(y4, y5, y6, y7, y8, y9, y10, y11, BYTE(bsw1, 1), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 1))) {
                YR(2) = (uint8_t)y6; YR(3) = (uint8_t)y7; YR(4) = (uint8_t)y8; YR(5) = (uint8_t)y9;
            }
            if (luma_line<false>(y8, y9, y10, y11, y12, y13, y14, y15, BYTE(bsw1, 2), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 2))) {
                YR(6) = (uint8_t)y10; YR(7) = (uint8_t)y11; YR(8) = (uint8_t)y12; YR(9) = (uint8_t)y13;
            }
            int y19 = 0;
            if (luma_line<false>(y12, y13, y14, y15, y16, y17, y18, y19, BYTE(bsw1, 3), AB_A(ab_i), AB_B(ab_i), BYTE(tci1, 3))) {
                YR(10) = (uint8_t)y14; YR(11) = (uint8_t)y15; YR(12) = (uint8_t)y16; YR(13) = (uint8_t)y17;
            }
            if (chroma_line(u0, u1, u2, u3, BYTE(bsc1, 0), AB_A(cab_t), AB_B(cab_t), BYTE(cct1, 0))) {
                CAR(7) = (uint8_t)u1; CR(0) = (uint8_t)u2;
            }
            if (chroma_line(u4, u5, u6, u7, BYTE(bsc1, 2), AB_A(cab_i), AB_B(cab_i), BYTE(cci1, 2))) {
                CR(3) = (uint8_t)u5; CR(4) = (uint8_t)u6;
            }
#undef YJ
#undef AJ
#undef YR
#undef AR
#undef CR
#undef CAR
        }
    };
    if constexpr (NW == 1) {
        await_above(0);
        dma_issue(0);
        prefetch(pre, 0);
        hl = pre.h;
#pragma nounroll
        for (int t = 0; t < nsteps; t++) {
            /* ---- everything requested during the last step is here; what was stored during it is out ------------------------------ */
            agent_drain_stores();
            if (hand && t >= 8) {
                /* hand_down(t - 1) is out: group 3's macroblocks 0 .. t - 8 */
                const int done = t - 7 < W ? t - 7 : W;
                if (lane_id() == 0) agent_store_u32(mi355_global_v(prog_self), (uint32_t)done);
            }
            MI355_WAVE_SYNC();
            if (t > 0) stores(t - 1);
            await_above(t + 1);
            dma_issue(t + 1);
            uint32_t bsw0, bsw1, bsc0, bsc1;
            strengths(t, 0, bsw0, bsw1, bsc0, bsc1);
            /* the next macroblock's records and vectors: the registers of this one's are free */
            prefetch(pre, t + 1);
            MI355_WAVE_SYNC();
            edges_v(t, 0, bsw0, bsc0);
            MI355_WAVE_SYNC();
            if (hand) hand_down(t);
            edges_h(t, 0, bsw1, bsc1);
        }
        /* ---- the last step's results, and the closing word to the band below ----------------------------------------------------- */
        agent_drain_stores();
        MI355_WAVE_SYNC();
        stores(nsteps - 1);
    } else {
        /* wave 1 first: the loads of step 0, the strengths of step 0 into the mailbox, the records of step 1 */
        if (srv_wave) {
            await_above(0);
            dma_issue(0);
            prefetch(pre, 0);
            hl = pre.h;
            uint32_t b[4];
            strengths(0, 0, b[0], b[1], b[2], b[3]);
            *reinterpret_cast<mi355_u32x4 *>(mail->bs[0][lane_id()]) = mi355_u32x4{ b[0], b[1], b[2], b[3] };
            prefetch(pre_next, 1);
        }
#pragma nounroll
        for (int t = 0; t < nsteps; t++) {
            /* both waves: what this wave stored during the last step is out; wave 1: the tiles of step t have landed, the records of step t + 1 are here */
            agent_drain_stores();
            if (edge_wave && hand && t >= 8) {
                /* the hand-down is the edge wave's (right behind its vertical edges, as in the one-wave form): hand_down(t - 1) is out */
                const int done = t - 7 < W ? t - 7 : W;
                if (lane_id() == 0) agent_store_u32(mi355_global_v(prog_self), (uint32_t)done);
            }
            MI355_WG_BARRIER_LDS();
            if (edge_wave) {
                const mi355_u32x4 b = *reinterpret_cast<const mi355_u32x4 *>(mail->bs[t & 1][lane_id()]);
                edges_v(t, t & 1, b[0], b[2]);
                MI355_WAVE_SYNC();
                if (hand) hand_down(t);
                edges_h(t, t & 1, b[1], b[3]);
            } else {
                /* a second set of record registers: the loads of step t + 2 go out FIRST and have the whole step (this wave has registers to spare;
                 * issued last, they would be waited for at once by the drain that begins the next step) */
                pre = pre_next;
                prefetch(pre_next, t + 2);
                if (t > 0) stores(t - 1);
                await_above(t + 1);
                dma_issue(t + 1);
                uint32_t b[4];
                strengths(t + 1, (t + 1) & 1, b[0], b[1], b[2], b[3]);
                *reinterpret_cast<mi355_u32x4 *>(mail->bs[(t + 1) & 1][lane_id()]) = mi355_u32x4{ b[0], b[1], b[2], b[3] };
            }
            MI355_WG_BARRIER_LDS();
        }
        if (!srv_wave) {
            /* the edge wave's last hand-down (macroblock W - 1) and the closing word to the band below */
            if (hand) {
                agent_drain_stores();
                if (lane_id() == 0) agent_store_u32(mi355_global_v(prog_self), (uint32_t)W);
            }
            return;
        }
        agent_drain_stores();
        stores(nsteps - 1);
        return;
    }
#undef AB_A
#undef AB_B
#undef BYTE
    if (hand) {
        agent_drain_stores();
        if (lane_id() == 0) agent_store_u32(mi355_global_v(prog_self), (uint32_t)W);
    }
}

/* Band-major tickets: all pictures' band 0, then band 1, ...; a wave's only dependence is the ticket nframes before its own, taken by a
 * wave that is running or done.  sync[0]: the ticket counter; sync[16 ...]: one progress word per picture and band.  Pictures that
 * are not tiled are left to k_deblock / k_deblock_bands (their waves take a ticket, look at the descriptor and leave). */
#ifdef MI355_DB3_WAVES
__attribute__((amdgpu_waves_per_eu(MI355_DB3_WAVES, MI355_DB3_WAVES)))
#endif
__global__ void __launch_bounds__(64)
k_deblock_tiled(const mi355_h264_frame *__restrict__ frames, int nframes, int nbands, uint32_t *sync)
{
    __shared__ Deblock3Lds s;
    uint32_t tk = 0;
    if (lane_id() == 0) tk = atomicAdd(mi355_global(sync), 1u);
    tk = (uint32_t)lane_value((int)tk, 0);
    const int band = (int)(tk / (uint32_t)nframes), pic = (int)(tk - (uint32_t)band * (uint32_t)nframes);
    if (band >= nbands) return;
    const mi355_h264_frame &fr = frames[pic];
    if (4 * band >= uniform(fr.mb_height) || uniform(fr.surface_layout) != MI355_SURFACE_TILED) return;
    uint32_t *prog = mi355_global(sync) + 16 + (size_t)pic * (size_t)nbands;
    if (mi355_global(fr.mv[1]) != nullptr) deblock3_band<true, 1>(s, nullptr, fr, band, prog, 0);
    else deblock3_band<false, 1>(s, nullptr, fr, band, prog, 0);
}
/* the same with two waves per band (few pictures): see deblock3_band */
__global__ void __launch_bounds__(128)
k_deblock_tiled2(const mi355_h264_frame *__restrict__ frames, int nframes, int nbands, uint32_t *sync)
{
    __shared__ Deblock3Lds s;
    __shared__ Deblock3Mail mail;
    __shared__ uint32_t ticket;
    if (threadIdx.x == 0) ticket = atomicAdd(mi355_global(sync), 1u);
    __syncthreads();
    const uint32_t tk = (uint32_t)uniform((int)ticket);
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int band = (int)(tk / (uint32_t)nframes), pic = (int)(tk - (uint32_t)band * (uint32_t)nframes);
    if (band >= nbands) return;
    const mi355_h264_frame &fr = frames[pic];
    if (4 * band >= uniform(fr.mb_height) || uniform(fr.surface_layout) != MI355_SURFACE_TILED) return;
    uint32_t *prog = mi355_global(sync) + 16 + (size_t)pic * (size_t)nbands;
    if (mi355_global(fr.mv[1]) != nullptr) deblock3_band<true, 2>(s, &mail, fr, band, prog, wave);
    else deblock3_band<false, 2>(s, &mail, fr, band, prog, wave);
}

}  // namespace

#ifdef MI355_PROF
extern "C" void mi355_debug_prof(unsigned long long *out, int reset)
{
    unsigned long long z[16] = {};
    MI355_CHECK(hipDeviceSynchronize());
    MI355_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(z)));
    if (reset) MI355_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)));
}
#endif

/* The single-launch form's counters: one ticket word and one progress word per picture and band, zeroed on the stream before every
 * launch.  A buffer per (thread, device, stream), grow-only: launches on one stream run one after the other, so a stream's buffer
 * is never shared by two launches in flight. */
namespace {
struct SyncBuf {
    int device;
    hipStream_t stream;
    uint32_t *dev;
    size_t words;
};
}  // namespace
static std::vector<SyncBuf> &sync_pool() { static thread_local std::vector<SyncBuf> pool; return pool; }
uint32_t *mi355::sync_words(hipStream_t st, size_t words)
{
    std::vector<SyncBuf> &pool = sync_pool();
    const int device = mi355::current_device();
    for (SyncBuf &b : pool) {
        if (b.device != device || b.stream != st) continue;
        if (b.words >= words) return b.dev;
        (void)hipFree(b.dev);                  /* waits for what still uses it */
        b.dev = nullptr; b.words = 0;
        size_t n = 4096;
        while (n < words) n *= 2;
        if (hipMalloc(reinterpret_cast<void **>(&b.dev), n * sizeof(uint32_t)) != hipSuccess) return nullptr;
        b.words = n;
        return b.dev;
    }
    SyncBuf b{ device, st, nullptr, 4096 };
    while (b.words < words) b.words *= 2;
    if (hipMalloc(reinterpret_cast<void **>(&b.dev), b.words * sizeof(uint32_t)) != hipSuccess) return nullptr;
    pool.push_back(b);
    return b.dev;
}
void mi355::sync_words_release(hipStream_t st)
{
    std::vector<SyncBuf> &pool = sync_pool();
    const int device = mi355::current_device();
    for (size_t i = 0; i < pool.size(); i++)
        if (pool[i].device == device && pool[i].stream == st) {
            if (pool[i].dev) (void)hipFree(pool[i].dev);
            pool.erase(pool.begin() + (long)i);
            return;
        }
}

extern "C" int mi355_h264_deblock_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream)
{
    return mi355_h264_deblock_layouts_dev(d_frames, nframes, max_mb_width, max_mb_height, MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED, stream);
}

extern "C" int mi355_h264_deblock_layouts_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, int layouts, void *stream)
{
    if (!mi355::bind() || !d_frames || nframes <= 0 || !(layouts & (MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED))) return -1;
    const int nbands = (max_mb_height + 3) / 4, nsteps = max_mb_width + 6;
    if (nbands <= 0 || max_mb_width <= 0) return -1;
    /* Tiled pictures: ONE launch for all bands of all pictures (k_deblock_tiled).  Pictures with line strides (field pictures, 4:4:4 plane
     * passes, callers that keep AVFrame-like planes): the forms of rounds 1-3, whose chunked row pieces suit that layout — a launch per
     * band, or 2 to 6 bands per workgroup when the pictures are few.  MI355_DEBLOCK_FORM (developer switch) = 1 / 2 / 3 / 4 / 6 pins the
     * latter's bands per workgroup and sends tiled pictures through it as well (what round 3 measured). */
    static const int force = std::getenv("MI355_DEBLOCK_FORM") ? std::atoi(std::getenv("MI355_DEBLOCK_FORM")) : 0;
    hipStream_t st = (hipStream_t)stream;
    const bool tiled_launch = (layouts & MI355_LAYOUTS_TILED) && force == 0;
    if (tiled_launch) {
        if ((long long)nframes * nbands > 0x7FFFFFFFLL) return -3;
        const size_t words = 16 + (size_t)nframes * (size_t)nbands;
        uint32_t *sync = mi355::sync_words(st, words);
        if (!sync) return -4;
        MI355_TRY(hipMemsetAsync(sync, 0, words * sizeof(uint32_t), st), -4);
        /* few bands in all (fewer than two per SIMD): two waves per band, the edge phases beside everything else (MI355_DEBLOCK_WAVES = 1 / 2 pins the form) */
        static const int pin = std::getenv("MI355_DEBLOCK_WAVES") ? std::atoi(std::getenv("MI355_DEBLOCK_WAVES")) : 0;
        static int simds = 0;
        if (!simds) {
            hipDeviceProp_t prop;
            int dev = 0;
            simds = 4 * (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
        }
        const bool two = pin == 2 || (pin != 1 && (long long)nframes * nbands < 2LL * simds);
        if (two) hipLaunchKernelGGL(k_deblock_tiled2, dim3((unsigned)(nframes * nbands)), dim3(128), 0, st, d_frames, nframes, nbands, sync);
        else hipLaunchKernelGGL(k_deblock_tiled, dim3((unsigned)(nframes * nbands)), dim3(64), 0, st, d_frames, nframes, nbands, sync);
        if (!(layouts & MI355_LAYOUTS_LINEAR)) return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        cus = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    /* sequential steps a form needs: launches x (steps of one walk), times the rounds it takes to get all pictures through the
     * CUs (8 band tiles fit a CU); a step of the multi-band form is dearer by its barrier (measured: 2.9 against 2.3 us) */
    int best = 1;
    double best_cost = 0;
    const int forms[5] = { 1, 2, 3, 4, 6 };
    for (int i = 0; i < 5; i++) {
        const int kw = forms[i];
        const long resident = (long)cus * (DEBLOCK_WAVES_PER_CU / kw);
        const double rounds = (double)((nframes + resident - 1) / resident);
        const double cost = rounds * ((nbands + kw - 1) / kw) * (nsteps + DEBLOCK_LAG * (kw - 1)) * (kw > 1 ? 1.25 : 1.0);
        if (i == 0 || cost < best_cost) { best = kw; best_cost = cost; }
    }
    if (force == 1 || force == 2 || force == 3 || force == 4 || force == 6) best = force;
    const int skip = tiled_launch ? 1 : 0;
    for (int band = 0; band < nbands; band += best) {
        const dim3 grid((unsigned)nframes);
        switch (best) {
        case 2: hipLaunchKernelGGL(k_deblock_bands<2>, grid, dim3(128), 0, st, d_frames, band, skip); break;
        case 3: hipLaunchKernelGGL(k_deblock_bands<3>, grid, dim3(192), 0, st, d_frames, band, skip); break;
        case 4: hipLaunchKernelGGL(k_deblock_bands<4>, grid, dim3(256), 0, st, d_frames, band, skip); break;
        case 6: hipLaunchKernelGGL(k_deblock_bands<6>, grid, dim3(384), 0, st, d_frames, band, skip); break;
        default: hipLaunchKernelGGL(k_deblock, grid, dim3(64), 0, st, d_frames, band, skip); break;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
