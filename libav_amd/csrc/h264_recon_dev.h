/*
 * h264_recon_dev.h — the device code of the inter reconstruction pass (one macroblock per wave: record + coefficients -> LDS,
 * quarter-sample motion compensation, weighted prediction, inverse transforms, residual add, store), shared by the translation
 * units that instantiate its kernels: h264_frame.hip (every surface layout; lane id opaque to the optimiser, see lane_id() in
 * h264_dev.h) and h264_frame_tiled.hip (the tiled form alone, lane id plain: that instance has registers to spare for what the
 * compiler then hoists).  Reference behaviour restated: hl_decode_mb (h264_mb_template.c:41-257), hl_motion
 * (h264_mc_template.c:64-163), mc_part_* / mc_dir_part (h264_mb.c:204-471), hl_decode_mb_idct_luma (h264_mb.c:726-795).
 * Everything lives in an anonymous namespace: each translation unit gets its own copy.
 */
#ifndef MI355_H264_RECON_DEV_H
#define MI355_H264_RECON_DEV_H

#include "h264_frame_dev.h"

using namespace mi355;

namespace {


/* byte offsets of the prediction tiles in MbLds (behind the MbCore part; MbLds is no standard-layout type, offsetof() is not for it) */
constexpr int MB_PY_OFF = 960, MB_PC_OFF = 960 + 256;
/* Scratch of the general partition path on tiled surfaces (hl_motion4): the macroblock as sixteen 4x4 blocks, each with its own
 * vector, reference and block-aligned window */
struct __attribute__((aligned(16))) Mc4Scratch {
    uint32_t winY[16][9][3];     /* block b: rows iy - 2 .. iy + 6, byte j = picture column ix - 4 + j (block column 0 on a dword) */
    uint32_t winC[2][16][3];     /* plane, block: rows cy .. cy + 2, byte j = column cx + j (j = 0..2) */
    int16_t tmp[16][9][4];       /* unclipped horizontal 6-tap sums of a block's nine rows (centre positions) */
    uint64_t qref[2][4][2];      /* [list][quadrant]: luma and chroma plane of the quadrant's reference picture */
};
/* what every kernel keeps of a macroblock: record, vectors, coefficients (the intra kernel's LDS holds this part alone) */
struct __attribute__((aligned(16))) MbCore {
    mi355_h264_mb hdr;
    uint32_t mv[2][16];                      /* (x | y << 16) per 4x4 block, raster order, per list */
    int16_t coef[384];
};
static_assert(sizeof(MbCore) == 960, "record + vectors + coefficients");
/* ... and the inter kernels' prediction tiles and motion scratch behind it */
struct __attribute__((aligned(16))) MbLds : MbCore {
    uint8_t py[16 * 16], pc[2][8 * 8];       /* prediction -> reconstruction */
    uint8_t qy[16 * 16], qc[2][8 * 8];       /* second prediction for weighted bi-pred */
    union {
        McScratch mc;                        /* one partition's windows (the 16x16 path; every path on surfaces with line strides) */
        Mc4Scratch mc4;                      /* the windows of sixteen 4x4 blocks (the general path on tiled surfaces) */
    };
};


/* The fields of a picture descriptor the reconstruction kernels use, read ONCE into scalar registers at kernel start.
 * Reading them through the descriptor inside a loop that also stores samples makes every read a fresh (vector) load
 * that the stores might alias: a dependent memory access in front of each macroblock. */
struct FrameHot {
    const mi355_h264_mb *mb;
    const int16_t *mv[2];
    const int16_t *coef;
    const mi355_h264_slice *slices;
    uint8_t *recon[3];
    int32_t recon_stride[2], ref_stride[2];
    int32_t mb_width, mb_height;
    const mi355_h264_frame *desc;     /* for the reference table when a kernel keeps no copy of it in LDS */
};
__device__ __forceinline__ FrameHot frame_hot(const mi355_h264_frame &fr)
{
    FrameHot h;
    h.desc = &fr;
    h.mb = mi355_global(fr.mb);
    h.mv[0] = mi355_global(fr.mv[0]); h.mv[1] = mi355_global(fr.mv[1]);
    h.coef = mi355_global(fr.coef);
    h.slices = mi355_global(fr.slices);
    h.recon[0] = mi355_global(fr.recon[0]); h.recon[1] = mi355_global(fr.recon[1]); h.recon[2] = mi355_global(fr.recon[2]);
    h.recon_stride[0] = uniform(fr.recon_stride[0]); h.recon_stride[1] = uniform(fr.recon_stride[1]);
    h.ref_stride[0] = uniform(fr.dst_stride[0]); h.ref_stride[1] = uniform(fr.dst_stride[1]);
    h.mb_width = uniform(fr.mb_width); h.mb_height = uniform(fr.mb_height);
    return h;
}
/* the picture's reference table in LDS: [slot][plane] */
typedef const uint8_t *const (*RefTable)[3];

/* record (64 B), motion vectors (64 B per list) and coefficients (768 B) -> LDS: every load is
 * issued before the first wait, so the wave pays one memory round trip for all of them.  The two
 * halves can be separated: a strip kernel issues the loads of the next macroblock before it works on
 * the current one. */
struct MbLoad {
    uint32_t hw, mw, c0, c1, c2;
};
/* Every lane loads something valid, nothing is predicated: lanes 0..15 the record, 16..31 / 32..47 the list-0 / list-1
 * vectors (a missing list reads the record instead and is zeroed in commit), the rest repeat the record — with loads
 * under lane conditions the compiler merged each result with its zero default before issuing the next load, i.e. the wave
 * paid the record's memory latency and then the coefficients' again. */
__device__ __forceinline__ void load_mb_issue(MbLoad &r, const FrameHot &fr, int mb_xy, bool with_coefs, bool ok)
{
    const int lane = lane_id();
    r.hw = r.mw = r.c0 = r.c1 = r.c2 = 0;
    if (!ok) return;
    const uint32_t *hp = reinterpret_cast<const uint32_t *>(&fr.mb[mb_xy]);
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(fr.coef + (size_t)mb_xy * MI355_H264_COEFS_PER_MB);
    const uint32_t *m0 = fr.mv[0] ? reinterpret_cast<const uint32_t *>(fr.mv[0]) + (size_t)mb_xy * 16 : hp;
    const uint32_t *m1 = fr.mv[1] ? reinterpret_cast<const uint32_t *>(fr.mv[1]) + (size_t)mb_xy * 16 : hp;
    r.hw = hp[lane & 15];
    r.mw = (lane & 32 ? m1 : m0)[lane & 15];
    if (with_coefs) { r.c0 = cp[lane]; r.c1 = cp[lane + 64]; r.c2 = cp[lane + 128]; }
    MI355_ISSUE_FENCE();      /* keep the five loads here: the compiler otherwise sinks each one into the branch of commit() that uses it */
}
__device__ __forceinline__ void load_mb_commit(MbCore &s, const MbLoad &r, bool with_coefs, bool has0, bool has1)
{
    const int lane = lane_id();
    if (lane < 16) reinterpret_cast<uint32_t *>(&s.hdr)[lane] = r.hw;
    else if (lane < 48) s.mv[(lane >> 4) - 1][lane & 15] = (lane & 32 ? has1 : has0) ? r.mw : 0u;
    if (with_coefs) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.coef);
        dst[lane] = r.c0; dst[lane + 64] = r.c1; dst[lane + 128] = r.c2;
    }
    MI355_WAVE_SYNC();
}
__device__ inline void load_mb(MbCore &s, const FrameHot &fr, int mb_xy, bool with_coefs)
{
    MbLoad r;
    load_mb_issue(r, fr, mb_xy, with_coefs, true);
    load_mb_commit(s, r, with_coefs, fr.mv[0] != nullptr, fr.mv[1] != nullptr);
}

/* The same 960 bytes as ONE access of 16 bytes per lane: lanes 0-3 the record, 4-7 / 8-11 the list-0 / list-1 vectors (a
 * missing list reads sixteen zero bytes), 12-59 the coefficients, 60-63 repeat lane 59 — and one 16-byte LDS write per
 * lane (done by the memory pipeline itself: lds_dma16), because hdr, mv and coef follow each other in MbLds in that order.  The L1 handles a wave's access four lanes at a
 * time whatever their width: five dword accesses of 64 lanes are 80 such groups, this is 15 (k_recon_inter's memory
 * pipeline was busy 60 % of the time, profiles/r02g_pmc3_h264_f2048.json). */
static_assert(sizeof(mi355_h264_mb) == 64 && offsetof(MbCore, mv) == 64 && offsetof(MbCore, coef) == 192 && MI355_H264_COEFS_PER_MB == 384, "MbLds begins with the 960 bytes load_mb_wide fills");
__device__ const uint32_t k_zero16[4] = { 0u, 0u, 0u, 0u };           /* what the lanes of a missing vector list read */
__device__ __forceinline__ void load_mb_wide(MbLds &s, const FrameHot &fr, int mb_xy)
{
    const int lane = lane_id(), l = lane < 60 ? lane : 59;
    const uint8_t *hp = reinterpret_cast<const uint8_t *>(&fr.mb[mb_xy]);
    const uint8_t *cp = reinterpret_cast<const uint8_t *>(fr.coef + (size_t)mb_xy * MI355_H264_COEFS_PER_MB);
    const uint8_t *zp = reinterpret_cast<const uint8_t *>(k_zero16);
    const bool has0 = fr.mv[0] != nullptr, has1 = fr.mv[1] != nullptr;
    const uint8_t *m0 = has0 ? reinterpret_cast<const uint8_t *>(fr.mv[0]) + (size_t)mb_xy * 64 + 16 * (l - 4) : zp;
    const uint8_t *m1 = has1 ? reinterpret_cast<const uint8_t *>(fr.mv[1]) + (size_t)mb_xy * 64 + 16 * (l - 8) : zp;
    /* lane l reads 16 bytes at offset 16 * (l - first lane of its part) of its part, and the memory pipeline writes them to LDS at 16 * lane:
     * no register, no LDS write instruction (lanes 60-63 land in s.py, which nothing has written yet: one macroblock per wave) */
    const uint8_t *src = l < 4 ? hp + 16 * l : (l < 8 ? m0 : (l < 12 ? m1 : cp + 16 * (l - 12)));
    lds_dma16<false>(src, reinterpret_cast<uint8_t *>(&s));
    lds_dma_wait();
    MI355_WAVE_SYNC();
}

/* one prediction direction of one partition: mc_dir_part, h264_mb.c:204-318 */
template <bool TILED>
__device__ __forceinline__ void mc_dir(MbLds &s, const FrameHot &fr, RefTable refs, const mi355_h264_slice &sl, int mb_x, int mb_y,
                              int mb_xy, int list, int n_raster, int refn, int bx, int by, int w, int h,
                              uint8_t *py, uint8_t *pcb, uint8_t *pcr, int avg)
{
    (void)sl; (void)refn; (void)mb_xy;
    const uint32_t mvw = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.mv[list][n_raster]);
    const int slot = __builtin_amdgcn_readfirstlane((int)s.hdr.u.inter.ref_pic[list][(bx >> 3) + 2 * (by >> 3)]);
    const int mx = (int16_t)(mvw & 0xFFFF) + (mb_x * 16 + bx) * 4;
    const int my = (int16_t)(mvw >> 16) + (mb_y * 16 + by) * 4;
    /* the chroma vector of a field predicted from a field of the other parity: h264_mb.c:287-291 (0 in frame pictures) */
    const int myc = my + __builtin_amdgcn_readfirstlane((int)s.hdr.u.inter.chroma_dy[list][(bx >> 3) + 2 * (by >> 3)]);
    (void)refs;
    const uint8_t *const *rp = fr.desc->ref[slot < MI355_H264_MAX_SLOTS ? slot : 0];
    if (TILED) {
        const TiledRef tr{mi355_global(rp[0]), mi355_global(rp[1]), fr.ref_stride[0], fr.ref_stride[1], fr.mb_width, fr.mb_height};
        stage_windows_tiled(s.mc, tr, mx >> 2, my >> 2, w, h, mx >> 3, myc >> 3, w >> 1, h >> 1);
        RPROF(3);
        mc_luma_compute(s.mc, mx & 3, my & 3, w, h, py, 16, bx, by, avg);
        RPROF(4);
        if (w == 16 && h == 16) mc_chroma16(s.mc, mx & 7, myc & 7, pcb, pcr, 8, avg);
        else mc_chroma_compute(s.mc, 2, mx & 7, myc & 7, w >> 1, h >> 1, pcb, pcr, 8, bx >> 1, by >> 1, avg);
        return;
    }
    PlaneRef ry{mi355_global(rp[0]), fr.ref_stride[0], 16 * fr.mb_width, 16 * fr.mb_height};
    PlaneRef rb{mi355_global(rp[1]), fr.ref_stride[1], 8 * fr.mb_width, 8 * fr.mb_height};
    PlaneRef rr{mi355_global(rp[2]), fr.ref_stride[1], 8 * fr.mb_width, 8 * fr.mb_height};
    if (w == 16 && h == 16) stage_windows16(s.mc, ry, mx >> 2, my >> 2, rb, rr, mx >> 3, myc >> 3);
    else stage_windows(s.mc, &ry, mx >> 2, my >> 2, w, h, &rb, &rr, mx >> 3, myc >> 3, w >> 1, h >> 1);
    mc_luma_compute(s.mc, mx & 3, my & 3, w, h, py, 16, bx, by, avg);
    RPROF(4);
    if (w == 16 && h == 16) mc_chroma16(s.mc, mx & 7, myc & 7, pcb, pcr, 8, avg);
    else mc_chroma_compute(s.mc, 2, mx & 7, myc & 7, w >> 1, h >> 1, pcb, pcr, 8, bx >> 1, by >> 1, avg);
}

/* mc_part (h264_mc_template.c:44-62) -> mc_part_std / mc_part_weighted (h264_mb.c:320-471).  Both lists go
 * through ONE call site of mc_dir (inlined): a second prediction lands in the q* tiles when the two have to be
 * blended with weights, on top of the first one (rounded average) otherwise. */
template <bool TILED>
__device__ __forceinline__ void mc_part(MbLds &s, const FrameHot &fr, RefTable refs, const mi355_h264_slice &sl, int mb_x, int mb_y,
                                        int mb_xy, int n_raster, int quadrant, int bx, int by, int w, int h, int l0, int l1)
{
    const int r0 = uniform(s.hdr.ref_idx[0][quadrant]), r1 = uniform(s.hdr.ref_idx[1][quadrant]);
    const bool weighted = (uniform(s.hdr.flags) & MI355_MBF_WEIGHTED) &&
                          ((sl.use_weight == 2 && l0 && l1 && sl.implicit_weight[r0][r1] != 32) || sl.use_weight == 1);
    const bool two = l0 && l1;
#pragma nounroll
    for (int list = 0; list < 2; list++) {
        if (!(list ? l1 : l0)) continue;
        const bool second = list == 1 && two;
        const bool to_q = second && weighted;
        mc_dir<TILED>(s, fr, refs, sl, mb_x, mb_y, mb_xy, list, n_raster, list ? r1 : r0, bx, by, w, h, to_q ? s.qy : s.py, to_q ? s.qc[0] : s.pc[0],
               to_q ? s.qc[1] : s.pc[1], second && !weighted);
    }
    if (!weighted) return;
    uint8_t *dy = s.py + by * 16 + bx, *dcb = s.pc[0] + (by >> 1) * 8 + (bx >> 1), *dcr = s.pc[1] + (by >> 1) * 8 + (bx >> 1);
    if (two) {
        const uint8_t *ty = s.qy + by * 16 + bx, *tcb = s.qc[0] + (by >> 1) * 8 + (bx >> 1), *tcr = s.qc[1] + (by >> 1) * 8 + (bx >> 1);
        if (sl.use_weight == 2) {
            const int w0 = sl.implicit_weight[r0][r1], w1 = 64 - w0;
            biweight_block(dy, ty, 16, w, h, 5, w0, w1, 0);
            biweight_block(dcb, tcb, 8, w >> 1, h >> 1, 5, w0, w1, 0);
            biweight_block(dcr, tcr, 8, w >> 1, h >> 1, 5, w0, w1, 0);
        } else {
            biweight_block(dy, ty, 16, w, h, sl.luma_log2_weight_denom, sl.luma_weight[r0][0][0], sl.luma_weight[r1][1][0],
                           sl.luma_weight[r0][0][1] + sl.luma_weight[r1][1][1]);
            biweight_block(dcb, tcb, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][0][0],
                           sl.chroma_weight[r1][1][0][0], sl.chroma_weight[r0][0][0][1] + sl.chroma_weight[r1][1][0][1]);
            biweight_block(dcr, tcr, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][1][0],
                           sl.chroma_weight[r1][1][1][0], sl.chroma_weight[r0][0][1][1] + sl.chroma_weight[r1][1][1][1]);
        }
    } else {
        const int list = l1 ? 1 : 0, refn = list ? r1 : r0;
        weight_block(dy, 16, w, h, sl.luma_log2_weight_denom, sl.luma_weight[refn][list][0], sl.luma_weight[refn][list][1]);
        if (sl.use_weight_chroma) {
            weight_block(dcb, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][0][0], sl.chroma_weight[refn][list][0][1]);
            weight_block(dcr, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][1][0], sl.chroma_weight[refn][list][1][1]);
        }
    }
}

/* ---- the general partition path on macroblock-tiled surfaces ---------------------------------------------------------------------
 * hl_motion's partition loop (below) spends a whole-wave pass of ~230 instructions on every partition whatever its size: a macroblock
 * of 4x4 sub-partitions is sixteen of them per list (SURVEY 8d's mixed run: 5.6 prediction blocks per macroblock, 36 ms per 2048
 * pictures against 10.4 for 16x16).  Here every macroblock that is not one 16x16 partition is sixteen 4x4 blocks, FOUR LANES each
 * (lane 4b + r = row r of block b, raster order), and a lane works with its own block's vector, reference and quarter-sample case:
 *   - per list, the nine 12-byte window rows of every block are fetched by 144 lane tasks (three rounds): one unaligned 12-byte load
 *     from the tile holding the row's first byte and, for rows that run over the tile's edge, a second one from the next tile, merged
 *     by byte masks (v_bfi); chroma: 3 x 3 bytes per block and plane, 96 tasks;
 *   - the filters are mc_luma_compute's arithmetic (h264qpel_template.c:77-531) with the position flags per lane: every component
 *     (integer sample, horizontal / vertical half sample, centre) is evaluated where ANY lane needs it and added where THIS lane does;
 *   - weights per 8x8 quadrant afterwards (mc_part's formulas: one reference per quadrant and list).
 * Windows that reach over the left or right picture border take per-byte clamped loads (emulated_edge_mc's replication), rows clamp by
 * their row number.  Same results as the partition loop, which stays for surfaces with line strides. */
__device__ __forceinline__ void mc4_list(MbLds &s, const FrameHot &fr, int mb_x, int mb_y, int list, uint32_t use, uint32_t to_q, uint32_t avg)
{
    /* use / to_q / avg: bit b = block b is predicted from this list / its prediction goes to the q tiles (weighted second list) /
     * is averaged into what the first list left */
    Mc4Scratch &m = s.mc4;
    const int wpix = 16 * fr.mb_width, hpix = 16 * fr.mb_height, wc = wpix >> 1, hc = hpix >> 1;
    /* ---- luma windows: task i = 9 b + k ---------------------------------------------------------------------------------- */
#pragma unroll
    for (int round = 0; round < 3; round++) {
        const int i = lane_id() + 64 * round;
        const int ic = i < 144 ? i : 143;
        const int b = (ic * 57) >> 9, k = ic - 9 * b;
        const uint32_t mvw = s.mv[list][b];
        const int x0 = mb_x * 16 + 4 * (b & 3) + ((int)(int16_t)(mvw & 0xFFFF) >> 2) - 4;
        const int y = clip3(mb_y * 16 + 4 * (b >> 2) + ((int)(int16_t)(mvw >> 16) >> 2) - 2 + k, 0, hpix - 1);
        const uint8_t *ref = reinterpret_cast<const uint8_t *>(m.qref[list][(b & 2 ? 1 : 0) + (b & 8 ? 2 : 0)][0]);
        const uint8_t *row = mi355_global_v(ref) + (uint32_t)(__mul24(y >> 4, fr.ref_stride[0]) + (y & 15) * 16);
        const bool cross = x0 < 0 || x0 + 11 > wpix - 1;
        const int xs = clip3(x0, 0, wpix - 12), tx = xs >> 4, ox = xs & 15;
        struct __attribute__((packed)) B12 { uint32_t a, b, c; };
        B12 va, vb;
        __builtin_memcpy(&va, row + tx * 256 + ox, 12);
        /* the part of the row that lies in the next tile: its byte j sits 16 - ox bytes in front of that tile's row */
        __builtin_memcpy(&vb, row + (ox > 4 ? (tx + 1) * 256 - (16 - ox) : tx * 256 + ox), 12);
        MI355_ISSUE_FENCE();
        const int na = 16 - ox;                              /* bytes of the window in the first tile (>= 12: all) */
        uint32_t w[3] = { va.a, va.b, va.c };
        const uint32_t wb[3] = { vb.a, vb.b, vb.c };
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int n = na - 4 * d;                        /* bytes of dword d that come from the first tile */
            const uint32_t keep = n >= 4 ? 0xFFFFFFFFu : (n <= 0 ? 0u : ((1u << (8 * n)) - 1u));
            w[d] = (w[d] & keep) | (wb[d] & ~keep);
        }
        if (__any(cross)) {
            if (cross) {
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    uint32_t v = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int x = clip3(x0 + 4 * d + j, 0, wpix - 1);
                        v |= (uint32_t)row[(x >> 4) * 256 + (x & 15)] << (8 * j);
                    }
                    w[d] = v;
                }
            }
        }
        if (i < 144) { m.winY[b][k][0] = w[0]; m.winY[b][k][1] = w[1]; m.winY[b][k][2] = w[2]; }
    }
    /* ---- chroma windows: task i = 48 plane + 3 b + k ------------------------------------------------------------------------- */
#pragma unroll
    for (int round = 0; round < 2; round++) {
        const int i = lane_id() + 64 * round;
        const int ic = i < 96 ? i : 95;
        const int plane = ic >= 48, rem = ic - 48 * plane, b = (rem * 43) >> 7, k = rem - 3 * b;
        const uint32_t mvw = s.mv[list][b];
        const int x0 = mb_x * 8 + 2 * (b & 3) + ((int)(int16_t)(mvw & 0xFFFF) >> 3);
        const int y = clip3(mb_y * 8 + 2 * (b >> 2) + ((int)(int16_t)(mvw >> 16) >> 3) + k, 0, hc - 1);
        const uint8_t *ref = reinterpret_cast<const uint8_t *>(m.qref[list][(b & 2 ? 1 : 0) + (b & 8 ? 2 : 0)][1]);
        const uint8_t *row = mi355_global_v(ref) + (uint32_t)(__mul24(y >> 3, fr.ref_stride[1]) + plane * 64 + (y & 7) * 8);
        const bool cross = x0 < 0 || x0 + 2 > wc - 1;
        const int xs = clip3(x0, 0, wc - 4), tx = xs >> 3, ox = xs & 7;
        uint32_t va, vb;
        __builtin_memcpy(&va, row + tx * 128 + ox, 4);
        __builtin_memcpy(&vb, row + (ox > 4 ? (tx + 1) * 128 - (8 - ox) : tx * 128 + ox), 4);
        MI355_ISSUE_FENCE();
        const int na = 8 - ox;
        const uint32_t keep = na >= 4 ? 0xFFFFFFFFu : ((1u << (8 * na)) - 1u);
        /* a window that ends on the picture's last column starts one byte behind the last four-byte load that stays inside the row */
        uint32_t w = ((va & keep) | (vb & ~keep)) >> (8 * (cross ? 0 : x0 - xs));
        if (__any(cross)) {
            if (cross) {
                w = 0;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int x = clip3(x0 + j, 0, wc - 1);
                    w |= (uint32_t)row[(x >> 3) * 128 + (x & 7)] << (8 * j);
                }
            }
        }
        if (i < 96) m.winC[plane][b][k] = w;
    }
    MI355_WAVE_SYNC();
    /* ---- per lane: block b = lane >> 2, row r = lane & 3 -------------------------------------------------------------------------- */
    const int lane = lane_id(), b = lane >> 2, r = lane & 3;
    const uint32_t mvw = s.mv[list][b];
    const int mx = (int)(mvw & 3u), my = (int)((mvw >> 16) & 3u);
    const bool used = (use >> b) & 1;
    const bool use_j = used && ((mx == 2 && my != 0) || (my == 2 && mx != 0));
    const bool use_b = used && mx != 0 && my != 2;
    const bool use_h = used && my != 0 && mx != 2;
    const bool use_g = used && (mx == 0 || my == 0) && ((mx | my) != 2);
    if (__any(use_j)) {
        /* unclipped horizontal sums of the nine rows of every block that has a centre position: task i = 9 b + k again */
#pragma unroll
        for (int round = 0; round < 3; round++) {
            const int i = lane_id() + 64 * round;
            const int ic = i < 144 ? i : 143;
            const int tb = (ic * 57) >> 9, k = ic - 9 * tb;
            const uint32_t *w = m.winY[tb][k];
            uint32_t te, to;
            pk_htaps(w[0], w[1], w[2], te, to);
            if (i < 144) {
                uint32_t *t = reinterpret_cast<uint32_t *>(m.tmp[tb][k]);
                t[0] = (te & 0xFFFFu) | (to << 16);
                t[1] = (te >> 16) | (to & 0xFFFF0000u);
            }
        }
        MI355_WAVE_SYNC();
    }
    uint32_t se = 0, so = 0;                                 /* sums of the components, samples (0,2) and (1,3) */
    if (__any(use_g)) {
        const int gdx = (my == 0 && mx == 3), gdy = (mx == 0 && my == 3);
        const uint32_t *w = m.winY[b][r + 2 + gdy];
        const uint32_t g = gdx ? mi355_alignbyte(w[2], w[1], 1) : w[1];
        if (use_g) { se = pk_even(g); so = pk_odd(g); }
    }
    if (__any(use_b)) {
        const int bdy = my == 3;
        uint32_t te, to;
        if (use_j) {
            const uint32_t *t = reinterpret_cast<const uint32_t *>(m.tmp[b][r + 2 + bdy]);
            te = (t[0] & 0xFFFFu) | (t[1] << 16);
            to = (t[0] >> 16) | (t[1] & 0xFFFF0000u);
        } else {
            const uint32_t *w = m.winY[b][r + 2 + bdy];
            pk_htaps(w[0], w[1], w[2], te, to);
        }
        if (use_b) { se = pk_add(se, pk_round5(te)); so = pk_add(so, pk_round5(to)); }
    }
    if (__any(use_h)) {
        const int hdx = mx == 3;
        uint32_t e[6], o[6];
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const uint32_t *w = m.winY[b][r + q];
            const uint32_t c = hdx ? mi355_alignbyte(w[2], w[1], 1) : w[1];
            e[q] = pk_even(c); o[q] = pk_odd(c);
        }
        const uint32_t he = pk_round5(pk_tap6(e[0], e[1], e[2], e[3], e[4], e[5])), ho = pk_round5(pk_tap6(o[0], o[1], o[2], o[3], o[4], o[5]));
        if (use_h) { se = pk_add(se, he); so = pk_add(so, ho); }
    }
    if (__any(use_j)) {
        /* second pass over the unclipped sums: mc_luma_compute's staged-shift form (exact, see there) */
        const uint32_t *t = reinterpret_cast<const uint32_t *>(m.tmp[b][r]);
        uint32_t jv[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
            const uint32_t af = pk_add(t[hlf], t[10 + hlf]), be = pk_add(t[2 + hlf], t[8 + hlf]), cd = pk_add(t[4 + hlf], t[6 + hlf]);
            const uint32_t t1 = pk_ashr(pk_sub(af, be), 2);
            const uint32_t t2 = pk_ashr(pk_adds(pk_sub(t1, be), cd), 2);
            jv[hlf] = pk_clip_u8(pk_ashr(pk_add(pk_add(t2, cd), 0x00200020u), 6));
        }
        if (use_j) { se = pk_add(se, byte_perm(jv[1], jv[0], 0x05040100u)); so = pk_add(so, byte_perm(jv[1], jv[0], 0x07060302u)); }
    }
    if ((int)use_g + (int)use_b + (int)use_h + (int)use_j == 2) {
        se = pk_ashr(pk_add(se, 0x00010001u), 1); so = pk_ashr(pk_add(so, 0x00010001u), 1);
    }
    const bool q_dst = (to_q >> b) & 1, do_avg = (avg >> b) & 1;
    {
        const uint32_t v = pk_bytes(se, so);
        uint32_t *d = reinterpret_cast<uint32_t *>((q_dst ? s.qy : s.py) + (4 * (b >> 2) + r) * 16 + 4 * (b & 3));
        if (used) *d = do_avg ? rnd_avg4(*d, v) : v;
    }
    /* chroma (h264chroma_template.c:27-173): lane (b, r) = plane r >> 1, row r & 1 of the block's 2x2 samples */
    {
        const int plane = r >> 1, cy = r & 1;
        const int fx = (int)(mvw & 7u), fy = (int)((mvw >> 16) & 7u);
        const int A = (8 - fx) * (8 - fy), B = fx * (8 - fy), C = (8 - fx) * fy, D = fx * fy;
        const uint32_t r0 = m.winC[plane][b][cy], r1 = m.winC[plane][b][cy + 1];
        const uint32_t a0 = byte_perm(0, r0, 0x0C010C00u), a1 = byte_perm(0, r0, 0x0C020C01u);
        const uint32_t b0 = byte_perm(0, r1, 0x0C010C00u), b1 = byte_perm(0, r1, 0x0C020C01u);
        const uint32_t v = pk_ashr(pk_mad(a0, A, pk_mad(a1, B, pk_mad(b0, C, pk_mad(b1, D, 0x00200020u)))), 6);
        uint32_t two = byte_perm(0, v, 0x0C0C0200u);
        uint16_t *d = reinterpret_cast<uint16_t *>((q_dst ? s.qc[plane] : s.pc[plane]) + (2 * (b >> 2) + cy) * 8 + 2 * (b & 3));
        if (used) {
            if (do_avg) two = rnd_avg4(*d, two);
            *d = (uint16_t)two;
        }
    }
    MI355_WAVE_SYNC();
}

__device__ inline void hl_motion4(MbLds &s, const FrameHot &fr, const mi355_h264_slice &sl, int mb_x, int mb_y)
{
    const uint32_t t = (uint32_t)uniform((int)s.hdr.mb_type);
    const int kind = (t & MI355_MB_16x8) ? 1 : ((t & MI355_MB_8x16) ? 2 : ((t & MI355_MB_16x16) ? 0 : 3));
    /* the quadrants' reference planes -> LDS (lane 4 * list + quadrant) */
    {
        const int lane = lane_id(), l8 = lane & 7, list = l8 >> 2, q = l8 & 3;
        const int slot = s.hdr.u.inter.ref_pic[list][q];
        const uint8_t *const *rp = fr.desc->ref[slot < MI355_H264_MAX_SLOTS ? slot : 0];
        const uint64_t py = (uint64_t)reinterpret_cast<uintptr_t>(rp[0]), pc = (uint64_t)reinterpret_cast<uintptr_t>(rp[1]);
        MI355_ISSUE_FENCE();
        if (lane < 8) { s.mc4.qref[list][q][0] = py; s.mc4.qref[list][q][1] = pc; }
    }
    /* per quadrant (wave-uniform): which lists, which weights */
    uint32_t use0 = 0, use1 = 0, wq = 0;
    const bool slice_w = (uniform(s.hdr.flags) & MI355_MBF_WEIGHTED) != 0;
    for (int q = 0; q < 4; q++) {
        int l0, l1;
        if (kind == 3) { const int st = uniform(s.hdr.sub_mb_type[q]); l0 = (st & MI355_SUB_L0) != 0; l1 = (st & MI355_SUB_L1) != 0; }
        else {
            const int part = kind == 0 ? 0 : (kind == 1 ? q >> 1 : q & 1);
            l0 = (int)((t >> (12 + part)) & 1); l1 = (int)((t >> (14 + part)) & 1);
        }
        const uint32_t blocks = 0x33u << (2 * (q & 1) + 8 * (q >> 1));          /* the quadrant's four 4x4 blocks (raster bits) */
        if (l0) use0 |= blocks;
        if (l1) use1 |= blocks;
        const int r0 = uniform(s.hdr.ref_idx[0][q]), r1 = uniform(s.hdr.ref_idx[1][q]);
        if (slice_w && ((sl.use_weight == 2 && l0 && l1 && sl.implicit_weight[r0][r1] != 32) || sl.use_weight == 1)) wq |= blocks;
    }
    MI355_WAVE_SYNC();
    if (use0) mc4_list(s, fr, mb_x, mb_y, 0, use0, 0, 0);
    if (use1) mc4_list(s, fr, mb_x, mb_y, 1, use1, use0 & use1 & wq, use0 & use1 & ~wq);
    if (!wq) return;
    /* weights, per quadrant: mc_part_weighted, h264_mb.c:369-471 */
    for (int q = 0; q < 4; q++) {
        const uint32_t blocks = 0x33u << (2 * (q & 1) + 8 * (q >> 1));
        if (!(wq & blocks)) continue;
        const int bx = 8 * (q & 1), by = 8 * (q >> 1);
        const bool l0 = (use0 & blocks) != 0, l1 = (use1 & blocks) != 0;
        const int r0 = uniform(s.hdr.ref_idx[0][q]), r1 = uniform(s.hdr.ref_idx[1][q]);
        uint8_t *dy = s.py + by * 16 + bx, *dcb = s.pc[0] + (by >> 1) * 8 + (bx >> 1), *dcr = s.pc[1] + (by >> 1) * 8 + (bx >> 1);
        if (l0 && l1) {
            const uint8_t *ty = s.qy + by * 16 + bx, *tcb = s.qc[0] + (by >> 1) * 8 + (bx >> 1), *tcr = s.qc[1] + (by >> 1) * 8 + (bx >> 1);
            if (sl.use_weight == 2) {
                const int w0 = sl.implicit_weight[r0][r1], w1 = 64 - w0;
                biweight_block(dy, ty, 16, 8, 8, 5, w0, w1, 0);
                biweight_block(dcb, tcb, 8, 4, 4, 5, w0, w1, 0);
                biweight_block(dcr, tcr, 8, 4, 4, 5, w0, w1, 0);
            } else {
                biweight_block(dy, ty, 16, 8, 8, sl.luma_log2_weight_denom, sl.luma_weight[r0][0][0], sl.luma_weight[r1][1][0],
                               sl.luma_weight[r0][0][1] + sl.luma_weight[r1][1][1]);
                biweight_block(dcb, tcb, 8, 4, 4, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][0][0],
                               sl.chroma_weight[r1][1][0][0], sl.chroma_weight[r0][0][0][1] + sl.chroma_weight[r1][1][0][1]);
                biweight_block(dcr, tcr, 8, 4, 4, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][1][0],
                               sl.chroma_weight[r1][1][1][0], sl.chroma_weight[r0][0][1][1] + sl.chroma_weight[r1][1][1][1]);
            }
        } else {
            const int list = l1 ? 1 : 0, refn = list ? r1 : r0;
            weight_block(dy, 16, 8, 8, sl.luma_log2_weight_denom, sl.luma_weight[refn][list][0], sl.luma_weight[refn][list][1]);
            if (sl.use_weight_chroma) {
                weight_block(dcb, 8, 4, 4, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][0][0], sl.chroma_weight[refn][list][0][1]);
                weight_block(dcr, 8, 4, 4, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][1][0], sl.chroma_weight[refn][list][1][1]);
            }
        }
    }
}

/* hl_motion, h264_mc_template.c:64-163.  The partitions are enumerated by one loop so that mc_part has
 * a single (inlined) call site. */
template <bool TILED>
__device__ inline void hl_motion(MbLds &s, const FrameHot &fr, RefTable refs, const mi355_h264_slice &sl, int mb_x, int mb_y, int mb_xy)
{
    const uint32_t t = (uint32_t)uniform((int)s.hdr.mb_type);
#define DIRF(part, list) (int)((t >> (12 + (part) + 2 * (list))) & 1)
    const int kind = (t & MI355_MB_16x16) ? 0 : ((t & MI355_MB_16x8) ? 1 : ((t & MI355_MB_8x16) ? 2 : 3));
    if (kind == 0) {
        /* the common shape gets its own copy of the (inlined) motion code: block size and position are literals there,
         * so tile loops have one iteration, window sizes are constants and the small-block branches disappear */
        const int l0 = DIRF(0, 0), l1 = DIRF(0, 1);
        if (l0 && !l1 && !(uniform(s.hdr.flags) & MI355_MBF_WEIGHTED)) {
            /* ... and the plain P_16x16 / P_Skip macroblock goes straight to one list-0 prediction written in place */
            mc_dir<TILED>(s, fr, refs, sl, mb_x, mb_y, mb_xy, 0, 0, 0, 0, 0, 16, 16, s.py, s.pc[0], s.pc[1], 0);
            return;
        }
        mc_part<TILED>(s, fr, refs, sl, mb_x, mb_y, mb_xy, 0, 0, 0, 0, 16, 16, l0, l1);
        return;
    }
    if (TILED) { hl_motion4(s, fr, sl, mb_x, mb_y); return; }
    const int nparts = kind == 3 ? 16 : 2;
    for (int p = 0; p < nparts; p++) {
        int n, quad, bx, by, w, h, l0, l1;
        if (kind == 1) { n = 8 * p; quad = 2 * p; bx = 0; by = 8 * p; w = 16; h = 8; l0 = DIRF(p, 0); l1 = DIRF(p, 1); }
        else if (kind == 2) { n = 2 * p; quad = p; bx = 8 * p; by = 0; w = 8; h = 16; l0 = DIRF(p, 0); l1 = DIRF(p, 1); }
        else {
            const int i = p >> 2, j = p & 3;
            const int st = uniform(s.hdr.sub_mb_type[i]), shape = st & 3;
            const int cnt = shape == MI355_SUB_8x8 ? 1 : (shape == MI355_SUB_4x4 ? 4 : 2);
            if (j >= cnt) continue;
            l0 = (st & MI355_SUB_L0) != 0; l1 = (st & MI355_SUB_L1) != 0;
            const int x = (i & 1) * 8, y = (i >> 1) * 8;
            quad = i;
            w = (shape == MI355_SUB_8x8 || shape == MI355_SUB_8x4) ? 8 : 4;
            h = (shape == MI355_SUB_8x8 || shape == MI355_SUB_4x8) ? 8 : 4;
            bx = x + (shape == MI355_SUB_4x8 ? 4 * j : (shape == MI355_SUB_4x4 ? 4 * (j & 1) : 0));
            by = y + (shape == MI355_SUB_8x4 ? 4 * j : (shape == MI355_SUB_4x4 ? 4 * (j >> 1) : 0));
            n = (bx >> 2) + 4 * (by >> 2);
        }
        mc_part<TILED>(s, fr, refs, sl, mb_x, mb_y, mb_xy, n, quad, bx, by, w, h, l0, l1);
    }
#undef DIRF
}

/* luma residual of a non-Intra4x4/8x8 MB onto a picture tile: hl_decode_mb_idct_luma
 * (h264_mb.c:726-795) with the dc / full / skip choice of h264idct_template.c:174-201 folded
 * into "transform the block iff it carries a coefficient" (identical results) */
template <bool ALIGNED>   /* ALIGNED: every 4-sample row segment of `y` starts on a dword (both frame kernels lay their tiles out that way) */
__device__ inline void residual_luma(MbCore &s, uint8_t *y, int pitch, bool intra16)
{
    const int lane = lane_id();
    const uint32_t mask = (uint32_t)uniform((int)s.hdr.nnz_mask);
    if (uniform((int)s.hdr.mb_type) & MI355_MB_8x8DCT) {
        const int b = (lane >> 3) & 3, i = lane & 7;
        const bool active = lane < 32;
        int r[8];
        idct8_lds(s.coef + b * 64, i, active, r);
        if (active && ((mask >> (4 * b)) & 1))
            add_col(y + (8 * (b >> 1)) * pitch + 8 * (b & 1) + i, pitch, r, 8);
    } else {
        const int b = lane >> 2, q = lane & 3;
        int c[4], r[4], row;
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = s.coef[b * 16 + q + 4 * i];
        const int dc = s.coef[b * 16];
        idct4_quad(c, q, r, row);
        /* a block without the nnz bit but with a DC value (Intra16x16) takes h264_idct_dc_add
         * (h264idct_template.c:144-156): (dc + 32) >> 6 in int, not the int16 wrap of the full transform */
        const bool coded = (mask >> b) & 1;
        if (!coded) r[0] = r[1] = r[2] = r[3] = (dc + 32) >> 6;
        if (coded || (intra16 && dc))
            add_row4<ALIGNED>(y + (4 * blk_y4(b) + row) * pitch + 4 * blk_x4(b), r);
    }
    MI355_WAVE_SYNC();
}

/* chroma residual: h264_mb_template.c:196-247 */
template <bool ALIGNED>
__device__ inline void residual_chroma(MbCore &s, uint8_t *cb, uint8_t *cr, int pitch)
{
    if (!(uniform(s.hdr.cbp) & 0x30)) return;
    const int lane = lane_id();
    const uint32_t mask = (uint32_t)uniform((int)s.hdr.nnz_mask);
    if (lane < 2 && ((mask >> (MI355_NNZ_CB_DC + lane)) & 1)) {
        int16_t *p = s.coef + 256 + 64 * lane;
        int a = p[0], b = p[16], c = p[32], d = p[48];
        chroma_dc_dequant(a, b, c, d, (int)s.hdr.dc_qmul[1 + lane]);
        p[0] = (int16_t)a; p[16] = (int16_t)b; p[32] = (int16_t)c; p[48] = (int16_t)d;
    }
    MI355_WAVE_SYNC();
    const int j = (lane >> 2) & 7, q = lane & 3;
    int c[4], r[4], row;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = s.coef[256 + j * 16 + q + 4 * i];
    const int dc = s.coef[256 + j * 16];
    idct4_quad(c, q, r, row);
    const bool coded = (mask >> (16 + j)) & 1;
    if (!coded) r[0] = r[1] = r[2] = r[3] = (dc + 32) >> 6;       /* DC only: h264_idct_dc_add, no int16 wrap */
    if (lane < 32 && (coded || dc)) {
        uint8_t *p = (j >> 2) ? cr : cb;
        const int jj = j & 3;
        add_row4<ALIGNED>(p + (4 * (jj >> 1) + row) * pitch + 4 * (jj & 1), r);
    }
    MI355_WAVE_SYNC();
}

/* Residual of an inter macroblock with 4x4 transforms (h264_mb.c:726-795, h264_mb_template.c:196-247; same results as
 * residual_luma + residual_chroma above).  Coefficient block b starts at coef[16 * b] and its nnz bit is bit b for luma,
 * Cb and Cr alike.  A block takes the full transform when its nnz bit is set (h264idct_template.c:33-67) and, for chroma
 * only, h264_idct_dc_add (:144-156, (dc + 32) >> 6 in int) when the bit is clear but ff_h264_chroma_dc_dequant_idct left
 * a DC value.
 *
 * All 24 blocks at once, two lanes per block, two 16-bit values per register: lane h of block b holds columns 2h, 2h + 1
 * (coef[16b + 4i + 2h], [.. + 1] for i = 0..3: one dword each).  The first pass runs along i inside the lane and wraps at
 * 16 bits BY DEFINITION (the reference stores it back into its int16 block), so v_pk_* arithmetic is exact.  The second
 * pass is int in the reference: the four values of a row — the two halves of this lane's register and of the partner
 * lane's — meet in v_dot2_i32_i16 (16-bit factors, 32-bit sum) with factors +-1024, which leaves (sum >> 6) in the upper
 * half of the result (|sum| <= 3.5 * 32768).  A block without anything, and the coefficients other than the DC of a DC-only
 * chroma block, are masked to zero: the transform of a lone DC is (dc + 32) >> 6 in every position, that of nothing is 0.
 *
 * What a lane needs that depends on nothing but its number — tile addresses of its two rows, address of its coefficients,
 * signs — is a ResidLane: computed from the lane number on the device (resid_lane_compute), read from a table built at
 * compile time in the emulator build, which checks the two against each other. */
struct ResidLane {
    uint32_t off_a, off_b;   /* byte offsets in MbLds of this lane's two destination rows (4 samples each) */
    uint32_t cw;             /* byte offset in MbLds of coef[16 b + 2 h] */
    uint32_t dc16;           /* 0xFFFF where the lane's first dword starts with a chroma DC */
    uint32_t misc;           /* bits 0-7: 32 on lane h = 0 (the rounding constant goes onto element 0); bits 8-9: 1 luma, 2 chroma lane */
    uint32_t rc;             /* 32 * 1024 on chroma lanes: rounding of the DC-only form, in the second pass's scale */
    uint32_t ka, kb;         /* factors of this lane's own pair for its two rows */
};
struct ResidLaneTable { ResidLane l[64]; };
constexpr ResidLaneTable make_resid_lanes()
{
    ResidLaneTable t{};
    for (int lane = 0; lane < 64; lane++) {
        const int b = lane >> 1, h = lane & 1, bc = b < 24 ? b : 23, jj = bc & 3;
        const bool chroma = bc >= 16;
        const int x4 = (bc & 1) + 2 * ((bc >> 2) & 1), y4 = ((bc >> 1) & 1) + 2 * (bc >> 3);
        const int base = chroma ? (int)MB_PC_OFF + 64 * ((bc >> 2) & 1) + 32 * (jj >> 1) + 4 * (jj & 1) : (int)MB_PY_OFF + 64 * y4 + 4 * x4;
        const int pitch = chroma ? 8 : 16;
        ResidLane &e = t.l[lane];
        e.off_a = (uint32_t)(base + (h ? 1 : 0) * pitch);
        e.off_b = (uint32_t)(base + (h ? 2 : 3) * pitch);
        e.cw = (uint32_t)((int)offsetof(MbCore, coef) + (bc * 8 + h) * 4);
        e.dc16 = b < 24 && chroma && h == 0 ? 0xFFFFu : 0u;
        e.misc = (h == 0 ? 32u : 0u) | (b >= 24 ? 0u : (chroma ? 0x200u : 0x100u));
        e.rc = b < 24 && chroma ? 32u * 1024u : 0u;
        e.ka = h ? 0xFC00FC00u : 0x04000400u;        /* lane 1: -(v2) - (v3);  lane 0: v0 + v1 */
        e.kb = h ? 0x0400FC00u : 0xFC000400u;        /* lane 1: -(v2) + (v3);  lane 0: v0 - v1 */
    }
    return t;
}
__device__ const ResidLaneTable k_resid_lanes = make_resid_lanes();
/* the same values from the lane number (lanes past block 23 get addresses inside MbLds and no `live` bit) */
__device__ __forceinline__ void resid_lane_compute(ResidLane &r)
{
    const uint32_t lane = (uint32_t)lane_id(), h = lane & 1u;
    const bool chroma = lane >= 32;
    const uint32_t x = (lane & 2u) * 2u;                                                    /* 4 * (b & 1) */
    const uint32_t yl = ((lane & 4u) << 4) | (lane & 8u) | ((lane & 16u) << 3);             /* 64 * b1 + 8 * b2 + 128 * b3 */
    const uint32_t yc = (lane & 12u) << 3;                                                  /* 32 * b1 + 64 * b2 */
    const uint32_t base = chroma ? (uint32_t)MB_PC_OFF + yc + x : (uint32_t)MB_PY_OFF + yl + x;
    const uint32_t pitch = chroma ? 8u : 16u;
    r.off_a = base + h * pitch;
    r.off_b = base + (3u - h) * pitch;
    r.cw = (uint32_t)offsetof(MbCore, coef) + lane * 16u - h * 12u;
    r.dc16 = (lane & 33u) == 32u ? 0xFFFFu : 0u;
    r.misc = (h ? 0u : 32u) | (lane < 32 ? 0x100u : (lane < 48 ? 0x200u : 0u));
    r.rc = chroma ? 32u * 1024u : 0u;
    r.ka = h ? 0xFC00FC00u : 0x04000400u;
    r.kb = h ? 0x0400FC00u : 0xFC000400u;
}
__device__ __forceinline__ void resid_lane_issue(ResidLane &r)
{
    /* the table, two 16-byte loads per lane issued with the record.  Round 2 measured this form 4 % SLOWER than computing the
     * values (37 VALU) — the kernel was waiting for memory then; on tiled surfaces it runs at the VALU's issue rate and the same
     * change is 2.7 % faster (10.77 -> 10.48 ms, profiles/r03_experiments.md).  A second table for the motion code's lane constants
     * (23 more VALU for three more loads) no longer moved the time: 10.52 -> 10.51 ms, not kept.
     * MI355_RESID_LANES_COMPUTED keeps the other form. */
#if defined(MI355_RESID_LANES_COMPUTED) && !defined(MI355_HIP_EMU_H)
    (void)r;
#else
    r = k_resid_lanes.l[lane_id()];
#endif
#if defined(MI355_HIP_EMU_H)
    /* the emulator checks the arithmetic form against the table */
    ResidLane c;
    resid_lane_compute(c);
    if (lane_id() < 48 && (c.off_a != r.off_a || c.off_b != r.off_b || c.cw != r.cw || c.dc16 != r.dc16 || c.misc != r.misc || c.rc != r.rc || c.ka != r.ka || c.kb != r.kb)) abort();
#endif
}
template <bool ALIGNED>
__device__ inline void residual_blocks(MbLds &s, const ResidLane &rl_in)
{
    static_assert(ALIGNED, "the tile rows of the frame kernels start on dwords");
    const int lane = lane_id();
    const uint32_t nnz = (uint32_t)uniform((int)s.hdr.nnz_mask);
    const bool has_chroma = (uniform(s.hdr.cbp) & 0x30) != 0;
    if (!(nnz & 0xFFFFu) && !has_chroma) return;
    if (has_chroma) {
        if (lane < 2 && ((nnz >> (MI355_NNZ_CB_DC + lane)) & 1)) {
            int16_t *p = s.coef + 256 + 64 * lane;
            int a = p[0], b = p[16], c = p[32], d = p[48];
            chroma_dc_dequant(a, b, c, d, (int)s.hdr.dc_qmul[1 + lane]);
            p[0] = (int16_t)a; p[16] = (int16_t)b; p[32] = (int16_t)c; p[48] = (int16_t)d;
        }
        MI355_WAVE_SYNC();
    }
#if !defined(MI355_RESID_LANES_COMPUTED) || defined(MI355_HIP_EMU_H)
    const ResidLane &rl = rl_in;
#else
    ResidLane rl;
    (void)rl_in;
    resid_lane_compute(rl);
#endif
    /* scalars: the nnz bits that count (lanes past block 23 look at bits 24..31: none), chroma switched off as a whole */
    const uint32_t nnz24 = nnz & (has_chroma ? 0xFFFFFFu : 0xFFFFu), hc = has_chroma ? 0xFFFFFFFFu : 0u;
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    const uint32_t keep = bit_mask(nnz24, lane >> 1);                 /* full transform: all coefficients */
    const uint32_t keep0 = keep | (rl.dc16 & hc);                     /* a chroma DC stays without the nnz bit */
    /* + 32 goes onto element 0 in 16 bits for the full transform (block[0] += 1 << 5, :38) but in int for the DC-only form
     * (:148): there it joins the second pass's sums */
    const int rnd = (int)(~keep & rl.rc & hc);
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(base + rl.cw);
    const uint32_t c0 = pk_add(cw[0] & keep0, keep & rl.misc & 0xFFu), c1 = cw[2] & keep, c2 = cw[4] & keep, c3 = cw[6] & keep;
    /* first pass (:42-52), 16-bit wrap */
    const uint32_t z0 = pk_add(c0, c2), z1 = pk_sub(c0, c2), z2 = pk_sub(pk_ashr(c1, 1), c3), z3 = pk_add(c1, pk_ashr(c3, 1));
    const uint32_t w[4] = { pk_add(z0, z3), pk_add(z1, z2), pk_sub(z1, z2), pk_sub(z0, z3) };
    /* second pass (:54-66) for row i: the lane pair holds (v0, v1 | v2, v3).  With xh = the partner's (lo, hi >> 1):
     * lane 0 (own = v0, v1; xh = v2, v3 >> 1): rows 0 and 3 = (v0 + v2) +- (v1 + (v3 >> 1));
     * lane 1 (own = v2, v3; xh = v0, v1 >> 1): rows 1 and 2 = (v0 - v2) +- ((v1 >> 1) - v3) */
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t xh = pk_ashr_hi1((uint32_t)quad_xor1((int)w[i]));
        const int ra = pk_dot2k(xh, 0x04000400u, pk_dot2(w[i], rl.ka, rnd)), rb = pk_dot2k(xh, 0xFC000400u, pk_dot2(w[i], rl.kb, rnd));
        o[i] = byte_perm((uint32_t)rb, (uint32_t)ra, 0x07060302u);          /* the upper halves: (residual a, residual b) */
    }
    if (rl.misc & (has_chroma ? 0x300u : 0x100u)) {
        uint32_t *pa = reinterpret_cast<uint32_t *>(base + rl.off_a), *pb = reinterpret_cast<uint32_t *>(base + rl.off_b);
        const uint32_t va = *pa, vb = *pb;
        /* column i: (sample of row a, sample of row b) + (residual a, residual b), clipped to bytes (a | b << 8) */
        const uint32_t s0 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C040C00u), o[0])), s1 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C050C01u), o[1]));
        const uint32_t s2 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C060C02u), o[2])), s3 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C070C03u), o[3]));
        const uint32_t m01 = byte_perm(s1, s0, 0x05040100u), m23 = byte_perm(s3, s2, 0x05040100u);      /* a0 b0 a1 b1 / a2 b2 a3 b3 */
        *pa = byte_perm(m23, m01, 0x06040200u);
        *pb = byte_perm(m23, m01, 0x07050301u);
    }
    MI355_WAVE_SYNC();
}

/* The same transform arrangement (two lanes per block, residual_blocks above) onto tiles with any row pitch — the intra kernel's — and with the DC-only
 * rule for LUMA as well when `intra16` (an Intra16x16 block without the nnz bit whose DC level ff_h264_luma_dc_dequant_idct left non-zero takes
 * h264_idct_dc_add, (dc + 32) >> 6 in int: h264_mb.c:726-760 with h264idct_template.c:144-156).  Every tile row segment starts on a dword. */
__device__ inline void residual_tile(MbCore &s, uint8_t *y, int ypitch, uint8_t *cb, uint8_t *cr, int cpitch, bool intra16)
{
    const int lane = lane_id(), b = lane >> 1, h = lane & 1;
    const uint32_t nnz = (uint32_t)uniform((int)s.hdr.nnz_mask);
    const bool has_chroma = (uniform(s.hdr.cbp) & 0x30) != 0;
    if (has_chroma) {
        if (lane < 2 && ((nnz >> (MI355_NNZ_CB_DC + lane)) & 1)) {
            int16_t *p = s.coef + 256 + 64 * lane;
            int a = p[0], bb = p[16], c = p[32], d = p[48];
            chroma_dc_dequant(a, bb, c, d, (int)s.hdr.dc_qmul[1 + lane]);
            p[0] = (int16_t)a; p[16] = (int16_t)bb; p[32] = (int16_t)c; p[48] = (int16_t)d;
        }
        MI355_WAVE_SYNC();
    }
    const bool chroma = b >= 16;
    const bool live = b < 16 || (b < 24 && has_chroma);
    const bool dc_rule = chroma ? has_chroma : intra16;                         /* a block of this lane may be DC only */
    const uint32_t nnz24 = nnz & (has_chroma ? 0xFFFFFFu : 0xFFFFu);
    const uint32_t keep = bit_mask(nnz24, b < 24 ? b : 31);
    const uint32_t keep0 = keep | (h == 0 && dc_rule && b < 24 ? 0xFFFFu : 0u);
    const int rnd = (int)(~keep & (dc_rule && b < 24 ? 32u * 1024u : 0u));
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(s.coef) + (b < 24 ? b : 23) * 8 + h;
    const uint32_t c0 = pk_add(cw[0] & keep0, keep & (h ? 0u : 32u)), c1 = cw[2] & keep, c2 = cw[4] & keep, c3 = cw[6] & keep;
    const uint32_t z0 = pk_add(c0, c2), z1 = pk_sub(c0, c2), z2 = pk_sub(pk_ashr(c1, 1), c3), z3 = pk_add(c1, pk_ashr(c3, 1));
    const uint32_t w[4] = { pk_add(z0, z3), pk_add(z1, z2), pk_sub(z1, z2), pk_sub(z0, z3) };
    const uint32_t ka = h ? 0xFC00FC00u : 0x04000400u, kb = h ? 0x0400FC00u : 0xFC000400u;
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t xh = pk_ashr_hi1((uint32_t)quad_xor1((int)w[i]));
        const int ra = pk_dot2k(xh, 0x04000400u, pk_dot2(w[i], ka, rnd)), rb = pk_dot2k(xh, 0xFC000400u, pk_dot2(w[i], kb, rnd));
        o[i] = byte_perm((uint32_t)rb, (uint32_t)ra, 0x07060302u);
    }
    if (live) {
        const int jj = b & 3;
        uint8_t *base = chroma ? ((b & 4) ? cr : cb) + (4 * (jj >> 1)) * cpitch + 4 * (jj & 1)
                               : y + (4 * blk_y4(b)) * ypitch + 4 * blk_x4(b);
        const int pitch = chroma ? cpitch : ypitch;
        uint32_t *pa = reinterpret_cast<uint32_t *>(base + (h ? 1 : 0) * pitch), *pb = reinterpret_cast<uint32_t *>(base + (h ? 2 : 3) * pitch);
        const uint32_t va = *pa, vb = *pb;
        const uint32_t s0 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C040C00u), o[0])), s1 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C050C01u), o[1]));
        const uint32_t s2 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C060C02u), o[2])), s3 = pk_sat_u8(pk_add(byte_perm(vb, va, 0x0C070C03u), o[3]));
        const uint32_t m01 = byte_perm(s1, s0, 0x05040100u), m23 = byte_perm(s3, s2, 0x05040100u);
        *pa = byte_perm(m23, m01, 0x06040200u);
        *pb = byte_perm(m23, m01, 0x07050301u);
    }
    MI355_WAVE_SYNC();
}

/* tile (LDS) -> picture, 4 bytes per lane */
template <bool ALIGNED>
__device__ __forceinline__ uint32_t tile_dword(const uint8_t *p)
{
    if (ALIGNED || (reinterpret_cast<uintptr_t>(p) & 3) == 0) return *reinterpret_cast<const uint32_t *>(p);
    return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
}
template <bool ALIGNED = false>
__device__ inline void store_mb(const uint8_t *y, int ypitch, const uint8_t *cb, const uint8_t *cr, int cpitch,
                                const FrameHot &fr, int mb_x, int mb_y)
{
    const int lane = lane_id();
    {
        const int row = lane >> 2, seg = lane & 3;
        *reinterpret_cast<uint32_t *>(fr.recon[0] + (uint32_t)(__mul24(mb_y * 16 + row, fr.recon_stride[0]) + mb_x * 16 + 4 * seg)) = tile_dword<ALIGNED>(y + row * ypitch + 4 * seg);
    }
    if (lane < 32) {
        const int plane = lane >> 4, row = (lane >> 1) & 7, seg = lane & 1;
        *reinterpret_cast<uint32_t *>((plane ? fr.recon[2] : fr.recon[1]) + (uint32_t)(__mul24(mb_y * 8 + row, fr.recon_stride[1]) + mb_x * 8 + 4 * seg)) =
            tile_dword<ALIGNED>((plane ? cr : cb) + row * cpitch + 4 * seg);
    }
}

/* the inter kernel's tile (py: 16 rows of 16 bytes, then pc: 2 x 8 rows of 8 bytes) -> picture: a row per lane, 16 + 16 lanes */
__device__ __forceinline__ void store_mb_rows(const MbLds &s, const FrameHot &fr, int mb_x, int mb_y)
{
    static_assert(MB_PC_OFF == MB_PY_OFF + 256, "py and pc are one run of rows");
    const int lane = lane_id();
    if (lane < 16) {
        const mi355_u32x4 v = *reinterpret_cast<const mi355_u32x4 *>(s.py + 16 * lane);
        *reinterpret_cast<mi355_u32x4u *>(fr.recon[0] + (uint32_t)(__mul24(mb_y * 16 + lane, fr.recon_stride[0]) + mb_x * 16)) = mi355_u32x4u{ v[0], v[1], v[2], v[3] };
    } else if (lane < 32) {
        const int plane = (lane >> 3) & 1, row = lane & 7;
        const mi355_u32x2 v = *reinterpret_cast<const mi355_u32x2 *>(s.py + 256 + 8 * (lane - 16));
        *reinterpret_cast<mi355_u32x2u *>((plane ? fr.recon[2] : fr.recon[1]) + (uint32_t)(__mul24(mb_y * 8 + row, fr.recon_stride[1]) + mb_x * 8)) = mi355_u32x2u{ v[0], v[1] };
    }
}

/* the same tile into a macroblock-tiled surface: py and pc ARE the tile (16 x 16, then 8 x 8 Cb, 8 x 8 Cr): 24 lanes, 16 bytes each,
 * three whole cache lines */
__device__ __forceinline__ void store_mb_tiled(const MbLds &s, const FrameHot &fr, int mb_x, int mb_y)
{
    static_assert(MB_PC_OFF == MB_PY_OFF + 256, "py and pc are one run of rows");
    const int lane = lane_id();
    if (lane < 24) {
        const mi355_u32x4 v = *reinterpret_cast<const mi355_u32x4 *>(s.py + 16 * lane);
        uint8_t *d = lane < 16 ? fr.recon[0] + tile_y_off(mb_x, mb_y, fr.recon_stride[0]) + 16 * lane
                               : fr.recon[1] + tile_c_off(mb_x, mb_y, fr.recon_stride[1]) + 16 * (lane - 16);
        *reinterpret_cast<mi355_u32x4 *>(d) = v;
    }
}
/* a tile with pitches (the intra kernel's, or raw I_PCM samples) into a macroblock-tiled surface: a dword per lane, 64 + 32 lanes */
template <bool ALIGNED = false>
__device__ inline void store_mb_pitched_tiled(const uint8_t *y, int ypitch, const uint8_t *cb, const uint8_t *cr, int cpitch,
                                              const FrameHot &fr, int mb_x, int mb_y)
{
    const int lane = lane_id();
    *reinterpret_cast<uint32_t *>(fr.recon[0] + tile_y_off(mb_x, mb_y, fr.recon_stride[0]) + 4 * lane) = tile_dword<ALIGNED>(y + (lane >> 2) * ypitch + 4 * (lane & 3));
    if (lane < 32) {
        const int plane = lane >> 4, row = (lane >> 1) & 7, seg = lane & 1;
        *reinterpret_cast<uint32_t *>(fr.recon[1] + tile_c_off(mb_x, mb_y, fr.recon_stride[1]) + 4 * lane) = tile_dword<ALIGNED>((plane ? cr : cb) + row * cpitch + 4 * seg);
    }
}

/* ------------------------------------------------------------------------- */
/* Workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Give each XCD one
 * contiguous run of macroblocks so that horizontally adjacent MBs — which share reference
 * cache lines and write the same 128-byte lines of `recon` — meet in the same L2. */
__device__ __forceinline__ int xcd_linear(int b, int per_xcd) { return (b & 7) * per_xcd + (b >> 3); }

/* n / d for a launch-constant divisor d: the host passes m = ceil(2^40 / d); exact for n < 2^24 */
__device__ __forceinline__ int div_magic(int n, unsigned long long m) { return (int)(((unsigned long long)(unsigned)n * m) >> 40); }

/* One macroblock per wave.  Measured and not kept (round 2, profiles/r02_experiments.md): a wave walking a run of 4-15
 * macroblocks with the next macroblock's windows and the one after's record in flight (two register sets, exact
 * partial waits) was 25-30 % SLOWER although it hid both memory round trips — the kernel is bound by the request rate
 * of the reference fetch (removing the window loads alone: -30 % time at -8 % VALU), which a deeper pipeline does not
 * lower, and the loop cost 50 more VALU per macroblock.  Nor does a plain loop over 2 / 4 consecutive macroblocks per wave help
 * (+9 % / +24 % time: the next record's wait also waits for the previous macroblock's stores), nor a prefetch of the record
 * 1024-5000 macroblocks ahead into the L2 (+5 %). */
/* SPARSE: the coefficient array lives in device-visible HOST memory (a bridge's staging block read in place): a macroblock
 * fetches its 768 bytes only when its record says it has coefficients (cbp), at the price of a second, dependent round of
 * loads for those that do — in P / B pictures of real streams most macroblocks carry none, and the link is the narrow
 * place there.  With everything in HBM (the dense form) all five loads of a macroblock go out together. */
template <bool SPARSE, bool TILED>
__device__ __forceinline__ void recon_inter_mb(MbLds &s, const mi355_h264_frame &frd, int mb_x, int mb_y)
{
    const FrameHot fr = frame_hot(frd);
    if (mb_x >= fr.mb_width || mb_y >= fr.mb_height || (uniform(frd.flags) & MI355_FRAME_NO_INTER)) return;
    const int mb_xy = mb_y * fr.mb_width + mb_x;
    RPROF(0);
    ResidLane rl;
    resid_lane_issue(rl);            /* lane constants of residual_blocks: in flight with the record */
    if (SPARSE) load_mb(s, fr, mb_xy, false);
    else load_mb_wide(s, fr, mb_xy);
    RPROF(1);
    if (uniform((int)s.hdr.mb_type) & MI355_MB_INTRA) return;
    if (SPARSE && (uniform((int)s.hdr.cbp) & 0x3F)) {
        /* ... and of those only the coded parts: an 8x8 luma quadrant (blocks 4q..4q+3 = 32 dwords) per coded_block_pattern bit, both chroma planes
         * for its bits 4-5; a lane whose part is not coded reads a zero word from HBM instead (no predicated load) */
        const int lane = lane_id(), cbp = uniform((int)s.hdr.cbp);
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(fr.coef + (size_t)mb_xy * MI355_H264_COEFS_PER_MB);
        const uint32_t *z = k_zero16;
        const uint32_t c0 = *(((cbp >> (lane >> 5)) & 1) ? cp + lane : z), c1 = *(((cbp >> (2 + (lane >> 5))) & 1) ? cp + lane + 64 : z),
                       c2 = *((cbp & 0x30) ? cp + lane + 128 : z);
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.coef);
        dst[lane] = c0; dst[lane + 64] = c1; dst[lane + 128] = c2;
        MI355_WAVE_SYNC();
    }
    const mi355_h264_slice &sl = fr.slices[uniform(s.hdr.slice_id)];
    hl_motion<TILED>(s, fr, nullptr, sl, mb_x, mb_y, mb_xy);
    RPROF(5);
    /* inter MBs without luma coefficients (cbp & 15 == 0: skip and most of real P/B pictures) have nothing to add */
    if (uniform((int)s.hdr.mb_type) & MI355_MB_8x8DCT) {
        if (uniform((int)s.hdr.nnz_mask) & 0xFFFF) residual_luma<true>(s, s.py, 16, false);
        residual_chroma<true>(s, s.pc[0], s.pc[1], 8);
    } else {
        residual_blocks<true>(s, rl);
    }
    RPROF(6);
    if (TILED) store_mb_tiled(s, fr, mb_x, mb_y);
    else store_mb_rows(s, fr, mb_x, mb_y);
    RPROF(7);
}
/* LAYOUTS: the surface layouts the launch may meet (MI355_LAYOUTS_*).  A picture says which one it has and the general kernel carries both forms of the
 * macroblock code; a caller that knows its batch is tiled throughout launches the instance that holds the tiled form alone — 61 instead of 139 scalar
 * registers spilled at eight waves per SIMD (each spill and reload is a VALU instruction), half the code */
template <bool SPARSE, int LAYOUTS = MI355_LAYOUTS_LINEAR | MI355_LAYOUTS_TILED>
__device__ __forceinline__ void recon_inter_wave(MbLds &s, const mi355_h264_frame *__restrict__ frames, int max_w, int max_h,
                                                 unsigned long long inv_w, unsigned long long inv_h, int nblocks, int per_xcd)
{
    RPROF_START();
#ifdef MI355_HIP_EMU_H
    if (reinterpret_cast<uint8_t *>(s.py) - reinterpret_cast<uint8_t *>(&s) != MB_PY_OFF || reinterpret_cast<uint8_t *>(s.pc) - reinterpret_cast<uint8_t *>(&s) != MB_PC_OFF) abort();
#endif
    const int lin = xcd_linear((int)blockIdx.x, per_xcd);
    if (lin >= nblocks) return;
    /* lin = (f * max_h + mb_y) * max_w + mb_x */
    const int row = div_magic(lin, inv_w), mb_x = lin - row * max_w;
    const int f = div_magic(row, inv_h), mb_y = row - f * max_h;
    /* the surface layout is a property of the picture: both forms of the macroblock code live in the kernel, a wave takes one */
    if (LAYOUTS == MI355_LAYOUTS_TILED) {
        if (uniform(frames[f].surface_layout) == MI355_SURFACE_TILED) recon_inter_mb<SPARSE, true>(s, frames[f], mb_x, mb_y);
        return;                                              /* a picture of the other layout in a launch that promised none: left alone */
    }
    if (uniform(frames[f].surface_layout) == MI355_SURFACE_TILED) recon_inter_mb<SPARSE, true>(s, frames[f], mb_x, mb_y);
    else recon_inter_mb<SPARSE, false>(s, frames[f], mb_x, mb_y);
}
/* Eight waves per SIMD: left alone the compiler takes 106 scalar registers (seven waves).  Capped at 96 it spills more of them
 * to vector lanes (+45 VALU per macroblock) and the kernel is still 2.5 % faster: it waits on three dependent memory round
 * trips per macroblock (descriptor, record, reference window), which only more waves in flight hide. */
#ifndef MI355_RECON_WAVES
#define MI355_RECON_WAVES 8
#endif

}  // namespace
#endif
