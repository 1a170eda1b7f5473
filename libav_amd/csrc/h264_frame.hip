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
#include "mi355_rt.h"
#include "h264_dev.h"
#include "../../include/mi355_h264_frame.h"

using namespace mi355;

namespace {

struct MbLds {
    mi355_h264_mb hdr;
    uint32_t mv[2][16];                      /* (x | y << 16) per 4x4 block, raster order, per list */
    int16_t coef[384];
    uint8_t py[16 * 16], pc[2][8 * 8];       /* prediction -> reconstruction */
    uint8_t qy[16 * 16], qc[2][8 * 8];       /* second prediction for weighted bi-pred */
    McScratch mc;
};

__device__ __forceinline__ int blk_x4(int i) { return (i & 1) + 2 * ((i >> 2) & 1); }
__device__ __forceinline__ int blk_y4(int i) { return ((i >> 1) & 1) + 2 * (i >> 3); }
__device__ __forceinline__ int blk_index(int x4, int y4) { return (x4 & 1) + 2 * (y4 & 1) + 4 * (x4 >> 1) + 8 * (y4 >> 1); }

/* record (64 B), motion vectors (64 B per list) and coefficients (768 B) -> LDS: every load is
 * issued before the first wait, so the wave pays one memory round trip for all of them */
__device__ inline void load_mb(MbLds &s, const mi355_h264_frame &fr, int mb_xy, bool with_coefs)
{
    const int lane = lane_id();
    const uint32_t *hp = reinterpret_cast<const uint32_t *>(&fr.mb[mb_xy]);
    const uint32_t *cp = reinterpret_cast<const uint32_t *>(fr.coef + (size_t)mb_xy * MI355_H264_COEFS_PER_MB);
    uint32_t hw = 0, mw = 0, c0 = 0, c1 = 0, c2 = 0;
    if (lane < 16) hw = hp[lane];
    else if (lane < 32) { if (fr.mv[0]) mw = reinterpret_cast<const uint32_t *>(fr.mv[0])[(size_t)mb_xy * 16 + lane - 16]; }
    else if (lane < 48) { if (fr.mv[1]) mw = reinterpret_cast<const uint32_t *>(fr.mv[1])[(size_t)mb_xy * 16 + lane - 32]; }
    if (with_coefs) { c0 = cp[lane]; c1 = cp[lane + 64]; c2 = cp[lane + 128]; }
    if (lane < 16) reinterpret_cast<uint32_t *>(&s.hdr)[lane] = hw;
    else if (lane < 48) s.mv[(lane >> 4) - 1][lane & 15] = mw;
    if (with_coefs) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(s.coef);
        dst[lane] = c0; dst[lane + 64] = c1; dst[lane + 128] = c2;
    }
    __syncthreads();
}

/* one prediction direction of one partition: mc_dir_part, h264_mb.c:204-318 */
__device__ inline void mc_dir(MbLds &s, const mi355_h264_frame &fr, const mi355_h264_slice &sl, int mb_x, int mb_y,
                              int mb_xy, int list, int n_raster, int refn, int bx, int by, int w, int h,
                              uint8_t *py, uint8_t *pcb, uint8_t *pcr, int avg)
{
    (void)sl; (void)refn; (void)mb_xy;
    const uint32_t mvw = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.mv[list][n_raster]);
    const int slot = __builtin_amdgcn_readfirstlane((int)s.hdr.u.inter.ref_pic[list][(bx >> 3) + 2 * (by >> 3)]);
    const int mx = (int16_t)(mvw & 0xFFFF) + (mb_x * 16 + bx) * 4;
    const int my = (int16_t)(mvw >> 16) + (mb_y * 16 + by) * 4;
    PlaneRef ry{fr.ref[slot][0], fr.dst_stride[0], 16 * fr.mb_width, 16 * fr.mb_height};
    PlaneRef rb{fr.ref[slot][1], fr.dst_stride[1], 8 * fr.mb_width, 8 * fr.mb_height};
    PlaneRef rr{fr.ref[slot][2], fr.dst_stride[1], 8 * fr.mb_width, 8 * fr.mb_height};
    stage_windows(s.mc, &ry, mx >> 2, my >> 2, w, h, &rb, &rr, mx >> 3, my >> 3, w >> 1, h >> 1);
    mc_luma_compute(s.mc, mx & 3, my & 3, w, h, py, 16, bx, by, avg);
    mc_chroma_compute(s.mc, 0, mx & 7, my & 7, w >> 1, h >> 1, pcb, 8, bx >> 1, by >> 1, avg);
    mc_chroma_compute(s.mc, 1, mx & 7, my & 7, w >> 1, h >> 1, pcr, 8, bx >> 1, by >> 1, avg);
}

/* mc_part (h264_mc_template.c:44-62) -> mc_part_std / mc_part_weighted (h264_mb.c:320-471) */
__device__ inline void mc_part(MbLds &s, const mi355_h264_frame &fr, const mi355_h264_slice &sl, int mb_x, int mb_y,
                               int mb_xy, int n_raster, int quadrant, int bx, int by, int w, int h, int l0, int l1)
{
    const int r0 = s.hdr.ref_idx[0][quadrant], r1 = s.hdr.ref_idx[1][quadrant];
    const bool weighted = (s.hdr.flags & MI355_MBF_WEIGHTED) &&
                          ((sl.use_weight == 2 && l0 && l1 && sl.implicit_weight[r0][r1] != 32) || sl.use_weight == 1);
    if (!weighted) {
        int avg = 0;
        if (l0) { mc_dir(s, fr, sl, mb_x, mb_y, mb_xy, 0, n_raster, r0, bx, by, w, h, s.py, s.pc[0], s.pc[1], 0); avg = 1; }
        if (l1) mc_dir(s, fr, sl, mb_x, mb_y, mb_xy, 1, n_raster, r1, bx, by, w, h, s.py, s.pc[0], s.pc[1], avg);
        return;
    }
    uint8_t *dy = s.py + by * 16 + bx, *dcb = s.pc[0] + (by >> 1) * 8 + (bx >> 1), *dcr = s.pc[1] + (by >> 1) * 8 + (bx >> 1);
    if (l0 && l1) {
        mc_dir(s, fr, sl, mb_x, mb_y, mb_xy, 0, n_raster, r0, bx, by, w, h, s.py, s.pc[0], s.pc[1], 0);
        mc_dir(s, fr, sl, mb_x, mb_y, mb_xy, 1, n_raster, r1, bx, by, w, h, s.qy, s.qc[0], s.qc[1], 0);
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
        mc_dir(s, fr, sl, mb_x, mb_y, mb_xy, list, n_raster, refn, bx, by, w, h, s.py, s.pc[0], s.pc[1], 0);
        weight_block(dy, 16, w, h, sl.luma_log2_weight_denom, sl.luma_weight[refn][list][0], sl.luma_weight[refn][list][1]);
        if (sl.use_weight_chroma) {
            weight_block(dcb, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][0][0], sl.chroma_weight[refn][list][0][1]);
            weight_block(dcr, 8, w >> 1, h >> 1, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][1][0], sl.chroma_weight[refn][list][1][1]);
        }
    }
}

/* hl_motion, h264_mc_template.c:64-163 */
__device__ inline void hl_motion(MbLds &s, const mi355_h264_frame &fr, const mi355_h264_slice &sl, int mb_x, int mb_y, int mb_xy)
{
    const uint32_t t = s.hdr.mb_type;
#define DIRF(part, list) (int)((t >> (12 + (part) + 2 * (list))) & 1)
    if (t & MI355_MB_16x16) {
        mc_part(s, fr, sl, mb_x, mb_y, mb_xy, 0, 0, 0, 0, 16, 16, DIRF(0, 0), DIRF(0, 1));
    } else if (t & MI355_MB_16x8) {
        mc_part(s, fr, sl, mb_x, mb_y, mb_xy, 0, 0, 0, 0, 16, 8, DIRF(0, 0), DIRF(0, 1));
        mc_part(s, fr, sl, mb_x, mb_y, mb_xy, 8, 2, 0, 8, 16, 8, DIRF(1, 0), DIRF(1, 1));
    } else if (t & MI355_MB_8x16) {
        mc_part(s, fr, sl, mb_x, mb_y, mb_xy, 0, 0, 0, 0, 8, 16, DIRF(0, 0), DIRF(0, 1));
        mc_part(s, fr, sl, mb_x, mb_y, mb_xy, 2, 1, 8, 0, 8, 16, DIRF(1, 0), DIRF(1, 1));
    } else {
        for (int i = 0; i < 4; i++) {
            const int st = s.hdr.sub_mb_type[i], shape = st & 3;
            const int l0 = (st & MI355_SUB_L0) != 0, l1 = (st & MI355_SUB_L1) != 0;
            const int x = (i & 1) * 8, y = (i >> 1) * 8, n = (x >> 2) + 4 * (y >> 2);
            if (shape == MI355_SUB_8x8) {
                mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n, i, x, y, 8, 8, l0, l1);
            } else if (shape == MI355_SUB_8x4) {
                mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n, i, x, y, 8, 4, l0, l1);
                mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n + 4, i, x, y + 4, 8, 4, l0, l1);
            } else if (shape == MI355_SUB_4x8) {
                mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n, i, x, y, 4, 8, l0, l1);
                mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n + 1, i, x + 4, y, 4, 8, l0, l1);
            } else {
                for (int j = 0; j < 4; j++)
                    mc_part(s, fr, sl, mb_x, mb_y, mb_xy, n + (j & 1) + 4 * (j >> 1), i, x + 4 * (j & 1), y + 4 * (j >> 1), 4, 4, l0, l1);
            }
        }
    }
#undef DIRF
}

/* luma residual of a non-Intra4x4/8x8 MB onto a picture tile: hl_decode_mb_idct_luma
 * (h264_mb.c:726-795) with the dc / full / skip choice of h264idct_template.c:174-201 folded
 * into "transform the block iff it carries a coefficient" (identical results) */
__device__ inline void residual_luma(MbLds &s, uint8_t *y, int pitch, bool intra16)
{
    const int lane = lane_id();
    const uint32_t mask = s.hdr.nnz_mask;
    if (s.hdr.mb_type & MI355_MB_8x8DCT) {
        const int b = (lane >> 3) & 3, i = lane & 7;
        const bool active = lane < 32;
        int r[8];
        idct8_lds(s.coef + b * 64, i, active, r);
        if (active && ((mask >> (4 * b)) & 1))
            add_col(y + (8 * (b >> 1)) * pitch + 8 * (b & 1) + i, pitch, r, 8);
    } else {
        const int b = lane >> 2, q = lane & 3;
        int c[4], r[4], col;
#pragma unroll
        for (int i = 0; i < 4; i++) c[i] = s.coef[b * 16 + 4 * q + i];
        const int dc = s.coef[b * 16];
        idct4_quad(c, q, r, col);
        if (((mask >> b) & 1) || (intra16 && dc))
            add_col(y + (4 * blk_y4(b)) * pitch + 4 * blk_x4(b) + col, pitch, r, 4);
    }
    __syncthreads();
}

/* chroma residual: h264_mb_template.c:196-247 */
__device__ inline void residual_chroma(MbLds &s, uint8_t *cb, uint8_t *cr, int pitch)
{
    if (!(s.hdr.cbp & 0x30)) return;
    const int lane = lane_id();
    const uint32_t mask = s.hdr.nnz_mask;
    if (lane < 2 && ((mask >> (MI355_NNZ_CB_DC + lane)) & 1)) {
        int16_t *p = s.coef + 256 + 64 * lane;
        int a = p[0], b = p[16], c = p[32], d = p[48];
        chroma_dc_dequant(a, b, c, d, (int)s.hdr.dc_qmul[1 + lane]);
        p[0] = (int16_t)a; p[16] = (int16_t)b; p[32] = (int16_t)c; p[48] = (int16_t)d;
    }
    __syncthreads();
    const int j = (lane >> 2) & 7, q = lane & 3;
    int c[4], r[4], col;
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = s.coef[256 + j * 16 + 4 * q + i];
    const int dc = s.coef[256 + j * 16];
    idct4_quad(c, q, r, col);
    if (lane < 32 && (((mask >> (16 + j)) & 1) || dc)) {
        uint8_t *p = (j >> 2) ? cr : cb;
        const int jj = j & 3;
        add_col(p + (4 * (jj >> 1)) * pitch + 4 * (jj & 1) + col, pitch, r, 4);
    }
    __syncthreads();
}

/* tile (LDS) -> picture, 4 bytes per lane */
__device__ inline void store_mb(const uint8_t *y, int ypitch, const uint8_t *cb, const uint8_t *cr, int cpitch,
                                uint8_t *const dst[3], const int32_t stride[2], int mb_x, int mb_y)
{
    const int lane = lane_id();
    {
        const int row = lane >> 2, seg = lane & 3;
        const uint8_t *p = y + row * ypitch + 4 * seg;
        uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
        *reinterpret_cast<uint32_t *>(dst[0] + (size_t)(mb_y * 16 + row) * stride[0] + mb_x * 16 + 4 * seg) = v;
    }
    if (lane < 32) {
        const int plane = lane >> 4, row = (lane >> 1) & 7, seg = lane & 1;
        const uint8_t *p = (plane ? cr : cb) + row * cpitch + 4 * seg;
        uint32_t v = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
        *reinterpret_cast<uint32_t *>(dst[1 + plane] + (size_t)(mb_y * 8 + row) * stride[1] + mb_x * 8 + 4 * seg) = v;
    }
}

/* ------------------------------------------------------------------------- */
/* Workgroup b runs on XCD b % 8 (observed dispatch order; speed only).  Give each XCD one
 * contiguous run of macroblocks so that horizontally adjacent MBs — which share reference
 * cache lines and write the same 128-byte lines of `recon` — meet in the same L2. */
__device__ __forceinline__ int xcd_linear(int b, int per_xcd) { return (b & 7) * per_xcd + (b >> 3); }

__global__ void __launch_bounds__(64)
k_recon_inter(const mi355_h264_frame *__restrict__ frames, int max_nmb, int nblocks, int per_xcd)
{
    __shared__ MbLds s;
    const int lin = xcd_linear((int)blockIdx.x, per_xcd);
    if (lin >= nblocks) return;
    const int f = lin / max_nmb, mb_xy = lin - f * max_nmb;
    const mi355_h264_frame &fr = frames[f];
    if (mb_xy >= fr.mb_width * fr.mb_height) return;
    load_mb(s, fr, mb_xy, true);
    if (s.hdr.mb_type & MI355_MB_INTRA) return;
    const int mb_x = mb_xy % fr.mb_width, mb_y = mb_xy / fr.mb_width;
    const mi355_h264_slice &sl = fr.slices[s.hdr.slice_id];
    hl_motion(s, fr, sl, mb_x, mb_y, mb_xy);
    residual_luma(s, s.py, 16, false);
    residual_chroma(s, s.pc[0], s.pc[1], 8);
    store_mb(s.py, 16, s.pc[0], s.pc[1], 8, fr.recon, fr.recon_stride, mb_x, mb_y);
}

/* ------------------------------------------------------------------------- */
/* intra                                                                        */
/* ------------------------------------------------------------------------- */
constexpr int TP = 28;   /* luma tile pitch: columns -1..23 (8x8 blocks read 16 samples of the row above) */
constexpr int CP = 12;   /* chroma tile pitch: columns -1..7 */
struct IntraLds {
    MbLds mb;
    uint8_t tile[17 * TP];
    uint8_t ctile[2][9 * CP];
    PredScratch ps;
};
#define TILE(x, y) s.tile[((y) + 1) * TP + (x) + 1]

__global__ void __launch_bounds__(64)
k_recon_intra(const mi355_h264_frame *frames, int level, int width)
{
    __shared__ IntraLds s;
    const int lane = lane_id();
    const int f = blockIdx.x / width, k = blockIdx.x - f * width;
    const mi355_h264_frame &fr = frames[f];
    if (level > fr.max_intra_level) return;
    const int first = fr.intra_level_start[level - 1], count = fr.intra_level_start[level] - first;
    if (k >= count) return;
    const int mb_xy = (int)fr.intra_list[first + k];
    const int mb_x = mb_xy % fr.mb_width, mb_y = mb_xy / fr.mb_width;
    load_mb(s.mb, fr, mb_xy, true);
    const mi355_h264_mb &h = s.mb.hdr;
    const uint32_t t = h.mb_type;
    const int ys = fr.recon_stride[0], cs = fr.recon_stride[1];
    uint8_t *const ry = fr.recon[0] + (size_t)mb_y * 16 * ys + mb_x * 16;

    if (t & MI355_MB_INTRA_PCM) {        /* h264_mb_template.c:139-153 */
        const uint8_t *src = reinterpret_cast<const uint8_t *>(s.mb.coef);
        store_mb(src, 16, src + 256, src + 320, 8, fr.recon, fr.recon_stride, mb_x, mb_y);
        return;
    }
    /* edge samples of the unfiltered neighbours -> tiles */
    const int pic_w = 16 * fr.mb_width;
    if (mb_y > 0 && lane < 25) {
        int x = lane - 1;
        if (mb_x * 16 + x >= 0 && mb_x * 16 + x < pic_w) TILE(x, -1) = ry[x - ys];
    }
    if (mb_x > 0 && lane >= 32 && lane < 48) TILE(-1, lane - 32) = ry[(lane - 32) * ys - 1];
    for (int p = 0; p < 2; p++) {
        const uint8_t *rc = fr.recon[1 + p] + (size_t)mb_y * 8 * cs + mb_x * 8;
        if (mb_y > 0 && lane < 9 && (mb_x > 0 || lane > 0)) s.ctile[p][lane] = rc[lane - 1 - cs];
        if (mb_x > 0 && lane >= 16 && lane < 24) s.ctile[p][(lane - 16 + 1) * CP] = rc[(lane - 16) * cs - 1];
    }
    __syncthreads();

    /* chroma prediction: hpc.pred8x8[chroma_pred_mode], h264_mb_template.c:161-164 */
    for (int p = 0; p < 2; p++) {
        if (lane < 9) s.ps.T[lane] = s.ctile[p][lane];
        if (lane >= 16 && lane < 25) s.ps.L[lane - 16] = s.ctile[p][(lane - 16) * CP];
        __syncthreads();
        intra_pred_wave(s.ps, 2, h.chroma_pred_mode, 0, 0, &s.ctile[p][CP + 1], CP);
    }

    if (t & MI355_MB_INTRA16x16) {       /* h264_mb.c:701-722 */
        if (lane < 17) s.ps.T[lane] = TILE(lane - 1, -1);
        if (lane >= 32 && lane < 49) s.ps.L[lane - 32] = TILE(-1, lane - 33);
        __syncthreads();
        intra_pred_wave(s.ps, 3, h.intra16x16_pred_mode, 0, 0, &TILE(0, 0), TP);
        if ((h.nnz_mask >> MI355_NNZ_LUMA_DC) & 1) {
            if (lane == 0) {
                int in[16], out[16];
                for (int k2 = 0; k2 < 16; k2++) in[k2] = s.mb.coef[luma_dc_slot(k2)];
                luma_dc_dequant(in, (int)h.dc_qmul[0], out);
                for (int k2 = 0; k2 < 16; k2++) s.mb.coef[luma_dc_slot(k2)] = (int16_t)out[k2];
            }
            __syncthreads();
        }
        residual_luma(s.mb, &TILE(0, 0), TP, true);
    } else if (t & MI355_MB_8x8DCT) {    /* Intra 8x8: h264_mb.c:626-656 */
        for (int i8 = 0; i8 < 4; i8++) {
            const int x0 = 8 * (i8 & 1), y0 = 8 * (i8 >> 1), i = 4 * i8;
            if (lane < 17) s.ps.T[lane] = TILE(x0 + lane - 1, y0 - 1);
            if (lane >= 32 && lane < 41) s.ps.L[lane - 32] = TILE(x0 - 1, y0 + lane - 33);
            __syncthreads();
            intra_pred_wave(s.ps, 1, h.u.intra4x4_pred_mode[i], (h.topleft_samples_available << i) & 0x8000,
                            (h.topright_samples_available << i) & 0x4000, &TILE(x0, y0), TP);
            int r[8];
            idct8_lds(s.mb.coef + i8 * 64, lane & 7, lane < 8, r);
            if (lane < 8 && ((h.nnz_mask >> i) & 1)) add_col(&TILE(x0 + lane, y0), TP, r, 8);
            __syncthreads();
        }
    } else {                              /* Intra 4x4: h264_mb.c:657-700 */
        for (int i = 0; i < 16; i++) {
            const int x0 = 4 * blk_x4(i), y0 = 4 * blk_y4(i);
            const int tr_ok = (h.topright_samples_available << i) & 0x8000;
            if (lane < 5) s.ps.T[lane] = TILE(x0 + lane - 1, y0 - 1);
            else if (lane < 9) s.ps.T[lane] = tr_ok ? TILE(x0 + lane - 1, y0 - 1) : TILE(x0 + 3, y0 - 1);
            if (lane >= 32 && lane < 37) s.ps.L[lane - 32] = TILE(x0 - 1, y0 + lane - 33);
            __syncthreads();
            intra_pred_wave(s.ps, 0, h.u.intra4x4_pred_mode[i], 0, 0, &TILE(x0, y0), TP);
            const int q = lane & 3;
            int c[4], r[4], col;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) c[k2] = s.mb.coef[i * 16 + 4 * q + k2];
            idct4_quad(c, q, r, col);
            if (lane < 4 && ((h.nnz_mask >> i) & 1)) add_col(&TILE(x0 + col, y0), TP, r, 4);
            __syncthreads();
        }
    }
    residual_chroma(s.mb, &s.ctile[0][CP + 1], &s.ctile[1][CP + 1], CP);
    store_mb(&TILE(0, 0), TP, &s.ctile[0][CP + 1], &s.ctile[1][CP + 1], CP, fr.recon, fr.recon_stride, mb_x, mb_y);
}
#undef TILE

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

constexpr int DP = 24;   /* luma tile pitch: columns -4..15, rows -4..15 */
constexpr int DCP = 12;  /* chroma tile pitch: columns -2..7, rows -2..7 */
struct DeblockLds {
    mi355_h264_mb hdr[3];     /* this MB, left neighbour, top neighbour */
    uint8_t y[20 * DP];
    uint8_t c[2][10 * DCP];
    int8_t bs[2][4][4];
};
#define YT(x, y_) s.y[((y_) + 4) * DP + (x) + 4]
#define CT(p, x, y_) s.c[p][((y_) + 2) * DCP + (x) + 2]

__global__ void __launch_bounds__(64)
k_deblock(const mi355_h264_frame *__restrict__ frames, int diag, int max_mb_height)
{
    __shared__ DeblockLds s;
    const int lane = lane_id();
    const int f = blockIdx.x / max_mb_height, mb_y = blockIdx.x - f * max_mb_height;
    const mi355_h264_frame &fr = frames[f];
    const int mb_x = diag - 2 * mb_y;
    if (mb_y >= fr.mb_height || mb_x < 0 || mb_x >= fr.mb_width) return;
    const int mb_xy = mb_x + mb_y * fr.mb_width;
    const bool has_l = mb_x > 0, has_t = mb_y > 0;
    const int rs = fr.recon_stride[0], rcs = fr.recon_stride[1], ds = fr.dst_stride[0], dcs = fr.dst_stride[1];
    const uint8_t *src = fr.recon[0] + (size_t)mb_y * 16 * rs + mb_x * 16;
    uint8_t *dy = fr.dst[0] + (size_t)mb_y * 16 * ds + mb_x * 16;

    /* ---- phase A: every load this wave needs, issued before the first wait -------------------
     * records of this MB and of its left / top neighbours; own samples from `recon`; the neighbour
     * columns / rows (already filtered by earlier diagonals) from `dst`; the motion vectors of the
     * two 4x4 blocks each (dir, edge, segment) lane compares. */
    uint32_t hw = 0;
    {
        const int which = lane >> 4, w = lane & 15;
        const int xy = which == 0 ? mb_xy : (which == 1 ? mb_xy - 1 : mb_xy - fr.mb_width);
        if (which == 0 || (which == 1 && has_l) || (which == 2 && has_t))
            hw = reinterpret_cast<const uint32_t *>(&fr.mb[xy])[w];
    }
    const uint32_t own_y = *reinterpret_cast<const uint32_t *>(src + (lane >> 2) * rs + 4 * (lane & 3));
    uint32_t own_c = 0, nb_a = 0, nb_b = 0;
    if (lane < 32) {
        const int p = lane >> 4, crow = (lane >> 1) & 7, cseg = lane & 1;
        own_c = *reinterpret_cast<const uint32_t *>(fr.recon[1 + p] + (size_t)(mb_y * 8 + crow) * rcs + mb_x * 8 + 4 * cseg);
    }
    if (has_l) {
        if (lane < 16) nb_a = *reinterpret_cast<const uint32_t *>(dy + lane * ds - 4);
        else if (lane < 32) {
            const int p = (lane >> 3) & 1, r = lane & 7;
            nb_a = *reinterpret_cast<const uint16_t *>(fr.dst[1 + p] + (size_t)(mb_y * 8 + r) * dcs + mb_x * 8 - 2);
        }
    }
    if (has_t) {
        if (lane >= 32 && lane < 48) {
            const int r = (lane - 32) >> 2, sg = lane & 3;
            nb_b = *reinterpret_cast<const uint32_t *>(dy + (r - 4) * ds + 4 * sg);
        } else if (lane >= 48 && lane < 56) {
            const int p = (lane >> 2) & 1, r = (lane >> 1) & 1, sg = lane & 1;
            nb_b = *reinterpret_cast<const uint32_t *>(fr.dst[1 + p] + (size_t)(mb_y * 8 + r - 2) * dcs + mb_x * 8 + 4 * sg);
        }
    }
    /* bS lanes: (dir, edge, i) -> block p = (x4,y4) of this MB, block q = its neighbour across the edge */
    const int dir = (lane >> 4) & 1, edge = (lane >> 2) & 3, seg = lane & 3;
    const int px4 = dir ? seg : edge, py4 = dir ? edge : seg;
    const bool q_out = edge == 0;                      /* q lives in the neighbouring MB */
    const int qx4 = dir ? seg : (q_out ? 3 : edge - 1), qy4 = dir ? (q_out ? 3 : edge - 1) : seg;
    const int q_xy = !q_out ? mb_xy : (dir ? mb_xy - fr.mb_width : mb_xy - 1);
    const bool q_ok = !q_out || (dir ? has_t : has_l);
    BlkMotion mp, mq;
    mp.mv[0] = mp.mv[1] = mq.mv[0] = mq.mv[1] = 0;
    if (lane < 32) {
        if (fr.mv[0]) {
            mp.mv[0] = reinterpret_cast<const uint32_t *>(fr.mv[0])[(size_t)mb_xy * 16 + px4 + 4 * py4];
            if (q_ok) mq.mv[0] = reinterpret_cast<const uint32_t *>(fr.mv[0])[(size_t)q_xy * 16 + qx4 + 4 * qy4];
        }
        if (fr.mv[1]) {
            mp.mv[1] = reinterpret_cast<const uint32_t *>(fr.mv[1])[(size_t)mb_xy * 16 + px4 + 4 * py4];
            if (q_ok) mq.mv[1] = reinterpret_cast<const uint32_t *>(fr.mv[1])[(size_t)q_xy * 16 + qx4 + 4 * qy4];
        }
    }

    /* ---- phase B: registers -> LDS ------------------------------------------------------------ */
    if (lane < 48) reinterpret_cast<uint32_t *>(&s.hdr[lane >> 4])[lane & 15] = hw;
    {
        const int row = lane >> 2, sg = lane & 3;
        YT(4 * sg + 0, row) = (uint8_t)own_y; YT(4 * sg + 1, row) = (uint8_t)(own_y >> 8);
        YT(4 * sg + 2, row) = (uint8_t)(own_y >> 16); YT(4 * sg + 3, row) = (uint8_t)(own_y >> 24);
    }
    if (lane < 32) {
        const int p = lane >> 4, crow = (lane >> 1) & 7, cseg = lane & 1;
        CT(p, 4 * cseg + 0, crow) = (uint8_t)own_c; CT(p, 4 * cseg + 1, crow) = (uint8_t)(own_c >> 8);
        CT(p, 4 * cseg + 2, crow) = (uint8_t)(own_c >> 16); CT(p, 4 * cseg + 3, crow) = (uint8_t)(own_c >> 24);
    }
    if (has_l) {
        if (lane < 16) {
            YT(-4, lane) = (uint8_t)nb_a; YT(-3, lane) = (uint8_t)(nb_a >> 8); YT(-2, lane) = (uint8_t)(nb_a >> 16); YT(-1, lane) = (uint8_t)(nb_a >> 24);
        } else if (lane < 32) {
            const int p = (lane >> 3) & 1, r = lane & 7;
            CT(p, -2, r) = (uint8_t)nb_a; CT(p, -1, r) = (uint8_t)(nb_a >> 8);
        }
    }
    if (has_t) {
        if (lane >= 32 && lane < 48) {
            const int r = (lane - 32) >> 2, sg = lane & 3;
            YT(4 * sg + 0, r - 4) = (uint8_t)nb_b; YT(4 * sg + 1, r - 4) = (uint8_t)(nb_b >> 8);
            YT(4 * sg + 2, r - 4) = (uint8_t)(nb_b >> 16); YT(4 * sg + 3, r - 4) = (uint8_t)(nb_b >> 24);
        } else if (lane >= 48 && lane < 56) {
            const int p = (lane >> 2) & 1, r = (lane >> 1) & 1, sg = lane & 1;
            CT(p, 4 * sg + 0, r - 2) = (uint8_t)nb_b; CT(p, 4 * sg + 1, r - 2) = (uint8_t)(nb_b >> 8);
            CT(p, 4 * sg + 2, r - 2) = (uint8_t)(nb_b >> 16); CT(p, 4 * sg + 3, r - 2) = (uint8_t)(nb_b >> 24);
        }
    }
    __syncthreads();

    const mi355_h264_mb &h = s.hdr[0];
    const bool filter = !(h.flags & MI355_MBF_NO_DEBLOCK);
    const bool have_left = filter && (h.flags & MI355_MBF_LEFT_EDGE), have_top = filter && (h.flags & MI355_MBF_TOP_EDGE);

    /* ---- phase C: boundary strengths, filter_mb_dir (h264_loopfilter.c:472-713), one per lane -- */
    if (filter && lane < 32) {
        const bool intra = (h.mb_type & MI355_MB_INTRA) != 0;
        const int list_count = fr.mv[1] ? 2 : 1;     /* sl->list_count == 2 exactly when list-1 vectors exist */
        int bs = 0;
        if (edge == 0) {
            if (dir ? have_top : have_left) {
                const mi355_h264_mb &nb = s.hdr[1 + dir];
                if (intra || (nb.mb_type & MI355_MB_INTRA)) bs = 4;
                else if (((h.nnz_mask >> blk_index(px4, py4)) | (nb.nnz_mask >> blk_index(qx4, qy4))) & 1) bs = 2;
                else {
                    mp.ref[0] = ref_identity(h, 0, px4, py4); mp.ref[1] = ref_identity(h, 1, px4, py4);
                    mq.ref[0] = ref_identity(nb, 0, qx4, qy4); mq.ref[1] = ref_identity(nb, 1, qx4, qy4);
                    bs = check_mv(mp, mq, list_count);
                }
            }
        } else if (!((h.mb_type & MI355_MB_8x8DCT) && (edge & 1))) {
            if (intra) bs = 3;
            else if (((h.nnz_mask >> blk_index(px4, py4)) | (h.nnz_mask >> blk_index(qx4, qy4))) & 1) bs = 2;
            else {
                mp.ref[0] = ref_identity(h, 0, px4, py4); mp.ref[1] = ref_identity(h, 1, px4, py4);
                mq.ref[0] = ref_identity(h, 0, qx4, qy4); mq.ref[1] = ref_identity(h, 1, qx4, qy4);
                bs = check_mv(mp, mq, list_count);
            }
        }
        s.bs[dir][edge][seg] = (int8_t)bs;
    }
    __syncthreads();

    /* ---- phase D: the 8 luma + 4 chroma edges, in the reference's order --------------------- */
    if (filter) {
        /* lanes 0..15: luma lines; 16..23: Cb lines; 24..31: Cr lines */
        const int plane = lane < 16 ? 0 : (lane < 24 ? 1 : 2);
        const int line = plane == 0 ? lane : (lane & 7);
        for (int d2 = 0; d2 < 2; d2++) {
            for (int e = 0; e < 4; e++) {
                if (lane < 32 && !(plane && (e & 1))) {
                    const int bs = s.bs[d2][e][plane ? line >> 1 : line >> 2];
                    if (bs) {
                        const mi355_h264_mb &nb = s.hdr[1 + d2];
                        int qp;
                        if (plane == 0) qp = e ? h.qp : (h.qp + nb.qp + 1) >> 1;
                        else if (e) qp = h.qpc[plane - 1];
                        else {
                            /* the reference maps the neighbour's QP through the CURRENT slice's table
                             * (h264_loopfilter.c:628-629); identical to the neighbour's own qpc unless the
                             * two MBs sit in slices with different PPS chroma offsets */
                            const int nq = nb.slice_id == h.slice_id ? nb.qpc[plane - 1]
                                                                     : fr.slices[h.slice_id].chroma_qp_table[plane - 1][nb.qp];
                            qp = (h.qpc[plane - 1] + nq + 1) >> 1;
                        }
                        const int ia = clip3(qp + h.slice_alpha_c0_offset, 0, 51), ib = clip3(qp + h.slice_beta_offset, 0, 51);
                        const int alpha = k_alpha[ia], beta = k_beta[ib];
                        if (alpha && beta) {
                            if (plane == 0) {
                                uint8_t *c = d2 ? &YT(line, 4 * e) : &YT(4 * e, line);
                                const int xs = d2 ? DP : 1;
                                int p3 = c[-4 * xs], p2 = c[-3 * xs], p1 = c[-2 * xs], p0 = c[-xs];
                                int q0 = c[0], q1 = c[xs], q2 = c[2 * xs], q3 = c[3 * xs];
                                if (bs < 4) lf_luma_line(p2, p1, p0, q0, q1, q2, alpha, beta, k_tc0[ia][bs - 1]);
                                else lf_luma_intra_line(p3, p2, p1, p0, q0, q1, q2, q3, alpha, beta);
                                c[-3 * xs] = (uint8_t)p2; c[-2 * xs] = (uint8_t)p1; c[-xs] = (uint8_t)p0;
                                c[0] = (uint8_t)q0; c[xs] = (uint8_t)q1; c[2 * xs] = (uint8_t)q2;
                            } else {
                                uint8_t *c = d2 ? &CT(plane - 1, line, 2 * e) : &CT(plane - 1, 2 * e, line);
                                const int xs = d2 ? DCP : 1;
                                int p1 = c[-2 * xs], p0 = c[-xs], q0 = c[0], q1 = c[xs];
                                if (bs < 4) lf_chroma_line(p1, p0, q0, q1, alpha, beta, k_tc0[ia][bs - 1] + 1);
                                else lf_chroma_intra_line(p1, p0, q0, q1, alpha, beta);
                                c[-xs] = (uint8_t)p0; c[0] = (uint8_t)q0;
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    /* ---- phase E: the MB, plus the neighbour samples its edge-0 filters may have changed ------- */
    uint8_t *const dst3[3] = {fr.dst[0], fr.dst[1], fr.dst[2]};
    store_mb(&YT(0, 0), DP, &CT(0, 0, 0), &CT(1, 0, 0), DCP, dst3, fr.dst_stride, mb_x, mb_y);
    if (have_left) {
        if (lane < 16) { dy[lane * ds - 3] = YT(-3, lane); dy[lane * ds - 2] = YT(-2, lane); dy[lane * ds - 1] = YT(-1, lane); }
        else if (lane < 32) {
            const int p = (lane >> 3) & 1, r = lane & 7;
            fr.dst[1 + p][(size_t)(mb_y * 8 + r) * dcs + mb_x * 8 - 1] = CT(p, -1, r);
        }
    }
    if (have_top) {
        if (lane >= 32 && lane < 48) {
            const int x = lane - 32;
            dy[-3 * ds + x] = YT(x, -3); dy[-2 * ds + x] = YT(x, -2); dy[-ds + x] = YT(x, -1);
        } else if (lane >= 48) {
            const int p = (lane >> 3) & 1, x = lane & 7;
            fr.dst[1 + p][(size_t)(mb_y * 8 - 1) * dcs + mb_x * 8 + x] = CT(p, x, -1);
        }
    }
}
#undef YT
#undef CT

}  // namespace

/* ------------------------------------------------------------------------- */
/* host entry points                                                            */
/* ------------------------------------------------------------------------- */
extern "C" int mi355_h264_recon_inter_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream)
{
    if (!mi355::ready() || !d_frames || nframes <= 0) return -1;
    const int max_nmb = max_mb_width * max_mb_height;
    const int nblocks = nframes * max_nmb, per_xcd = (nblocks + 7) / 8;
    hipLaunchKernelGGL(k_recon_inter, dim3((unsigned)(8 * per_xcd)), dim3(64), 0, (hipStream_t)stream,
                       d_frames, max_nmb, nblocks, per_xcd);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_h264_recon_intra_dev(const mi355_h264_frame *d_frames, int nframes, int max_intra_level, int max_level_width, void *stream)
{
    if (!mi355::ready() || !d_frames || nframes <= 0) return -1;
    if (max_intra_level <= 0 || max_level_width <= 0) return 0;
    for (int level = 1; level <= max_intra_level; level++)
        hipLaunchKernelGGL(k_recon_intra, dim3((unsigned)(nframes * max_level_width)), dim3(64), 0, (hipStream_t)stream,
                           d_frames, level, max_level_width);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_h264_deblock_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height, void *stream)
{
    if (!mi355::ready() || !d_frames || nframes <= 0) return -1;
    const int ndiag = (max_mb_width - 1) + 2 * (max_mb_height - 1) + 1;
    for (int d = 0; d < ndiag; d++)
        hipLaunchKernelGGL(k_deblock, dim3((unsigned)(nframes * max_mb_height)), dim3(64), 0, (hipStream_t)stream,
                           d_frames, d, max_mb_height);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_h264_decode_frames_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                            int max_intra_level, int max_level_width, void *stream)
{
    int rc = mi355_h264_recon_inter_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
    if (rc) return rc;
    rc = mi355_h264_recon_intra_dev(d_frames, nframes, max_intra_level, max_level_width, stream);
    if (rc) return rc;
    return mi355_h264_deblock_dev(d_frames, nframes, max_mb_width, max_mb_height, stream);
}

extern "C" int mi355_h264_decode_frames(const mi355_h264_frame *frames, int nframes, void *stream)
{
    if (!mi355::ready() || !frames || nframes <= 0) return -1;
    int mw = 0, mh = 0, ml = 0;
    for (int i = 0; i < nframes; i++) {
        if (frames[i].mb_width > mw) mw = frames[i].mb_width;
        if (frames[i].mb_height > mh) mh = frames[i].mb_height;
        if (frames[i].max_intra_level > ml) ml = frames[i].max_intra_level;
    }
    /* descriptors travel through a per-call device buffer; the level width is bounded by the
     * picture's anti-diagonal only for all-intra pictures, so use the MB count as the safe bound */
    mi355_h264_frame *d = nullptr;
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&d), sizeof(*d) * (size_t)nframes));
    MI355_CHECK(hipMemcpyAsync(d, frames, sizeof(*d) * (size_t)nframes, hipMemcpyHostToDevice, (hipStream_t)stream));
    int rc = mi355_h264_decode_frames_dev(d, nframes, mw, mh, ml, mw * mh, stream);
    MI355_CHECK(hipStreamSynchronize((hipStream_t)stream));
    MI355_CHECK(hipFree(d));
    return rc;
}

extern "C" int mi355_h264_intra_schedule(mi355_h264_mb *mb, int mb_width, int mb_height,
                                         uint32_t *list, int32_t *level_start, int *max_level_width)
{
    int maxl = 0;
    const int nmb = mb_width * mb_height;
    for (int y = 0; y < mb_height; y++)
        for (int x = 0; x < mb_width; x++) {
            mi355_h264_mb &m = mb[x + y * mb_width];
            if (!(m.mb_type & MI355_MB_INTRA)) { m.intra_level = 0; continue; }
            int lv = 0;
            const int dx[4] = {-1, -1, 0, 1}, dy[4] = {0, -1, -1, -1};
            for (int k = 0; k < 4; k++) {
                int nx = x + dx[k], ny = y + dy[k];
                if (nx >= 0 && nx < mb_width && ny >= 0 && ny < mb_height && mb[nx + ny * mb_width].intra_level > lv)
                    lv = mb[nx + ny * mb_width].intra_level;
            }
            m.intra_level = (uint8_t)(lv + 1);   /* bounded by mb_width + 2*mb_height - 2 <= 254 for <= 4096x2304 */
            if (lv + 1 > maxl) maxl = lv + 1;
        }
    int n = 0, width = 0;
    level_start[0] = 0;
    for (int l = 1; l <= maxl; l++) {
        for (int i = 0; i < nmb; i++)
            if (mb[i].intra_level == l) list[n++] = (uint32_t)i;
        level_start[l] = n;
        if (level_start[l] - level_start[l - 1] > width) width = level_start[l] - level_start[l - 1];
    }
    if (max_level_width) *max_level_width = width;
    return maxl;
}

extern "C" void *mi355_malloc(size_t bytes)
{
    void *p = nullptr;
    if (!mi355::ready() || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mi355_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int mi355_memcpy_h2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
extern "C" int mi355_memcpy_d2h(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
extern "C" int mi355_memcpy_d2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice) == hipSuccess ? 0 : -1; }
extern "C" int mi355_sync(void *stream) { return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : -1; }

extern "C" void *mi355_event_create(void)
{
    hipEvent_t e;
    MI355_CHECK(hipEventCreate(&e));
    return e;
}
extern "C" void mi355_event_destroy(void *e) { (void)hipEventDestroy((hipEvent_t)e); }
extern "C" int mi355_event_record(void *e, void *stream) { return hipEventRecord((hipEvent_t)e, (hipStream_t)stream) == hipSuccess ? 0 : -1; }
/* milliseconds between two recorded events; waits for `b` */
extern "C" float mi355_event_elapsed_ms(void *a, void *b)
{
    float ms = 0.f;
    MI355_CHECK(hipEventSynchronize((hipEvent_t)b));
    MI355_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b));
    return ms;
}
