/*
 * h264_frame_wide.hip — Tier-2 for the H.264 formats outside the 8-bit 4:2:0 kernels (h264_frame.hip, h264_deblock.hip):
 * 9- and 10-bit samples (High 10) and 4:2:2 chroma (High 4:2:2), frame or field pictures without MBAFF and without
 * transform bypass.  C ABI: mi355_h264_decode_frames_wide_dev (include/mi355_h264_frame.h).
 *
 * Reference behaviour restated: hl_decode_mb with PIXEL_SHIFT / CHROMA422 (h264_mb_template.c:27-58, :174-232, :239-257),
 * mc_dir_part / mc_part_std / mc_part_weighted (h264_mb.c:204-471, the chroma_idc == 2 branches :284-315), hl_motion
 * (h264_mc_template.c:64-163), hl_decode_mb_predict_luma / _idct_luma (h264_mb.c:612-795), the BIT_DEPTH 9 / 10 templates of
 * h264idct_template.c:33-310 (chroma422_dc_dequant_idct :275-310, idct_add8_422 :216-238), h264qpel_template.c:77-300,
 * h264chroma_template.c:28-200, h264dsp_template.c:30-330 (weights, loop-filter lines, the sixteen-line chroma422 edge),
 * h264pred_template.c (pred8x16_* :502-846), ff_h264_filter_mb / filter_mb_dir / check_mv (h264_loopfilter.c:442-847).
 *
 * This is the SECOND kernel set DESIGN.md §8 announces, in its first form: one wave per macroblock, samples widened to 16 bits
 * and coefficients to 32 bits in LDS whatever the picture holds, the arithmetic per sample as the templates write it (the
 * device functions the 9 / 10-bit Tier-1 tables already run, h264_tier1_hbd.hip, pinned against the reference's own objects).
 * Three passes as in the 8-bit set: every inter macroblock in one launch; intra macroblocks level by level
 * (mi355_h264_intra_schedule); the loop filter as one launch per anti-diagonal d = x + 2y (the reference's raster order only
 * needs left, top and top-right done).  Surfaces are planes with byte strides (MI355_SURFACE_LINEAR).  No byte packing, no
 * tiled surfaces, no single-launch loop filter yet: parity first (bench point config2_high10 in bench.py).
 */
#include <cstdlib>
#include <type_traits>
#include "h264_frame_dev.h"

using namespace mi355;

namespace {

template <int BD, int CF> struct Fmt {
    typedef typename std::conditional<(BD > 8), uint16_t, uint8_t>::type PX;      /* `pixel` */
    typedef typename std::conditional<(BD > 8), int32_t, int16_t>::type COEF;     /* `dctcoef` */
    static constexpr int CH = CF == 2 ? 16 : 8;          /* chroma rows of a macroblock */
    static constexpr int NCB = CF == 2 ? 8 : 4;          /* 4x4 blocks of a chroma plane */
    static constexpr int NCOEF = 256 + 2 * 16 * NCB;     /* coefficients per macroblock: 384 / 512 */
    static constexpr int MAXV = (1 << BD) - 1;
};

/* N samples between a picture row (4-byte aligned address, N * sizeof(PX) a multiple of 4) and 16-bit samples in LDS */
/* AGENT: agent-scope accesses (another workgroup of the same launch wrote / will read these samples: the row hand-over of k_wide_deblock_rows) */
/* LA: what the caller knows about the alignment of the LDS side (bytes).  With LA >= 8 a row piece moves as ONE memory instruction of
 * N * sizeof(PX) bytes (any alignment in memory: the hardware splits what straddles) and one or two LDS instructions — the loop filter's
 * tiles: a lane's piece is its own cache line, so every further instruction for the same piece is another pass of that line through the L1 */
template <typename PX, int N, bool AGENT = false, int LA = 2>
__device__ __forceinline__ void wide_ld_row(const uint8_t *p, uint16_t *d)
{
    if (!AGENT && LA >= 8) {
        PX v[N];
        uint16_t t[N];
        __builtin_memcpy(v, p, sizeof(v));
#pragma unroll
        for (int k = 0; k < N; k++) t[k] = v[k];
        __builtin_memcpy(__builtin_assume_aligned(d, LA), t, sizeof(t));
        return;
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    if (sizeof(PX) == 2) {
#pragma unroll
        for (int k = 0; k < N / 2; k++) { const uint32_t v = AGENT ? agent_load_u32(w + k) : w[k]; d[2 * k] = (uint16_t)(v & 0xFFFF); d[2 * k + 1] = (uint16_t)(v >> 16); }
    } else {
#pragma unroll
        for (int k = 0; k < N / 4; k++) {
            const uint32_t v = AGENT ? agent_load_u32(w + k) : w[k];
            d[4 * k] = (uint16_t)(v & 0xFF); d[4 * k + 1] = (uint16_t)((v >> 8) & 0xFF); d[4 * k + 2] = (uint16_t)((v >> 16) & 0xFF); d[4 * k + 3] = (uint16_t)(v >> 24);
        }
    }
}
template <typename PX, int N, bool AGENT = false, int LA = 2>
__device__ __forceinline__ void wide_st_row(uint8_t *p, const uint16_t *d)
{
    if (!AGENT && LA >= 8) {
        PX v[N];
        uint16_t t[N];
        __builtin_memcpy(t, __builtin_assume_aligned(d, LA), sizeof(t));
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = (PX)t[k];
        __builtin_memcpy(p, v, sizeof(v));
        return;
    }
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    if (sizeof(PX) == 2) {
#pragma unroll
        for (int k = 0; k < N / 2; k++) { const uint32_t v = (uint32_t)d[2 * k] | ((uint32_t)d[2 * k + 1] << 16); if (AGENT) agent_store_u32(w + k, v); else w[k] = v; }
    } else {
#pragma unroll
        for (int k = 0; k < N / 4; k++) {
            const uint32_t v = (uint32_t)d[4 * k] | ((uint32_t)d[4 * k + 1] << 8) | ((uint32_t)d[4 * k + 2] << 16) | ((uint32_t)d[4 * k + 3] << 24);
            if (AGENT) agent_store_u32(w + k, v); else w[k] = v;
        }
    }
}

/* position of chroma block j of a plane in 4-sample units: the reference's block_offset[16 + j] (4:2:0, j = 0..3) and
 * block_offset[16 + j] / [16 + j + 4] (4:2:2, h264idct_template.c:222-236) */
__device__ __forceinline__ int cblk_x4(int j) { return j & 1; }
__device__ __forceinline__ int cblk_y4(int j) { return ((j >> 1) & 1) + 2 * (j >> 2); }

/* ---- inverse transforms on 32-bit LDS copies of the coefficients; stores between the two passes have the width of dctcoef ------ */
template <typename COEF>
__device__ inline void wide_idct4(const int32_t *c, int r[16])       /* h264idct_template.c:33-66; r[4 * row + column] */
{
    int t[16];
    for (int i = 0; i < 4; i++) {
        const int c0 = (COEF)(c[i] + (i == 0 ? 32 : 0));
        const int z0 = c0 + c[i + 8], z1 = c0 - c[i + 8], z2 = (c[i + 4] >> 1) - c[i + 12], z3 = c[i + 4] + (c[i + 12] >> 1);
        t[i] = (COEF)(z0 + z3); t[i + 4] = (COEF)(z1 + z2); t[i + 8] = (COEF)(z1 - z2); t[i + 12] = (COEF)(z0 - z3);
    }
    for (int i = 0; i < 4; i++) {
        const int z0 = t[4 * i] + t[4 * i + 2], z1 = t[4 * i] - t[4 * i + 2], z2 = (t[4 * i + 1] >> 1) - t[4 * i + 3], z3 = t[4 * i + 1] + (t[4 * i + 3] >> 1);
        r[i] = (z0 + z3) >> 6; r[4 + i] = (z1 + z2) >> 6; r[8 + i] = (z1 - z2) >> 6; r[12 + i] = (z0 - z3) >> 6;
    }
}
/* The residual of one 4x4 block as the reference's dispatchers apply it (idct_add16 / idct_add16intra / idct_add8, h264idct_template.c:174-238):
 * the full transform when the block holds an AC coefficient, the DC-only form (dc + 32) >> 6 when it holds its DC alone — the same
 * sums except that the DC-only form does not pass through a dctcoef store, which wraps at 16 bits — nothing otherwise.  false: nothing to add */
template <typename COEF>
__device__ inline bool wide_block4(const int32_t *c, int r[16])
{
    int ac = 0;
    for (int k = 1; k < 16; k++) ac |= c[k];
    if (ac) { wide_idct4<COEF>(c, r); return true; }
    if (!c[0]) return false;
    const int dc = (c[0] + 32) >> 6;
    for (int k = 0; k < 16; k++) r[k] = dc;
    return true;
}
/* The same transform on the four lanes of a quad (the arrangement of idct4_quad, h264_dev.h, with the width of dctcoef a parameter): lane j
 * holds storage column j of the block, c[k] = block[j + 4 k]; the first loop of h264idct_template.c:33-66 is per lane, the second runs across
 * the quad with two exchanges.  On return lane j holds the residuals (>> 6) of destination row {0, 3, 1, 2}[j], columns 0..3. */
template <typename COEF>
__device__ __forceinline__ void wide_idct4_quad(const int c[4], int j, int r[4], int &row)
{
    const int c0 = j == 0 ? (COEF)(c[0] + 32) : c[0];
    const int e0 = c0 + c[2], e1 = c0 - c[2], e2 = (c[1] >> 1) - c[3], e3 = c[1] + (c[3] >> 1);
    const int b[4] = { (COEF)(e0 + e3), (COEF)(e1 + e2), (COEF)(e1 - e2), (COEF)(e0 - e3) };
    const int sh = j & 1;
    const bool n1 = j >= 2, n2 = (j & 1) != 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int v = b[k];
        const int s = (n1 ? -v : v) + (quad_xor2(v) >> sh);        /* z0, z3, z1, z2 on lanes 0..3 */
        r[k] = ((n2 ? -s : s) + quad_xor1(s)) >> 6;
    }
    row = j == 0 ? 0 : (j == 1 ? 3 : (j == 2 ? 1 : 2));
}
/* up to sixteen 4x4 blocks at once, four lanes each (lane 4 * b + j: column j of block b's coefficients, then one row of its samples) with the
 * dispatchers' choice per block (wide_block4 above): an AC coefficient anywhere in the block -> the transform; the DC alone -> (dc + 32) >> 6;
 * nothing -> nothing.  dst_of(b): the block's first sample.  Every lane of the wave calls. */
template <int BD, int CF, typename Dst>
__device__ __forceinline__ void wide_add_blocks4(const int32_t *coef, int nblocks, Dst dst_of, int pitch)
{
    typedef Fmt<BD, CF> F;
    const int lane = lane_id(), b = lane >> 2, j = lane & 3;
    const bool on = b < nblocks;
    const int32_t *c = coef + 16 * (on ? b : 0);
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = c[j + 4 * k];
    int ac = v[1] | v[2] | v[3] | (j ? v[0] : 0);
    ac |= quad_xor1(ac);
    ac |= quad_xor2(ac);
    const int dc = quad_bcast<0>(v[0]);
    int r[4], row;
    wide_idct4_quad<typename F::COEF>(v, j, r, row);
    if (!ac) r[0] = r[1] = r[2] = r[3] = (dc + 32) >> 6;
    if (on && (ac | dc)) {
        uint16_t *d = dst_of(b) + row * pitch;
#pragma unroll
        for (int k = 0; k < 4; k++) d[k] = (uint16_t)med3i(d[k] + r[k], 0, F::MAXV);
    }
    MI355_WAVE_SYNC();
}
/* ... of the luma plane (blocks in the reference's order) / of a chroma plane */
template <int BD, int CF>
__device__ __forceinline__ void wide_add_blocks4(const int32_t *coef, int nblocks, bool chroma, uint16_t *dst, int pitch)
{
    wide_add_blocks4<BD, CF>(coef, nblocks, [&](int b) { return dst + 4 * (chroma ? cblk_y4(b) : blk_y4(b)) * pitch + 4 * (chroma ? cblk_x4(b) : blk_x4(b)); }, pitch);
}
/* the four 8x8 blocks of the luma plane (h264idct_template.c:69-141, dispatch :189-201): lane 8 * b + i transforms column i of block b */
template <int BD, int CF>
__device__ inline void wide_add_blocks8(const int32_t *coef, int32_t (*t8)[64], int first, int nblocks, uint16_t *dst, int pitch)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::COEF COEF;
    const int lane = lane_id(), b = first + (lane >> 3), i = lane & 7;
    const bool on = (lane >> 3) < nblocks;
    int in[8], out[8];
    /* does the block hold an AC coefficient?  every lane looks at its column, the eight answers meet in LDS */
    if (on) {
        int ac = 0;
        for (int k = 0; k < 8; k++) { in[k] = coef[64 * b + i + 8 * k]; if (i | k) ac |= in[k]; }
        t8[b & 3][i] = ac;
    }
    MI355_WAVE_SYNC();
    int ac = 0;
    if (on) for (int k = 0; k < 8; k++) ac |= t8[b & 3][k];
    MI355_WAVE_SYNC();
    const int dc0 = on ? coef[64 * b] : 0;
    if (on && ac) {
        if (i == 0) in[0] = (COEF)(in[0] + 32);
        idct8_1d(in, out);
        for (int k = 0; k < 8; k++) t8[b & 3][i + 8 * k] = (COEF)out[k];
    }
    MI355_WAVE_SYNC();
    if (on && (ac || dc0)) {
        if (ac) {
            for (int k = 0; k < 8; k++) in[k] = t8[b & 3][k + 8 * i];
            idct8_1d(in, out);
        } else
            for (int k = 0; k < 8; k++) out[k] = dc0 + 32;          /* idct8_dc_add: (dc + 32) >> 6 on every sample */
        uint16_t *d = dst + 8 * (b >> 1) * pitch + 8 * (b & 1) + i;
        for (int k = 0; k < 8; k++) d[k * pitch] = (uint16_t)clip3(d[k * pitch] + (out[k] >> 6), 0, F::MAXV);
    }
    MI355_WAVE_SYNC();
}

/* DC transforms (h264idct_template.c:242-324), one lane each, in place in the DC slots of the LDS coefficient copy */
template <typename COEF>
__device__ inline void wide_luma_dc(int32_t *coef, int qmul)
{
    int v[16], t[16];
    for (int k = 0; k < 16; k++) v[k] = coef[luma_dc_slot(k)];
    for (int i = 0; i < 4; i++) {
        const int s = v[4 * i] + v[4 * i + 1], d = v[4 * i] - v[4 * i + 1], e = v[4 * i + 2] - v[4 * i + 3], u = v[4 * i + 2] + v[4 * i + 3];
        t[4 * i] = s + u; t[4 * i + 1] = s - u; t[4 * i + 2] = d - e; t[4 * i + 3] = d + e;
    }
    for (int i = 0; i < 4; i++) {
        const int s = t[i] + t[8 + i], d = t[i] - t[8 + i], e = t[4 + i] - t[12 + i], u = t[4 + i] + t[12 + i];
        coef[luma_dc_slot(4 * i + 0)] = (COEF)(((s + u) * qmul + 128) >> 8);
        coef[luma_dc_slot(4 * i + 1)] = (COEF)(((d + e) * qmul + 128) >> 8);
        coef[luma_dc_slot(4 * i + 2)] = (COEF)(((d - e) * qmul + 128) >> 8);
        coef[luma_dc_slot(4 * i + 3)] = (COEF)(((s - u) * qmul + 128) >> 8);
    }
}
template <typename COEF, int CF>
__device__ inline void wide_chroma_dc(int32_t *c, int qmul)       /* c: the plane's first coefficient */
{
    if (CF == 2) {                                                 /* chroma422_dc_dequant_idct :275-310 */
        int v[8], t[8];
        for (int i = 0; i < 4; i++) { v[2 * i] = c[32 * i]; v[2 * i + 1] = c[32 * i + 16]; }
        for (int i = 0; i < 4; i++) { t[2 * i] = v[2 * i] + v[2 * i + 1]; t[2 * i + 1] = v[2 * i] - v[2 * i + 1]; }
        for (int i = 0; i < 2; i++) {
            const int z0 = t[i] + t[4 + i], z1 = t[i] - t[4 + i], z2 = t[2 + i] - t[6 + i], z3 = t[2 + i] + t[6 + i];
            v[0 + i] = (COEF)(((z0 + z3) * qmul + 128) >> 8);
            v[2 + i] = (COEF)(((z1 + z2) * qmul + 128) >> 8);
            v[4 + i] = (COEF)(((z1 - z2) * qmul + 128) >> 8);
            v[6 + i] = (COEF)(((z0 - z3) * qmul + 128) >> 8);
        }
        for (int i = 0; i < 4; i++) { c[32 * i] = v[2 * i]; c[32 * i + 16] = v[2 * i + 1]; }
    } else {                                                       /* chroma_dc_dequant_idct :312-324 */
        const int a = c[0], b = c[16], cc = c[32], d = c[48];
        const int s0 = a + b, d0 = a - b, s1 = cc + d, d1 = cc - d;
        c[0] = (COEF)(((s0 + s1) * qmul) >> 7); c[16] = (COEF)(((d0 + d1) * qmul) >> 7);
        c[32] = (COEF)(((s0 - s1) * qmul) >> 7); c[48] = (COEF)(((d0 - d1) * qmul) >> 7);
    }
}

/* ---- transform bypass (sl->qscale == 0 in a sequence with qpprime_y_zero_transform_bypass_flag) ----------------------------------------------
 * the coefficients are the residual: added sample by sample, no transform, no clipping — the sum wraps like the sample type (`pixel`),
 * h264addpx_template.c:30-72.  Blocks without coefficients hold zeros, so every block is added. */
template <int BD, int CF>
__device__ __forceinline__ uint16_t wide_wrap(int v) { return (uint16_t)(typename Fmt<BD, CF>::PX)v; }
/* nblocks 4x4 blocks (lane 4 * b + row) or 8x8 blocks (lane 8 * b + row) of one plane */
template <int BD, int CF>
__device__ inline void wide_add_raw(const int32_t *coef, int nblocks, int size, bool chroma, uint16_t *dst, int pitch)
{
    const int lane = lane_id();
    if (size == 4) {
        const int b = lane >> 2, q = lane & 3;
        if (b < nblocks) {
            const int x4 = chroma ? cblk_x4(b) : blk_x4(b), y4 = chroma ? cblk_y4(b) : blk_y4(b);
            uint16_t *d = dst + (4 * y4 + q) * pitch + 4 * x4;
            for (int k = 0; k < 4; k++) d[k] = wide_wrap<BD, CF>(d[k] + coef[16 * b + 4 * q + k]);
        }
    } else {
        const int b = lane >> 3, q = lane & 7;
        if (b < nblocks) {
            uint16_t *d = dst + (8 * (b >> 1) + q) * pitch + 8 * (b & 1);
            for (int k = 0; k < 8; k++) d[k] = wide_wrap<BD, CF>(d[k] + coef[64 * b + 8 * q + k]);
        }
    }
    MI355_WAVE_SYNC();
}
/* the running sums of the lossless vertical / horizontal predictors over an n_cols x n_rows region whose 4x4 blocks lie in `coef` as the
 * plane's blocks (pred4x4_*_add chained by pred16x16_*_add / pred8x8_*_add / pred8x16_*_add, h264pred_template.c:1209-1354: a block starts
 * from the sample above / left of it, which the block before it has just written — one sum down each column / along each row).
 * start(k): the sample above column k / left of row k.  Lanes 0..n-1. */
template <int BD, int CF, typename Start>
__device__ inline void wide_lossless_run(const int32_t *coef, bool chroma, bool vertical, int n_cols, int n_rows, uint16_t *dst, int pitch, Start start)
{
    const int lane = lane_id();
    if (lane < (vertical ? n_cols : n_rows)) {
        int v = start(lane);
        for (int k = 0; k < (vertical ? n_rows : n_cols); k++) {
            const int x = vertical ? lane : k, y = vertical ? k : lane, x4 = x >> 2, y4 = y >> 2;
            const int b = chroma ? x4 + 2 * (y4 & 1) + 4 * (y4 >> 1) : blk_index(x4, y4);
            v = wide_wrap<BD, CF>(v + coef[16 * b + 4 * (y & 3) + (x & 3)]);
            dst[y * pitch + x] = (uint16_t)v;
        }
    }
    MI355_WAVE_SYNC();
}

/* The chroma residual of a macroblock (h264_mb_template.c:225-257): DC transforms where the record says DC levels were coded, then
 * every block through wide_block4 — idct_add8's choice between idct_add, idct_dc_add and nothing (h264idct_template.c:203-238) read off the
 * coefficients themselves (a chroma block's count is the count of its AC coefficients).  cb / cr: the planes' 8 x CH tiles. */
template <int BD, int CF>
__device__ inline void wide_residual_chroma(int32_t *coef, const mi355_h264_mb &h, uint16_t *cb, uint16_t *cr, int pitch)
{
    typedef Fmt<BD, CF> F;
    if (!(h.cbp & 0x30)) return;
    const int lane = lane_id();
    if (h.flags & MI355_MBF_BYPASS) {            /* h264_mb_template.c:199-224 */
        for (int p = 0; p < 2; p++) {
            const int32_t *c = coef + 256 + 16 * F::NCB * p;
            uint16_t *d = p ? cr : cb;
            if ((h.mb_type & MI355_MB_INTRA) && (h.flags & MI355_MBF_BYPASS_PRED) && (h.chroma_pred_mode == 1 || h.chroma_pred_mode == 2)) {
                const bool vertical = h.chroma_pred_mode == 2;           /* VERT_PRED8x8 = 2, HOR_PRED8x8 = 1 */
                wide_lossless_run<BD, CF>(c, true, vertical, 8, F::CH, d, pitch, [&](int k) { return (int)(vertical ? d[-pitch + k] : d[k * pitch - 1]); });
            } else
                wide_add_raw<BD, CF>(c, F::NCB, 4, true, d, pitch);
        }
        return;
    }
    if (lane < 2 && ((h.nnz_mask >> (MI355_NNZ_CB_DC + lane)) & 1))
        wide_chroma_dc<typename F::COEF, CF>(coef + 256 + 16 * F::NCB * lane, (int)h.dc_qmul[1 + lane]);
    MI355_WAVE_SYNC();
    /* both planes in one pass: blocks 0..NCB-1 Cb, NCB..2 NCB-1 Cr (their coefficients follow each other) */
    wide_add_blocks4<BD, CF>(coef + 256, 2 * F::NCB, [&](int b) {
        const int bb = b >= F::NCB ? b - F::NCB : b;
        return (b >= F::NCB ? cr : cb) + 4 * cblk_y4(bb) * pitch + 4 * cblk_x4(bb);
    }, pitch);
}

/* Where a macroblock's rows are.  A frame or field PICTURE: rows 16 * mb_y + r of its planes.  An MBAFF frame (MI355_FRAME_MBAFF): macroblock rows
 * 2k, 2k + 1 are pair k; a FRAME macroblock is as above, a FIELD macroblock (MB_TYPE_INTERLACED) owns every other line of its pair — the top
 * macroblock the even ones — and lives in field coordinates for motion compensation: field row 16 * k, the reference's stride doubled, its height
 * halved (h264_mb_template.c:61-76, h264_mb.c:59-101, :212-224). */
struct WideGeom { int y0, cy0, ystep, mcy, hs; };
template <int BD, int CF>
__device__ __forceinline__ WideGeom wide_geom(const mi355_h264_frame &fr, uint32_t mb_type, int mb_y)
{
    typedef Fmt<BD, CF> F;
    const bool field = (fr.flags & MI355_FRAME_MBAFF) && (mb_type & 0x80u);
    WideGeom g;
    g.ystep = field ? 2 : 1;
    g.y0 = field ? 32 * (mb_y >> 1) + (mb_y & 1) : 16 * mb_y;
    g.cy0 = field ? 2 * F::CH * (mb_y >> 1) + (mb_y & 1) : F::CH * mb_y;
    g.mcy = field ? mb_y >> 1 : mb_y;
    g.hs = field ? 1 : 0;
    return g;
}

/* the macroblock's coefficients -> the 32-bit LDS copy.  parts: bits 0..3 the luma 8x8 quadrants (blocks 4q..4q+3 = coefficients 64q..64q+63: what
 * coded_block_pattern bit q says), bit 4 both chroma planes; a part that is not coded is not fetched (it holds zeros) — with the bridge the array is
 * pinned host memory, and the coefficients of a 10-bit macroblock are 1.5 KB across PCIe */
template <int BD, int CF>
__device__ inline void wide_load_coefs(int32_t *dst, const mi355_h264_frame &fr, int mb_xy, int parts)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::COEF COEF;
    const COEF *cp = reinterpret_cast<const COEF *>(fr.coef) + (size_t)mb_xy * F::NCOEF;
    /* sixteen bytes of the array per lane and step (four 32-bit or eight 16-bit coefficients: one load, any alignment), 16-byte LDS stores */
    constexpr int PER = 16 / (int)sizeof(COEF);
    for (int i = lane_id(); i < F::NCOEF / PER; i += 64) {
        const int first = PER * i;
        COEF v[PER];
        int32_t w[PER];
        if ((parts >> (first < 256 ? first >> 6 : 4)) & 1) __builtin_memcpy(v, cp + first, sizeof(v));
        else for (int k = 0; k < PER; k++) v[k] = 0;
#pragma unroll
        for (int k = 0; k < PER; k++) w[k] = v[k];
        __builtin_memcpy(__builtin_assume_aligned(dst + first, 16), w, sizeof(w));
    }
    MI355_WAVE_SYNC();
}

/* The same in two halves, for a caller whose LDS copy shares its place with something it uses first (k_wide_inter: the motion compensation's windows):
 * the loads leave at once, into registers (up to two 16-byte pieces per lane), and become the LDS copy later. */
template <int BD, int CF> struct WideCoefRegs {
    typedef typename Fmt<BD, CF>::COEF COEF;
    static constexpr int PER = 16 / (int)sizeof(COEF), PIECES = Fmt<BD, CF>::NCOEF / PER;
    COEF v[2][PER];
};
template <int BD, int CF>
__device__ __forceinline__ void wide_fetch_coefs(WideCoefRegs<BD, CF> &r, const mi355_h264_frame &fr, int mb_xy, int parts)
{
    typedef WideCoefRegs<BD, CF> R;
    const typename R::COEF *cp = reinterpret_cast<const typename R::COEF *>(fr.coef) + (size_t)mb_xy * Fmt<BD, CF>::NCOEF;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = lane_id() + 64 * k, first = R::PER * i;
        if (i < R::PIECES && ((parts >> (first < 256 ? first >> 6 : 4)) & 1)) __builtin_memcpy(r.v[k], cp + first, sizeof(r.v[k]));
        else for (int j = 0; j < R::PER; j++) r.v[k][j] = 0;
    }
}
template <int BD, int CF>
__device__ __forceinline__ void wide_commit_coefs(int32_t *dst, const WideCoefRegs<BD, CF> &r)
{
    typedef WideCoefRegs<BD, CF> R;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = lane_id() + 64 * k;
        int32_t w[R::PER];
#pragma unroll
        for (int j = 0; j < R::PER; j++) w[j] = r.v[k][j];
        if (i < R::PIECES) __builtin_memcpy(__builtin_assume_aligned(dst + R::PER * i, 16), w, sizeof(w));
    }
    MI355_WAVE_SYNC();
}

template <int BD, int CF>
__device__ inline void wide_store_mb(const mi355_h264_frame &fr, int mb_x, const WideGeom &g, const uint16_t *y, int ypitch, const uint16_t *cb, const uint16_t *cr, int cpitch)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    constexpr int PXB = (int)sizeof(PX);
    const int lane = lane_id();
    {   /* luma: lane = 4 * row + quarter */
        const int r = lane >> 2, c = 4 * (lane & 3);
        wide_st_row<PX, 4>(fr.recon[0] + (size_t)(g.y0 + g.ystep * r) * fr.recon_stride[0] + (16 * mb_x + c) * PXB, y + r * ypitch + c);
    }
    if (lane < 4 * F::CH) {   /* chroma: lane = (2 * row + half) of Cb, then of Cr */
        const int p = lane >= 2 * F::CH, k = lane - 2 * F::CH * p, r = k >> 1, c = 4 * (k & 1);
        wide_st_row<PX, 4>(fr.recon[1 + p] + (size_t)(g.cy0 + g.ystep * r) * fr.recon_stride[1] + (8 * mb_x + c) * PXB, (p ? cr : cb) + r * cpitch + c);
    }
}

/* ------------------------------------------------------------------------- */
/* inter                                                                        */
/* ------------------------------------------------------------------------- */
constexpr int WP = 24;          /* pitch of the luma window: 16 + 5 columns, fetched as up to three pieces of eight */
constexpr int CWP = 16, CWIN = 17 * CWP;   /* chroma windows: 9 x 17 per plane (two pieces of eight per row), the second plane at CWIN */
/* 4.3 KB: the kernel holds 32 registers, so its waves per SIMD are what its LDS leaves — eight (the limit) instead of five with the coefficients and the
 * windows side by side (7.1 KB).  The coefficients wait in registers while the prediction is made (wide_fetch_coefs / wide_commit_coefs). */
struct WideInterLds {
    mi355_h264_mb hdr;
    uint32_t mv[2][16];
    uint16_t py[256], pc[2][128];
    union {
        struct {                            /* motion compensation */
            uint16_t qy[256], qc[2][128];   /* the second prediction of a weighted two-reference partition */
            alignas(16) uint16_t win[2 * CWIN + 8];     /* >= 21 * WP */
            int16_t tmp[21 * 16];           /* the horizontal pass of the 2-D quarter positions: rows -2..h+2 of the block */
        };
        struct {                            /* the residual, once the prediction stands in py / pc */
            alignas(16) int32_t coef[512];
            int32_t t8[4][64];
        };
    };
};

/* one luma sample at quarter position (mx, my): h264qpel_template.c:77-300 as the standard writes it; the first pass of the 2-D
 * positions is kept in 16 bits around the reference's bias (:119-146), so that samples outside the bit depth's range wrap as they do there */
__device__ inline int wide_qpel_px(const uint16_t *win, const int16_t *tmp, int x, int y, int mx, int my, int maxv)
{
#define S(xx, yy) ((int)win[((yy) + 2) * WP + (xx) + 2])
    auto rawh = [&](int xx, int yy) { return tap6(S(xx - 2, yy), S(xx - 1, yy), S(xx, yy), S(xx + 1, yy), S(xx + 2, yy), S(xx + 3, yy)); };
    auto hh = [&](int xx, int yy) { return med3i((rawh(xx, yy) + 16) >> 5, 0, maxv); };
    auto vv = [&](int xx, int yy) { return med3i((tap6(S(xx, yy - 2), S(xx, yy - 1), S(xx, yy), S(xx, yy + 1), S(xx, yy + 2), S(xx, yy + 3)) + 16) >> 5, 0, maxv); };
    const int pad = maxv > 511 ? -10 * maxv : 0;
    auto tmph = [&](int xx, int yy) { return (int)tmp[(yy + 2) * 16 + xx] - pad; };      /* (int16_t)(rawh + pad), stored by the caller */
    auto hv = [&](int xx, int yy) {
        return med3i((tap6(tmph(xx, yy - 2), tmph(xx, yy - 1), tmph(xx, yy), tmph(xx, yy + 1), tmph(xx, yy + 2), tmph(xx, yy + 3)) + 512) >> 10, 0, maxv);
    };
    int v;
    if (my == 0) v = mx == 0 ? S(x, y) : (mx == 2 ? hh(x, y) : f2(S(x + (mx == 3), y), hh(x, y)));
    else if (mx == 0) v = my == 2 ? vv(x, y) : f2(S(x, y + (my == 3)), vv(x, y));
    else if (mx == 2 && my == 2) v = hv(x, y);
    else if (mx == 2) v = f2(hh(x, y + (my == 3)), hv(x, y));
    else if (my == 2) v = f2(vv(x + (mx == 3), y), hv(x, y));
    else v = f2(hh(x, y + (my == 3)), vv(x + (mx == 3), y));
#undef S
    return v;
}

/* one prediction direction of one partition: mc_dir_part, h264_mb.c:204-318.  Windows are fetched with clamped coordinates, which is
 * what emulated_edge_mc produces (videodsp_template.c:24-96).  dy / dcb / dcr: the macroblock's 16-pitch luma and 8-pitch chroma tiles */
template <int BD, int CF>
__device__ inline void wide_mc_dir(WideInterLds &s, const mi355_h264_frame &fr, int mb_x, const WideGeom &g, int list, int n_raster, int quadrant,
                                   int bx, int by, int w, int h, uint16_t *dy, uint16_t *dcb, uint16_t *dcr, int avg)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    const int lane = lane_id();
    const uint32_t mvw = (uint32_t)uniform((int)s.mv[list][n_raster]);
    const int slot = uniform((int)s.hdr.u.inter.ref_pic[list][quadrant]);
    const int mx = (int16_t)(mvw & 0xFFFF) + (mb_x * 16 + bx) * 4;
    const int my = (int16_t)(mvw >> 16) + (g.mcy * 16 + by) * 4;
    const uint8_t *const *rp = fr.ref[slot < MI355_H264_MAX_SLOTS ? slot : 0];
    const int W = 16 * fr.mb_width, H = (16 * fr.mb_height) >> g.hs;
    const size_t rys = (size_t)fr.dst_stride[0] << g.hs, rcs = (size_t)fr.dst_stride[1] << g.hs;      /* a field macroblock of an MBAFF frame predicts from fields */
    const int lw = w == 16 ? 4 : (w == 8 ? 3 : 2);            /* log2 of the block's width */
#ifndef MI355_WIDE_EXP_NOLUMA
    {
        const int x0 = (mx >> 2) - 2, y0 = (my >> 2) - 2, ww = w + 5, hh = h + 5, nc = (ww + 7) >> 3;
        if (x0 >= 0 && y0 >= 0 && x0 + 8 * nc <= W && y0 + hh <= H) {
            /* the window lies inside the picture: one piece of eight samples per lane (any alignment), <= 21 rows x 3 pieces */
            const int r = nc == 3 ? lane / 3 : (nc == 2 ? lane >> 1 : lane), c = lane - r * nc;
            if (r < hh) {
                PX v[8];
                __builtin_memcpy(v, rp[0] + (size_t)(y0 + r) * rys + (size_t)(x0 + 8 * c) * sizeof(PX), sizeof(v));
#pragma unroll
                for (int k = 0; k < 8; k++) s.win[r * WP + 8 * c + k] = v[k];
            }
        } else
        for (int i = lane; i < ww * hh; i += 64) {
            const int r = i / ww, c = i - r * ww;
            const int xx = clip3(x0 + c, 0, W - 1), yy = clip3(y0 + r, 0, H - 1);
            s.win[r * WP + c] = reinterpret_cast<const PX *>(rp[0] + (size_t)yy * rys)[xx];
        }
        MI355_WAVE_SYNC();
        if (((mx & 3) == 2 && (my & 3)) || ((my & 3) == 2 && (mx & 3))) {
            /* the 2-D positions: the horizontal 6-tap sums of rows -2..h+2 once, in 16 bits around the reference's bias (h264qpel_template.c:119-146) */
            const int pad = F::MAXV > 511 ? -10 * F::MAXV : 0;
            for (int i = lane; i < w * (h + 5); i += 64) {
                const int r = i >> lw, x = i & (w - 1);
                const uint16_t *q = s.win + r * WP + x;
                s.tmp[r * 16 + x] = (int16_t)(tap6(q[0], q[1], q[2], q[3], q[4], q[5]) + pad);
            }
            MI355_WAVE_SYNC();
        }
        for (int i = lane; i < w * h; i += 64) {
            const int y = i >> lw, x = i & (w - 1);
            const int v = wide_qpel_px(s.win, s.tmp, x, y, mx & 3, my & 3, F::MAXV);
            uint16_t *d = dy + (by + y) * 16 + bx + x;
            *d = (uint16_t)(avg ? f2(*d, v) : v);
        }
        MI355_WAVE_SYNC();
    }
#endif
#ifdef MI355_WIDE_EXP_NOCHROMA
    return;
#endif
    /* chroma: eighth-sample bilinear (h264chroma_template.c:28-200); 4:2:2 keeps the luma's vertical resolution (h264_mb.c:284-315) */
    const int cw = w >> 1, ch = CF == 2 ? h : h >> 1, cby = CF == 2 ? by : by >> 1;
    const int myc = CF == 1 ? my + uniform((int)s.hdr.u.inter.chroma_dy[list][quadrant]) : my;
    const int cx = mx >> 3, cy = CF == 2 ? myc >> 2 : myc >> 3, fx = mx & 7, fy = CF == 2 ? (myc << 1) & 7 : myc & 7;
    const int CWd = 8 * fr.mb_width, CHt = (F::CH * fr.mb_height) >> g.hs, cww = cw + 1, chh = ch + 1, ncc = (cww + 7) >> 3;
    if (cx >= 0 && cy >= 0 && cx + 8 * ncc <= CWd && cy + chh <= CHt) {
        const int r = ncc == 2 ? lane >> 1 : lane, c = lane - r * ncc;
        for (int p = 0; p < 2; p++)
            if (r < chh) {
                PX v[8];
                __builtin_memcpy(v, rp[1 + p] + (size_t)(cy + r) * rcs + (size_t)(cx + 8 * c) * sizeof(PX), sizeof(v));
#pragma unroll
                for (int k = 0; k < 8; k++) s.win[p * CWIN + r * CWP + 8 * c + k] = v[k];
            }
    } else
    for (int p = 0; p < 2; p++)
        for (int i = lane; i < cww * chh; i += 64) {
            const int r = i / cww, c = i - r * cww;
            const int xx = clip3(cx + c, 0, CWd - 1), yy = clip3(cy + r, 0, CHt - 1);
            s.win[p * CWIN + r * CWP + c] = reinterpret_cast<const PX *>(rp[1 + p] + (size_t)yy * rcs)[xx];
        }
    MI355_WAVE_SYNC();
    const int A = (8 - fx) * (8 - fy), B = fx * (8 - fy), C = (8 - fx) * fy, D = fx * fy;
    for (int p = 0; p < 2; p++)
        for (int i = lane; i < cw * ch; i += 64) {
            const int y = i >> (lw - 1), x = i & (cw - 1);
            const uint16_t *q = s.win + p * CWIN + y * CWP + x;
            const int v = (A * q[0] + B * q[1] + C * q[CWP] + D * q[CWP + 1] + 32) >> 6;
            uint16_t *d = (p ? dcr : dcb) + (cby + y) * 8 + (bx >> 1) + x;
            *d = (uint16_t)(avg ? f2(*d, v) : v);
        }
    MI355_WAVE_SYNC();
}

/* weighted prediction on a w x h block of an LDS tile: h264dsp_template.c:30-98 */
template <int BD>
__device__ inline void wide_weight(uint16_t *p, int pitch, int w, int h, int ld, int wt, int off)
{
    int o = (int)((unsigned)off << (ld + (BD - 8)));
    if (ld) o += 1 << (ld - 1);
    const int lw = w == 16 ? 4 : (w == 8 ? 3 : (w == 4 ? 2 : 1));
    for (int i = lane_id(); i < w * h; i += 64) {
        const int y = i >> lw, x = i & (w - 1);
        p[y * pitch + x] = (uint16_t)med3i((p[y * pitch + x] * wt + o) >> ld, 0, (1 << BD) - 1);
    }
    MI355_WAVE_SYNC();
}
template <int BD>
__device__ inline void wide_biweight(uint16_t *d, const uint16_t *s, int pitch, int w, int h, int ld, int wd, int ws, int off)
{
    const int o = (int)((unsigned)((((int)((unsigned)off << (BD - 8))) + 1) | 1) << ld);
    const int lw = w == 16 ? 4 : (w == 8 ? 3 : (w == 4 ? 2 : 1));
    for (int i = lane_id(); i < w * h; i += 64) {
        const int y = i >> lw, x = i & (w - 1);
        d[y * pitch + x] = (uint16_t)med3i((s[y * pitch + x] * ws + d[y * pitch + x] * wd + o) >> (ld + 1), 0, (1 << BD) - 1);
    }
    MI355_WAVE_SYNC();
}

/* mc_part (h264_mc_template.c:44-62) -> mc_part_std / mc_part_weighted (h264_mb.c:320-471) */
template <int BD, int CF>
__device__ inline void wide_mc_part(WideInterLds &s, const mi355_h264_frame &fr, const mi355_h264_slice &sl, int mb_x, const WideGeom &g,
                                    int n_raster, int quadrant, int bx, int by, int w, int h, int l0, int l1)
{
    /* a field macroblock of an MBAFF frame counts fields: entry 16 + 2i (+ 1) of the reference's weight tables repeats frame i's (h264_slice.c pred_weight_table) */
    const int ri0 = uniform((int)s.hdr.ref_idx[0][quadrant]), ri1 = uniform((int)s.hdr.ref_idx[1][quadrant]), mbf = uniform((int)s.hdr.flags);
    const int r0 = ri0 >> g.hs, r1 = ri1 >> g.hs;
    /* implicit weights come from the distances between FIELDS for such a macroblock: a table per parity of the macroblock row (h264_slice.c:623-682) */
    const int iw = g.hs ? sl.implicit_weight_field[g.y0 & 1][ri0 & 31][ri1 & 31] : sl.implicit_weight[r0 & 15][r1 & 15];
    const bool weighted = (mbf & MI355_MBF_WEIGHTED) && ((sl.use_weight == 2 && l0 && l1 && iw != 32) || sl.use_weight == 1);
    const bool two = l0 && l1;
    for (int list = 0; list < 2; list++) {
        if (!(list ? l1 : l0)) continue;
        const bool second = list == 1 && two, to_q = second && weighted;
        wide_mc_dir<BD, CF>(s, fr, mb_x, g, list, n_raster, quadrant, bx, by, w, h, to_q ? s.qy : s.py, to_q ? s.qc[0] : s.pc[0], to_q ? s.qc[1] : s.pc[1],
                            second && !weighted);
    }
    if (!weighted) return;
    const int cw = w >> 1, ch = CF == 2 ? h : h >> 1, co = (CF == 2 ? by : by >> 1) * 8 + (bx >> 1);
    uint16_t *dy = s.py + by * 16 + bx, *dcb = s.pc[0] + co, *dcr = s.pc[1] + co;
    if (two) {
        const uint16_t *ty = s.qy + by * 16 + bx, *tcb = s.qc[0] + co, *tcr = s.qc[1] + co;
        if (sl.use_weight == 2) {
            const int w0 = iw, w1 = 64 - w0;
            wide_biweight<BD>(dy, ty, 16, w, h, 5, w0, w1, 0);
            wide_biweight<BD>(dcb, tcb, 8, cw, ch, 5, w0, w1, 0);
            wide_biweight<BD>(dcr, tcr, 8, cw, ch, 5, w0, w1, 0);
        } else {
            wide_biweight<BD>(dy, ty, 16, w, h, sl.luma_log2_weight_denom, sl.luma_weight[r0][0][0], sl.luma_weight[r1][1][0],
                              sl.luma_weight[r0][0][1] + sl.luma_weight[r1][1][1]);
            wide_biweight<BD>(dcb, tcb, 8, cw, ch, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][0][0], sl.chroma_weight[r1][1][0][0],
                              sl.chroma_weight[r0][0][0][1] + sl.chroma_weight[r1][1][0][1]);
            wide_biweight<BD>(dcr, tcr, 8, cw, ch, sl.chroma_log2_weight_denom, sl.chroma_weight[r0][0][1][0], sl.chroma_weight[r1][1][1][0],
                              sl.chroma_weight[r0][0][1][1] + sl.chroma_weight[r1][1][1][1]);
        }
    } else {
        const int list = l1 ? 1 : 0, refn = list ? r1 : r0;
        wide_weight<BD>(dy, 16, w, h, sl.luma_log2_weight_denom, sl.luma_weight[refn][list][0], sl.luma_weight[refn][list][1]);
        if (sl.use_weight_chroma) {
            wide_weight<BD>(dcb, 8, cw, ch, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][0][0], sl.chroma_weight[refn][list][0][1]);
            wide_weight<BD>(dcr, 8, cw, ch, sl.chroma_log2_weight_denom, sl.chroma_weight[refn][list][1][0], sl.chroma_weight[refn][list][1][1]);
        }
    }
}

template <int BD, int CF>
__global__ void __launch_bounds__(64)
k_wide_inter(const mi355_h264_frame *__restrict__ frames, int max_w, int max_h)
{
    __shared__ WideInterLds s;
    const int per = max_w * max_h, f = (int)blockIdx.x / per, rem = (int)blockIdx.x - f * per, mb_y = rem / max_w, mb_x = rem - mb_y * max_w;
    const mi355_h264_frame &fr = frames[f];
    if (mb_x >= fr.mb_width || mb_y >= fr.mb_height || (fr.flags & MI355_FRAME_NO_INTER)) return;
    const int mb_xy = mb_y * fr.mb_width + mb_x, lane = lane_id();
    if (lane < 16) reinterpret_cast<uint32_t *>(&s.hdr)[lane] = reinterpret_cast<const uint32_t *>(&fr.mb[mb_xy])[lane];
    else if (lane < 48) {
        const int list = (lane >> 4) - 1;
        s.mv[list][lane & 15] = fr.mv[list] ? reinterpret_cast<const uint32_t *>(fr.mv[list])[(size_t)mb_xy * 16 + (lane & 15)] : 0u;
    }
    MI355_WAVE_SYNC();
    /* one macroblock per wave: what the record says is the same for every lane — as scalars the partition loop, the quarter-sample position
     * and the window test below are branches of the wave, not masks of its lanes */
    const uint32_t t = (uint32_t)uniform((int)s.hdr.mb_type);
    if (t & MI355_MB_INTRA) return;
    const int cbp = uniform((int)s.hdr.cbp);
    const bool luma_coded = (cbp & 15) != 0, chroma_coded = (cbp & 0x30) != 0;
    WideCoefRegs<BD, CF> cregs;
    if (luma_coded || chroma_coded) wide_fetch_coefs<BD, CF>(cregs, fr, mb_xy, (cbp & 15) | (chroma_coded ? 16 : 0));
    const mi355_h264_slice &sl = fr.slices[uniform((int)s.hdr.slice_id)];
    const WideGeom g = wide_geom<BD, CF>(fr, t, mb_y);

    /* hl_motion, h264_mc_template.c:64-163 */
#define DIRF(part, list) (int)((t >> (12 + (part) + 2 * (list))) & 1)
    const int kind = (t & MI355_MB_16x16) ? 0 : ((t & MI355_MB_16x8) ? 1 : ((t & MI355_MB_8x16) ? 2 : 3));
    const int nparts = kind == 0 ? 1 : (kind == 3 ? 16 : 2);
    for (int p = 0; p < nparts; p++) {
        int n, quad, bx, by, w, h, l0, l1;
        if (kind == 0) { n = 0; quad = 0; bx = by = 0; w = h = 16; l0 = DIRF(0, 0); l1 = DIRF(0, 1); }
        else if (kind == 1) { n = 8 * p; quad = 2 * p; bx = 0; by = 8 * p; w = 16; h = 8; l0 = DIRF(p, 0); l1 = DIRF(p, 1); }
        else if (kind == 2) { n = 2 * p; quad = p; bx = 8 * p; by = 0; w = 8; h = 16; l0 = DIRF(p, 0); l1 = DIRF(p, 1); }
        else {
            const int i = p >> 2, j = p & 3;
            const int st = uniform((int)s.hdr.sub_mb_type[i]), shape = st & 3;
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
        wide_mc_part<BD, CF>(s, fr, sl, mb_x, g, n, quad, bx, by, w, h, l0, l1);
    }
#undef DIRF
#ifdef MI355_WIDE_EXP_NORES
    wide_store_mb<BD, CF>(fr, mb_x, g, s.py, 16, s.pc[0], s.pc[1], 8);
    return;
#endif
    if (luma_coded || chroma_coded) wide_commit_coefs<BD, CF>(s.coef, cregs);       /* the windows' place is free now */
    /* hl_decode_mb_idct_luma (h264_mb.c:726-795): idct_add16 / idct8_add4 choose between full, DC-only and nothing per block; so does
     * wide_block4 / wide_add_blocks8, from the coefficients (a block whose count is 1 with a DC level holds nothing else) */
    if (luma_coded) {
        if (s.hdr.flags & MI355_MBF_BYPASS) wide_add_raw<BD, CF>(s.coef, (t & MI355_MB_8x8DCT) ? 4 : 16, (t & MI355_MB_8x8DCT) ? 8 : 4, false, s.py, 16);
        else if (t & MI355_MB_8x8DCT) wide_add_blocks8<BD, CF>(s.coef, s.t8, 0, 4, s.py, 16);
        else wide_add_blocks4<BD, CF>(s.coef, 16, false, s.py, 16);
    }
    wide_residual_chroma<BD, CF>(s.coef, s.hdr, s.pc[0], s.pc[1], 8);
    wide_store_mb<BD, CF>(fr, mb_x, g, s.py, 16, s.pc[0], s.pc[1], 8);
}

/* ------------------------------------------------------------------------- */
/* intra                                                                        */
/* ------------------------------------------------------------------------- */
constexpr int TPW = 32;   /* luma tile pitch: columns -1..23 at TOW - 1 .. */
constexpr int CPW = 16;   /* chroma tile pitch: columns -1..7 */
constexpr int TOW = 4;
struct WideIntraLds {
    mi355_h264_mb hdr;
    alignas(16) int32_t coef[512];
    uint16_t tile[17 * TPW];
    uint16_t ctile[2][17 * CPW];
    PredScratch ps;
    int32_t t8[4][64];
};
#define WTILE(x, y) s.tile[((y) + 1) * TPW + (x) + TOW]
#define WCTILE(p, x, y) s.ctile[p][((y) + 1) * CPW + (x) + TOW]

template <int BD, int CF>
__global__ void __launch_bounds__(64)
k_wide_intra(const mi355_h264_frame *frames, int level, int width)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    __shared__ WideIntraLds s;
    const int f = (int)blockIdx.x / width, k = (int)blockIdx.x - f * width;
    const mi355_h264_frame &fr = frames[f];
    if (level > fr.max_intra_level) return;
    const int first = fr.intra_level_start[level - 1], count = fr.intra_level_start[level] - first;
    if (k >= count) return;
    const int mb_xy = (int)fr.intra_list[first + k], mb_x = mb_xy % fr.mb_width, mb_y = mb_xy / fr.mb_width, lane = lane_id();
    if (lane < 16) reinterpret_cast<uint32_t *>(&s.hdr)[lane] = reinterpret_cast<const uint32_t *>(&fr.mb[mb_xy])[lane];
    MI355_WAVE_SYNC();
    const mi355_h264_mb &h = s.hdr;
    const uint32_t t = h.mb_type;
    /* Intra16x16 carries DC levels in every block whatever its pattern says; I_PCM is all samples */
    wide_load_coefs<BD, CF>(s.coef, fr, mb_xy, (t & MI355_MB_INTRA_PCM) ? 31 : (((t & MI355_MB_INTRA16x16) ? 15 : (h.cbp & 15)) | ((h.cbp & 0x30) ? 16 : 0)));
    const WideGeom g = wide_geom<BD, CF>(fr, t, mb_y);
    const bool bypass = (h.flags & MI355_MBF_BYPASS) != 0, bypass_pred = (h.flags & MI355_MBF_BYPASS_PRED) != 0, x264old = (h.flags & MI355_MBF_BYPASS_X264OLD) != 0;
    if (t & MI355_MB_INTRA_PCM) {        /* h264_mb_template.c:101-153: the samples themselves, one per coefficient slot: Y, Cb, Cr */
        for (int i = lane; i < 256; i += 64) WTILE(i & 15, i >> 4) = (uint16_t)s.coef[i];
        for (int i = lane; i < 8 * F::CH; i += 64) { WCTILE(0, i & 7, i >> 3) = (uint16_t)s.coef[256 + i]; WCTILE(1, i & 7, i >> 3) = (uint16_t)s.coef[256 + 8 * F::CH + i]; }
        MI355_WAVE_SYNC();
        wide_store_mb<BD, CF>(fr, mb_x, g, &WTILE(0, 0), TPW, &WCTILE(0, 0, 0), &WCTILE(1, 0, 0), CPW);
        return;
    }
    /* the unfiltered edge samples of the neighbours: the row above (columns -1..23), the column to the left */
    const int ys = fr.recon_stride[0], cs = fr.recon_stride[1], pic_w = 16 * fr.mb_width;
    /* (a field macroblock of an MBAFF frame: the line above in ITS field, the column to its left on ITS lines — the samples the reference reads with
     * the doubled line size, whatever the neighbouring pairs' coding) */
    if (g.y0 - g.ystep >= 0 && lane < 25 && mb_x * 16 + lane - 1 >= 0 && mb_x * 16 + lane - 1 < pic_w)
        WTILE(lane - 1, -1) = reinterpret_cast<const PX *>(fr.recon[0] + (size_t)(g.y0 - g.ystep) * ys)[16 * mb_x + lane - 1];
    if (mb_x > 0 && lane >= 32 && lane < 48)
        WTILE(-1, lane - 32) = reinterpret_cast<const PX *>(fr.recon[0] + (size_t)(g.y0 + g.ystep * (lane - 32)) * ys)[16 * mb_x - 1];
    for (int p = 0; p < 2; p++) {
        if (g.cy0 - g.ystep >= 0 && lane < 9 && (mb_x > 0 || lane > 0))
            WCTILE(p, lane - 1, -1) = reinterpret_cast<const PX *>(fr.recon[1 + p] + (size_t)(g.cy0 - g.ystep) * cs)[8 * mb_x + lane - 1];
        if (mb_x > 0 && lane >= 16 && lane < 16 + F::CH)
            WCTILE(p, -1, lane - 16) = reinterpret_cast<const PX *>(fr.recon[1 + p] + (size_t)(g.cy0 + g.ystep * (lane - 16)) * cs)[8 * mb_x - 1];
    }
    MI355_WAVE_SYNC();

    /* chroma prediction: hpc.pred8x8[chroma_pred_mode] — the pred8x16 functions when chroma_format_idc == 2 (h264pred.c:439-470) */
    for (int p = 0; p < 2; p++) {
        if (lane < 9) s.ps.T[lane] = WCTILE(p, lane - 1, -1);
        if (lane >= 16 && lane < 17 + F::CH) s.ps.L[lane - 16] = WCTILE(p, -1, lane - 17);
        MI355_WAVE_SYNC();
        intra_pred_wave<uint16_t, BD>(s.ps, CF == 2 ? 4 : 2, h.chroma_pred_mode, 0, 0, &WCTILE(p, 0, 0), CPW);
    }
    if (t & MI355_MB_INTRA16x16) {       /* h264_mb.c:701-722 */
        if (lane < 17) s.ps.T[lane] = WTILE(lane - 1, -1);
        if (lane >= 32 && lane < 49) s.ps.L[lane - 32] = WTILE(-1, lane - 33);
        MI355_WAVE_SYNC();
        intra_pred_wave<uint16_t, BD>(s.ps, 3, h.intra16x16_pred_mode, 0, 0, &WTILE(0, 0), TPW);
        if (bypass) {                     /* h264_mb.c:712-722 (the DC levels are in their blocks already), :733-750 */
            if (bypass_pred && (h.intra16x16_pred_mode == 1 || h.intra16x16_pred_mode == 2)) {
                const bool vertical = h.intra16x16_pred_mode == 2;
                wide_lossless_run<BD, CF>(s.coef, false, vertical, 16, 16, &WTILE(0, 0), TPW, [&](int k) { return (int)(vertical ? WTILE(k, -1) : WTILE(-1, k)); });
            } else
                wide_add_raw<BD, CF>(s.coef, 16, 4, false, &WTILE(0, 0), TPW);
        } else {
        if (lane == 0 && ((h.nnz_mask >> MI355_NNZ_LUMA_DC) & 1)) wide_luma_dc<typename F::COEF>(s.coef, (int)h.dc_qmul[0]);
        MI355_WAVE_SYNC();
        wide_add_blocks4<BD, CF>(s.coef, 16, false, &WTILE(0, 0), TPW);      /* idct_add16intra: full, DC-only or nothing per block */
        }
    } else if (t & MI355_MB_8x8DCT) {    /* Intra 8x8: h264_mb.c:626-656 */
        for (int i8 = 0; i8 < 4; i8++) {
            const int x0 = 8 * (i8 & 1), y0 = 8 * (i8 >> 1), i = 4 * i8;
            if (lane < 17) s.ps.T[lane] = WTILE(x0 + lane - 1, y0 - 1);
            if (lane >= 32 && lane < 41) s.ps.L[lane - 32] = WTILE(x0 - 1, y0 + lane - 33);
            MI355_WAVE_SYNC();
            const int dir = h.u.intra4x4_pred_mode[i];
            intra_pred_wave<uint16_t, BD>(s.ps, 1, dir, (h.topleft_samples_available << i) & 0x8000,
                                          (h.topright_samples_available << i) & 0x4000, &WTILE(x0, y0), TPW);
            if (bypass) {                 /* h264_mb.c:628-643: the sums start from the filtered edge the predictor has just left in s.ps (pred8x8l_*_filter_add), or — x264 before build 151 — from the samples themselves */
                if (bypass_pred && (dir == 0 || dir == 1)) {
                    if (lane < 8) {
                        const bool vertical = dir == 0;
                        int v = x264old ? (int)(vertical ? WTILE(x0 + lane, y0 - 1) : WTILE(x0 - 1, y0 + lane)) : (int)(vertical ? s.ps.fT[1 + lane] : s.ps.fL[1 + lane]);
                        for (int k = 0; k < 8; k++) {
                            const int x = vertical ? lane : k, y = vertical ? k : lane;
                            v = wide_wrap<BD, CF>(v + s.coef[64 * i8 + 8 * y + x]);
                            WTILE(x0 + x, y0 + y) = (uint16_t)v;
                        }
                    }
                    MI355_WAVE_SYNC();
                } else {
                    if (lane < 8) for (int k = 0; k < 8; k++) WTILE(x0 + k, y0 + lane) = wide_wrap<BD, CF>(WTILE(x0 + k, y0 + lane) + s.coef[64 * i8 + 8 * lane + k]);
                    MI355_WAVE_SYNC();
                }
            } else
            wide_add_blocks8<BD, CF>(s.coef, s.t8, i8, 1, &WTILE(0, 0), TPW);
        }
    } else {                              /* Intra 4x4: h264_mb.c:657-700 */
        for (int i = 0; i < 16; i++) {
            const int x0 = 4 * blk_x4(i), y0 = 4 * blk_y4(i);
            const int tr_ok = (h.topright_samples_available << i) & 0x8000;
            if (lane < 5) s.ps.T[lane] = WTILE(x0 + lane - 1, y0 - 1);
            else if (lane < 9) s.ps.T[lane] = tr_ok ? WTILE(x0 + lane - 1, y0 - 1) : WTILE(x0 + 3, y0 - 1);
            if (lane >= 32 && lane < 37) s.ps.L[lane - 32] = WTILE(x0 - 1, y0 + lane - 33);
            MI355_WAVE_SYNC();
            const int dir = h.u.intra4x4_pred_mode[i];
            intra_pred_wave<uint16_t, BD>(s.ps, 0, dir, 0, 0, &WTILE(x0, y0), TPW);
            if (bypass) {                 /* h264_mb.c:665-669, :690-698 */
                if (lane < 4) {
                    if (bypass_pred && (dir == 0 || dir == 1)) {
                        const bool vertical = dir == 0;
                        int v = vertical ? WTILE(x0 + lane, y0 - 1) : WTILE(x0 - 1, y0 + lane);
                        for (int k = 0; k < 4; k++) {
                            const int x = vertical ? lane : k, y = vertical ? k : lane;
                            v = wide_wrap<BD, CF>(v + s.coef[16 * i + 4 * y + x]);
                            WTILE(x0 + x, y0 + y) = (uint16_t)v;
                        }
                    } else
                        for (int k = 0; k < 4; k++) WTILE(x0 + k, y0 + lane) = wide_wrap<BD, CF>(WTILE(x0 + k, y0 + lane) + s.coef[16 * i + 4 * lane + k]);
                }
            } else
            if (lane < 4) {
                int r[16];
                if (wide_block4<typename F::COEF>(s.coef + 16 * i, r)) {
                    uint16_t *d = &WTILE(x0, y0 + lane);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int rc = lane == 0 ? r[c] : (lane == 1 ? r[4 + c] : (lane == 2 ? r[8 + c] : r[12 + c]));
                        d[c] = (uint16_t)clip3(d[c] + rc, 0, F::MAXV);
                    }
                }
            }
            MI355_WAVE_SYNC();
        }
    }
    wide_residual_chroma<BD, CF>(s.coef, h, &WCTILE(0, 0, 0), &WCTILE(1, 0, 0), CPW);
    wide_store_mb<BD, CF>(fr, mb_x, g, &WTILE(0, 0), TPW, &WCTILE(0, 0, 0), &WCTILE(1, 0, 0), CPW);
}
#undef WTILE
#undef WCTILE

/* ------------------------------------------------------------------------- */
/* loop filter                                                                  */
/* ------------------------------------------------------------------------- */
__device__ const uint8_t kw_alpha[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28,
    32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
__device__ const uint8_t kw_beta[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8,
    9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
__device__ const uint8_t kw_tc0[52][3] = {
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,0},
    {0,0,0},{0,0,0},{0,0,0},{0,0,0},{0,0,1},{0,0,1},{0,0,1},{0,0,1},{0,1,1},{0,1,1},{1,1,1},{1,1,1},{1,1,1},
    {1,1,1},{1,1,2},{1,1,2},{1,1,2},{1,1,2},{1,2,3},{1,2,3},{2,2,3},{2,2,4},{2,3,4},{2,3,4},{3,3,5},{3,4,6},
    {3,4,6},{4,5,7},{4,5,8},{4,6,9},{5,7,10},{6,8,11},{6,8,13},{7,10,14},{8,11,16},{9,12,18},{10,13,20},
    {11,15,23},{13,17,25} };

constexpr int DYP = 20, DCPW = 12;       /* pitches of the MBAFF filter's luma (-4..15) and chroma (-4..7) tiles */
/* The frame / field filter's tiles hold a UNIT: up to WIDE_UNIT macroblocks side by side (round 6).  A lane's piece of a 10-bit luma row is 32 bytes of a 128-byte line; fetched
 * macroblock by macroblock the rest of the line came again for the next macroblock unless the L2 had kept it — it had not (3072 groups of sixteen lines each in flight per XCD's
 * 4 MB): 20 GB crossed the fabric per 512 pictures for 4.6 GB of samples (profiles/r04zzz_wide_first_form_counters.txt).  Now a unit's rows come in ONCE, as runs of whole lines
 * (four macroblocks = 128 bytes of a luma row, 64 of a chroma row), the macroblocks are filtered where they lie — macroblock u at columns 16 u .., its left neighbour's columns
 * beside it — and the rows leave once. */
#ifndef MI355_WIDE_UNIT_MAX
#define MI355_WIDE_UNIT_MAX 4            /* developer switch: 2 = tiles of two macroblocks (half the LDS, half lines per piece) */
#endif
constexpr int WIDE_UNIT = MI355_WIDE_UNIT_MAX;
#ifndef MI355_WIDE_TILE_MARGIN
#define MI355_WIDE_TILE_MARGIN 4        /* columns left of the unit in a tile row: the four the filter reaches (8: a macroblock's sample 0 on a 16-byte boundary, 320 bytes more per group) */
#endif
constexpr int DBM = MI355_WIDE_TILE_MARGIN, DBLA = DBM % 8 == 0 ? 16 : 8;      /* ... and what that says about the alignment of a macroblock's row piece in LDS */
constexpr int DBYP = DBM + 16 * WIDE_UNIT, DBCP = DBM + 8 * WIDE_UNIT;      /* pitches: columns -DBM .. 16 * WIDE_UNIT - 1 / -DBM .. 8 * WIDE_UNIT - 1 */
template <int CF> struct WideDbLds {
    mi355_h264_mb m[3];                  /* this macroblock, its left and its top neighbour */
    int32_t ref[2][25];                  /* the filter's view of the motion, (y + 1) * 5 + (x + 1), x, y = -1..3: picture identity (-1: none) */
    uint32_t mv[2][25];
    uint8_t nnz[25];
    alignas(8) uint8_t bs[4][8];         /* [i][4 * dir + edge]: the eight strengths a line meets are one 64-bit read */
    alignas(16) uint16_t y[20 * DBYP];   /* rows -4..15; a macroblock's sample 0 on a 16-byte boundary: a row piece is one LDS instruction */
    alignas(16) uint16_t c[2][(CF == 2 ? 18 : 10) * DBCP];       /* rows -2..7 (4:2:2: ..15), columns -4.. (the filter reaches two to the left; four make the write-back whole dwords) */
};
/* (xo, xc: the macroblock's first column inside the unit's tile, luma and chroma — locals of the code that uses the macros) */
#define DY(x, yy) s.y[((yy) + 4) * DBYP + (x) + DBM + xo]
#define DC(p, x, yy) s.c[p][((yy) + 2) * DBCP + (x) + DBM + xc]

__device__ __forceinline__ bool wide_mv_far(uint32_t a, uint32_t b, int ylim)
{
    return pk_absdiff_far(a, b, ylim == 2 ? 0xFFFEFFFCu : 0xFFFCFFFCu);
}

/* The line filters without branches — the form of h264_deblock.hip's luma_line / chroma_line for samples up to MAXV: a line's conditions are
 * folded into the clipping bounds (tc = 0 leaves a sample as it is), |a - b| is one v_sad_u16 (samples and thresholds lie below 65536), a
 * clip one v_med3_i32; the bS 4 forms run only where a lane of the wave has that strength (MAY_INTRA: macroblock edges).  Return: 0 — no line
 * of the WAVE passes the conditions (nothing changed), 1, or 2 when the bS 4 filter ran (p2 / q2 may have changed).
 * h264dsp_template.c:103-163 (normal), :165-210 (bS 4), :212-265 (chroma). */
/* VOTE false: for callers inside lane-dependent control flow (the MBAFF filter) — no wave vote, every lane computes; returns 1, or 2 for a bS 4 lane */
template <int MAXV, bool MAY_INTRA, bool VOTE = true>
__device__ __forceinline__ int wide_luma_line(int p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int q3, int bs, int alpha, int beta, int tc0)
{
    const bool f = bs != 0 && absdiff8(p0, q0) < alpha && absdiff8(p1, p0) < beta && absdiff8(q1, q0) < beta;
    if (VOTE && !__any(f)) return 0;
    const bool ap = absdiff8(p2, p0) < beta, aq = absdiff8(q2, q0) < beta;
    const bool fn = f && bs < 4;
    const int avg = (p0 + q0 + 1) >> 1;
    const int tp = fn && ap ? tc0 : 0, tq = fn && aq ? tc0 : 0, tc = fn ? tc0 + (int)ap + (int)aq : 0;
    const int np1 = p1 + med3i(((p2 + avg) >> 1) - p1, -tp, tp), nq1 = q1 + med3i(((q2 + avg) >> 1) - q1, -tq, tq);
    const int delta = med3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    const int P0 = p0, P1 = p1, P2 = p2, Q0 = q0, Q1 = q1, Q2 = q2;
    p1 = np1; q1 = nq1;
    p0 = med3i(P0 + delta, 0, MAXV);
    q0 = med3i(Q0 - delta, 0, MAXV);
    if (MAY_INTRA) {
        const bool fi = f && bs == 4;
        if (VOTE ? (bool)__any(fi) : fi) {
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
/* tc1: the caller's tc0 + 1 (h264_loopfilter.c:126-129) */
template <int MAXV, bool VOTE = true>
__device__ __forceinline__ bool wide_chroma_line(int p1, int &p0, int &q0, int q1, int bs, int alpha, int beta, int tc1)
{
    const bool f = bs != 0 && absdiff8(p0, q0) < alpha && absdiff8(p1, p0) < beta && absdiff8(q1, q0) < beta;
    if (VOTE && !__any(f)) return false;
    const int tc = f && bs < 4 ? tc1 : 0;
    const int delta = med3i((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    const bool fi = f && bs == 4;
    const int np0 = fi ? (2 * p1 + p0 + q1 + 2) >> 2 : med3i(p0 + delta, 0, MAXV);
    const int nq0 = fi ? (2 * q1 + q0 + p1 + 2) >> 2 : med3i(q0 - delta, 0, MAXV);
    p0 = np0; q0 = nq0;
    return true;
}
/* alpha, beta and the tc0 row (table 8-16 / 8-17; tc0 of strength bs: byte bs - 1) of an edge with average QP qp: three of them per plane of a
 * macroblock (left edge, top edge, inner edges), worked out once before the edge loop */
struct WideThr { int alpha, beta; uint32_t tc0; };
template <int BD>
__device__ __forceinline__ WideThr wide_thr(const uint8_t *t_alpha, const uint8_t *t_beta, const uint8_t (*t_tc0)[4], int qp, int a_off, int b_off)
{
    const int ia = med3i(qp - 6 * (BD - 8) + a_off, 0, 51), ib = med3i(qp - 6 * (BD - 8) + b_off, 0, 51);
    WideThr t;
    t.alpha = t_alpha[ia] << (BD - 8);
    t.beta = t_beta[ib] << (BD - 8);
    t.tc0 = *reinterpret_cast<const uint32_t *>(t_tc0[ia]);
    return t;
}
template <int BD>
__device__ __forceinline__ int wide_tc0(const WideThr &t, int bs) { return (int)((t.tc0 >> (8 * ((bs - 1) & 3))) & 0xFF) << (BD - 8); }
/* check_mv, h264_loopfilter.c:442-470 */
template <typename LDS>
__device__ inline int wide_check_mv(const LDS &s, int b, int bn, int list_count, int ylim)
{
    bool v = s.ref[0][b] != s.ref[0][bn];
    if (!v && s.ref[0][b] != -1) v = wide_mv_far(s.mv[0][b], s.mv[0][bn], ylim);
    if (list_count == 2) {
        if (!v) v = s.ref[1][b] != s.ref[1][bn] || wide_mv_far(s.mv[1][b], s.mv[1][bn], ylim);
        if (v) {
            if (s.ref[0][b] != s.ref[1][bn] || s.ref[0][bn] != s.ref[1][b]) return 1;
            return wide_mv_far(s.mv[0][b], s.mv[1][bn], ylim) || wide_mv_far(s.mv[1][b], s.mv[0][bn], ylim);
        }
    }
    return v;
}
/* the same without branches (the frame / field filter's strengths: every lane evaluates one pair of blocks) */
template <typename LDS>
__device__ __forceinline__ int wide_check_mv_bf(const LDS &s, int b, int bn, bool two_lists, int ylim)
{
    const int r0b = s.ref[0][b], r0n = s.ref[0][bn], r1b = s.ref[1][b], r1n = s.ref[1][bn];
    const uint32_t m0b = s.mv[0][b], m0n = s.mv[0][bn], m1b = s.mv[1][b], m1n = s.mv[1][bn];
    const bool v1 = (r0b != r0n) | ((r0b != -1) & wide_mv_far(m0b, m0n, ylim));
    /* (int operands: every term is evaluated, no short-circuit branches) */
    const int v2 = (int)v1 | (int)(r1b != r1n) | (int)wide_mv_far(m1b, m1n, ylim);
    const int cross = (int)(r0b != r1n) | (int)(r0n != r1b) | (int)wide_mv_far(m0b, m1n, ylim) | (int)wide_mv_far(m1b, m0n, ylim);
    return two_lists ? (v2 & cross) : (int)v1;
}
__device__ __forceinline__ int wide_ref_identity(const mi355_h264_mb &m, int list, int x4, int y4)
{
    if (m.mb_type & MI355_MB_INTRA) return -1;
    const int r = m.u.inter.ref_pic[list][(x4 >> 1) + 2 * (y4 >> 1)];
    return r == 0xFF ? -1 : r;
}

/* ---- the frame / field loop filter: what a group of sixteen lanes needs of its picture, of a macroblock, and the macroblock itself ---------- */
/* the picture's descriptor in registers: read once per wave (a field of `frames[f]` read inside the macroblock loop is a load again after every
 * store, for the compiler cannot know that the pictures do not overlap it) */
struct WideDbPic {
    const uint8_t *dst[3], *recon[3];
    const mi355_h264_mb *mb;
    const uint32_t *mv[2];
    const mi355_h264_slice *slices;
    int ys, cs, yd, cd, mbw, mbh, field, nslices;
};
__device__ __forceinline__ WideDbPic wide_db_pic(const mi355_h264_frame &fr)
{
    WideDbPic p;
#pragma unroll
    for (int k = 0; k < 3; k++) { p.dst[k] = fr.dst[k]; p.recon[k] = fr.recon[k]; }
    p.mb = fr.mb;
    p.mv[0] = reinterpret_cast<const uint32_t *>(fr.mv[0]); p.mv[1] = reinterpret_cast<const uint32_t *>(fr.mv[1]);
    p.slices = fr.slices;
    p.ys = fr.recon_stride[0]; p.cs = fr.recon_stride[1]; p.yd = fr.dst_stride[0]; p.cd = fr.dst_stride[1];
    p.mbw = fr.mb_width; p.mbh = fr.mb_height; p.field = fr.field_picture; p.nslices = fr.nslices;
    return p;
}
/* N samples of a picture row in registers, as memory holds them */
template <typename PX, int N> struct WidePiece { uint32_t w[N * sizeof(PX) / 4]; };
template <typename PX, int N>
__device__ __forceinline__ void wide_get(WidePiece<PX, N> &v, const uint8_t *p) { __builtin_memcpy(v.w, p, sizeof(v.w)); }      /* one load, any alignment */
template <typename PX, int N, int LA>
__device__ __forceinline__ void wide_put(const WidePiece<PX, N> &v, uint16_t *d)                                                /* -> 16-bit samples in LDS */
{
    PX t[N];
    uint16_t u[N];
    __builtin_memcpy(t, v.w, sizeof(t));
#pragma unroll
    for (int k = 0; k < N; k++) u[k] = t[k];
    __builtin_memcpy(__builtin_assume_aligned(d, LA), u, sizeof(u));
}
/* one macroblock's records and vectors as the loads deliver them: a group fetches those of the NEXT macroblock of its unit while it filters this one */
template <int BD, int CF> struct WideDbIn {
    uint32_t rec[3];                               /* dword l of this / the left / the top macroblock's record */
    uint32_t mv[2][2];                             /* [list][own block l | lanes 0..3 the left neighbour's last column, 4..7 the top neighbour's last row] */
};
template <int BD, int CF>
__device__ __forceinline__ void wide_db_fetch(WideDbIn<BD, CF> &in, const WideDbPic &pic, bool ok, int mb_x, int mb_y, int l)
{
    if (!ok) return;
    const int mb_xy = mb_y * pic.mbw + mb_x;
    const bool has_left = mb_x > 0, has_top = mb_y > 0;
    const int xl = has_left ? mb_xy - 1 : mb_xy, xt = has_top ? mb_xy - pic.mbw : mb_xy;
    in.rec[0] = reinterpret_cast<const uint32_t *>(&pic.mb[mb_xy])[l];
    in.rec[1] = reinterpret_cast<const uint32_t *>(&pic.mb[xl])[l];
    in.rec[2] = reinterpret_cast<const uint32_t *>(&pic.mb[xt])[l];
    /* vectors: lane l its own 4x4 block, lanes 0..3 / 4..7 also a block of the left neighbour's last column / the top neighbour's last row */
    const int xy2 = l < 4 ? xl : xt, i2 = l < 4 ? 3 + 4 * l : 8 + l;
#pragma unroll
    for (int list = 0; list < 2; list++) {
        in.mv[list][0] = pic.mv[list] ? pic.mv[list][(size_t)mb_xy * 16 + l] : 0u;
        in.mv[list][1] = pic.mv[list] && l < 8 ? pic.mv[list][(size_t)xy2 * 16 + i2] : 0u;
    }
}

/* The samples of a unit — macroblocks x0 .. x0 + nu - 1 of row mb_y (nu <= WIDE_UNIT) — into the group's tile: their own from `recon`; four rows of the top neighbours and
 * four columns of the left neighbour from `dst`, as those macroblocks' own passes left them (earlier launches).  Lane l: luma row l (nu pieces of sixteen samples), a quarter
 * of one of the four rows above, a chroma row (4:2:0: plane l >> 3, row l & 7; 4:2:2: row l of both planes), lanes 0..7 half of one of the two chroma rows above. */
template <int BD, int CF>
__device__ __forceinline__ void wide_db_fetch_unit(WideDbLds<CF> &s, const WideDbPic &pic, int x0, int nu, int mb_y, int l)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    constexpr int PXB = (int)sizeof(PX), xo = 0, xc = 0;
    const bool has_left = x0 > 0, has_top = mb_y > 0;
    WidePiece<PX, 16> y[WIDE_UNIT], yt;
    WidePiece<PX, 4> yl;
    const uint8_t *row = pic.recon[0] + (size_t)(16 * mb_y + l) * pic.ys + 16 * x0 * PXB;
#pragma unroll
    for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_get(y[k], row + 16 * PXB * k);
    if (has_left) wide_get(yl, pic.dst[0] + (size_t)(16 * mb_y + l) * pic.yd + (16 * x0 - 4) * PXB);
    const int tr = (l >> 2) - 4, tk = l & 3;
    if (has_top && tk < nu) wide_get(yt, pic.dst[0] + (size_t)(16 * mb_y + tr) * pic.yd + (16 * x0 + 16 * tk) * PXB);
    constexpr int NP = CF == 2 ? 2 : 1;
    WidePiece<PX, 8> c[NP][WIDE_UNIT];
    WidePiece<PX, 4> cl[NP];
    WidePiece<PX, 8> ct;
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const int p = CF == 2 ? q : l >> 3, r = CF == 2 ? l : l & 7;
        const uint8_t *crow = (p ? pic.recon[2] : pic.recon[1]) + (size_t)(F::CH * mb_y + r) * pic.cs + 8 * x0 * PXB;
#pragma unroll
        for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_get(c[q][k], crow + 8 * PXB * k);
        if (has_left) wide_get(cl[q], (p ? pic.dst[2] : pic.dst[1]) + (size_t)(F::CH * mb_y + r) * pic.cd + (8 * x0 - 4) * PXB);
    }
    /* the two chroma rows above: lane = plane (l >> 3), row ((l >> 2) & 1) - 2, macroblock l & 3 of the unit */
    const int cp = l >> 3, cr = ((l >> 2) & 1) - 2, ck = l & 3;
    const bool ct_on = has_top && ck < nu;
    if (ct_on) wide_get(ct, (cp ? pic.dst[2] : pic.dst[1]) + (size_t)(F::CH * mb_y + cr) * pic.cd + (8 * x0 + 8 * ck) * PXB);
    /* ---- into the tile ---- */
#pragma unroll
    for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_put<PX, 16, DBLA>(y[k], &DY(16 * k, l));
    if (has_left) wide_put<PX, 4, 8>(yl, &DY(-4, l));
    if (has_top && tk < nu) wide_put<PX, 16, DBLA>(yt, &DY(16 * tk, tr));
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const int p = CF == 2 ? q : l >> 3, r = CF == 2 ? l : l & 7;
#pragma unroll
        for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_put<PX, 8, DBLA>(c[q][k], &DC(p, 8 * k, r));
        if (has_left) wide_put<PX, 4, 8>(cl[q], &DC(p, -4, r));
    }
    if (ct_on) wide_put<PX, 8, DBLA>(ct, &DC(cp, 8 * ck, cr));
}
/* ... and back: the unit's macroblocks, the three columns its first left edge changed (four: whole dwords) and the three rows (one in chroma) its top edges changed */
template <int BD, int CF>
__device__ __forceinline__ void wide_db_store_unit(const WideDbLds<CF> &s, const WideDbPic &pic, int x0, int nu, int mb_y, int l)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    constexpr int PXB = (int)sizeof(PX), xo = 0, xc = 0;
    const bool has_left = x0 > 0, has_top = mb_y > 0;
    const int yd = pic.yd, cd = pic.cd;
    uint8_t *row = const_cast<uint8_t *>(pic.dst[0]) + (size_t)(16 * mb_y + l) * yd + 16 * x0 * PXB;
#pragma unroll
    for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_st_row<PX, 16, false, DBLA>(row + 16 * PXB * k, &DY(16 * k, l));
    if (has_left) wide_st_row<PX, 4, false, 8>(row - 4 * PXB, &DY(-4, l));
    if (has_top && l < 12) {
        const int r = (l >> 2) - 3, k = l & 3;
        if (k < nu) wide_st_row<PX, 16, false, DBLA>(const_cast<uint8_t *>(pic.dst[0]) + (size_t)(16 * mb_y + r) * yd + (16 * x0 + 16 * k) * PXB, &DY(16 * k, r));
    }
#pragma unroll
    for (int q = 0; q < (CF == 2 ? 2 : 1); q++) {
        const int p = CF == 2 ? q : l >> 3, r = CF == 2 ? l : l & 7;
        uint8_t *crow = const_cast<uint8_t *>(p ? pic.dst[2] : pic.dst[1]) + (size_t)(F::CH * mb_y + r) * cd + 8 * x0 * PXB;
#pragma unroll
        for (int k = 0; k < WIDE_UNIT; k++) if (k < nu) wide_st_row<PX, 8, false, DBLA>(crow + 8 * PXB * k, &DC(p, 8 * k, r));
        if (has_left) wide_st_row<PX, 4, false, 8>(crow - 4 * PXB, &DC(p, -4, r));
    }
    if (has_top && l < 2 * WIDE_UNIT) {           /* chroma row -1: lane = plane * WIDE_UNIT + macroblock */
        const int p = l / WIDE_UNIT, k = l % WIDE_UNIT;
        if (k < nu) wide_st_row<PX, 8, false, DBLA>(const_cast<uint8_t *>(p ? pic.dst[2] : pic.dst[1]) + (size_t)(F::CH * mb_y - 1) * cd + (8 * x0 + 8 * k) * PXB, &DC(p, 8 * k, -1));
    }
}

/* One macroblock (mb_x, mb_y) per sixteen-lane group — ff_h264_filter_mb (h264_loopfilter.c:716-847) for frame and field pictures without
 * MBAFF.  The macroblock's own samples come from `recon`, four rows of the top neighbour from `dst` (as that macroblock's own pass left
 * them) and four columns of the left neighbour from `dst` or — FIRST false: the group filtered that macroblock with this tile a moment ago
 * — from the tile's last columns; the macroblock, three columns and three rows (one of each in chroma) go back to `dst`.
 * `in`: what wide_db_fetch brought for this macroblock; once the tile is filled, `next` may fetch the following macroblock into it. */
template <int BD, int CF, typename Next>
__device__ __forceinline__ void wide_deblock_mb(WideDbLds<CF> &s, const uint8_t *t_alpha, const uint8_t *t_beta, const uint8_t (*t_tc0)[4], const uint8_t *t_lc,
                                                const WideDbPic &pic, const WideDbIn<BD, CF> &in, bool ok, int u, int l, Next next)
{
    typedef Fmt<BD, CF> F;
    const int xo = 16 * u, xc = 8 * u;           /* the macroblock's place in the unit's tile (DY / DC) */
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; k++) reinterpret_cast<uint32_t *>(&s.m[k])[l] = in.rec[k];
        /* the vectors' places in the filter's view of the motion: (y + 1) * 5 + (x + 1) */
        const int c0 = ((l >> 2) + 1) * 5 + (l & 3) + 1, c1 = l < 4 ? (l + 1) * 5 : l - 3;
#pragma unroll
        for (int list = 0; list < 2; list++) { s.mv[list][c0] = in.mv[list][0]; if (l < 8) s.mv[list][c1] = in.mv[list][1]; }
    }
    MI355_WAVE_SYNC();
    next();                                      /* the next macroblock's loads leave now and arrive during the filter */
    const mi355_h264_mb &m = s.m[0];
    const bool filter = ok && !(m.flags & MI355_MBF_NO_DEBLOCK);
    int list_count = 1;
    const int ylim = pic.field ? 2 : 4;
    if (filter) {
        /* the slice's list_count: of the picture's first sixteen slices in LDS since the wave began */
        list_count = m.slice_id < 16 ? t_lc[m.slice_id] : pic.slices[m.slice_id].list_count;
        /* picture identities and coefficient flags the strengths are derived from (fill_filter_caches, h264_slice.c:2056-2196), from the records */
        for (int k = 0; k < 2; k++) {
            if (k && l >= 8) break;
            int which, x4, y4, cx, cy;
            if (!k) { which = 0; x4 = l & 3; y4 = l >> 2; cx = x4; cy = y4; }
            else if (l < 4) { which = 1; x4 = 3; y4 = l; cx = -1; cy = y4; }
            else { which = 2; x4 = l - 4; y4 = 3; cx = x4; cy = -1; }
            const int ci = (cy + 1) * 5 + cx + 1;
            for (int list = 0; list < 2; list++) s.ref[list][ci] = wide_ref_identity(s.m[which], list, x4, y4);
            s.nnz[ci] = (uint8_t)((s.m[which].nnz_mask >> blk_index(x4, y4)) & 1);
        }
    }
    MI355_WAVE_SYNC();
    if (filter) {
        /* boundary strengths: filter_mb_dir, h264_loopfilter.c:472-714; lane l = 4 * edge + i, both directions.  Without branches: the one
         * pair of blocks whose motion decides (the lane's own pair, or the partition's first pair where the macroblock moves as a whole across
         * the edge) is compared in any case, the cases of :553-607 / :638-690 pick among 0, 2, 3 / 4 and that comparison */
        const uint32_t t = m.mb_type;
        const int edge = l >> 2, i = l & 3, tk = (t >> 3) & 7;
        const bool e0 = edge == 0, two = list_count == 2, intra_cur = (t & MI355_MB_INTRA) != 0;
#pragma unroll
        for (int dir = 0; dir < 2; dir++) {
            const int mask_edge = dir == 0 ? (tk == 0 ? 0 : (tk < 4 ? 3 : 1)) : (tk == 0 ? 0 : (tk == 1 ? 3 : (tk < 4 ? 1 : 3)));
            const int edges = (mask_edge == 3 && !(m.cbp & 15)) ? 1 : 4;
            const uint32_t par_types = MI355_MB_16x16 | (MI355_MB_8x16 >> dir);
            const bool mask_par0 = (t & par_types) != 0;
            const int x = dir == 0 ? edge : i, y = dir == 0 ? i : edge, step = dir ? 5 : 1;
            const int b = (y + 1) * 5 + x + 1, bn = b - step;
            const uint32_t mmt = s.m[1 + dir].mb_type;
            const bool par = mask_par0 & (!e0 | ((mmt & par_types) != 0));
            const int b0 = e0 ? 6 : (dir == 0 ? 6 + edge : (edge + 1) * 5 + 1), cb = par ? b0 : b;
            const int mvd = wide_check_mv_bf(s, cb, cb - step, two, ylim);
            const bool nz = (s.nnz[b] | s.nnz[bn]) != 0;
            const bool deblock_edge = !((t & MI355_MB_8x8DCT) && (edge & 1));
            const bool active = e0 ? (m.flags & (dir ? MI355_MBF_TOP_EDGE : MI355_MBF_LEFT_EDGE)) != 0 : (edge < edges) & (deblock_edge | (CF == 2 && dir == 1));
            const bool intra = e0 ? ((t | mmt) & MI355_MB_INTRA) != 0 : intra_cur;
            const int bs_intra = e0 ? ((!pic.field || dir == 0) ? 4 : 3) : 3;
            const int bs_inter = nz ? 2 : ((!e0 & ((edge & mask_edge) != 0)) ? 0 : mvd);
            const int bs = active ? (intra ? bs_intra : bs_inter) : 0;
            s.bs[i][4 * dir + edge] = (uint8_t)bs;
        }
    }
    MI355_WAVE_SYNC();
    {
        /* what the edge loop needs of the three records, read once (the tile's 16-bit stores below may alias a record's bytes for the compiler) */
        const int a_off = m.slice_alpha_c0_offset, b_off = m.slice_beta_offset, qp0 = m.qp;
        const bool dct8 = (m.mb_type & MI355_MB_8x8DCT) != 0;
        WideThr ty[3], tc[2][3];                 /* [inner, left, top]; chroma: [the plane(s) this lane filters] */
        for (int e = 0; e < 3; e++) ty[e] = wide_thr<BD>(t_alpha, t_beta, t_tc0, e ? (qp0 + s.m[e].qp + 1) >> 1 : qp0, a_off, b_off);
        for (int k = 0; k < (CF == 2 ? 2 : 1); k++) {
            const int p = CF == 2 ? k : l >> 3;
            for (int e = 0; e < 3; e++) tc[k][e] = wide_thr<BD>(t_alpha, t_beta, t_tc0, e ? (m.qpc[p] + s.m[e].qpc[p] + 1) >> 1 : m.qpc[p], a_off, b_off);
        }
        /* The edges, one direction at a time with the line in registers: lane l holds row l (then column l) of the luma tile, columns / rows
         * -4..15, and a row (then a column) of chroma; the four edges of a direction pass over it without a trip to LDS in between — what a lone
         * macroblock spends is latency (eight steps of read, filter, write, rendezvous were ~3 of its ~4 us), not arithmetic.  Writes go out
         * as an edge changes samples; one rendezvous between the directions. */
        const uint64_t bsy = filter ? *reinterpret_cast<const uint64_t *>(s.bs[l >> 2]) : 0;            /* strengths of luma line l: byte 4 * dir + edge */
        const uint64_t bsc = filter ? *reinterpret_cast<const uint64_t *>(s.bs[(l & 7) >> 1]) : 0;       /* of chroma line l & 7 (two lines per strength) */
        auto strength = [](uint64_t w, int dir, int edge) { return (int)((uint32_t)(w >> (8 * (4 * dir + edge))) & 0xFF); };
#pragma unroll
        for (int dir = 0; dir < 2; dir++) {
            /* luma */
            {
                uint16_t *base = dir == 0 ? &DY(0, l) : &DY(l, 0);
                const int st = dir == 0 ? 1 : DBYP;
                int r[20];
                bool any_bs = false;
#pragma unroll
                for (int edge = 0; edge < 4; edge++) any_bs |= strength(bsy, dir, edge) != 0 && (edge == 0 || !(dct8 && (edge & 1)));
                if (__any(any_bs)) {
#pragma unroll
                    for (int j = 0; j < 20; j++) r[j] = base[(j - 4) * st];
#pragma unroll
                    for (int edge = 0; edge < 4; edge++) {
                        const int te = edge ? 0 : 1 + dir;
                        const bool luma_on = edge == 0 || !(dct8 && (edge & 1));
                        const int bs = luma_on ? strength(bsy, dir, edge) : 0;
                        if (__any(bs != 0)) {
                            int *q = r + 4 + 4 * edge;
                            const int rc = edge == 0 ? wide_luma_line<F::MAXV, true>(q[-4], q[-3], q[-2], q[-1], q[0], q[1], q[2], q[3], bs, ty[te].alpha, ty[te].beta, wide_tc0<BD>(ty[te], bs))
                                                     : wide_luma_line<F::MAXV, false>(q[-4], q[-3], q[-2], q[-1], q[0], q[1], q[2], q[3], bs, ty[te].alpha, ty[te].beta, wide_tc0<BD>(ty[te], bs));
                            uint16_t *w = base + 4 * edge * st;
                            if (rc) { w[-2 * st] = (uint16_t)q[-2]; w[-st] = (uint16_t)q[-1]; w[0] = (uint16_t)q[0]; w[st] = (uint16_t)q[1]; }
                            if (rc == 2) { w[-3 * st] = (uint16_t)q[-3]; w[2 * st] = (uint16_t)q[2]; }
                        }
                    }
                }
            }
            /* chroma: vertical edges 0 and 2 at columns 0 and 4 (sixteen lines in 4:2:2: four per strength); horizontal edges 0 and 2
             * at rows 0 and 4 in 4:2:0, all four at rows 0, 4, 8, 12 in 4:2:2 (:679-686) */
            {
                constexpr int NE = 4;                                       /* edge slots; 4:2:0 and the vertical edges use 0 and 2 */
                const bool both = dir == 0 && CF == 2;                      /* 4:2:2, vertical edges: lane l = row l of both planes */
                const int pl = l >> 3, k = l & 7;
                const uint64_t bw = both ? bsy : bsc;
                const int est = dir == 0 ? 2 : (CF == 2 ? 4 : 2);           /* samples between edge slots */
                bool any_bs = false;
#pragma unroll
                for (int edge = 0; edge < NE; edge++) if (dir == 0 ? !(edge & 1) : (CF == 2 || !(edge & 1))) any_bs |= strength(bw, dir, edge) != 0;
                if (__any(any_bs)) {
#pragma unroll
                    for (int pp = 0; pp < (both ? 2 : 1); pp++) {
                        const int p = both ? pp : pl;
                        uint16_t *base = dir == 0 ? &DC(p, 0, both ? l : k) : &DC(p, k, 0);
                        const int st = dir == 0 ? 1 : DBCP;
                        constexpr int NS = 2 + 3 * 4 + 2;                   /* samples -2 .. 13 along the line */
                        int r[NS];
#pragma unroll
                        for (int j = 0; j < NS; j++) if (j - 2 < (dir == 0 ? 6 : (CF == 2 ? 14 : 6))) r[j] = base[(j - 2) * st];
#pragma unroll
                        for (int edge = 0; edge < NE; edge++) {
                            if (!(dir == 0 ? !(edge & 1) : (CF == 2 || !(edge & 1)))) continue;
                            const int te = edge ? 0 : 1 + dir;
                            const int bs = strength(bw, dir, edge);
                            if (__any(bs != 0)) {
                                const WideThr t = both ? tc[CF == 2 ? pp : 0][te] : (CF == 2 && pl ? tc[CF == 2 ? 1 : 0][te] : tc[0][te]);
                                int *q = r + 2 + est * edge;
                                if (wide_chroma_line<F::MAXV>(q[-2], q[-1], q[0], q[1], bs, t.alpha, t.beta, wide_tc0<BD>(t, bs) + 1)) {
                                    uint16_t *w = base + est * edge * st;
                                    w[-st] = (uint16_t)q[-1]; w[0] = (uint16_t)q[0];
                                }
                            }
                        }
                    }
                }
            }
            MI355_WAVE_SYNC();
        }
    }
}

/* Anti-diagonal d of FOUR pictures per wave: lanes 16g..16g+15 filter a unit of picture 4 * k + g.  Sixteen lanes are what one macroblock has
 * to offer (the sixteen lines across a luma edge; 8 + 8 chroma lines); four pictures fill the wave, and a lane moves whole rows (32 bytes of a
 * 10-bit luma row) between memory and the group's LDS tiles.  One launch per anti-diagonal (the reference's raster order needs left, top and
 * top-right done).
 * A unit (X, y) = macroblocks unit * X .. unit * X + unit - 1 of row y, filtered one after the other by the same group; the anti-diagonals
 * count units (d = X + 2 y: the unit to the left is launch d - 1, the units above and above-right are d - 2 and d - 1).  unit = 1: the shortest
 * launches (few pictures: a launch is as long as its longest wave).  unit = 4: what a group fetches is a run of whole cache lines (four 10-bit
 * macroblocks side by side are 128 bytes of a luma row) instead of a 32-byte piece of every line — measured: one macroblock per group moves
 * 8.8 x the bytes it uses across the fabric (72 lines of 128 bytes for 1.1 KB) — the left neighbour's columns stay in the tile, and the loads
 * of macroblock u + 1 are in flight while u is filtered. */
template <int BD, int CF>
__global__ void __launch_bounds__(64)
k_wide_deblock(const mi355_h264_frame *frames, int nframes, int d, int y_first, int rows, int unit)
{
    __shared__ WideDbLds<CF> sh[4];
    /* tables 8-16 / 8-17 in LDS: the edge loop looks alpha, beta and tc0 up per lane — from memory that is a dependent load of a microsecond */
    __shared__ uint8_t t_alpha[52], t_beta[52], t_lc[4][16];
    __shared__ __attribute__((aligned(4))) uint8_t t_tc0[52][4];       /* a row is read as one dword (wide_thr) */
    const int lane = lane_id(), g = lane >> 4, l = lane & 15;
    if (lane < 52) { t_alpha[lane] = kw_alpha[lane]; t_beta[lane] = kw_beta[lane]; t_tc0[lane][0] = kw_tc0[lane][0]; t_tc0[lane][1] = kw_tc0[lane][1]; t_tc0[lane][2] = kw_tc0[lane][2]; t_tc0[lane][3] = 0; }
    /* the launch holds the rows that have a unit on this anti-diagonal (of the largest picture): y_first .. y_first + rows - 1 */
    const int f = 4 * ((int)blockIdx.x / rows) + g, mb_y = y_first + (int)blockIdx.x % rows, x0 = (d - 2 * mb_y) * unit;
    const WideDbPic pic = wide_db_pic(frames[f < nframes ? f : nframes - 1]);
    const bool row_ok = f < nframes && mb_y < pic.mbh && x0 >= 0;
    /* the unit's macroblocks inside the picture: nu (0: this group has nothing on this anti-diagonal) */
    const int nu = row_ok && x0 < pic.mbw ? imin(unit, pic.mbw - x0) : 0;
    WideDbIn<BD, CF> in;
    wide_db_fetch<BD, CF>(in, pic, nu > 0, x0, mb_y, l);
    if (nu > 0) wide_db_fetch_unit<BD, CF>(sh[g], pic, x0, nu, mb_y, l);
    t_lc[g][l] = row_ok && pic.nslices > 0 ? pic.slices[l < pic.nslices ? l : pic.nslices - 1].list_count : 1;
    MI355_WAVE_SYNC();
#pragma unroll 1
    for (int u = 0; u < unit; u++) {
        const int mb_x = x0 + u;
        const bool ok = u < nu, more = u + 1 < nu;
        wide_deblock_mb<BD, CF>(sh[g], t_alpha, t_beta, t_tc0, t_lc[g], pic, in, ok, u, l,
                                [&]() { if (more) wide_db_fetch<BD, CF>(in, pic, true, mb_x + 1, mb_y, l); });
        MI355_WAVE_SYNC();                       /* the next macroblock's records overwrite what this one's strengths read */
    }
    if (nu > 0) wide_db_store_unit<BD, CF>(sh[g], pic, x0, nu, mb_y, l);
}
#undef DY
#undef DC

/* ------------------------------------------------------------------------- */
/* loop filter of MBAFF frames                                                  */
/* ------------------------------------------------------------------------- */
/* ff_h264_filter_mb for macroblock PAIRS (h264_loopfilter.c:716-847 with its FRAME_MBAFF branches, fill_filter_caches h264_slice.c:2056-2196):
 * lanes 16g..16g+15 filter pair (x, pr) of picture 4k + g — its top macroblock, then its bottom one, in a tile that holds the pair's 32 lines,
 * four columns of the left pair and six lines of the pair above.  A FRAME macroblock's lines are 16 * pos + r of the pair, a FIELD macroblock's
 * pos + 2r.  What differs from a plain frame:
 *   - the left edge between pairs of different coding: eight strengths from coefficient flags alone, two QPs (one per left macroblock), eight lines
 *     per call of the *_mbaff edge filters (:733-806);
 *   - the top edge of a frame macroblock under a field pair: filtered twice, once per field of the pair above, strengths 1 / 2 (3 intra), never the
 *     strong filter (:497-540);
 *   - which macroblock is "above": the same pair's top macroblock, the pair above's bottom macroblock, or for a top field macroblock the pair above's
 *     top field macroblock (fill_filter_caches :2066-2080); frame / field neighbours in the vertical direction give strength 1 without looking at
 *     vectors (:568-571); intra strength 4 only between frame macroblocks or on vertical edges (:551-556); the vertical vector limit is 2 for a field
 *     macroblock (:723).
 * One launch per anti-diagonal of PAIRS (d = x + 2 * pr). */
struct WideMbaffLds {
    mi355_h264_mb m[6];                  /* this pair (top, bottom), the left pair, the pair above */
    int32_t ref[2][25];
    uint32_t mv[2][25];
    uint8_t nnz[25];
    uint8_t bs[2][4][4];
    uint8_t bs8[8];                      /* the left edge between pairs of different coding */
    uint8_t bsd[2][4];                   /* the twice-filtered top edge: [field of the pair above][column / 4] */
    uint16_t y[38 * DYP];                /* lines -6..31 of the pair, columns -4..15 */
    uint16_t c[2][36 * DCPW];            /* lines -4..31 (4:2:0: ..15), columns -4..7 */
};
#define PY(x, R) s.y[((R) + 6) * DYP + (x) + 4]
#define PC(p, x, R) s.c[p][((R) + 4) * DCPW + (x) + 4]

template <int BD, int CF>
__global__ void __launch_bounds__(64)
k_wide_deblock_mbaff(const mi355_h264_frame *frames, int nframes, int d, int max_pr)
{
    typedef Fmt<BD, CF> F;
    typedef typename F::PX PX;
    constexpr int PXB = (int)sizeof(PX), CHP = 2 * F::CH;       /* chroma lines of a pair */
    __shared__ WideMbaffLds sh[4];
    __shared__ uint8_t t_alpha[52], t_beta[52];
    __shared__ __attribute__((aligned(4))) uint8_t t_tc0[52][4];       /* a row is read as one dword (wide_thr) */
    const int lane = lane_id(), g = lane >> 4, l = lane & 15;
    WideMbaffLds &s = sh[g];
    if (lane < 52) { t_alpha[lane] = kw_alpha[lane]; t_beta[lane] = kw_beta[lane]; t_tc0[lane][0] = kw_tc0[lane][0]; t_tc0[lane][1] = kw_tc0[lane][1]; t_tc0[lane][2] = kw_tc0[lane][2]; t_tc0[lane][3] = 0; }
    const int f = 4 * ((int)blockIdx.x / max_pr) + g, pr = (int)blockIdx.x % max_pr, x = d - 2 * pr;
    const mi355_h264_frame &fr = frames[f < nframes ? f : nframes - 1];
    const int W = fr.mb_width;
    const bool ok = f < nframes && 2 * pr + 1 < fr.mb_height && x >= 0 && x < W;
    const bool has_left = x > 0, has_top = pr > 0;
    const int xy0 = 2 * pr * W + x;                              /* the pair's top macroblock */
    const int yd = fr.dst_stride[0], cd = fr.dst_stride[1], ys = fr.recon_stride[0], cs = fr.recon_stride[1];
    if (ok) {
        const int idx[6] = { xy0, xy0 + W, has_left ? xy0 - 1 : xy0, has_left ? xy0 + W - 1 : xy0, has_top ? xy0 - 2 * W : xy0, has_top ? xy0 - W : xy0 };
        for (int k = 0; k < 6; k++) reinterpret_cast<uint32_t *>(&s.m[k])[l] = reinterpret_cast<const uint32_t *>(&fr.mb[idx[k]])[l];
        for (int h2 = 0; h2 < 2; h2++) {
            const int R = l + 16 * h2;
            wide_ld_row<PX, 16>(fr.recon[0] + (size_t)(32 * pr + R) * ys + 16 * x * PXB, &PY(0, R));
            if (has_left) wide_ld_row<PX, 4>(fr.dst[0] + (size_t)(32 * pr + R) * yd + (16 * x - 4) * PXB, &PY(-4, R));
            const int k = l + 16 * h2;                           /* 24 pieces of four samples: the six lines above */
            if (has_top && k < 24) { const int R2 = (k >> 2) - 6, c = 4 * (k & 3); wide_ld_row<PX, 4>(fr.dst[0] + (size_t)(32 * pr + R2) * yd + (16 * x + c) * PXB, &PY(c, R2)); }
        }
        for (int p = 0; p < 2; p++) {
            for (int R = l; R < CHP; R += 16) {
                wide_ld_row<PX, 8>(fr.recon[1 + p] + (size_t)(CHP * pr + R) * cs + 8 * x * PXB, &PC(p, 0, R));
                if (has_left) wide_ld_row<PX, 4>(fr.dst[1 + p] + (size_t)(CHP * pr + R) * cd + (8 * x - 4) * PXB, &PC(p, -4, R));
            }
            if (has_top && l < 8) { const int R2 = (l >> 1) - 4, c = 4 * (l & 1); wide_ld_row<PX, 4>(fr.dst[1 + p] + (size_t)(CHP * pr + R2) * cd + (8 * x + c) * PXB, &PC(p, c, R2)); }
        }
    }
    MI355_WAVE_SYNC();
    /* the line filters of the frame / field kernel in their voteless form: these run inside lane-dependent control flow */
    auto luma_line = [&](uint16_t *q, int st, int bs, int qp, int a_off, int b_off) {
        const WideThr t = wide_thr<BD>(t_alpha, t_beta, t_tc0, qp, a_off, b_off);
        /* (the fourth sample of a side only where bS 4 reads it: at the twice-filtered top edge it would lie outside the tile) */
        int p3 = bs == 4 ? q[-4 * st] : 0, p2 = q[-3 * st], p1 = q[-2 * st], p0 = q[-st], q0 = q[0], q1 = q[st], q2 = q[2 * st], q3 = bs == 4 ? q[3 * st] : 0;
        const int r = wide_luma_line<F::MAXV, true, false>(p3, p2, p1, p0, q0, q1, q2, q3, bs, t.alpha, t.beta, wide_tc0<BD>(t, bs));
        q[-2 * st] = (uint16_t)p1; q[-st] = (uint16_t)p0; q[0] = (uint16_t)q0; q[st] = (uint16_t)q1;
        if (r == 2) { q[-3 * st] = (uint16_t)p2; q[2 * st] = (uint16_t)q2; }
    };
    auto chroma_line = [&](uint16_t *q, int st, int bs, int qp, int a_off, int b_off) {
        const WideThr t = wide_thr<BD>(t_alpha, t_beta, t_tc0, qp, a_off, b_off);
        int p1 = q[-2 * st], p0 = q[-st], q0 = q[0], q1 = q[st];
        wide_chroma_line<F::MAXV, false>(p1, p0, q0, q1, bs, t.alpha, t.beta, wide_tc0<BD>(t, bs) + 1);
        q[-st] = (uint16_t)p0; q[0] = (uint16_t)q0;
    };
    for (int pos = 0; pos < 2; pos++) {
        const mi355_h264_mb &m = s.m[pos];
        const bool filter = ok && !(m.flags & MI355_MBF_NO_DEBLOCK);
        const uint32_t t = m.mb_type;
        const bool cur_field = (t & 0x80u) != 0;
        const int R0 = cur_field ? pos : 16 * pos, step = cur_field ? 2 : 1;           /* luma (and 4:2:2 chroma) lines of this macroblock in the tile */
        const int C0 = cur_field ? pos : F::CH * pos;
        const int mb_xy = xy0 + pos * W;
        const bool own_slice = (m.flags & MI355_MBF_FILTER_OWN_SLICE) != 0;
        /* the neighbours: fill_filter_caches */
        const bool left_ok = has_left && (!own_slice || s.m[2].slice_id == m.slice_id);
        const bool left_field = (s.m[2].mb_type & 0x80u) != 0;
        const bool left_mixed = left_ok && left_field != cur_field;
        /* above: 0 = the same pair's top macroblock, 4 / 5 = the pair above's top / bottom macroblock */
        int top_k = -1;
        if (!cur_field && pos == 1) top_k = 0;
        else if (has_top) top_k = (cur_field && pos == 0 && (s.m[4].mb_type & 0x80u)) ? 4 : 5;
        if (top_k >= 4 && own_slice && s.m[top_k].slice_id != m.slice_id) top_k = -1;
        const bool top_double = top_k == 5 && !cur_field && pos == 0 && (s.m[5].mb_type & 0x80u);       /* a frame macroblock under a field pair */
        const int top_xy = top_k == 0 ? xy0 : (top_k == 4 ? xy0 - 2 * W : xy0 - W);
        int list_count = 1;
        const int ylim = cur_field ? 2 : 4;
        if (filter) {
            list_count = fr.slices[m.slice_id].list_count;
            for (int k = 0; k < 2; k++) {
                if (k && l >= 8) break;
                int which, nxy, x4, y4, cx, cy;
                if (!k) { which = pos; nxy = mb_xy; x4 = l & 3; y4 = l >> 2; cx = x4; cy = y4; }
                else if (l < 4) { which = 2 + pos; nxy = mb_xy - 1; x4 = 3; y4 = l; cx = -1; cy = y4; }
                else { which = top_k < 0 ? pos : top_k; nxy = top_k < 0 ? mb_xy : top_xy; x4 = l - 4; y4 = 3; cx = x4; cy = -1; }
                if (k && l < 4 && !has_left) { which = pos; nxy = mb_xy; }
                const int ci = (cy + 1) * 5 + cx + 1;
                for (int list = 0; list < 2; list++) {
                    s.ref[list][ci] = wide_ref_identity(s.m[which], list, x4, y4);
                    s.mv[list][ci] = fr.mv[list] ? reinterpret_cast<const uint32_t *>(fr.mv[list])[(size_t)nxy * 16 + x4 + 4 * y4] : 0u;
                }
                s.nnz[ci] = (uint8_t)((s.m[which].nnz_mask >> blk_index(x4, y4)) & 1);
            }
        }
        MI355_WAVE_SYNC();
        if (filter) {
            const int edge = l >> 2, i = l & 3, tk = (t >> 3) & 7;
            for (int dir = 0; dir < 2; dir++) {
                const int mask_edge = dir == 0 ? (tk == 0 ? 0 : (tk < 4 ? 3 : 1)) : (tk == 0 ? 0 : (tk == 1 ? 3 : (tk < 4 ? 1 : 3)));
                const int edges = (mask_edge == 3 && !(m.cbp & 15)) ? 1 : 4;
                const uint32_t par_types = MI355_MB_16x16 | (MI355_MB_8x16 >> dir);
                const bool mask_par0 = (t & par_types) != 0;
                const int bx = dir == 0 ? edge : i, by = dir == 0 ? i : edge;
                const int b = (by + 1) * 5 + bx + 1, bn = b - (dir ? 5 : 1);
                int bs = 0;
                if (edge == 0) {
                    const bool avail = dir == 0 ? (left_ok && !left_mixed) : (top_k >= 0 && !top_double);     /* the mixed left edge and the twice-filtered top edge: below */
                    if (avail) {
                        const mi355_h264_mb &mm = dir == 0 ? s.m[2 + pos] : s.m[top_k];
                        const uint32_t tm = mm.mb_type;
                        if ((t | tm) & MI355_MB_INTRA) bs = (!((t | tm) & 0x80u) || dir == 0) ? 4 : 3;
                        else if (dir == 1 && ((t ^ tm) & 0x80u)) bs = (s.nnz[b] | s.nnz[bn]) ? 2 : 1;                /* frame above field or field above frame: no look at the vectors */
                        else if (s.nnz[b] | s.nnz[bn]) bs = 2;
                        else if (mask_par0 && (tm & par_types)) bs = wide_check_mv(s, 6, 6 - (dir ? 5 : 1), list_count, ylim);
                        else bs = wide_check_mv(s, b, bn, list_count, ylim);
                    }
                } else if (edge < edges) {
                    const bool deblock_edge = !((t & MI355_MB_8x8DCT) && (edge & 1));
                    if (deblock_edge || (CF == 2 && dir == 1)) {
                        if (t & MI355_MB_INTRA) bs = 3;
                        else if (s.nnz[b] | s.nnz[bn]) bs = 2;
                        else if (edge & mask_edge) bs = 0;
                        else if (mask_par0) { const int b0 = dir == 0 ? 5 + edge + 1 : (edge + 1) * 5 + 1; bs = wide_check_mv(s, b0, b0 - (dir ? 5 : 1), list_count, ylim); }
                        else bs = wide_check_mv(s, b, bn, list_count, ylim);
                    }
                }
                s.bs[dir][edge][i] = (uint8_t)bs;
            }
            if (l < 8) {
                /* the left edge between pairs of different coding, :748-770: strength i pairs this macroblock's block row i >> 1 with a block of one of the
                 * left pair's macroblocks */
                int bs = 0;
                if (left_mixed) {
                    const int j = cur_field ? l >> 2 : l & 1;                                     /* which macroblock of the left pair */
                    const int lrow = cur_field ? (l & 3) : 2 * pos + (l >> 2);                    /* ... and which of its block rows (column 3) */
                    const mi355_h264_mb &mn = s.m[2 + j];
                    if ((t | mn.mb_type) & MI355_MB_INTRA) bs = 4;
                    else bs = 1 + (int)((((m.nnz_mask >> blk_index(0, l >> 1)) | (mn.nnz_mask >> blk_index(3, lrow))) & 1) != 0);
                }
                s.bs8[l] = (uint8_t)bs;
                /* the top edge of a frame macroblock under a field pair, :497-540: [field j of the pair above][column / 4] */
                int bd2 = 0;
                if (top_double) {
                    const int j = l >> 2, col = l & 3;
                    const mi355_h264_mb &mn = s.m[4 + j];
                    if ((t | mn.mb_type) & MI355_MB_INTRA) bd2 = 3;
                    else bd2 = 1 + (int)((((m.nnz_mask >> blk_index(col, 0)) | (mn.nnz_mask >> blk_index(col, 3))) & 1) != 0);
                }
                s.bsd[l >> 2][l & 3] = (uint8_t)bd2;
            }
        }
        MI355_WAVE_SYNC();
        {
            const int a_off = m.slice_alpha_c0_offset, b_off = m.slice_beta_offset;
            const bool dct8 = (t & MI355_MB_8x8DCT) != 0;
            for (int dir = 0; dir < 2; dir++)
                for (int edge = 0; edge < 4; edge++) {
                    if (filter && dir == 0 && edge == 0 && left_mixed) {
                        /* eight strengths, two QPs: luma lines l = 0..15 of this macroblock */
                        {
                            const int call = cur_field ? l >> 3 : l & 1, k = cur_field ? l & 7 : l >> 1;
                            const int bs = s.bs8[cur_field ? 4 * call + (k >> 1) : call + 2 * (k >> 1)];
                            if (bs) luma_line(&PY(0, R0 + step * l), 1, bs, (m.qp + s.m[2 + call].qp + 1) >> 1, a_off, b_off);
                        }
                        for (int p = 0; p < 2; p++) {
                            if (l < F::CH) {
                                const int call = cur_field ? l / (F::CH / 2) : l & 1, k = cur_field ? l % (F::CH / 2) : l >> 1;
                                const int within = CF == 2 ? k >> 1 : k;                          /* 4:2:2: eight lines per call, two per strength */
                                const int bs = s.bs8[cur_field ? 4 * call + within : call + 2 * within];
                                if (bs) chroma_line(&PC(p, 0, C0 + step * l), 1, bs, (m.qpc[p] + s.m[2 + call].qpc[p] + 1) >> 1, a_off, b_off);
                            }
                        }
                    } else if (filter && dir == 1 && edge == 0 && top_double) {
                        /* below, with a rendezvous between the two fields */
                    } else if (filter) {
                        const mi355_h264_mb &mm = dir == 0 ? s.m[2 + pos] : s.m[top_k < 0 ? pos : top_k];
                        {
                            const bool luma_on = edge == 0 || !(dct8 && (edge & 1));
                            const int bs = s.bs[dir][edge][l >> 2];
                            if (luma_on && bs) {
                                const int qp = edge == 0 ? (m.qp + mm.qp + 1) >> 1 : m.qp;
                                if (dir == 0) luma_line(&PY(4 * edge, R0 + step * l), 1, bs, qp, a_off, b_off);
                                else luma_line(&PY(l, R0 + step * 4 * edge), step * DYP, bs, qp, a_off, b_off);
                            }
                        }
                        if (dir == 0 ? !(edge & 1) : (CF == 2 || !(edge & 1))) {
                            if (dir == 0) {
                                for (int p = 0; p < 2; p++)
                                    if (l < F::CH) {
                                        const int bs = s.bs[0][edge][CF == 2 ? l >> 2 : l >> 1];
                                        if (bs) chroma_line(&PC(p, 2 * edge, C0 + step * l), 1, bs, edge == 0 ? (m.qpc[p] + mm.qpc[p] + 1) >> 1 : m.qpc[p], a_off, b_off);
                                    }
                            } else if (l < 8) {
                                const int bs = s.bs[1][edge][l >> 1];
                                for (int p = 0; p < 2; p++)
                                    if (bs) chroma_line(&PC(p, l, C0 + step * (CF == 2 ? 4 * edge : 2 * edge)), step * DCPW, bs, edge == 0 ? (m.qpc[p] + mm.qpc[p] + 1) >> 1 : m.qpc[p], a_off, b_off);
                            }
                        }
                    }
                    if (dir == 1 && edge == 0)
                        for (int j = 0; j < 2; j++) {      /* the twice-filtered top edge: the top field's lines, then the bottom field's */
                            if (filter && top_double) {
                                const mi355_h264_mb &mn = s.m[4 + j];
                                { const int bs = s.bsd[j][l >> 2]; if (bs) luma_line(&PY(l, j), 2 * DYP, bs, (m.qp + mn.qp + 1) >> 1, a_off, b_off); }
                                if (l < 8) { const int bs = s.bsd[j][l >> 1]; for (int p = 0; p < 2; p++) if (bs) chroma_line(&PC(p, l, j), 2 * DCPW, bs, (m.qpc[p] + mn.qpc[p] + 1) >> 1, a_off, b_off); }
                            }
                            MI355_WAVE_SYNC();
                        }
                    MI355_WAVE_SYNC();
                }
        }
    }
    if (ok) {
        for (int h2 = 0; h2 < 2; h2++) {
            const int R = l + 16 * h2;
            wide_st_row<PX, 16>(fr.dst[0] + (size_t)(32 * pr + R) * yd + 16 * x * PXB, &PY(0, R));
            if (has_left) wide_st_row<PX, 4>(fr.dst[0] + (size_t)(32 * pr + R) * yd + (16 * x - 4) * PXB, &PY(-4, R));
        }
        if (has_top) { const int R2 = (l >> 2) - 4, c = 4 * (l & 3); wide_st_row<PX, 4>(fr.dst[0] + (size_t)(32 * pr + R2) * yd + (16 * x + c) * PXB, &PY(c, R2)); }
        for (int p = 0; p < 2; p++) {
            for (int R = l; R < CHP; R += 16) {
                wide_st_row<PX, 8>(fr.dst[1 + p] + (size_t)(CHP * pr + R) * cd + 8 * x * PXB, &PC(p, 0, R));
                if (has_left) wide_st_row<PX, 4>(fr.dst[1 + p] + (size_t)(CHP * pr + R) * cd + (8 * x - 4) * PXB, &PC(p, -4, R));
            }
            if (has_top && l < 4) { const int R2 = (l >> 1) - 2, c = 4 * (l & 1); wide_st_row<PX, 4>(fr.dst[1 + p] + (size_t)(CHP * pr + R2) * cd + (8 * x + c) * PXB, &PC(p, c, R2)); }
        }
    }
}
#undef PY
#undef PC

template <int BD, int CF>
int wide_launch(const mi355_h264_frame *d_frames, int nframes, int mw, int mh, int max_intra_level, const int32_t *level_widths, int passes, hipStream_t st)
{
    if (passes & 1) {
        if ((long long)nframes * mw * mh > 0x7FFFFFFFLL) return -3;
        hipLaunchKernelGGL((k_wide_inter<BD, CF>), dim3((unsigned)(nframes * mw * mh)), dim3(64), 0, st, d_frames, mw, mh);
    }
    if (passes & 2)
        for (int level = 1; level <= max_intra_level; level++) {
            const int width = level_widths[level - 1];
            if (width <= 0) continue;
            if ((long long)nframes * width > 0x7FFFFFFFLL) return -3;
            hipLaunchKernelGGL((k_wide_intra<BD, CF>), dim3((unsigned)(nframes * width)), dim3(64), 0, st, d_frames, level, width);
        }
    if ((passes & 4) && (passes & 8)) {          /* a batch of MBAFF frames: pairs */
        const unsigned nquads = (unsigned)((nframes + 3) / 4);
        const int npr = mh / 2;
        for (int d = 0; d <= (mw - 1) + 2 * (npr - 1); d++)
            hipLaunchKernelGGL((k_wide_deblock_mbaff<BD, CF>), dim3(nquads * (unsigned)npr), dim3(64), 0, st, d_frames, nframes, d, npr);
    } else
    if (passes & 4) {
        const unsigned nquads = (unsigned)((nframes + 3) / 4);
        {
            /* macroblocks per group and launch: 1 for small batches (a launch is as long as its longest wave: ~3.8 us a macroblock, ~3.7 us a
             * launch), 4 once the launches fill the device (the fabric traffic decides); measured on 1080p 10-bit batches of 32 .. 2048 pictures
             * (profiles/r04_experiments.md, section 9).  MI355_WIDE_UNIT overrides */
            const char *ue = getenv("MI355_WIDE_UNIT");
            int unit = ue ? atoi(ue) : (nframes >= 192 ? 4 : (nframes >= 64 ? 2 : 1));      /* profiles/r06o_wide_unit_sweep.txt: 256 pictures 4.42 (4) against 5.18 ms (2), 128 pictures 3.67 against 3.28 */
            if (unit < 1 || unit > WIDE_UNIT) unit = unit > WIDE_UNIT ? WIDE_UNIT : 1;
            const int uw = (mw + unit - 1) / unit;
            for (int d = 0; d <= (uw - 1) + 2 * (mh - 1); d++) {
                const int y_first = d - (uw - 1) > 0 ? (d - (uw - 1) + 1) / 2 : 0, y_last = d / 2 < mh - 1 ? d / 2 : mh - 1;     /* 0 <= d - 2 y <= uw - 1 */
                if (y_last < y_first) continue;              /* a picture one unit wide has nothing on the odd anti-diagonals */
                hipLaunchKernelGGL((k_wide_deblock<BD, CF>), dim3(nquads * (unsigned)(y_last - y_first + 1)), dim3(64), 0, st, d_frames, nframes, d, y_first,
                                   y_last - y_first + 1, unit);
            }
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" int mi355_h264_decode_frames_wide_dev(const mi355_h264_frame *d_frames, int nframes, int max_mb_width, int max_mb_height,
                                                 int max_intra_level, const int32_t *level_widths, int bit_depth, int chroma_format_idc,
                                                 int passes, void *stream)
{
    if (!mi355::bind() || !d_frames || nframes <= 0 || max_mb_width <= 0 || max_mb_height <= 0 || (max_intra_level > 0 && !level_widths)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const int key = bit_depth * 10 + chroma_format_idc;
    switch (key) {
    case 81:  return wide_launch<8, 1>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);   /* what the 8-bit kernels decode too: this set against that one (tests) */
    case 91:  return wide_launch<9, 1>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);
    case 101: return wide_launch<10, 1>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);
    case 82:  return wide_launch<8, 2>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);
    case 92:  return wide_launch<9, 2>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);
    case 102: return wide_launch<10, 2>(d_frames, nframes, max_mb_width, max_mb_height, max_intra_level, level_widths, passes, st);
    default:  return -1;
    }
}
