/*
 * hevc_dev.h — wave-cooperative HEVC building blocks, bit depth as a
 * runtime parameter (8, 9, 10): samples are uint8_t or uint16_t, coefficients int16_t.
 * Same execution model as h264_dev.h: one 64-lane wavefront per workgroup, LDS private to the wave.
 * Reference: libavcodec/hevcdsp_template.c, hevcpred_template.c (lines quoted per function).
 */
#ifndef MI355_HEVC_DEV_H
#define MI355_HEVC_DEV_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include "h264_dev.h"   /* clip3, iabs, lane_id */

/* Which rendezvous the wave-level building blocks below use between their LDS phases.  Default: __syncthreads() — the kernels of
 * hevc_batch.hip / hevc_tier1.hip are one wavefront per workgroup.  hevc_ctb.hip (several wavefronts per workgroup, each on its own
 * job and its own scratch) includes this file with MI355_HEVC_SYNC() = MI355_WAVE_SYNC() and MI355_HEVC_NS = a namespace of its own,
 * so that the two instantiations of these inline functions are different entities (the emulator build links them into one object). */
#ifndef MI355_HEVC_NS
#define MI355_HEVC_NS mi355
#endif
#ifndef MI355_HEVC_SYNC
#define MI355_HEVC_SYNC() __syncthreads()
#endif

namespace MI355_HEVC_NS {
using namespace mi355;

__device__ __forceinline__ int clip_i16(int v) { return clip3(v, -32768, 32767); }
__device__ __forceinline__ int clip_px(int v, int bd) { return clip3(v, 0, (1 << bd) - 1); }
/* sample i of a plane whose element size depends on the bit depth */
__device__ __forceinline__ int ldpx(const uint8_t *p, int i, int bd) { return bd > 8 ? reinterpret_cast<const uint16_t *>(p)[i] : p[i]; }
__device__ __forceinline__ void stpx(uint8_t *p, int i, int v, int bd)
{
    if (bd > 8) reinterpret_cast<uint16_t *>(p)[i] = (uint16_t)v; else p[i] = (uint8_t)v;
}

/* ---- inverse DCT matrix: transMatrix[k][n] = c(k) cos((2n+1) k pi/64) in the standard's integers,
 * generated from the 33 magnitudes of angle index a = (2n+1)k mod 128 — compile-time constants, so the
 * unrolled transforms below multiply by literals ----------------------------------------------- */
__host__ __device__ constexpr int dct_mag(int a)
{
    constexpr int8_t mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
    return mag[a];
}
__host__ __device__ constexpr int dct_coef(int k, int n)
{
    if (!k) return 64;
    const int a = ((2 * n + 1) * k) & 127;
    return a <= 32 ? dct_mag(a) : (a <= 64 ? -dct_mag(64 - a) : (a <= 96 ? -dct_mag(a - 64) : dct_mag(128 - a)));
}
/* Two 16-bit x 16-bit products and an accumulate in one instruction (v_dot2_i32_i16): a dword holds two
 * neighbouring samples, the other operand two neighbouring taps. */
#ifdef MI355_HIP_EMU_H
static inline int mi355_dot2(uint32_t a, uint32_t b, int c)
{
    return c + (int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int16_t)(a >> 16) * (int16_t)(b >> 16);
}
static inline uint32_t mi355_alignbit16(uint32_t hi, uint32_t lo) { return (lo >> 16) | (hi << 16); }
#else
typedef short mi355_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mi355_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(mi355_short2, a), __builtin_bit_cast(mi355_short2, b), c, false);
}
__device__ __forceinline__ uint32_t mi355_alignbit16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#endif

#ifdef MI355_HIP_EMU_H
static inline uint32_t mi355_pair_lo(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }
static inline uint32_t mi355_pair_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xFFFF0000u); }
static inline uint32_t mi355_widen_lo(uint32_t w) { return (w & 0xFFu) | ((w & 0xFF00u) << 8); }
static inline uint32_t mi355_widen_hi(uint32_t w) { return ((w >> 16) & 0xFFu) | ((w >> 8) & 0xFF0000u); }
#else
/* v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0C = zero */
__device__ __forceinline__ uint32_t mi355_pair_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t mi355_pair_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ uint32_t mi355_widen_lo(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0C010C00u); }
__device__ __forceinline__ uint32_t mi355_widen_hi(uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0C030C02u); }
#endif

/* rows of the input a 1-D pass of size H looks at when pruned to `end` (hevcdsp_template.c:140-206) */
__device__ __forceinline__ bool dct_row_used(int H, int j, int end)
{
    if (H == 4) return true;
    if (j & 1) return j < end;
    if (H == 32 && ((j >> 1) & 1)) return (j >> 1) < (end >> 1);
    return true;
}

/* One H-point inverse transform in registers, as the even/odd decomposition of the reference's TR_n
 * macros (hevcdsp_template.c:140-206): out[n] = E[n] + O[n], out[H-1-n] = E[n] - O[n], where O takes the
 * odd inputs through an (H/2 x H/2) block of the matrix and E is the H/2-point transform of the even
 * inputs.  Sums are exact integers, so the grouping does not change the result.  `x` holds the H inputs
 * of THIS size's transform (row j of size H = row j * 32/H of the 32-point matrix); S = 32 / H. */
/* END < H: the pruned form of the reference's TR_32 / TR_16 (`end` argument): the odd-part sums of the 32- and 16-point
 * levels stop at input END; the 8-point transform of the rows that feed it is always complete, exactly as there. */
template <int H, int S, int END = H> struct Idct1D {
    static __device__ __forceinline__ void run(const int *x, int *out)
    {
        int xe[H / 2], E[H / 2], O[H / 2];
#pragma unroll
        for (int k = 0; k < H / 2; k++) xe[k] = x[2 * k];
        Idct1D<H / 2, 2 * S, (H / 2 <= 8 ? H / 2 : (END + 1) / 2)>::run(xe, E);
        /* the odd part two inputs at a time (v_dot2_i32_i16: the inputs are int16 in both passes, the matrix entries fit a byte):
         * (x[4m+1], x[4m+3]) against the matrix pair of output n.  H = 4 has one such pair, larger sizes H / 4.  END is a
         * multiple of 4 wherever it prunes (16 or 8 of 32, 8 of 16): whole pairs drop out. */
        static_assert(H == 4 || END % 4 == 0, "pruning by whole input pairs");
        uint32_t xo[H / 4];
#pragma unroll
        for (int m = 0; m < H / 4; m++) xo[m] = mi355_pair_lo((uint32_t)x[4 * m + 1], (uint32_t)x[4 * m + 3]);
#pragma unroll
        for (int n = 0; n < H / 2; n++) {
            int o = 0;
#pragma unroll
            for (int m = 0; m < H / 4; m++)
                if (4 * m + 1 < END) {
                    const uint32_t kk = ((uint32_t)dct_coef((4 * m + 1) * S, n) & 0xFFFFu) |
                                        ((uint32_t)(4 * m + 3 < END ? dct_coef((4 * m + 3) * S, n) : 0) << 16);
                    o = mi355_dot2(xo[m], kk, o);
                }
            O[n] = o;
        }
#pragma unroll
        for (int n = 0; n < H / 2; n++) { out[n] = E[n] + O[n]; out[H - 1 - n] = E[n] - O[n]; }
    }
};
template <int S, int END> struct Idct1D<2, S, END> {
    static __device__ __forceinline__ void run(const int *x, int *out)
    {
        out[0] = 64 * x[0] + dct_coef(S, 0) * x[1];
        out[1] = 64 * x[0] + dct_coef(S, 1) * x[1];
    }
};

struct __attribute__((aligned(16))) IdctScratch {
    int16_t c[2][32 * 32];      /* one block per half-wave (filled and read back 16 / 8 bytes per lane) */
};

/* In-place 2-D inverse DCT of c (HxH, row-major) by ONE HALF-WAVE (32 lanes; `hl` = lane within the half,
 * both halves of a wave call this together, each on its own block) with the reference's col_limit
 * semantics (:208-236): first pass down the columns with limit2 shrinking every 4 columns, clip to int16
 * after (x+64)>>7; second pass along the rows, (x + (1<<(19-bd))) >> (20-bd). */
template <int H>
__device__ inline void hevc_idct_half(int16_t *c, int hl, bool active, int col_limit, int bd)
{
    const int limit = col_limit < H ? col_limit : H;
    const int l0 = col_limit + 4 < H ? col_limit + 4 : H;
    int in[H], out[H];
    if (active && hl < H) {
        const int i = hl;
        const int end = l0 < H ? l0 - 4 * (i > 0 ? (i - 1) >> 2 : 0) : H;
        /* l0 (>= every lane's `end`) in the lower half: the instantiation without the upper half's inputs and products */
        if (H >= 16 && l0 <= H / 2) {
#pragma unroll
            for (int j = 0; j < H; j++) in[j] = (j < H / 2 || (j * (32 / H)) % 4 == 0) && dct_row_used(H, j, end) ? c[i + H * j] : 0;
            Idct1D<H, 32 / H, (H >= 16 ? H / 2 : H)>::run(in, out);
        } else {
#pragma unroll
            for (int j = 0; j < H; j++) in[j] = dct_row_used(H, j, end) ? c[i + H * j] : 0;
            Idct1D<H, 32 / H>::run(in, out);
        }
#pragma unroll
        for (int n = 0; n < H; n++) c[i + H * n] = (int16_t)clip_i16((out[n] + 64) >> 7);
    }
    MI355_HEVC_SYNC();
    if (active && hl < H) {
        const int i = hl, shift = 20 - bd, add = 1 << (shift - 1);
        if (H >= 16 && limit <= H / 2) {
#pragma unroll
            for (int j = 0; j < H; j++) in[j] = (j < H / 2 || (j * (32 / H)) % 4 == 0) && dct_row_used(H, j, limit) ? c[H * i + j] : 0;
            Idct1D<H, 32 / H, (H >= 16 ? H / 2 : H)>::run(in, out);
        } else {
#pragma unroll
            for (int j = 0; j < H; j++) in[j] = dct_row_used(H, j, limit) ? c[H * i + j] : 0;
            Idct1D<H, 32 / H>::run(in, out);
        }
#pragma unroll
        for (int n = 0; n < H; n++) c[H * i + n] = (int16_t)clip_i16((out[n] + add) >> shift);
    }
    MI355_HEVC_SYNC();
}

/* 4x4 DST-VII for intra luma (:103-136), lanes 0..3 */
__device__ __forceinline__ void dst4_1d(const int in[4], int out[4])
{
    const int c0 = in[0] + in[2], c1 = in[2] + in[3], c2 = in[0] - in[3], c3 = 74 * in[1];
    out[0] = 29 * c0 + 55 * c1 + c3;
    out[1] = 55 * c2 - 29 * c1 + c3;
    out[2] = 74 * (in[0] - in[2] + in[3]);
    out[3] = 55 * c0 + 29 * c2 - c3;
}
__device__ inline void hevc_dst4_wave(int16_t *c, int bd, int lane, bool active = true)
{
    int in[4], out[4];
    if (active && lane < 4) {
        for (int k = 0; k < 4; k++) in[k] = c[lane + 4 * k];
        dst4_1d(in, out);
        for (int k = 0; k < 4; k++) c[lane + 4 * k] = (int16_t)clip_i16((out[k] + 64) >> 7);
    }
    MI355_HEVC_SYNC();
    if (active && lane < 4) {
        const int shift = 20 - bd, add = 1 << (shift - 1);
        for (int k = 0; k < 4; k++) in[k] = c[4 * lane + k];
        dst4_1d(in, out);
        for (int k = 0; k < 4; k++) c[4 * lane + k] = (int16_t)clip_i16((out[k] + add) >> shift);
    }
    MI355_HEVC_SYNC();
}

/* ---- a14: luma 8-tap / chroma 4-tap MC to the 14-bit intermediate ----------------------------- */
__device__ const int8_t k_qpel[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                         { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__device__ const int8_t k_epel[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                         { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

/* Four consecutive outputs of a 4- or 8-tap FIR along a line of 16-bit values: d[0..5] = the line from the
 * first output's first tap (dword aligned), t[k] = taps 2k, 2k+1 packed.  Outputs 0 and 2 use the dwords as
 * they are, 1 and 3 the same dwords shifted by one value. */
template <int TAPS>
__device__ __forceinline__ void fir4(const uint32_t *d, const uint32_t *t, int *out)
{
    uint32_t e[5];
#pragma unroll
    for (int k = 0; k < TAPS / 2 + 1; k++) e[k] = mi355_alignbit16(d[k + 1], d[k]);
    out[0] = out[1] = out[2] = out[3] = 0;
#pragma unroll
    for (int k = 0; k < TAPS / 2; k++) {
        out[0] = mi355_dot2(d[k], t[k], out[0]);
        out[1] = mi355_dot2(e[k], t[k], out[1]);
        out[2] = mi355_dot2(d[k + 1], t[k], out[2]);
        out[3] = mi355_dot2(e[k + 1], t[k], out[3]);
    }
}

/* src points at sample (0,0) of the block inside a plane/window with `ss` samples per row; dst is
 * int16 with `ds` elements per row.  taps = 8 (qpel, :729-937) or 4 (epel, :939-1089).  A block is worked
 * off in tiles of at most 32x32 outputs.  The samples the taps of a tile touch — and only those: a filter
 * that is not applied reads nothing beyond the block, exactly like the reference's per-direction functions —
 * are staged once in LDS as 16-bit values, eight bytes per lane and load.  Horizontal pass: a lane produces
 * four consecutive outputs of a row with dot-product instructions on (sample, sample+1) dwords.  Vertical
 * pass: a lane owns two neighbouring columns and eight rows; it builds the (row, row+1) dwords of each
 * column with one byte-permute per pair and feeds the same dot products, so both passes work on the
 * row-major layout and every result leaves as a dword (two int16) or wider. */
constexpr int HEVC_MC_TILE = 32;
constexpr int HEVC_MC_PITCH = 44;      /* values per LDS row: >= 32 + 7 + the 4 values a segment may read past its last tap */
constexpr int HEVC_MC_TPITCH = 36;     /* ... of the first pass's results: 32 columns (+ 4: rows start 8 bytes apart, 18 dwords: no two of the vertical
                                          pass's sixteen column pairs x groups of rows share a bank pattern) — 6.4 KB of scratch per wave instead of 7 */
#ifndef MI355_HEVC_MC_TILE_H
#define MI355_HEVC_MC_TILE_H 32
#endif
constexpr int HEVC_MC_TILE_H = MI355_HEVC_MC_TILE_H;      /* rows of a tile (32 wide) */
constexpr int HEVC_MC_ROWS = HEVC_MC_TILE_H + 8;
/* both chroma planes of a block in one tile pass (k_hevc_mcpred_batch, chroma == 2): tiles of at most HEVC_MC_PAIR_TILE_H rows, the second
 * plane's window and first-pass rows HEVC_MC_PAIR_ROW rows below the first's */
constexpr int HEVC_MC_PAIR_ROW = HEVC_MC_ROWS / 2, HEVC_MC_PAIR_TILE_H = HEVC_MC_PAIR_ROW - 4;
static_assert(HEVC_MC_PAIR_TILE_H + 3 <= HEVC_MC_PAIR_ROW && 2 * HEVC_MC_PAIR_ROW <= HEVC_MC_ROWS, "two 4-tap windows in the scratch");
struct __attribute__((aligned(16))) HevcMcScratch {
    uint16_t win[HEVC_MC_ROWS * HEVC_MC_PITCH];      /* staged samples */
    int16_t tmp[HEVC_MC_ROWS * HEVC_MC_TPITCH];      /* first-pass results of the 2-D case */
};
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
/* alignment class of the int16 destination: 8, 4 or 2 bytes for every row start */
__device__ __forceinline__ int hevc_mc_align(const int16_t *dst, int ds)
{
    const unsigned a = (unsigned)(uintptr_t)dst | (unsigned)(ds * 2);
    return (a & 7) == 0 ? 8 : ((a & 3) == 0 ? 4 : 2);
}
__device__ __forceinline__ void hevc_mc_st2(int16_t *p, uint32_t v, int amode)
{
    if (amode >= 4) *reinterpret_cast<uint32_t *>(p) = v;
    else { p[0] = (int16_t)(v & 0xFFFF); p[1] = (int16_t)(v >> 16); }
}
/* four results (n of them inside the block) */
__device__ __forceinline__ void hevc_mc_st4(int16_t *p, uint32_t lo, uint32_t hi, int n, int amode)
{
    if (n >= 4) {
        if (amode == 8) *reinterpret_cast<uint2 *>(p) = make_uint2(lo, hi);
        else { hevc_mc_st2(p, lo, amode); hevc_mc_st2(p + 2, hi, amode); }
    } else {
        if (n >= 2) hevc_mc_st2(p, lo, amode); else if (n == 1) p[0] = (int16_t)(lo & 0xFFFF);
        if (n == 3) p[2] = (int16_t)(hi & 0xFFFF);
    }
}
/* one round of hevc_mc_stage: U pieces of eight bytes per lane, all U loads issued before the first result is touched (a lane
 * past the end repeats the last piece and drops it) — one memory round trip per 64 U pieces */
/* PAIR: rows 0 .. rows_a - 1 come from w0 and land in LDS rows 0 .., rows rows_a .. from w0b and land HEVC_MC_PAIR_ROW rows further down
 * (the windows of both chroma planes of a block in one set of loads) */
template <int U, bool PAIR = false>
__device__ __forceinline__ void hevc_mc_stage_round(HevcMcScratch &s, const uint8_t *w0, ptrdiff_t sb, int base, int n, int K, int inv, int rowbytes, int bd, int lane,
                                                    const uint8_t *w0b = nullptr, int rows_a = 0)
{
    uint64_t v[U];
    int rr[U], kk[U], sh[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int i = base + 64 * u + lane, ic = i < n ? i : n - 1;
        int r = mi355_div20(ic, inv);
        const int k = ic - r * K;
        /* the last piece of a row is fetched so that it ends with the row, then shifted into place */
        const int want = 8 * k, start = want < rowbytes - 8 ? want : rowbytes - 8;
        const uint8_t *rowp = w0 + (ptrdiff_t)r * sb;
        if (PAIR && r >= rows_a) { rowp = w0b + (ptrdiff_t)(r - rows_a) * sb; r += HEVC_MC_PAIR_ROW - rows_a; }
        __builtin_memcpy(&v[u], rowp + start, 8);
        rr[u] = i < n ? r : -1; kk[u] = k; sh[u] = 8 * (want - start);
    }
    MI355_ISSUE_FENCE();
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (rr[u] < 0) continue;
        const uint64_t w64 = v[u] >> sh[u];
        const uint32_t lo = (uint32_t)w64, hi = (uint32_t)(w64 >> 32);
        if (bd > 8) {
            *reinterpret_cast<uint2 *>(&s.win[rr[u] * HEVC_MC_PITCH + 4 * kk[u]]) = make_uint2(lo, hi);
        } else {
            uint32_t *w = reinterpret_cast<uint32_t *>(&s.win[rr[u] * HEVC_MC_PITCH + 8 * kk[u]]);
            w[0] = mi355_widen_lo(lo); w[1] = mi355_widen_hi(lo); w[2] = mi355_widen_lo(hi); w[3] = mi355_widen_hi(hi);
        }
    }
}
/* rows x cols samples from `w0` (bytes, `sb` bytes per row) -> s.win as 16-bit values */
__device__ inline void hevc_mc_stage(HevcMcScratch &s, const uint8_t *w0, ptrdiff_t sb, int rows, int cols, int bd)
{
    const int lane = lane_id(), rowbytes = cols * (bd > 8 ? 2 : 1);
    if (rowbytes >= 8) {
        const int K = (rowbytes + 7) >> 3, inv = mi355_inv20(K), n = rows * K;
        /* rounds of four pieces per lane while that many remain, then three, two or one: a 16x16 chroma block's window is 95 pieces,
         * and every piece slot costs its ~20 instructions whether a lane uses it or not */
        int base = 0;
        for (; n - base > 192; base += 256) hevc_mc_stage_round<4>(s, w0, sb, base, n, K, inv, rowbytes, bd, lane);
        if (n - base > 128) hevc_mc_stage_round<3>(s, w0, sb, base, n, K, inv, rowbytes, bd, lane);
        else if (n - base > 64) hevc_mc_stage_round<2>(s, w0, sb, base, n, K, inv, rowbytes, bd, lane);
        else if (n - base > 0) hevc_mc_stage_round<1>(s, w0, sb, base, n, K, inv, rowbytes, bd, lane);
    } else {
        for (int i = lane; i < rows * cols; i += 64) {
            const int r = i / cols, c = i - r * cols;
            s.win[r * HEVC_MC_PITCH + c] = (uint16_t)(bd > 8 ? reinterpret_cast<const uint16_t *>(w0 + (ptrdiff_t)r * sb)[c] : w0[(ptrdiff_t)r * sb + c]);
        }
    }
}
/* the windows of two planes (same geometry) in one go; rows of the second plane HEVC_MC_PAIR_ROW rows below the first's */
__device__ inline void hevc_mc_stage_pair(HevcMcScratch &s, const uint8_t *w0a, const uint8_t *w0b, ptrdiff_t sb, int rows, int cols, int bd)
{
    const int lane = lane_id(), rowbytes = cols * (bd > 8 ? 2 : 1);
    if (rowbytes >= 8) {
        const int K = (rowbytes + 7) >> 3, inv = mi355_inv20(K), n = 2 * rows * K;
        int base = 0;
        for (; n - base > 192; base += 256) hevc_mc_stage_round<4, true>(s, w0a, sb, base, n, K, inv, rowbytes, bd, lane, w0b, rows);
        if (n - base > 128) hevc_mc_stage_round<3, true>(s, w0a, sb, base, n, K, inv, rowbytes, bd, lane, w0b, rows);
        else if (n - base > 64) hevc_mc_stage_round<2, true>(s, w0a, sb, base, n, K, inv, rowbytes, bd, lane, w0b, rows);
        else if (n - base > 0) hevc_mc_stage_round<1, true>(s, w0a, sb, base, n, K, inv, rowbytes, bd, lane, w0b, rows);
    } else {
        for (int i = lane; i < 2 * rows * cols; i += 64) {
            int r = i / cols;
            const int c = i - r * cols;
            const uint8_t *w0 = w0a;
            int lr = r;
            if (r >= rows) { w0 = w0b; r -= rows; lr = r + HEVC_MC_PAIR_ROW; }
            s.win[lr * HEVC_MC_PITCH + c] = (uint16_t)(bd > 8 ? reinterpret_cast<const uint16_t *>(w0 + (ptrdiff_t)r * sb)[c] : w0[(ptrdiff_t)r * sb + c]);
        }
    }
}
/* Where the results of a tile go.  put4: four horizontally adjacent results of row r from column x0 (n of them inside
 * the tile); put2: two from column x.  Tile-relative coordinates. */
struct HevcMcToI16 {          /* the reference's destination: int16, `ds` elements per row */
    int16_t *out; int ds, amode;
    __device__ __forceinline__ void put4(int r, int x0, uint32_t lo, uint32_t hi, int n) const { hevc_mc_st4(out + (ptrdiff_t)r * ds + x0, lo, hi, n, amode); }
    __device__ __forceinline__ void put2(int r, int x, uint32_t v) const { hevc_mc_st2(out + (ptrdiff_t)r * ds + x, v, amode); }
};
constexpr int HEVC_MC_KEEP_PITCH = 32;
/* two-reference predictions (k_hevc_mcpred_batch): tiles of HEVC_MC_BI_TILE_H rows use HEVC_MC_BI_ROWS rows of win / tmp; the kept tile of the
 * first reference sits in tmp behind them */
constexpr int HEVC_MC_BI_TILE_H = HEVC_MC_TILE_H / 2, HEVC_MC_BI_ROWS = HEVC_MC_BI_TILE_H + 7;
static_assert((HEVC_MC_BI_ROWS * HEVC_MC_TPITCH) % 4 == 0 && HEVC_MC_BI_ROWS * HEVC_MC_TPITCH + HEVC_MC_BI_TILE_H * HEVC_MC_KEEP_PITCH <= HEVC_MC_ROWS * HEVC_MC_TPITCH,
              "the kept tile fits behind the rows of a 16-row tile, 8-byte aligned");
struct HevcMcToTile {         /* kept in LDS for a second prediction to combine with */
    int16_t *t;
    __device__ __forceinline__ void put4(int r, int x0, uint32_t lo, uint32_t hi, int n) const
    {
        (void)n;                                             /* the tile has room for whole segments */
        *reinterpret_cast<uint2 *>(&t[r * HEVC_MC_KEEP_PITCH + x0]) = make_uint2(lo, hi);
    }
    __device__ __forceinline__ void put2(int r, int x, uint32_t v) const { *reinterpret_cast<uint32_t *>(&t[r * HEVC_MC_KEEP_PITCH + x]) = v; }
};

/* vertical pass of a tile from row-major 16-bit lines (HEVC_MC_PITCH values apart): a lane owns the column pair c2 and R output
 * rows; it builds the (row, row + 1) dwords of each column with one byte-permute per pair and feeds the dot products */
template <int TAPS, int R, class Sink>
__device__ __forceinline__ void hevc_mc_vpass(const Sink &sink, const uint32_t *lines, int pitch2, const uint32_t *tv, int cp, int th_, int vshift, int lane)
{   /* pitch2: dwords per line (HEVC_MC_PITCH / 2 for staged samples, HEVC_MC_TPITCH / 2 for first-pass results) */
    const int cinv = mi355_inv20(cp), groups = (th_ + R - 1) / R;
    for (int i = lane; i < cp * groups; i += 64) {
        const int q = mi355_div20(i, cinv), c2 = i - q * cp, y0 = R * q;
        const uint32_t *d = lines + y0 * pitch2 + c2;
        int a0[R], a1[R];
#pragma unroll
        for (int y = 0; y < R; y++) a0[y] = a1[y] = 0;
        uint32_t prev = d[0];
#pragma unroll
        for (int r = 0; r < R + TAPS - 2; r++) {
            const uint32_t cur = d[(r + 1) * pitch2];
            const uint32_t p0 = mi355_pair_lo(prev, cur), p1 = mi355_pair_hi(prev, cur);
            prev = cur;
#pragma unroll
            for (int k = 0; k < TAPS / 2; k++) {
                const int y = r - 2 * k;
                if (y >= 0 && y < R) { a0[y] = mi355_dot2(p0, tv[k], a0[y]); a1[y] = mi355_dot2(p1, tv[k], a1[y]); }
            }
        }
#pragma unroll
        for (int y = 0; y < R; y++)
            if (y0 + y < th_) sink.put2(y0 + y, 2 * c2, pack16(a0[y] >> vshift, a1[y] >> vshift));
    }
}

/* one tile of at most 32x32 outputs; w0 = byte address of the first sample the taps touch, sb = bytes per source row */
template <int TAPS, class Sink>
__device__ inline void hevc_mc_tile(const Sink &sink, const uint8_t *w0, ptrdiff_t sb, int tw, int th_, int mx, int my, int bd, HevcMcScratch &s)
{
    const int lane = lane_id();
    constexpr int extra = TAPS - 1;
    const int8_t *fh = TAPS == 8 ? k_qpel[mx] : k_epel[mx], *fv = TAPS == 8 ? k_qpel[my] : k_epel[my];
    uint32_t th[TAPS / 2], tv[TAPS / 2];
#pragma unroll
    for (int k = 0; k < TAPS / 2; k++) {
        th[k] = (uint32_t)(uint16_t)(int16_t)fh[2 * k] | ((uint32_t)(uint16_t)(int16_t)fh[2 * k + 1] << 16);
        tv[k] = (uint32_t)(uint16_t)(int16_t)fv[2 * k] | ((uint32_t)(uint16_t)(int16_t)fv[2 * k + 1] << 16);
    }
    const int rows = th_ + (my ? extra : 0);
    hevc_mc_stage(s, w0, sb, rows, tw + (mx ? extra : 0), bd);
    MI355_HEVC_SYNC();
    const int wseg = (tw + 3) >> 2, winv = mi355_inv20(wseg);
    if (!mx && !my) {
        /* put_hevc_*_pixels: sample << (14 - bd); two values per dword shift together (no carry across) */
        for (int i = lane; i < rows * wseg; i += 64) {
            const int r = mi355_div20(i, winv), x0 = 4 * (i - r * wseg);
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&s.win[r * HEVC_MC_PITCH + x0]);
            sink.put4(r, x0, d[0] << (14 - bd), d[1] << (14 - bd), tw - x0);
        }
    }
    if (mx) {
        /* horizontal pass over `rows` lines, four outputs per lane */
        for (int i = lane; i < rows * wseg; i += 64) {
            const int r = mi355_div20(i, winv), x0 = 4 * (i - r * wseg);
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&s.win[r * HEVC_MC_PITCH + x0]);
            uint32_t dd[6];
#pragma unroll
            for (int k = 0; k < TAPS / 2 + 2; k++) dd[k] = d[k];
            int o[4];
            fir4<TAPS>(dd, th, o);
            const uint32_t lo = pack16(o[0] >> (bd - 8), o[1] >> (bd - 8)), hi = pack16(o[2] >> (bd - 8), o[3] >> (bd - 8));
            if (my) *reinterpret_cast<uint2 *>(&s.tmp[r * HEVC_MC_TPITCH + x0]) = make_uint2(lo, hi);
            else sink.put4(r, x0, lo, hi, tw - x0);
        }
        if (my) MI355_HEVC_SYNC();
    }
    if (my) {
        /* vertical pass: lane = (column pair, R output rows).  R = 8 when that fills the wave (a 32x32 tile: 16 pairs x 4 groups);
         * smaller tiles take 4 or 2 rows per lane — a 16x16 chroma block with R = 8 is 16 busy lanes running the longest
         * instruction stream (R + TAPS - 1 rows in, R out), with R = 2 it is 64 lanes and a third of the instructions */
        const uint32_t *lines = reinterpret_cast<const uint32_t *>(mx ? reinterpret_cast<const uint16_t *>(s.tmp) : s.win);
        const int vshift = mx ? 6 : bd - 8;
        const int cp = tw >> 1;
        const int pitch2 = (mx ? HEVC_MC_TPITCH : HEVC_MC_PITCH) / 2;
        if (cp * ((th_ + 7) >> 3) > 32) hevc_mc_vpass<TAPS, 8>(sink, lines, pitch2, tv, cp, th_, vshift, lane);
        else if (cp * ((th_ + 3) >> 2) > 32) hevc_mc_vpass<TAPS, 4>(sink, lines, pitch2, tv, cp, th_, vshift, lane);
        else hevc_mc_vpass<TAPS, 2>(sink, lines, pitch2, tv, cp, th_, vshift, lane);
    }
    MI355_HEVC_SYNC();
}
/* the same tile of BOTH chroma planes (4-tap filters, one vector): windows staged together, every pass over the rows of both.
 * sink0 / sink1: where plane 0 / plane 1 results go; w0a / w0b: first sample the taps touch in each plane */
template <class Sink>
__device__ inline void hevc_mc_tile_pair(const Sink &sink0, const Sink &sink1, const uint8_t *w0a, const uint8_t *w0b, ptrdiff_t sb, int tw, int th_, int mx, int my, int bd,
                                         HevcMcScratch &s)
{
    constexpr int TAPS = 4, extra = TAPS - 1;
    const int lane = lane_id();
    const int8_t *fh = k_epel[mx], *fv = k_epel[my];
    uint32_t th[TAPS / 2], tv[TAPS / 2];
#pragma unroll
    for (int k = 0; k < TAPS / 2; k++) {
        th[k] = (uint32_t)(uint16_t)(int16_t)fh[2 * k] | ((uint32_t)(uint16_t)(int16_t)fh[2 * k + 1] << 16);
        tv[k] = (uint32_t)(uint16_t)(int16_t)fv[2 * k] | ((uint32_t)(uint16_t)(int16_t)fv[2 * k + 1] << 16);
    }
    const int rows = th_ + (my ? extra : 0);
    hevc_mc_stage_pair(s, w0a, w0b, sb, rows, tw + (mx ? extra : 0), bd);
    MI355_HEVC_SYNC();
    const int wseg = (tw + 3) >> 2, winv = mi355_inv20(wseg);
    /* item i of a row-wise pass: row r of plane pl, LDS row lr */
#define MI355_PAIR_ROW(i) const int r2 = mi355_div20(i, winv), x0 = 4 * (i - r2 * wseg), pl = r2 >= rows, r = r2 - (pl ? rows : 0), lr = r + (pl ? HEVC_MC_PAIR_ROW : 0)
    if (!mx && !my) {
        for (int i = lane; i < 2 * rows * wseg; i += 64) {
            MI355_PAIR_ROW(i);
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&s.win[lr * HEVC_MC_PITCH + x0]);
            const uint32_t lo = d[0] << (14 - bd), hi = d[1] << (14 - bd);
            if (pl) sink1.put4(r, x0, lo, hi, tw - x0); else sink0.put4(r, x0, lo, hi, tw - x0);
        }
    }
    if (mx) {
        for (int i = lane; i < 2 * rows * wseg; i += 64) {
            MI355_PAIR_ROW(i);
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&s.win[lr * HEVC_MC_PITCH + x0]);
            uint32_t dd[6];
#pragma unroll
            for (int k = 0; k < TAPS / 2 + 2; k++) dd[k] = d[k];
            int o[4];
            fir4<TAPS>(dd, th, o);
            const uint32_t lo = pack16(o[0] >> (bd - 8), o[1] >> (bd - 8)), hi = pack16(o[2] >> (bd - 8), o[3] >> (bd - 8));
            if (my) *reinterpret_cast<uint2 *>(&s.tmp[lr * HEVC_MC_TPITCH + x0]) = make_uint2(lo, hi);
            else if (pl) sink1.put4(r, x0, lo, hi, tw - x0);
            else sink0.put4(r, x0, lo, hi, tw - x0);
        }
        if (my) MI355_HEVC_SYNC();
    }
#undef MI355_PAIR_ROW
    if (my) {
        const uint32_t *lines = reinterpret_cast<const uint32_t *>(mx ? reinterpret_cast<const uint16_t *>(s.tmp) : s.win);
        const int vshift = mx ? 6 : bd - 8, pitch2 = (mx ? HEVC_MC_TPITCH : HEVC_MC_PITCH) / 2;
        const int cp = tw >> 1;
        const uint32_t *lines1 = lines + HEVC_MC_PAIR_ROW * pitch2;
        /* one call per plane, no barrier between them: together they fill the wave (a 16x16 block: 64 + 64 items of two rows) */
        if (cp * ((th_ + 3) >> 2) > 32) { hevc_mc_vpass<TAPS, 4>(sink0, lines, pitch2, tv, cp, th_, vshift, lane); hevc_mc_vpass<TAPS, 4>(sink1, lines1, pitch2, tv, cp, th_, vshift, lane); }
        else { hevc_mc_vpass<TAPS, 2>(sink0, lines, pitch2, tv, cp, th_, vshift, lane); hevc_mc_vpass<TAPS, 2>(sink1, lines1, pitch2, tv, cp, th_, vshift, lane); }
    }
    MI355_HEVC_SYNC();
}
template <int TAPS>
__device__ inline void hevc_mc_taps(int16_t *dst, int ds, const uint8_t *src, int ss, int width, int height,
                                    int mx, int my, int bd, HevcMcScratch &s)
{
    const int px = bd > 8 ? 2 : 1;
    constexpr int before = TAPS == 8 ? 3 : 1;
    const int bx = mx ? before : 0, by = my ? before : 0;
    const int amode = hevc_mc_align(dst, ds);
    const ptrdiff_t sb = (ptrdiff_t)ss * px;
    for (int ty = 0; ty < height; ty += HEVC_MC_TILE_H)
    for (int tx = 0; tx < width; tx += HEVC_MC_TILE) {
        const int tw = width - tx < HEVC_MC_TILE ? width - tx : HEVC_MC_TILE, th_ = height - ty < HEVC_MC_TILE_H ? height - ty : HEVC_MC_TILE_H;
        const HevcMcToI16 sink{ dst + (ptrdiff_t)ty * ds + tx, ds, amode };
        hevc_mc_tile<TAPS>(sink, src + (ptrdiff_t)(ty - by) * sb + (ptrdiff_t)(tx - bx) * px, sb, tw, th_, mx, my, bd, s);
    }
}
__device__ inline void hevc_mc_wave(int16_t *dst, int ds, const uint8_t *src, int ss, int width, int height,
                                    int mx, int my, int bd, int taps, HevcMcScratch &s)
{
    if (taps == 8) hevc_mc_taps<8>(dst, ds, src, ss, width, height, mx, my, bd, s);
    else hevc_mc_taps<4>(dst, ds, src, ss, width, height, mx, my, bd, s);
    MI355_HEVC_SYNC();
}

/* ---- a15: 14-bit intermediate -> samples (:1091-1242); mode 0 plain, 1 average, 2 weighted,
 * 3 weighted average ---------------------------------------------------------------------------- */
struct HevcPredParams {
    int mode, denom, w0, w1, o0, o1;
};
__device__ __forceinline__ int hevc_pred_px(const HevcPredParams &p, int a, int b, int bd)
{
    const int shift = 14 - bd;
    if (p.mode == 0) return clip_px((a + (1 << (shift - 1))) >> shift, bd);
    if (p.mode == 1) return clip_px((a + b + (1 << shift)) >> (shift + 1), bd);
    const int log2Wd = p.denom + shift;
    if (p.mode == 2) {
        const int ox = p.o0 * (1 << (bd - 8));
        return clip_px(log2Wd >= 1 ? ((a * p.w0 + (1 << (log2Wd - 1))) >> log2Wd) + ox : a * p.w0 + ox, bd);
    }
    const int o0 = p.o0 * (1 << (bd - 8)), o1 = p.o1 * (1 << (bd - 8));
    return clip_px((a * p.w0 + b * p.w1 + ((o0 + o1 + 1) << log2Wd)) >> (log2Wd + 1), bd);
}

/* ---- a16: deblocking of one 8-sample edge, lanes 0..7 = the 8 lines (:1264-1392) ----------------
 * `xs`: sample step across the edge, `ys`: along it.  Decisions use lines 0 and 3 of each 4-line
 * half, fetched from the neighbouring lanes with shuffles. */
/* the decisions and the arithmetic of one line of an edge (hevc_loop_filter_luma, hevcdsp_template.c:1520-1620): p[0..3] / q[0..3] = the line's samples from the
 * edge outwards, the lane's seven neighbours of its group of eight hold the other lines (lines 0 and 3 of each half decide for the half).  Every lane of the
 * group calls it (the shuffles); np / nq = p0..p2 / q0..q2 afterwards; false: the line is left as it is */
__device__ __forceinline__ bool hevc_lf_luma_core(const int p[4], const int q[4], int beta, const int *tc_, const uint8_t *no_p_, const uint8_t *no_q_, int bd, bool act,
                                                  int np[3], int nq[3])
{
    const int lane = lane_id(), l = lane & 7, j = l >> 2, g0 = lane & ~7;
    const int dp = iabs(p[2] - 2 * p[1] + p[0]), dq = iabs(q[2] - 2 * q[1] + q[0]);
    const int sflat = iabs(p[3] - p[0]) + iabs(q[3] - q[0]), sgap = iabs(p[0] - q[0]);
    /* values of line 0 and line 3 of this lane's half */
    const int dp0 = __shfl(dp, g0 + 4 * j), dp3 = __shfl(dp, g0 + 4 * j + 3), dq0 = __shfl(dq, g0 + 4 * j), dq3 = __shfl(dq, g0 + 4 * j + 3);
    const int f0 = __shfl(sflat, g0 + 4 * j), f3 = __shfl(sflat, g0 + 4 * j + 3), gp0 = __shfl(sgap, g0 + 4 * j), gp3 = __shfl(sgap, g0 + 4 * j + 3);
    if (!act) return false;
    beta <<= bd - 8;
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    const int tc = tc_[j] << (bd - 8), no_p = no_p_[j], no_q = no_q_[j];
    if (d0 + d3 >= beta) return false;
    const int beta_3 = beta >> 3, beta_2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    np[0] = p[0]; np[1] = p[1]; np[2] = p[2]; nq[0] = q[0]; nq[1] = q[1]; nq[2] = q[2];          /* the samples after filtering */
    if (f0 < beta_3 && gp0 < tc25 && f3 < beta_3 && gp3 < tc25 && (d0 << 1) < beta_2 && (d3 << 1) < beta_2) {
        const int tc2 = tc << 1;
        if (!no_p) {
            np[0] = p[0] + clip3(((p[2] + 2 * p[1] + 2 * p[0] + 2 * q[0] + q[1] + 4) >> 3) - p[0], -tc2, tc2);
            np[1] = p[1] + clip3(((p[2] + p[1] + p[0] + q[0] + 2) >> 2) - p[1], -tc2, tc2);
            np[2] = p[2] + clip3(((2 * p[3] + 3 * p[2] + p[1] + p[0] + q[0] + 4) >> 3) - p[2], -tc2, tc2);
        }
        if (!no_q) {
            nq[0] = q[0] + clip3(((p[1] + 2 * p[0] + 2 * q[0] + 2 * q[1] + q[2] + 4) >> 3) - q[0], -tc2, tc2);
            nq[1] = q[1] + clip3(((p[0] + q[0] + q[1] + q[2] + 2) >> 2) - q[1], -tc2, tc2);
            nq[2] = q[2] + clip3(((2 * q[3] + 3 * q[2] + q[1] + q[0] + p[0] + 4) >> 3) - q[2], -tc2, tc2);
        }
    } else {
        const int tc_2 = tc >> 1, thr = (beta + (beta >> 1)) >> 3;
        const int nd_p = dp0 + dp3 < thr ? 2 : 1, nd_q = dq0 + dq3 < thr ? 2 : 1;
        int delta0 = (9 * (q[0] - p[0]) - 3 * (q[1] - p[1]) + 8) >> 4;
        if (iabs(delta0) >= 10 * tc) return false;
        delta0 = clip3(delta0, -tc, tc);
        if (!no_p) np[0] = clip_px(p[0] + delta0, bd);
        if (!no_q) nq[0] = clip_px(q[0] - delta0, bd);
        if (!no_p && nd_p > 1) np[1] = clip_px(p[1] + clip3((((p[2] + p[0] + 1) >> 1) - p[1] + delta0) >> 1, -tc_2, tc_2), bd);
        if (!no_q && nd_q > 1) nq[1] = clip_px(q[1] + clip3((((q[2] + q[0] + 1) >> 1) - q[1] - delta0) >> 1, -tc_2, tc_2), bd);
    }
    return true;
}
/* `groups`: false = lanes 0..7 filter one edge (the other lanes idle); true = every group of eight lanes
 * filters its own edge (per-lane arguments) */
__device__ inline void hevc_lf_luma_wave(uint8_t *pix, int xs, int ys, int beta, const int *tc_, const uint8_t *no_p_,
                                         const uint8_t *no_q_, int bd, bool groups = false, bool enabled = true)
{
    const int lane = lane_id(), l = lane & 7, j = l >> 2, g0 = lane & ~7;
    const bool act = enabled && (groups || lane < 8);
    /* across a vertical edge (xs == 1) the eight samples of a line are contiguous: one 16-byte (8-byte for 8-bit samples)
     * access each way instead of eight loads and up to six stores */
    const bool row_wise = xs == 1;
    int p[4], q[4];
    if (act && row_wise) {
        if (bd > 8) {
            uint32_t w[4];
            __builtin_memcpy(w, pix + 2 * ((ptrdiff_t)l * ys - 4), 16);
            p[3] = w[0] & 0xFFFF; p[2] = w[0] >> 16; p[1] = w[1] & 0xFFFF; p[0] = w[1] >> 16;
            q[0] = w[2] & 0xFFFF; q[1] = w[2] >> 16; q[2] = w[3] & 0xFFFF; q[3] = w[3] >> 16;
        } else {
            uint32_t w[2];
            __builtin_memcpy(w, pix + ((ptrdiff_t)l * ys - 4), 8);
            p[3] = w[0] & 0xFF; p[2] = (w[0] >> 8) & 0xFF; p[1] = (w[0] >> 16) & 0xFF; p[0] = w[0] >> 24;
            q[0] = w[1] & 0xFF; q[1] = (w[1] >> 8) & 0xFF; q[2] = (w[1] >> 16) & 0xFF; q[3] = w[1] >> 24;
        }
    } else {
        for (int k = 0; k < 4; k++) {
            p[k] = act ? ldpx(pix, -(k + 1) * xs + l * ys, bd) : 0;
            q[k] = act ? ldpx(pix, k * xs + l * ys, bd) : 0;
        }
    }
    int np[3], nq[3];
    if (!hevc_lf_luma_core(p, q, beta, tc_, no_p_, no_q_, bd, act, np, nq)) return;
    if (row_wise) {
        /* p3 and q3 go back unchanged; no other job of a launch touches this line's eight samples */
        if (bd > 8) {
            const uint32_t w[4] = { (uint32_t)p[3] | ((uint32_t)np[2] << 16), (uint32_t)np[1] | ((uint32_t)np[0] << 16),
                                    (uint32_t)nq[0] | ((uint32_t)nq[1] << 16), (uint32_t)nq[2] | ((uint32_t)q[3] << 16) };
            __builtin_memcpy(pix + 2 * ((ptrdiff_t)l * ys - 4), w, 16);
        } else {
            const uint32_t w[2] = { (uint32_t)p[3] | ((uint32_t)np[2] << 8) | ((uint32_t)np[1] << 16) | ((uint32_t)np[0] << 24),
                                    (uint32_t)nq[0] | ((uint32_t)nq[1] << 8) | ((uint32_t)nq[2] << 16) | ((uint32_t)q[3] << 24) };
            __builtin_memcpy(pix + ((ptrdiff_t)l * ys - 4), w, 8);
        }
        return;
    }
    for (int k = 0; k < 3; k++) {
        if (np[k] != p[k]) stpx(pix, -(k + 1) * xs + l * ys, np[k], bd);
        if (nq[k] != q[k]) stpx(pix, k * xs + l * ys, nq[k], bd);
    }
}
__device__ inline void hevc_lf_chroma_wave(uint8_t *pix, int xs, int ys, const int *tc_, const uint8_t *no_p_, const uint8_t *no_q_, int bd,
                                           bool groups = false, bool enabled = true)
{
    const int lane = lane_id();
    if (!enabled || (!groups && lane >= 8)) return;
    const int l = lane & 7, j = l >> 2, tc = tc_[j] << (bd - 8);
    if (tc <= 0) return;
    const int p1 = ldpx(pix, -2 * xs + l * ys, bd), p0 = ldpx(pix, -xs + l * ys, bd);
    const int q0 = ldpx(pix, l * ys, bd), q1 = ldpx(pix, xs + l * ys, bd);
    const int delta0 = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
    if (!no_p_[j]) stpx(pix, -xs + l * ys, clip_px(p0 + delta0, bd), bd);
    if (!no_q_[j]) stpx(pix, l * ys, clip_px(q0 - delta0, bd), bd);
}

/* ---- a17: SAO (:270-718).  The region a (class, borders) pair owns is decided by the caller's
 * geometry helper; every sample of it is independent. ----------------------------------------- */
struct SaoJob {
    int width, height;            /* as passed by the reference caller */
    int c_idx, cls, bd, edge;     /* edge: 0 band, 1 edge */
    int borders[4];
    int vert_edge, horiz_edge, diag_edge;
    int eo_class, band_position;
    int offset_val[5];
};
/* two neighbouring samples at element offset o as one (possibly unaligned) access: low half = sample o */
__device__ __forceinline__ uint32_t sao_ld2(const uint8_t *p, ptrdiff_t o, int bd)
{
    if (bd > 8) { uint32_t v; __builtin_memcpy(&v, p + 2 * o, 4); return v; }
    uint16_t v; __builtin_memcpy(&v, p + o, 2);
    return (uint32_t)(v & 0xFF) | ((uint32_t)(v >> 8) << 16);
}
__device__ __forceinline__ void sao_st2(uint8_t *p, ptrdiff_t o, int v0, int v1, int bd)
{
    if (bd > 8) { const uint32_t v = (uint32_t)v0 | ((uint32_t)v1 << 16); __builtin_memcpy(p + 2 * o, &v, 4); }
    else { const uint16_t v = (uint16_t)(v0 | (v1 << 8)); __builtin_memcpy(p + o, &v, 2); }
}
/* src/dst point at the caller's (0,0) sample; `dt`/`st` = samples per row of dst/src; `tbl`: 32 ints of LDS */
/* offset_val[i] without an indexed read (an array indexed at run time would live in scratch memory) */
__device__ __forceinline__ int sao_offset(const SaoJob &j, int i)
{
    return i == 0 ? j.offset_val[0] : (i == 1 ? j.offset_val[1] : (i == 2 ? j.offset_val[2] : (i == 3 ? j.offset_val[3] : j.offset_val[4])));
}
__device__ inline void hevc_sao_wave(uint8_t *dst, int dt, const uint8_t *src, int st, const SaoJob &j, int *tbl)
{
    const int chroma = j.c_idx != 0, cw = (8 >> chroma) + 2, ch = (4 >> chroma) + 2, bd = j.bd, cls = j.cls;
    int x0 = 0, y0 = 0, w = j.width, h = j.height;
    if (cls & 1) { y0 = -ch; h = ch; } else if (!j.borders[3]) h -= ch;
    if (cls & 2) { x0 = -cw; w = cw; } else if (!j.borders[2]) w -= cw;
    const int w0 = w, h0 = h;
    const int lane = lane_id();
    /* Fast forms, two samples per lane, offsets from a table in LDS: the band filter always; the edge filter when no
     * picture border, slice or tile edge touches the region (every sample of it then has both neighbours and none is
     * restored) — the bulk of a picture. */
    const bool plain_edge = !(j.borders[0] | j.borders[1] | j.borders[2] | j.borders[3] | j.vert_edge | j.horiz_edge | j.diag_edge);
    if (w0 > 0 && (w0 & 1) == 0 && (!j.edge || plain_edge)) {
        if (!j.edge) { if (lane < 32) { const int k = (lane - j.band_position) & 31; tbl[lane] = k < 4 ? sao_offset(j, k + 1) : 0; } }
        else if (lane < 5) tbl[lane] = sao_offset(j, lane == 2 ? 0 : (lane == 0 ? 1 : (lane == 1 ? 2 : (lane == 3 ? 3 : 4))));   /* edge_idx[] = {1,2,0,3,4} */
        MI355_WAVE_SYNC();      /* a wave's function: the table is this wave's */
        const int eo = j.eo_class, hw = w0 >> 1, winv = mi355_inv20(hw), shift = bd - 5;
        const int dx0 = eo == 0 ? -1 : (eo == 1 ? 0 : (eo == 2 ? -1 : 1)), dy0 = eo == 0 ? 0 : -1;
        const ptrdiff_t da = dx0 + (ptrdiff_t)dy0 * st;
        for (int i = lane; i < hw * h0; i += 64) {
            const int y = mi355_div20(i, winv), x = 2 * (i - y * hw);
            const ptrdiff_t o = (ptrdiff_t)(y0 + y) * st + x0 + x;
            const uint32_t c2 = sao_ld2(src, o, bd);
            const int c0 = (int)(c2 & 0xFFFF), c1 = (int)(c2 >> 16);
            int v0, v1;
            if (!j.edge) {
                v0 = c0 + tbl[c0 >> shift]; v1 = c1 + tbl[c1 >> shift];
            } else {
                const uint32_t a2 = sao_ld2(src, o + da, bd), b2 = sao_ld2(src, o - da, bd);
                const int a0 = (int)(a2 & 0xFFFF), a1 = (int)(a2 >> 16), b0 = (int)(b2 & 0xFFFF), b1 = (int)(b2 >> 16);
                /* sign(c - a) + sign(c - b) + 2 */
                v0 = c0 + tbl[clip3(c0 - a0, -1, 1) + clip3(c0 - b0, -1, 1) + 2];
                v1 = c1 + tbl[clip3(c1 - a1, -1, 1) + clip3(c1 - b1, -1, 1) + 2];
            }
            sao_st2(dst, (ptrdiff_t)(y0 + y) * dt + x0 + x, clip_px(v0, bd), clip_px(v1, bd), bd);
        }
        MI355_WAVE_SYNC();      /* a wave's function: the table is this wave's */
        return;
    }
    if (!j.edge) {
        const int shift = bd - 5, winv = mi355_inv20(w > 0 ? w : 1);
        for (int i = lane_id(); i < w * h; i += 64) {
            const int y = mi355_div20(i, winv), x = i - y * w, o = (y0 + y) * st + x0 + x;
            const int v = ldpx(src, o, bd), k = ((v >> shift) - j.band_position) & 31;
            stpx(dst, (y0 + y) * dt + x0 + x, clip_px(v + (k < 4 ? sao_offset(j, k + 1) : 0), bd), bd);
        }
        return;
    }
    const int eo = j.eo_class;
    int init_x = 0, init_y = 0;
    const bool colpre = !(cls & 2) && eo != 1, rowpre = !(cls & 1) && eo != 0;
    if (colpre) { if (j.borders[0]) init_x = 1; if (j.borders[2]) w--; }
    if (rowpre) { if (j.borders[1]) init_y = 1; if (j.borders[3]) h--; }
    const int dx0 = eo == 0 ? -1 : (eo == 1 ? 0 : (eo == 2 ? -1 : 1)), dy0 = eo == 0 ? 0 : -1;
    const int dx1 = -dx0, dy1 = -dy0;
    /* restore rules: each class owns one corner of the CTB */
    const int ex = (cls & 2) ? w - 1 : 0, ey = (cls & 1) ? h - 1 : 0;
    const int diag_class = (cls == 0 || cls == 3) ? 2 : 3;
    int save;
    if (cls == 0) save = !j.diag_edge && eo == 2 && !j.borders[0] && !j.borders[1];
    else if (cls == 1) save = !j.diag_edge && eo == 3 && !j.borders[0];
    else if (cls == 2) save = !j.diag_edge && eo == 3 && !j.borders[1];
    else save = !j.diag_edge && eo == 2;
    const int ya = init_y + ((cls & 1) ? 0 : save), yb = h - ((cls & 1) ? save : 0);
    const int xa = init_x + ((cls & 2) ? 0 : save), xb = w - ((cls & 2) ? save : 0);
    const int winv = mi355_inv20(w0 > 0 ? w0 : 1);
    for (int i = lane_id(); i < w0 * h0; i += 64) {
        const int y = mi355_div20(i, winv), x = i - y * w0, o = (y0 + y) * st + x0 + x;
        const int c = ldpx(src, o, bd);
        int v;
        const bool in_main = x >= init_x && x < w && y >= init_y && y < h;
        if (in_main) {
            const int a = ldpx(src, o + dx0 + dy0 * st, bd), b = ldpx(src, o + dx1 + dy1 * st, bd);
            const int d = (c > a) - (c < a) + (c > b) - (c < b);          /* -2..2 */
            const int idx = d == 0 ? 0 : (d == -2 ? 1 : (d == -1 ? 2 : (d == 1 ? 3 : 4)));
            v = clip_px(c + sao_offset(j, idx), bd);
        } else {
            v = clip_px(c + j.offset_val[0], bd);   /* picture-border column / row */
        }
        if (j.vert_edge && eo != 1 && x == ex && y >= ya && y < yb) v = c;
        if (j.horiz_edge && eo != 0 && y == ey && x >= xa && x < xb) v = c;
        if (j.diag_edge && eo == diag_class && x == ex && y == ey) v = c;
        stpx(dst, (y0 + y) * dt + x0 + x, v, bd);
    }
}

/* ---- a18: pure intra predictors (hevcpred_template.c:349-516); top/left point at element 0 and
 * element -1 is addressable; `st` = samples per row of the destination ------------------------- */
__device__ const int8_t k_intra_angle[33] = { 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                              -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__device__ const int16_t k_inv_angle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

struct HevcPredScratch {
    int16_t top[1 + 65], left[1 + 65];   /* [0] = element -1 */
    int16_t ref[32 + 65];                /* ref[32 + k] = extended main reference k */
};
/* kind 0 planar, 1 dc, 2 angular */
__device__ inline void hevc_pred_wave(HevcPredScratch &s, uint8_t *dst, int st, int log2, int kind, int c_idx, int mode, int bd)
{
    const int lane = lane_id(), size = 1 << log2;
    const int16_t *top = s.top + 1, *left = s.left + 1;
    if (kind == 0) {
        for (int i = lane; i < size * size; i += 64) {
            const int y = i >> log2, x = i & (size - 1);
            stpx(dst, x + y * st, ((size - 1 - x) * left[y] + (x + 1) * top[size] + (size - 1 - y) * top[x] + (y + 1) * left[size] + size) >> (log2 + 1), bd);
        }
        return;
    }
    if (kind == 1) {
        int dc = size;
        for (int i = 0; i < size; i++) dc += left[i] + top[i];
        dc >>= log2 + 1;
        const bool smooth = c_idx == 0 && size < 32;
        for (int i = lane; i < size * size; i += 64) {
            const int y = i >> log2, x = i & (size - 1);
            int v = dc;
            if (smooth) {
                if (x == 0 && y == 0) v = (left[0] + 2 * dc + top[0] + 2) >> 2;
                else if (y == 0) v = (top[x] + 3 * dc + 2) >> 2;
                else if (x == 0) v = (left[y] + 3 * dc + 2) >> 2;
            }
            stpx(dst, x + y * st, v, bd);
        }
        return;
    }
    const int angle = k_intra_angle[mode - 2], last = (size * angle) >> 5;
    const bool vertical = mode >= 18;
    const int16_t *main_e = vertical ? top : left, *side_e = vertical ? left : top;
    int16_t *ref = s.ref + 32;
    for (int k = lane; k <= (angle < 0 ? size : 2 * size); k += 64) ref[k] = main_e[k - 1];
    if (angle < 0 && last < -1)
        for (int k = last + lane; k <= -1; k += 64) ref[k] = side_e[-1 + ((k * k_inv_angle[mode - 11] + 128) >> 8)];
    MI355_HEVC_SYNC();
    const bool edge_fix = c_idx == 0 && size < 32 && (mode == 26 || mode == 10);
    for (int i = lane; i < size * size; i += 64) {
        const int a = i >> log2, b = i & (size - 1);         /* a: along the minor axis */
        const int idx = ((a + 1) * angle) >> 5, fact = ((a + 1) * angle) & 31;
        int v = fact ? ((32 - fact) * ref[b + idx + 1] + fact * ref[b + idx + 2] + 16) >> 5 : ref[b + idx + 1];
        const int x = vertical ? b : a, y = vertical ? a : b;
        if (edge_fix) {
            if (mode == 26 && x == 0) v = clip_px(top[0] + ((left[y] - left[-1]) >> 1), bd);
            if (mode == 10 && y == 0) v = clip_px(left[0] + ((top[x] - top[-1]) >> 1), bd);
        }
        stpx(dst, x + y * st, v, bd);
    }
}

}  // namespace MI355_HEVC_NS
#endif
