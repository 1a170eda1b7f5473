/*
 * sws.hip — the libswscale part of the path for yuv420p -> rgb24:
 *   k_sws_generic   the generic scaler of swscale() (libswscale/swscale.c:343-722) for whole pictures,
 *                   fused per output tile: horizontal 8->15 bit FIR of the source lines the tile needs
 *                   (hScale8To15_c :133-147) into LDS, vertical FIR + yuv->rgb LUT
 *                   (yuv2rgb24_{1,2,X}_c output.c:937-1110) from LDS, RGB rows staged in LDS and
 *                   written as dwords.  No int16 intermediate ever goes to HBM.
 *   k_sws_c24       the unscaled converter yuv2rgb_c_24_rgb (yuv2rgb.c:335-363).
 *   k_sws_line_*    the individual inner loops for the Tier-1 entry points.
 * Filter banks and LUTs are inputs (built by the reference's init code, see include/mi355_sws.h).
 * Execution model: 256-thread workgroups (4 waves) sharing one LDS tile; integer only, no MFMA.
 */
#include "mi355_rt.h"
#include "../../include/mi355_sws.h"
#include "../../include/mi355dsp.h"

using namespace mi355;

namespace {

constexpr int TW = 128;      /* output samples per tile row */
constexpr int MAXTH = 16;    /* output rows per tile (upper bound) */
constexpr int MAXL = 48;     /* source luma lines a tile may need (upper bound: the LDS tile is sized per context, sws_plan) */
constexpr int MAXC = 24;     /* source chroma lines a tile may need */
constexpr int NT = 256;

struct SwsDev {
    int srcW, srcH, dstW, dstH, chrSrcW, chrSrcH, chrDstW, special;
    int hls, hcs, vls, vcs;                 /* filter sizes */
    const int16_t *hLumC, *hChrC, *vLumC, *vChrC;
    const int32_t *hLumP, *hChrP, *vLumP, *vChrP;
    int th;                                 /* output rows per tile chosen at create time */
    int lum_lines, chr_lines;               /* source lines the LDS tile holds (the largest span of a tile of th rows) */
    int hstage;                             /* horizontal filter positions never decrease: source spans can be staged in LDS */
    int hident_l, hident_c;                 /* the horizontal filter of the plane is the identity (one tap of 1 << 14 at position i: an unscaled
                                             * conversion through the generic path): hScale8To15 is then src << 7 */
    mi355_sws_luts luts;
};

struct LutLds {
    uint8_t y[1024];
    int16_t rV[256], gU[256], gV[256], bU[256];
};
__device__ __forceinline__ void lut_load(LutLds &s, const mi355_sws_luts *g, int tid, int nt)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(g);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s);
    static_assert(sizeof(LutLds) == 3 * 4 * NT, "three dwords per thread");
    (void)nt;
    const uint32_t a = src[tid], b = src[tid + NT], c = src[tid + 2 * NT];      /* in flight together */
    dst[tid] = a; dst[tid + NT] = b; dst[tid + 2 * NT] = c;
}
__device__ __forceinline__ int clip_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
/* yuv2rgb_write, rgb24 branch (output.c:853-866) */
__device__ __forceinline__ void write_pair(const LutLds &t, uint8_t *dest, int Y1, int Y2, int U, int V)
{
    const int r = t.rV[V], g = t.gU[U] + t.gV[V], b = t.bU[U];
    dest[0] = t.y[r + Y1]; dest[1] = t.y[g + Y1]; dest[2] = t.y[b + Y1];
    dest[3] = t.y[r + Y2]; dest[4] = t.y[g + Y2]; dest[5] = t.y[b + Y2];
}

/* one output pair of the three packed templates; rows are addressed through accessors so the same
 * code serves LDS tiles (whole pictures) and packed global rows (Tier-1 line calls) */
template <typename Rows>
__device__ __forceinline__ void rgb_pair(const LutLds &t, uint8_t *dest, const Rows &R, int i, int mode, const int16_t *lumF, int ls,
                                         const int16_t *chrF, int cs, int yalpha, int uvalpha)
{
    int Y1, Y2, U, V;
    if (mode == 1) {          /* yuv2rgb_1_c_template output.c:1043-1110 */
        Y1 = clip_u8(R.lum(0, 2 * i) >> 7); Y2 = clip_u8(R.lum(0, 2 * i + 1) >> 7);
        if (uvalpha < 2048) { U = clip_u8(R.cu(0, i) >> 7); V = clip_u8(R.cv(0, i) >> 7); }
        else { U = clip_u8((R.cu(0, i) + R.cu(1, i)) >> 8); V = clip_u8((R.cv(0, i) + R.cv(1, i)) >> 8); }
    } else if (mode == 2) {   /* yuv2rgb_2_c_template :998-1041 */
        const int ya1 = 4096 - yalpha, ua1 = 4096 - uvalpha;
        Y1 = clip_u8((R.lum(0, 2 * i) * ya1 + R.lum(1, 2 * i) * yalpha) >> 19);
        Y2 = clip_u8((R.lum(0, 2 * i + 1) * ya1 + R.lum(1, 2 * i + 1) * yalpha) >> 19);
        U = clip_u8((R.cu(0, i) * ua1 + R.cu(1, i) * uvalpha) >> 19);
        V = clip_u8((R.cv(0, i) * ua1 + R.cv(1, i) * uvalpha) >> 19);
    } else {                  /* yuv2rgb_X_c_template :937-996: clipped only if a value has bit 8 set */
        Y1 = Y2 = U = V = 1 << 18;
        for (int j = 0; j < ls; j++) { const int f = lumF[j]; Y1 += R.lum(j, 2 * i) * f; Y2 += R.lum(j, 2 * i + 1) * f; }
        for (int j = 0; j < cs; j++) { const int f = chrF[j]; U += R.cu(j, i) * f; V += R.cv(j, i) * f; }
        Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
        if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
    }
    write_pair(t, dest, Y1, Y2, U, V);
}
typedef uint32_t sws_u32x2 __attribute__((vector_size(8)));
typedef uint32_t sws_u32x4 __attribute__((vector_size(16)));
/* eight neighbouring samples of a line (four pairs sharing a chroma sample each): Y values 0..255 in Y[8], the pairs' LUT
 * row offsets in r/g/b -> 24 RGB bytes at d (8-byte aligned), three 8-byte stores */
__device__ __forceinline__ void rgb24_store8(const LutLds &t, uint8_t *d, const int *Y, const int *r, const int *g, const int *b)
{
    uint32_t o[6];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int Y1 = Y[2 * p], Y2 = Y[2 * p + 1];
        const uint32_t r1 = t.y[r[p] + Y1], g1 = t.y[g[p] + Y1], b1 = t.y[b[p] + Y1];
        const uint32_t r2 = t.y[r[p] + Y2], g2 = t.y[g[p] + Y2], b2 = t.y[b[p] + Y2];
        /* six bytes per pair: pairs 0,2 start on a dword, pairs 1,3 in the middle of one */
        if ((p & 1) == 0) {
            o[3 * (p >> 1)] = r1 | (g1 << 8) | (b1 << 16) | (r2 << 24);
            o[3 * (p >> 1) + 1] = g2 | (b2 << 8);
        } else {
            o[3 * (p >> 1) + 1] |= (r1 << 16) | (g1 << 24);
            o[3 * (p >> 1) + 2] = b1 | (r2 << 8) | (g2 << 16) | (b2 << 24);
        }
    }
    sws_u32x2 *q = reinterpret_cast<sws_u32x2 *>(d);
    q[0] = sws_u32x2{ o[0], o[1] }; q[1] = sws_u32x2{ o[2], o[3] }; q[2] = sws_u32x2{ o[4], o[5] };
}
__device__ __forceinline__ int packed_mode(int ls, int cs) { return (ls == 1 && cs <= 2) ? 1 : ((ls == 2 && cs == 2) ? 2 : 0); }  /* swscale.c:658-682 */

/* hScale8To15_c swscale.c:133-147 for one output sample */
__device__ __forceinline__ int hscale_one(const uint8_t *src, const int16_t *f, int pos, int fs)
{
    int val = 0;
    for (int j = 0; j < fs; j++) val += (int)src[pos + j] * f[j];
    val >>= 7;
    return val < 32767 ? val : 32767;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

struct TileRows {
    const int16_t (*lumT)[TW];
    const int16_t (*cuT)[TW / 2];
    const int16_t (*cvT)[TW / 2];
    int lfirst, llo, lmax, cfirst, clo, cmax;   /* first tap line, first staged line, last picture line */
    __device__ __forceinline__ int lum(int j, int x) const { return lumT[clampi(lfirst + j, 0, lmax) - llo][x]; }
    __device__ __forceinline__ int cu(int j, int x) const { return cuT[clampi(cfirst + j, 0, cmax) - clo][x]; }
    __device__ __forceinline__ int cv(int j, int x) const { return cvT[clampi(cfirst + j, 0, cmax) - clo][x]; }
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* Horizontal pass of one plane for a tile: COLS output columns starting at gx0, source lines lo..hi, results
 * to out[line - lo][x].  The source span the columns need ([pos[gx0], pos[last] + fs), monotonic positions)
 * is staged in LDS SG lines at a time with aligned 16-byte (or dword) loads — a few coalesced loads per thread
 * instead of fs single-byte loads per output sample; spans wider than the stage, unaligned planes and non-monotonic
 * filters take the direct path. */
#ifndef MI355_SWS_SG
#define MI355_SWS_SG 16
#endif
constexpr int SG = MI355_SWS_SG;      /* source lines per staging round (a round costs two workgroup barriers) */
constexpr int SRC_DW = 76;            /* dwords per staged line: 2:1 with 8 taps needs 128 * 2 + 8 bytes (+2 of slack for zero taps) */
constexpr int STAGE_BYTES = (int)sizeof(uint32_t) * SG * SRC_DW;
/* a chroma tile row is half as wide: its staged lines are about half as long (64 * 2 + 8 bytes + the 15 of a 16-byte aligned start) and a
 * round holds half as many again in the same storage (the 20 chroma lines of a 2:1 reduction: one round instead of two) */
template <int COLS> struct StageGeom {
    static constexpr int PITCH = COLS == TW ? SRC_DW : 40;             /* dwords per staged line: a multiple of four (16-byte LDS stores) */
    static constexpr int LINES = COLS == TW ? SG : (3 * SG) / 2;       /* staged lines per round */
    static_assert(PITCH % 4 == 0 && (size_t)PITCH * LINES * sizeof(uint32_t) <= (size_t)STAGE_BYTES, "a round fits the staging storage");
};
constexpr int OUT_ROWS = STAGE_BYTES / (TW * 3);            /* rows of a narrow tile written per pass (they reuse the staging lines) */
static_assert(OUT_ROWS >= 8, "the narrow form needs at most two passes over a tile of MAXTH rows");
/* byte funnel shift, byte permute and the two-term 16-bit dot product (v_alignbyte_b32, v_perm_b32, v_dot2_i32_i16); plain C
 * under the SIMT emulator */
#ifdef MI355_HIP_EMU_H
static inline uint32_t sws_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (s & 3))); }
static inline uint32_t sws_pair(uint32_t w, int k) { return ((w >> (16 * k)) & 0xFFu) | (((w >> (16 * k + 8)) & 0xFFu) << 16); }
static inline int sws_dot2(uint32_t a, uint32_t b, int c) { return c + (int16_t)(a & 0xFFFF) * (int16_t)(b & 0xFFFF) + (int16_t)(a >> 16) * (int16_t)(b >> 16); }
static inline uint32_t sws_lo2(uint32_t a, uint32_t b) { return (a & 0xFFFFu) | (b << 16); }            /* (a.lo, b.lo) */
static inline uint32_t sws_hi2(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xFFFF0000u); }        /* (a.hi, b.hi) */
#else
__device__ __forceinline__ uint32_t sws_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
/* bytes 2k, 2k + 1 of w as two 16-bit values */
__device__ __forceinline__ uint32_t sws_pair(uint32_t w, int k) { return __builtin_amdgcn_perm(0u, w, k ? 0x0C030C02u : 0x0C010C00u); }
__device__ __forceinline__ uint32_t sws_lo2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }      /* (a.lo, b.lo) */
__device__ __forceinline__ uint32_t sws_hi2(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }      /* (a.hi, b.hi) */
typedef short sws_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sws_dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(sws_short2, a), __builtin_bit_cast(sws_short2, b), c, false);
}
#endif
/* eight taps: the column's byte offset inside a dword is the same on every staged line, so a line's eight samples are
 * three aligned dwords funnel-shifted into two, expanded to four 16-bit pairs and multiplied with the coefficient pairs
 * (3 LDS reads and 10 arithmetic instructions per output instead of 8 byte reads and 8 multiply-adds).  Products and sums
 * are the same integers (samples 0..255, coefficients 16 bits, |sum| < 2^31). */
template <int COLS>
__device__ __forceinline__ void hscale_lines8(const uint8_t *row0, int16_t *out0, const uint32_t *cp, int left)
{
    constexpr int per = NT / COLS;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(row0) & 3);
    const uint32_t *w0 = reinterpret_cast<const uint32_t *>(row0 - sh);
    const uint32_t c01 = cp[0], c23 = cp[1], c45 = cp[2], c67 = cp[3];
    static_assert(COLS >= 64, "a wave's threads share their first line: `left` is the same on all of them");
    /* four (chroma: three) lines at a time: their LDS reads go out together, then the arithmetic of the group (one read-wait-compute chain
     * per line leaves the wave waiting for the LDS once per output).  A short last round ends at a branch of the wave, between
     * groups or inside one (the reads of a group are unconditional: the staged lines exist). */
    constexpr int N = StageGeom<COLS>::LINES / per, G = N % 4 == 0 ? 4 : 3, PITCH = StageGeom<COLS>::PITCH;
    static_assert(StageGeom<COLS>::LINES % per == 0 && N % G == 0, "whole groups of lines");
#pragma unroll
    for (int g = 0; g < N; g += G) {
        if (g * per > left) break;
        uint32_t d[G][3];
#pragma unroll
        for (int q = 0; q < G; q++) {
            const uint32_t *w = w0 + (g + q) * per * PITCH;
            d[q][0] = w[0]; d[q][1] = w[1]; d[q][2] = w[2];
        }
#pragma unroll
        for (int q = 0; q < G; q++) {
            if ((g + q) * per > left) break;                 /* of the wave, like the one between groups */
            const uint32_t lo = sws_alignbyte(d[q][1], d[q][0], sh), hi = sws_alignbyte(d[q][2], d[q][1], sh);
            int v = sws_dot2(sws_pair(lo, 0), c01, 0);
            v = sws_dot2(sws_pair(lo, 1), c23, v);
            v = sws_dot2(sws_pair(hi, 0), c45, v);
            v = sws_dot2(sws_pair(hi, 1), c67, v);
            v >>= 7;
            out0[(g + q) * per * COLS] = (int16_t)(v < 32767 ? v : 32767);
        }
    }
}
template <int COLS, int TAPS>
__device__ __forceinline__ void hscale_lines(const uint8_t *row0, int16_t *out0, const uint32_t *cp, int left)
{
    constexpr int per = NT / COLS;
    int cf[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; j++) cf[j] = (int16_t)(cp[j >> 1] >> (16 * (j & 1)));
#pragma unroll
    for (int k = 0; k < StageGeom<COLS>::LINES / per; k++) {
        if (k * per > left) break;
        const uint8_t *row = row0 + k * per * (StageGeom<COLS>::PITCH * 4);
        int val = 0;
#pragma unroll
        for (int j = 0; j < TAPS; j++) val += (int)row[j] * cf[j];
        val >>= 7;
        out0[k * per * COLS] = (int16_t)(val < 32767 ? val : 32767);
    }
}
template <int COLS>
__device__ __forceinline__ void hscale_tile(int16_t (*out)[COLS], const uint8_t *src, int stride, int srcW, const int32_t *posT,
                                            const int16_t *coefT, int fs, int gx0, int ncols, int lo, int hi,
                                            uint32_t *stage_mem, int tid, bool zero_tail, bool may_stage, bool identity)
{
    constexpr int PITCH = StageGeom<COLS>::PITCH, LINES = StageGeom<COLS>::LINES;
    uint32_t (*stage)[PITCH] = reinterpret_cast<uint32_t (*)[PITCH]>(stage_mem);
    if (identity) {
        /* one tap of 1 << 14 at position i: (src * 16384) >> 7 = src << 7 (below the 32767 clamp).  Eight columns per thread:
         * one 8-byte load (aligned planes, inside the line), one 16-byte LDS write */
        constexpr int TPL = COLS / 8;                  /* threads per line */
        const int xg = 8 * (tid % TPL), gxi = gx0 + xg;
        const bool al8 = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | (uintptr_t)gx0) & 7) == 0;
        for (int l = lo + tid / TPL; l <= hi; l += NT / TPL) {
            const uint8_t *p = src + (size_t)l * stride + gxi;
            uint32_t b0 = 0, b1 = 0;
            if (al8 && gxi + 8 <= srcW && gxi + 8 <= ncols) { const sws_u32x2 w = *reinterpret_cast<const sws_u32x2 *>(p); b0 = w[0]; b1 = w[1]; }
            else {
                for (int k = 0; k < 4; k++) {
                    if (gxi + k < ncols && gxi + k < srcW) b0 |= (uint32_t)p[k] << (8 * k);
                    if (gxi + 4 + k < ncols && gxi + 4 + k < srcW) b1 |= (uint32_t)p[4 + k] << (8 * k);
                }
            }
            /* bytes -> int16 << 7, two per dword */
            sws_u32x4 o;
            o[0] = ((b0 & 0xFFu) << 7) | ((b0 & 0xFF00u) << 15);
            o[1] = ((b0 >> 9) & 0x7F80u) | ((b0 >> 1) & 0x7F800000u);
            o[2] = ((b1 & 0xFFu) << 7) | ((b1 & 0xFF00u) << 15);
            o[3] = ((b1 >> 9) & 0x7F80u) | ((b1 >> 1) & 0x7F800000u);
            if (zero_tail || gxi < ncols) *reinterpret_cast<sws_u32x4 *>(&out[l - lo][xg]) = o;
        }
        return;
    }
    const int x = tid & (COLS - 1), gx = gx0 + x, per = NT / COLS;
    const bool col_ok = gx < ncols;
    /* no load below sits under a lane condition (a conditional load is a branch, the load and a wait for it: a memory round
     * trip per tap): columns past the picture read the last column's entries and do not use them */
    const int gxc = col_ok ? gx : ncols - 1;
    const int pos = posT[gxc];
    const int16_t *f = coefT + (size_t)gxc * fs;
    const int last = imin(gx0 + COLS, ncols) - 1;
    /* 16-byte pieces when the plane allows it, dwords otherwise */
    const bool al16 = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0;
    const int s0 = posT[gx0], s1 = posT[last] + fs, a0 = al16 ? (s0 & ~15) : (s0 & ~3), nd = (s1 - a0 + 3) >> 2;
    /* the column's filter in registers as four pairs of 16-bit taps (taps past fs are zero): the line loops below multiply by
     * them instead of reloading.  Eight taps: the column's entry of the bank is one aligned 16-byte word (the banks are
     * hipMalloc'ed by mi355_sws_create) */
    uint32_t cp[4];
    if (fs == 8) {
        const uint4 w = *reinterpret_cast<const uint4 *>(f);
        cp[0] = w.x; cp[1] = w.y; cp[2] = w.z; cp[3] = w.w;
    } else {
        int t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = f[j < fs ? j : 0];
#pragma unroll
        for (int j = 0; j < 4; j++) cp[j] = (2 * j < fs ? (uint32_t)t[2 * j] & 0xFFFFu : 0u) | (2 * j + 1 < fs ? (uint32_t)t[2 * j + 1] << 16 : 0u);
    }
    const bool staged = may_stage && nd <= PITCH - 2 && s1 >= s0 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 3) == 0;
    if (!staged) {
        if (col_ok) {
            for (int l = lo + tid / COLS; l <= hi; l += per) out[l - lo][x] = (int16_t)hscale_one(src + (size_t)l * stride, f, pos, fs);
        } else if (zero_tail) {
            for (int l = lo + tid / COLS; l <= hi; l += per) out[l - lo][x] = 0;
        }
        return;
    }
    /* idx / n as a 24-bit multiply and a shift (mi355_div20: exact for idx * n < 2^19; here idx < LINES * n, n <= PITCH) */
    static_assert(LINES * PITCH * PITCH < (1 << 19), "mi355_div20 range");
    const int np = al16 ? (nd + 3) >> 2 : nd, inv = mi355_inv20(np);
    /* 16-byte pieces travel through registers, one round ahead: the loads of round r + 1 are issued before round r's
     * arithmetic and stored to the staging lines after it (a round's loads would otherwise be waited for at its first
     * barrier with nothing to do: a third of the kernel's time on a 2:1 reduction).  At most PF pieces per thread and round. */
    constexpr int PF = (LINES * ((PITCH + 3) / 4) + NT - 1) / NT;
    uint4 pre[PF];
    /* which piece of a round a thread moves does not change from round to round: its staging row, source column and LDS address are
     * worked out once per plane.  Nothing sits under a lane condition: a piece past the round's last repeats the last one, a line past
     * the plane's last needed line repeats that one (the same bytes to the same place, or to a staging line nothing reads). */
    int p_row[PF], p_col[PF];
    uint32_t *p_lds[PF];
#pragma unroll
    for (int j = 0; j < PF; j++) {
        const int idx = imin(tid + j * NT, LINES * np - 1), r = mi355_div20(idx, inv), d = idx - r * np, off = a0 + 16 * d;
        p_row[j] = r;
        p_col[j] = imin(off, (srcW - 1) & ~15);
        p_lds[j] = &stage[r][4 * d];
    }
    auto fetch16 = [&](int base) {
#pragma unroll
        for (int j = 0; j < PF; j++) {
            if (j * NT >= LINES * np) break;                 /* a narrow plane's round is fewer pieces than threads: the same for every thread */
            /* the aligned 16 bytes lie inside the line's stride (both multiples of 16, column < srcW <= stride): always readable.
             * Bytes at and past srcW (padding) are whatever the plane holds there: no tap with a non-zero coefficient reads them —
             * mi355_sws_create stages only filter banks whose every position + size stays inside the line, as the reference's
             * initFilter builds them (utils.c "fix borders") — and a tap past the filter's size multiplies them by zero. */
            const uint4 w = *reinterpret_cast<const uint4 *>(src + (size_t)imin(base + p_row[j], hi) * stride + p_col[j]);
            pre[j] = w;
        }
    };
#ifndef MI355_SWS_EXP_NOSTAGE     /* developer experiments (tools/exp_sws_sg.sh): the pass without its loads / without its arithmetic */
    if (al16) fetch16(lo);
#endif
    for (int base = lo; base <= hi; base += LINES) {
#ifndef MI355_SWS_EXP_NOSTAGE
        if (al16) {
#pragma unroll
            for (int j = 0; j < PF; j++) {
                if (j * NT >= LINES * np) break;
                *reinterpret_cast<uint4 *>(p_lds[j]) = pre[j];
            }
        } else
        for (int idx = tid; idx < LINES * np; idx += NT) {
            const int r = mi355_div20(idx, inv), d = idx - r * np, line = base + r;
            if (line > hi) continue;
            const uint8_t *p = src + (size_t)line * stride + a0 + 4 * d;
            uint32_t w;
            if (a0 + 4 * d + 4 <= srcW) w = *reinterpret_cast<const uint32_t *>(p);
            else {
                w = 0;
                for (int b = 0; b < 4; b++) if (a0 + 4 * d + b < srcW) w |= (uint32_t)p[b] << (8 * b);
            }
            stage[r][d] = w;
        }
#endif
        __syncthreads();
#ifndef MI355_SWS_EXP_NOSTAGE
        if (al16 && base + LINES <= hi) fetch16(base + LINES);
#endif
        /* the thread's column over the staged lines: fixed trip count, so line and output addresses are
         * immediate offsets from one base each; tap count rounded up to 1 / 2 / 4 / 8 (taps past fs are zero, the
         * bytes exist: slack) */
        const int r0 = tid / COLS;
        const uint8_t *row0 = reinterpret_cast<const uint8_t *>(stage[r0]) + (pos - a0);
        int16_t *out0 = &out[base + r0 - lo][x];
        const int left = uniform(hi - base - r0);            /* lines r0, r0 + per, ... while k * per <= left (r0: one value per wave) */
#ifndef MI355_SWS_EXP_NOLINES
        if (col_ok) {
            if (fs == 1) hscale_lines<COLS, 1>(row0, out0, cp, left);
            else if (fs <= 2) hscale_lines<COLS, 2>(row0, out0, cp, left);
            else if (fs <= 4) hscale_lines<COLS, 4>(row0, out0, cp, left);
            else if (fs <= 8) hscale_lines8<COLS>(row0, out0, cp, left);
            else {
                for (int k = 0; k < LINES / per && k * per <= left; k++) {
                    const uint8_t *row = row0 + k * per * (PITCH * 4);
                    int val = 0;
                    for (int j = 0; j < fs; j++) val += (int)row[j] * f[j];
                    val >>= 7;
                    out0[k * per * COLS] = (int16_t)(val < 32767 ? val : 32767);
                }
            }
        } else if (zero_tail) {
            for (int k = 0; k < LINES / per && k * per <= left; k++) out0[k * per * COLS] = 0;
        }
#endif
        __syncthreads();
    }
}

/* Vertical pass + LUT for the output rows of a tile with the row's filter taps and source-line indices in
 * registers: NL / NC = luma / chroma tap counts rounded up to 1, 2, 4 or 8 (taps past the real size carry a
 * zero coefficient and a valid line index).  A thread owns one output row and every 16th pair of it. */
template <int NL, int NC, bool WIDE>
__device__ __forceinline__ void vertical_rows(const SwsDev &c, const LutLds &lut, const int16_t (*s_lum)[TW], const int16_t (*s_cu)[TW / 2],
                                              const int16_t (*s_cv)[TW / 2], uint8_t (*s_out)[TW * 3], int tid, int y0, int y1, int llo, int clo,
                                              int npairs, int mode, uint8_t *wide_dst, int dst_stride, int out_row0)
{
    /* out_row0: first tile row of this pass of the narrow form (s_out holds OUT_ROWS rows at a time) */
    const int row = tid >> 4, gy = y0 + row;
    if (gy > y1 || (!WIDE && (row < out_row0 || row >= out_row0 + OUT_ROWS))) return;
    const int ls = c.vls, cs = c.vcs;
    const int lfirst = imax(1 - ls, c.vLumP[gy]), cfirst = imax(1 - cs, c.vChrP[gy]);
    /* the taps: every load unconditional (a tap past the filter reads tap 0 and becomes zero), so that they are in flight together */
    int lf[NL], li[NL], cf[NC], ci[NC];
#pragma unroll
    for (int j = 0; j < NL; j++) lf[j] = c.vLumC[(size_t)gy * ls + (j < ls ? j : 0)];
#pragma unroll
    for (int j = 0; j < NC; j++) cf[j] = c.vChrC[(size_t)gy * cs + (j < cs ? j : 0)];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        lf[j] = j < ls ? lf[j] : 0;
        li[j] = clampi(lfirst + (j < ls ? j : 0), 0, c.srcH - 1) - llo;
    }
#pragma unroll
    for (int j = 0; j < NC; j++) {
        cf[j] = j < cs ? cf[j] : 0;
        ci[j] = clampi(cfirst + (j < cs ? j : 0), 0, c.chrSrcH - 1) - clo;
    }
    if (WIDE) {
        /* full tile, 8-byte aligned destination: a thread takes eight neighbouring samples (16 / 8 bytes per LDS read)
         * and stores its 24 RGB bytes directly */
        const int grp = tid & 15;
        int Y[8], U[4], V[4];
        if (mode == 1) {
            const int uvalpha = cs == 1 ? 0 : cf[NC > 1 ? 1 : 0];
            const sws_u32x4 l0 = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[0]][8 * grp]);
            const sws_u32x2 u0 = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[0]][4 * grp]), v0 = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[0]][4 * grp]);
            const sws_u32x2 u1 = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[NC > 1 ? 1 : 0]][4 * grp]), v1 = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[NC > 1 ? 1 : 0]][4 * grp]);
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] = clip_u8((int16_t)(l0[k >> 1] >> (16 * (k & 1))) >> 7);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a = (int16_t)(u0[k >> 1] >> (16 * (k & 1))), b = (int16_t)(v0[k >> 1] >> (16 * (k & 1)));
                const int a1 = (int16_t)(u1[k >> 1] >> (16 * (k & 1))), b1 = (int16_t)(v1[k >> 1] >> (16 * (k & 1)));
                if (uvalpha < 2048) { U[k] = clip_u8(a >> 7); V[k] = clip_u8(b >> 7); }
                else { U[k] = clip_u8((a + a1) >> 8); V[k] = clip_u8((b + b1) >> 8); }
            }
        } else if (mode == 2) {
            const int ya = lf[NL > 1 ? 1 : 0], ua = cf[NC > 1 ? 1 : 0], ya1 = 4096 - ya, ua1 = 4096 - ua;
            const sws_u32x4 l0 = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[0]][8 * grp]), l1 = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[NL > 1 ? 1 : 0]][8 * grp]);
            const sws_u32x2 u0 = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[0]][4 * grp]), v0 = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[0]][4 * grp]);
            const sws_u32x2 u1 = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[NC > 1 ? 1 : 0]][4 * grp]), v1 = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[NC > 1 ? 1 : 0]][4 * grp]);
#pragma unroll
            for (int k = 0; k < 8; k++)
                Y[k] = clip_u8(((int16_t)(l0[k >> 1] >> (16 * (k & 1))) * ya1 + (int16_t)(l1[k >> 1] >> (16 * (k & 1))) * ya) >> 19);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                U[k] = clip_u8(((int16_t)(u0[k >> 1] >> (16 * (k & 1))) * ua1 + (int16_t)(u1[k >> 1] >> (16 * (k & 1))) * ua) >> 19);
                V[k] = clip_u8(((int16_t)(v0[k >> 1] >> (16 * (k & 1))) * ua1 + (int16_t)(v1[k >> 1] >> (16 * (k & 1))) * ua) >> 19);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) Y[k] = 1 << 18;
#pragma unroll
            for (int k = 0; k < 4; k++) U[k] = V[k] = 1 << 18;
            /* two taps at a time: the same sample of two source lines side by side in a dword (one byte-permute) against the
             * tap pair, v_dot2_i32_i16 — the same integer sum as the reference's per-tap multiply-add (15-bit samples,
             * 16-bit coefficients, int accumulators) */
            if (NL >= 2) {
#pragma unroll
                for (int j = 0; j + 1 < NL; j += 2) {
                    const sws_u32x4 la = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[j]][8 * grp]), lb = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[j + 1]][8 * grp]);
                    const uint32_t cp = ((uint32_t)lf[j] & 0xFFFFu) | ((uint32_t)lf[j + 1] << 16);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        Y[2 * q] = sws_dot2(sws_lo2(la[q], lb[q]), cp, Y[2 * q]);
                        Y[2 * q + 1] = sws_dot2(sws_hi2(la[q], lb[q]), cp, Y[2 * q + 1]);
                    }
                }
            } else {
                const sws_u32x4 l = *reinterpret_cast<const sws_u32x4 *>(&s_lum[li[0]][8 * grp]);
#pragma unroll
                for (int k = 0; k < 8; k++) Y[k] += (int16_t)(l[k >> 1] >> (16 * (k & 1))) * lf[0];
            }
            if (NC >= 2) {
#pragma unroll
                for (int j = 0; j + 1 < NC; j += 2) {
                    const sws_u32x2 ua = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[j]][4 * grp]), ub = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[j + 1]][4 * grp]);
                    const sws_u32x2 va = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[j]][4 * grp]), vb = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[j + 1]][4 * grp]);
                    const uint32_t cp = ((uint32_t)cf[j] & 0xFFFFu) | ((uint32_t)cf[j + 1] << 16);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        U[2 * q] = sws_dot2(sws_lo2(ua[q], ub[q]), cp, U[2 * q]);
                        U[2 * q + 1] = sws_dot2(sws_hi2(ua[q], ub[q]), cp, U[2 * q + 1]);
                        V[2 * q] = sws_dot2(sws_lo2(va[q], vb[q]), cp, V[2 * q]);
                        V[2 * q + 1] = sws_dot2(sws_hi2(va[q], vb[q]), cp, V[2 * q + 1]);
                    }
                }
            } else {
                const sws_u32x2 u = *reinterpret_cast<const sws_u32x2 *>(&s_cu[ci[0]][4 * grp]), v = *reinterpret_cast<const sws_u32x2 *>(&s_cv[ci[0]][4 * grp]);
#pragma unroll
                for (int k = 0; k < 4; k++) { U[k] += (int16_t)(u[k >> 1] >> (16 * (k & 1))) * cf[0]; V[k] += (int16_t)(v[k >> 1] >> (16 * (k & 1))) * cf[0]; }
            }
#pragma unroll
            for (int p = 0; p < 4; p++) {       /* clipped per pair, only if one of its four values has bit 8 set (output.c:963) */
                int &Y1 = Y[2 * p], &Y2 = Y[2 * p + 1], &Up = U[p], &Vp = V[p];
                Y1 >>= 19; Y2 >>= 19; Up >>= 19; Vp >>= 19;
                if ((Y1 | Y2 | Up | Vp) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); Up = clip_u8(Up); Vp = clip_u8(Vp); }
            }
        }
        int r[4], g[4], b[4];
#pragma unroll
        for (int p = 0; p < 4; p++) { r[p] = lut.rV[V[p]]; g[p] = lut.gU[U[p]] + lut.gV[V[p]]; b[p] = lut.bU[U[p]]; }
        rgb24_store8(lut, wide_dst + (size_t)row * dst_stride + 24 * grp, Y, r, g, b);
        return;
    }
    /* fixed trip count: the pairs of a thread are 16 apart, so every LDS address is one base plus an immediate */
#pragma unroll
    for (int k = 0; k < TW / 32; k++) {
        const int i = (tid & 15) + 16 * k;
        if (i >= npairs) continue;
        int Y1, Y2, U, V;
        if (mode == 1) {          /* yuv2rgb_1_c_template output.c:1043-1110 (ls == 1, cs <= 2) */
            const int uvalpha = cs == 1 ? 0 : cf[NC > 1 ? 1 : 0];
            Y1 = clip_u8(s_lum[li[0]][2 * i] >> 7); Y2 = clip_u8(s_lum[li[0]][2 * i + 1] >> 7);
            if (uvalpha < 2048) { U = clip_u8(s_cu[ci[0]][i] >> 7); V = clip_u8(s_cv[ci[0]][i] >> 7); }
            else { U = clip_u8((s_cu[ci[0]][i] + s_cu[ci[NC > 1 ? 1 : 0]][i]) >> 8); V = clip_u8((s_cv[ci[0]][i] + s_cv[ci[NC > 1 ? 1 : 0]][i]) >> 8); }
        } else if (mode == 2) {   /* yuv2rgb_2_c_template :998-1041 (ls == cs == 2) */
            const int ya = lf[NL > 1 ? 1 : 0], ua = cf[NC > 1 ? 1 : 0], ya1 = 4096 - ya, ua1 = 4096 - ua;
            Y1 = clip_u8((s_lum[li[0]][2 * i] * ya1 + s_lum[li[NL > 1 ? 1 : 0]][2 * i] * ya) >> 19);
            Y2 = clip_u8((s_lum[li[0]][2 * i + 1] * ya1 + s_lum[li[NL > 1 ? 1 : 0]][2 * i + 1] * ya) >> 19);
            U = clip_u8((s_cu[ci[0]][i] * ua1 + s_cu[ci[NC > 1 ? 1 : 0]][i] * ua) >> 19);
            V = clip_u8((s_cv[ci[0]][i] * ua1 + s_cv[ci[NC > 1 ? 1 : 0]][i] * ua) >> 19);
        } else {                  /* yuv2rgb_X_c_template :937-996: clipped only if a value has bit 8 set */
            Y1 = Y2 = U = V = 1 << 18;
#pragma unroll
            for (int j = 0; j < NL; j++) {
                const uint32_t two = *reinterpret_cast<const uint32_t *>(&s_lum[li[j]][2 * i]);
                Y1 += (int16_t)(two & 0xFFFF) * lf[j]; Y2 += (int16_t)(two >> 16) * lf[j];
            }
#pragma unroll
            for (int j = 0; j < NC; j++) { U += s_cu[ci[j]][i] * cf[j]; V += s_cv[ci[j]][i] * cf[j]; }
            Y1 >>= 19; Y2 >>= 19; U >>= 19; V >>= 19;
            if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
        }
        write_pair(lut, &s_out[row - out_row0][i * 6], Y1, Y2, U, V);
    }
}

/* LDS of a workgroup, sized per context (sws_plan): the horizontal pass's results for the source lines a tile needs, the tables, and one
 * block shared by the staging lines (horizontal pass) and the output rows of tiles that cannot store from registers (after it). */
__host__ __device__ constexpr int sws_lds_bytes(int lum_lines, int chr_lines)
{
    return lum_lines * TW * 2 + 2 * chr_lines * (TW / 2) * 2 + (int)sizeof(LutLds) + STAGE_BYTES;
}

/* the sixteen tap-count instances of vertical_rows (taps in registers, rounded up to 1 / 2 / 4 / 8) */
#define MI355_VR(NL, NC) vertical_rows<NL, NC, MI355_VR_WIDE>(c, s_lut, s_lum, s_cu, s_cv, s_out, tid, y0, y1, llo, clo, npairs, mode, wide_dst, fr.dst_stride, r0)
#define MI355_VR_ALL \
    switch (bl * 4 + bc) { \
    case 0: MI355_VR(1, 1); break;   case 1: MI355_VR(1, 2); break;   case 2: MI355_VR(1, 4); break;   case 3: MI355_VR(1, 8); break; \
    case 4: MI355_VR(2, 1); break;   case 5: MI355_VR(2, 2); break;   case 6: MI355_VR(2, 4); break;   case 7: MI355_VR(2, 8); break; \
    case 8: MI355_VR(4, 1); break;   case 9: MI355_VR(4, 2); break;   case 10: MI355_VR(4, 4); break;  case 11: MI355_VR(4, 8); break; \
    case 12: MI355_VR(8, 1); break;  case 13: MI355_VR(8, 2); break;  case 14: MI355_VR(8, 4); break;  default: MI355_VR(8, 8); break; \
    }

/* Tiles that cannot store from registers (the picture's right edge, a destination that is not 8-byte aligned, filters of more than
 * eight taps): the rows go through s_out, OUT_ROWS at a time. */
__device__ __forceinline__ void vertical_narrow(const SwsDev *cp, const LutLds *lutp, const int16_t (*s_lum)[TW], const int16_t (*s_cu)[TW / 2],
                                                          const int16_t (*s_cv)[TW / 2], uint8_t (*s_out)[TW * 3], uint8_t *tile_dst, int dst_stride,
                                                          int x0, int y0, int y1, int llo, int clo)
{
    SwsDev c = *cp;
    c.vLumC = mi355_global(c.vLumC); c.vChrC = mi355_global(c.vChrC); c.vLumP = mi355_global(c.vLumP); c.vChrP = mi355_global(c.vChrP);
    const LutLds &s_lut = *lutp;
    const int tid = threadIdx.x, ls = c.vls, cs = c.vcs, mode = packed_mode(ls, cs);
    const int npairs = imin(TW, c.dstW - x0 + 1) >> 1;     /* (dstW + 1) >> 1 pairs in the picture */
    const int nbytes = imin(TW, c.dstW - x0) * 3, nrows_all = y1 - y0 + 1;
    const int bl = ls <= 1 ? 0 : (ls <= 2 ? 1 : (ls <= 4 ? 2 : 3)), bc = cs <= 1 ? 0 : (cs <= 2 ? 1 : (cs <= 4 ? 2 : 3));
    uint8_t *const wide_dst = nullptr;
    struct { int dst_stride; } fr{ dst_stride };
    for (int r0 = 0; r0 < nrows_all; r0 += OUT_ROWS) {
#ifndef MI355_SWS_NO_V
        if (ls <= 8 && cs <= 8) {
#define MI355_VR_WIDE false
            MI355_VR_ALL
#undef MI355_VR_WIDE
        } else
        for (int p = tid; p < OUT_ROWS * (TW / 2); p += NT) {
            const int row = r0 + (p >> 6), i = p & 63, gy = y0 + row;
            if (gy > y1 || i >= npairs) continue;
            TileRows R{ s_lum, s_cu, s_cv, imax(1 - ls, c.vLumP[gy]), llo, c.srcH - 1, imax(1 - cs, c.vChrP[gy]), clo, c.chrSrcH - 1 };
            int ya = 0, ua = 0;
            if (mode == 1) ua = cs == 1 ? 0 : c.vChrC[2 * gy + 1];
            else if (mode == 2) { ya = c.vLumC[2 * gy + 1]; ua = c.vChrC[2 * gy + 1]; }
            rgb_pair(s_lut, &s_out[row - r0][i * 6], R, i, mode, c.vLumC + (size_t)gy * ls, ls, c.vChrC + (size_t)gy * cs, cs, ya, ua);
        }
#endif
        __syncthreads();
        /* rows out: only samples below dstW (for odd dstW the reference also writes the phantom partner of
         * the last sample from uninitialised ring-buffer data; that sample is not reproduced) */
#ifndef MI355_SWS_NO_OUT
        {
            uint8_t *d0 = tile_dst + (size_t)r0 * fr.dst_stride;
            const int nrows = imin(OUT_ROWS, nrows_all - r0);
            const unsigned al = (unsigned)(uintptr_t)d0 | (unsigned)fr.dst_stride | (unsigned)nbytes;
            /* mi355_div20 below: idx < nrows * n with nrows <= 16 and n <= TW * 3 / 4 = 96: idx * n < 2^18 */
            if ((al & 15) == 0) {                     /* 16 bytes per thread and store */
                const int n = nbytes >> 4, inv = mi355_inv20(n);
                for (int idx = tid; idx < nrows * n; idx += NT) {
                    const int row = mi355_div20(idx, inv), k = idx - row * n;
                    reinterpret_cast<uint4 *>(d0 + (size_t)row * fr.dst_stride)[k] = reinterpret_cast<const uint4 *>(s_out[row])[k];
                }
            } else if ((al & 3) == 0) {
                const int n = nbytes >> 2, inv = mi355_inv20(n);
                for (int idx = tid; idx < nrows * n; idx += NT) {
                    const int row = mi355_div20(idx, inv), k = idx - row * n;
                    reinterpret_cast<uint32_t *>(d0 + (size_t)row * fr.dst_stride)[k] = reinterpret_cast<const uint32_t *>(s_out[row])[k];
                }
            } else {
                for (int row = 0; row < nrows; row++)
                    for (int k = tid; k < nbytes; k += NT) d0[(size_t)row * fr.dst_stride + k] = s_out[row][k];
            }
        }
#endif
        __syncthreads();                           /* the next pass overwrites s_out */
    }
}

/* LCAP / CCAP: source lines the LDS tile holds — the context's largest tile span picks the instance (sws_launch), and with it how
 * many workgroups a CU's 160 KB hold (one wave of each per SIMD); WAVES: the waves per SIMD the register allocation then aims at */
template <int LCAP, int CCAP, int WAVES>
#ifndef MI355_HIP_EMU_H
__attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
#endif
__global__ void __launch_bounds__(NT) k_sws_generic(const SwsDev *cp, const mi355_sws_frame *frames)
{
    __shared__ __attribute__((aligned(16))) int16_t s_lum[LCAP][TW];
    __shared__ __attribute__((aligned(16))) int16_t s_cu[CCAP][TW / 2], s_cv[CCAP][TW / 2];
    __shared__ LutLds s_lut;
    /* the staging lines of the horizontal pass; the output rows of tiles that cannot store from registers reuse them after it */
    __shared__ __attribute__((aligned(16))) uint8_t s_io[STAGE_BYTES];
    uint8_t (*s_out)[TW * 3] = reinterpret_cast<uint8_t (*)[TW * 3]>(s_io);
    SwsDev c = *cp;                                   /* pointers of the records: global address space (mi355_rt.h) */
    c.hLumC = mi355_global(c.hLumC); c.hChrC = mi355_global(c.hChrC); c.vLumC = mi355_global(c.vLumC); c.vChrC = mi355_global(c.vChrC);
    c.hLumP = mi355_global(c.hLumP); c.hChrP = mi355_global(c.hChrP); c.vLumP = mi355_global(c.vLumP); c.vChrP = mi355_global(c.vChrP);
    mi355_sws_frame fr = frames[blockIdx.z];
    for (int k = 0; k < 3; k++) fr.src[k] = mi355_global(fr.src[k]);
    fr.dst = mi355_global(fr.dst);
    const int tid = threadIdx.x, th = c.th;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * th, y1 = imin(y0 + th, c.dstH) - 1;
    const int ls = c.vls, cs = c.vcs;
    /* source lines this tile needs (swscale.c:459-468 for the first tap, :571-616 for the clamping) */
    const int lfirst0 = imax(1 - ls, c.vLumP[y0]), lfirst1 = imax(1 - ls, c.vLumP[y1]);
    const int cfirst0 = imax(1 - cs, c.vChrP[y0]), cfirst1 = imax(1 - cs, c.vChrP[y1]);
    const int llo = clampi(lfirst0, 0, c.srcH - 1), lhi = clampi(lfirst1 + ls - 1, 0, c.srcH - 1);
    const int clo = clampi(cfirst0, 0, c.chrSrcH - 1), chi = clampi(cfirst1 + cs - 1, 0, c.chrSrcH - 1);
    lut_load(s_lut, &cp->luts, tid, NT);
    /* horizontal pass: luma (the phantom partner of the last sample of an odd-width picture reads the
     * zero-initialised tail of the reference's line buffer, utils.c:1241-1262), then the chroma planes */
    uint32_t *s_stage = reinterpret_cast<uint32_t *>(s_io);
#ifndef MI355_SWS_NO_H
    hscale_tile<TW>(s_lum, fr.src[0], fr.src_stride[0], c.srcW, c.hLumP, c.hLumC, c.hls, x0, c.dstW, llo, lhi, s_stage, tid, true, c.hstage != 0, c.hident_l != 0);
    hscale_tile<TW / 2>(s_cu, fr.src[1], fr.src_stride[1], c.chrSrcW, c.hChrP, c.hChrC, c.hcs, x0 >> 1, c.chrDstW, clo, chi, s_stage, tid, false, c.hstage != 0, c.hident_c != 0);
    hscale_tile<TW / 2>(s_cv, fr.src[2], fr.src_stride[2], c.chrSrcW, c.hChrP, c.hChrC, c.hcs, x0 >> 1, c.chrDstW, clo, chi, s_stage, tid, false, c.hstage != 0, c.hident_c != 0);
#endif
    __syncthreads();
    /* vertical pass + LUT */
    const int mode = packed_mode(ls, cs);
    const int npairs = imin(TW, c.dstW - x0 + 1) >> 1;     /* (dstW + 1) >> 1 pairs in the picture */
    /* full tiles with an 8-byte aligned destination leave straight from registers; the others go through s_out, OUT_ROWS rows at a time */
    uint8_t *const tile_dst = fr.dst + (size_t)y0 * fr.dst_stride + (size_t)x0 * 3;
    uint8_t *const wide_dst = (ls <= 8 && cs <= 8 && c.dstW - x0 >= TW && ((reinterpret_cast<uintptr_t>(tile_dst) | (uintptr_t)fr.dst_stride) & 7) == 0)
                                  ? tile_dst : nullptr;
    /* tap counts in registers, rounded up to 1 / 2 / 4 / 8 */
    const int bl = ls <= 1 ? 0 : (ls <= 2 ? 1 : (ls <= 4 ? 2 : 3)), bc = cs <= 1 ? 0 : (cs <= 2 ? 1 : (cs <= 4 ? 2 : 3));
#ifndef MI355_SWS_NO_V
    if (wide_dst) {                               /* uniform over the workgroup: every row leaves from registers */
        const int r0 = 0;
#define MI355_VR_WIDE true
        MI355_VR_ALL
#undef MI355_VR_WIDE
        return;
    }
#endif
    vertical_narrow(cp, &s_lut, s_lum, s_cu, s_cv, s_out, tile_dst, fr.dst_stride, x0, y0, y1, llo, clo);
#undef MI355_VR_ALL
#undef MI355_VR
}

#ifndef MI355_C24_ROWS
#define MI355_C24_ROWS 16
#endif
#ifndef MI355_IDENT_ROWS
#define MI355_IDENT_ROWS 16
#endif
constexpr int C24_ROWS = MI355_C24_ROWS, C24_COLS = 512, IDENT_ROWS = MI355_IDENT_ROWS;      /* rows of a tile of k_sws_c24 / of k_sws_ident1 */
/* eight samples of one line: Y bytes in (y0, y1), the four pairs' LUT row offsets in r/g/b -> 24 RGB bytes */
__device__ __forceinline__ void c24_line(const LutLds &t, uint8_t *d, uint32_t y0, uint32_t y1, const int *r, const int *g, const int *b)
{
    int Y[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Y[k] = ((k < 4 ? y0 : y1) >> (8 * (k & 3))) & 0xFF;
    rgb24_store8(t, d, Y, r, g, b);
}
/* yuv2rgb_c_24_rgb (yuv2rgb.c:335-363): a block converts a 512 x 16 sample tile; a thread takes eight samples of two
 * lines per step (8-byte luma loads, 4-byte chroma loads, three 8-byte stores per line) — one (U,V) pair serves both
 * lines (LOADCHROMA :67-72, nearest chroma).  Unaligned planes and the right edge go pair by pair. */
__global__ void __launch_bounds__(NT) k_sws_c24(const mi355_sws_luts *luts, int dstW, int sliceH, int sliceY, const mi355_sws_frame *frames)
{
    __shared__ LutLds s_lut;
    const int tid = threadIdx.x;
    mi355_sws_frame fr = frames[blockIdx.z];
    for (int k = 0; k < 3; k++) fr.src[k] = mi355_global(fr.src[k]);
    fr.dst = mi355_global(fr.dst);
    const int x = blockIdx.x * C24_COLS + (tid & 63) * 8;   /* first of the thread's eight samples */
    const int npairs = dstW >> 1;                            /* pairs i < dstW >> 1 (8 + 4 + 2 sample groups, yuv2rgb.c:129-171) */
    const bool mine = (x >> 1) < npairs;
    const bool wide = mine && (x >> 1) + 4 <= npairs &&
                      ((reinterpret_cast<uintptr_t>(fr.src[0]) | (uintptr_t)fr.src_stride[0] | reinterpret_cast<uintptr_t>(fr.dst) | (uintptr_t)fr.dst_stride) & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(fr.src[1]) | (uintptr_t)fr.src_stride[1] | reinterpret_cast<uintptr_t>(fr.src[2]) | (uintptr_t)fr.src_stride[2]) & 3) == 0;
    /* the samples of both of the thread's line pairs are requested before the LUT copy below: the block pays one memory
     * round trip, not three (LUT, first pair, second pair) */
    constexpr int NR = C24_ROWS / 2 / (NT / 64);
    sws_u32x2 ya[NR], yc[NR];
    uint32_t u4[NR], v4[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int y = blockIdx.y * C24_ROWS + 2 * ((tid >> 6) + q * (NT / 64));
        ya[q] = yc[q] = sws_u32x2{ 0u, 0u }; u4[q] = v4[q] = 0;
        if (wide && y < sliceH) {
            const uint8_t *py1 = fr.src[0] + (size_t)y * fr.src_stride[0] + x;
            ya[q] = *reinterpret_cast<const sws_u32x2 *>(py1); yc[q] = *reinterpret_cast<const sws_u32x2 *>(py1 + fr.src_stride[0]);
            u4[q] = *reinterpret_cast<const uint32_t *>(fr.src[1] + (size_t)(y >> 1) * fr.src_stride[1] + (x >> 1));
            v4[q] = *reinterpret_cast<const uint32_t *>(fr.src[2] + (size_t)(y >> 1) * fr.src_stride[2] + (x >> 1));
        }
    }
    MI355_ISSUE_FENCE();
    lut_load(s_lut, luts, tid, NT);
    __syncthreads();
    if (!mine) return;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int rr = (tid >> 6) + q * (NT / 64);
        const int y = blockIdx.y * C24_ROWS + 2 * rr;
        if (y >= sliceH) break;
        const uint8_t *py1 = fr.src[0] + (size_t)y * fr.src_stride[0] + x, *py2 = py1 + fr.src_stride[0];
        const uint8_t *pu = fr.src[1] + (size_t)(y >> 1) * fr.src_stride[1] + (x >> 1), *pv = fr.src[2] + (size_t)(y >> 1) * fr.src_stride[2] + (x >> 1);
        uint8_t *d1 = fr.dst + (size_t)(y + sliceY) * fr.dst_stride + (size_t)x * 3, *d2 = d1 + fr.dst_stride;
        if (wide) {
            const sws_u32x2 a = ya[q], c = yc[q];
            int r[4], g[4], b[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int U = (u4[q] >> (8 * p)) & 0xFF, V = (v4[q] >> (8 * p)) & 0xFF;
                r[p] = s_lut.rV[V]; g[p] = s_lut.gU[U] + s_lut.gV[V]; b[p] = s_lut.bU[U];
            }
            c24_line(s_lut, d1, a[0], a[1], r, g, b);
            c24_line(s_lut, d2, c[0], c[1], r, g, b);
        } else {
            for (int p = 0; p < 4 && (x >> 1) + p < npairs; p++) {
                write_pair(s_lut, d1 + 6 * p, py1[2 * p], py1[2 * p + 1], pu[p], pv[p]);
                write_pair(s_lut, d2 + 6 * p, py2[2 * p], py2[2 * p + 1], pu[p], pv[p]);
            }
        }
    }
}

/* The generic scaler on a context that does not scale (round 6): identity horizontal filters (hident_l / hident_c) and ONE vertical luma tap — what swscale() runs for an
 * unscaled yuv420p -> rgb24 conversion that may not take the special converter (SWS_ACCURATE_RND): hScale8To15 is src << 7, and the vertical pass reads those 15-bit values of
 * one luma line and of the row's one to four chroma lines (bicubic: four taps on the chroma planes' half height).  The same integers k_sws_generic computes through its LDS tile
 * (vertical_rows), straight from the source bytes to the LUT:
 *   X false, yuv2rgb24_1_c (output.c:1043-1110; vChrFilterSize <= 2): Y = (src << 7) >> 7, U / V = the first chroma line's sample, or ((c0 << 7) + (c1 << 7)) >> 8 = the mean of
 *     the two lines when the row's second coefficient is >= 2048;
 *   X true, yuv2rgb24_X_c (:937-996; three or four chroma taps): Y = ((1 << 18) + (src << 7) * lumFilter[0]) >> 19, U / V = ((1 << 18) + sum (c_j << 7) * chrFilter[j]) >> 19,
 *     a pair's four values clipped only if one of them has bit 8 set.
 * 512 x 16 sample tiles, a thread eight samples of a line per step; the rows' table entries, then all the samples of a thread are requested before the LUT copy (two round
 * trips per workgroup instead of the tile's staging rounds and barriers).  Only for even dstW (the phantom partner of an odd width's last sample stays with k_sws_generic). */
template <bool X>
__global__ void __launch_bounds__(NT) k_sws_ident1(const SwsDev *cp, const mi355_sws_frame *frames)
{
    __shared__ LutLds s_lut;
    constexpr int NC = X ? 4 : 2;
    const int tid = threadIdx.x;
    const int dstW = cp->dstW, dstH = cp->dstH, cs = cp->vcs, srcH = cp->srcH, chrSrcH = cp->chrSrcH;
    const int32_t *vLumP = mi355_global(cp->vLumP), *vChrP = mi355_global(cp->vChrP);
    const int16_t *vLumC = mi355_global(cp->vLumC), *vChrC = mi355_global(cp->vChrC);
    mi355_sws_frame fr = frames[blockIdx.z];
    for (int k = 0; k < 3; k++) fr.src[k] = mi355_global(fr.src[k]);
    fr.dst = mi355_global(fr.dst);
    const int x = blockIdx.x * C24_COLS + (tid & 63) * 8;   /* first of the thread's eight samples */
    const int npairs = dstW >> 1;
    const bool mine = (x >> 1) < npairs;
    const bool wide = mine && (x >> 1) + 4 <= npairs &&
                      ((reinterpret_cast<uintptr_t>(fr.src[0]) | (uintptr_t)fr.src_stride[0] | reinterpret_cast<uintptr_t>(fr.dst) | (uintptr_t)fr.dst_stride) & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(fr.src[1]) | (uintptr_t)fr.src_stride[1] | reinterpret_cast<uintptr_t>(fr.src[2]) | (uintptr_t)fr.src_stride[2]) & 3) == 0;
    constexpr int NR = IDENT_ROWS / (NT / 64);
    /* the rows' lines and taps: every load unconditional (rows past the picture repeat its last row and are not written; a tap past the filter reads tap 0 and becomes zero) */
    int li[NR], lf[NR], c0[NR], cf[NR][NC];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int gy = imin(blockIdx.y * IDENT_ROWS + (tid >> 6) + q * (NT / 64), dstH - 1);
        li[q] = vLumP[gy]; lf[q] = vLumC[gy]; c0[q] = vChrP[gy];
#pragma unroll
        for (int j = 0; j < NC; j++) cf[q][j] = vChrC[(size_t)gy * cs + (j < cs ? j : 0)];
    }
    sws_u32x2 ya[NR];
    uint32_t u[NR][NC], v[NR][NC];
    int ci[NR][NC];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        li[q] = clampi(imax(0, li[q]), 0, srcH - 1);
        const int cfirst = imax(1 - cs, c0[q]);
        ya[q] = sws_u32x2{ 0u, 0u };
        if (wide) ya[q] = *reinterpret_cast<const sws_u32x2 *>(fr.src[0] + (size_t)li[q] * fr.src_stride[0] + x);
#pragma unroll
        for (int j = 0; j < NC; j++) {
            cf[q][j] = j < cs ? cf[q][j] : 0;
            ci[q][j] = clampi(cfirst + (j < cs ? j : 0), 0, chrSrcH - 1);
            u[q][j] = v[q][j] = 0;
            if (wide) {
                u[q][j] = *reinterpret_cast<const uint32_t *>(fr.src[1] + (size_t)ci[q][j] * fr.src_stride[1] + (x >> 1));
                v[q][j] = *reinterpret_cast<const uint32_t *>(fr.src[2] + (size_t)ci[q][j] * fr.src_stride[2] + (x >> 1));
            }
        }
    }
    MI355_ISSUE_FENCE();
    lut_load(s_lut, &cp->luts, tid, NT);
    __syncthreads();
    if (!mine) return;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int gy = blockIdx.y * IDENT_ROWS + (tid >> 6) + q * (NT / 64);
        if (gy >= dstH) break;
        uint8_t *d = fr.dst + (size_t)gy * fr.dst_stride + (size_t)x * 3;
        const bool mean = !X && cs > 1 && cf[q][1] >= 2048;
        /* one pair of the row from its bytes: two luma samples, the pair's chroma sample of each tap line */
        auto pair = [&](int y1, int y2, const int *us, const int *vs, int &Y1, int &Y2, int &U, int &V) {
            if (!X) {
                Y1 = y1; Y2 = y2;
                U = mean ? (us[0] + us[1]) >> 1 : us[0];
                V = mean ? (vs[0] + vs[1]) >> 1 : vs[0];
                return;
            }
            Y1 = ((1 << 18) + (y1 << 7) * lf[q]) >> 19; Y2 = ((1 << 18) + (y2 << 7) * lf[q]) >> 19;
            U = V = 1 << 18;
#pragma unroll
            for (int j = 0; j < NC; j++) { U += (us[j] << 7) * cf[q][j]; V += (vs[j] << 7) * cf[q][j]; }
            U >>= 19; V >>= 19;
            if ((Y1 | Y2 | U | V) & 0x100) { Y1 = clip_u8(Y1); Y2 = clip_u8(Y2); U = clip_u8(U); V = clip_u8(V); }
        };
        if (wide) {
            int Y[8], r[4], g[4], b[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                int us[NC], vs[NC], U, V;
#pragma unroll
                for (int j = 0; j < NC; j++) { us[j] = (u[q][j] >> (8 * p)) & 0xFF; vs[j] = (v[q][j] >> (8 * p)) & 0xFF; }
                const uint32_t w = ya[q][p >> 1] >> (16 * (p & 1));
                pair((int)(w & 0xFF), (int)((w >> 8) & 0xFF), us, vs, Y[2 * p], Y[2 * p + 1], U, V);
                r[p] = s_lut.rV[V]; g[p] = s_lut.gU[U] + s_lut.gV[V]; b[p] = s_lut.bU[U];
            }
            rgb24_store8(s_lut, d, Y, r, g, b);
        } else {
            const uint8_t *py = fr.src[0] + (size_t)li[q] * fr.src_stride[0] + x;
            for (int p = 0; p < 4 && (x >> 1) + p < npairs; p++) {
                int us[NC], vs[NC], Y1, Y2, U, V;
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    us[j] = fr.src[1][(size_t)ci[q][j] * fr.src_stride[1] + (x >> 1) + p];
                    vs[j] = fr.src[2][(size_t)ci[q][j] * fr.src_stride[2] + (x >> 1) + p];
                }
                pair(py[2 * p], py[2 * p + 1], us, vs, Y1, Y2, U, V);
                write_pair(s_lut, d + 6 * p, Y1, Y2, U, V);
            }
        }
    }
}

/* ---- Tier-1 line kernels ---------------------------------------------------------------------------- */
__global__ void __launch_bounds__(NT) k_sws_line_hscale(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *pos, int fs)
{
    for (int i = threadIdx.x; i < dstW; i += NT) dst[i] = (int16_t)hscale_one(src, filter + (size_t)i * fs, pos[i], fs);
}
/* yuv2planeX_8_c output.c:242-255 (fs >= 1 rows at `pitch` elements) / yuv2plane1_8_c :257-266 (fs == 0) */
__global__ void __launch_bounds__(NT) k_sws_line_plane(const int16_t *filter, int fs, const int16_t *rows, int pitch, uint8_t *dest, int dstW,
                                                       const uint8_t *dither, int offset)
{
    for (int i = threadIdx.x; i < dstW; i += NT) {
        if (fs == 0) { dest[i] = (uint8_t)clip_u8((rows[i] + dither[(i + offset) & 7]) >> 7); continue; }
        int val = dither[(i + offset) & 7] << 12;
        for (int j = 0; j < fs; j++) val += rows[(size_t)j * pitch + i] * filter[j];
        dest[i] = (uint8_t)clip_u8(val >> 19);
    }
}
struct PackedRows {
    const int16_t *l, *u, *v;
    int pitch;
    __device__ __forceinline__ int lum(int j, int x) const { return l[(size_t)j * pitch + x]; }
    __device__ __forceinline__ int cu(int j, int x) const { return u[(size_t)j * pitch + x]; }
    __device__ __forceinline__ int cv(int j, int x) const { return v[(size_t)j * pitch + x]; }
};
__global__ void __launch_bounds__(NT) k_sws_line_rgb(const mi355_sws_luts *luts, int mode, const int16_t *lumF, const int16_t *l, int ls,
                                                     const int16_t *chrF, const int16_t *u, const int16_t *v, int cs, int pitch, uint8_t *dest,
                                                     int dstW, int yalpha, int uvalpha)
{
    __shared__ LutLds s_lut;
    lut_load(s_lut, luts, threadIdx.x, NT);
    __syncthreads();
    PackedRows R{ l, u, v, pitch };
    for (int i = threadIdx.x; i < ((dstW + 1) >> 1); i += NT) rgb_pair(s_lut, dest + (size_t)i * 6, R, i, mode, lumF, ls, chrF, cs, yalpha, uvalpha);
}

}  // namespace

/* ---- context ------------------------------------------------------------------------------------------- */
struct mi355_sws_ctx {
    SwsDev h;
    SwsDev *d = nullptr;
    void *banks[8] = {};
    /* Tier-1 picture staging */
    uint8_t *d_src[3] = {}, *d_dst = nullptr;
    mi355_sws_frame *d_frame = nullptr;
    hipStream_t stream = nullptr;
    int device = -1;            /* the device of the thread that created the context: its entry points switch to it */
};

template <typename T> static const T *upload_bank(mi355_sws_ctx *c, int slot, const T *host, size_t n)
{
    if (!n || !host) return nullptr;
    void *p;
    MI355_CHECK(hipMalloc(&p, n * sizeof(T)));
    MI355_CHECK(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    c->banks[slot] = p;
    return static_cast<const T *>(p);
}

/* rows per tile: the largest power of two <= MAXTH for which no tile needs more source lines than the
 * LDS tile may hold; 0 if even single rows do not fit (filters larger than the tile: not supported).  lines[2]: the largest
 * luma / chroma span of a tile — what the context's LDS tile is sized for. */
static int choose_rows(const mi355_sws_desc *d, int lines[2])
{
    for (int th = MAXTH; th >= 1; th >>= 1) {
        bool ok = true;
        lines[0] = lines[1] = 1;
        for (int y0 = 0; y0 < d->dstH && ok; y0 += th) {
            const int y1 = (y0 + th < d->dstH ? y0 + th : d->dstH) - 1;
            auto span = [&](const mi355_sws_filter &f, int srcH) {
                int lo = f.pos[y0] > 1 - f.size ? f.pos[y0] : 1 - f.size, hi = (f.pos[y1] > 1 - f.size ? f.pos[y1] : 1 - f.size) + f.size - 1;
                for (int y = y0; y < y1; y++) if (f.pos[y + 1] < f.pos[y]) return 1 << 30;   /* not monotonic */
                lo = lo < 0 ? 0 : (lo > srcH - 1 ? srcH - 1 : lo);
                hi = hi < 0 ? 0 : (hi > srcH - 1 ? srcH - 1 : hi);
                return hi - lo + 1;
            };
            const int sl = span(d->vLum, d->srcH), sc = span(d->vChr, d->chrSrcH);
            ok = sl <= MAXL && sc <= MAXC;
            lines[0] = sl > lines[0] ? sl : lines[0];
            lines[1] = sc > lines[1] ? sc : lines[1];
        }
        if (ok) return th;
    }
    return 0;
}

extern "C" mi355_sws_ctx *mi355_sws_create(const mi355_sws_desc *desc)
{
    if (!bind()) { std::fprintf(stderr, "mi355dsp: mi355_sws_create without mi355_init(); no CPU fallback\n"); std::abort(); }
    mi355_sws_ctx *c = new mi355_sws_ctx;
    c->device = current_device();
    SwsDev &h = c->h;
    h.srcW = desc->srcW; h.srcH = desc->srcH; h.dstW = desc->dstW; h.dstH = desc->dstH;
    h.chrSrcW = desc->chrSrcW; h.chrSrcH = desc->chrSrcH; h.chrDstW = desc->chrDstW; h.special = desc->unscaled_special;
    h.hls = desc->hLum.size; h.hcs = desc->hChr.size; h.vls = desc->vLum.size; h.vcs = desc->vChr.size;
    h.luts = desc->luts;
    h.th = 0;
    h.lum_lines = h.chr_lines = 1;
    int lines[2] = { 1, 1 };
    h.hstage = 1;
    for (int i = 1; i < desc->hLum.n && desc->hLum.pos; i++) if (desc->hLum.pos[i] < desc->hLum.pos[i - 1]) h.hstage = 0;
    for (int i = 1; i < desc->hChr.n && desc->hChr.pos; i++) if (desc->hChr.pos[i] < desc->hChr.pos[i - 1]) h.hstage = 0;
    /* ... and every tap inside its line (the staged form does not clear what lies past the line's end) */
    for (int i = 0; i < desc->hLum.n && desc->hLum.pos; i++) if (desc->hLum.pos[i] < 0 || desc->hLum.pos[i] + desc->hLum.size > desc->srcW) h.hstage = 0;
    for (int i = 0; i < desc->hChr.n && desc->hChr.pos; i++) if (desc->hChr.pos[i] < 0 || desc->hChr.pos[i] + desc->hChr.size > desc->chrSrcW) h.hstage = 0;
    auto identity = [](const mi355_sws_filter &f, int src_w) {
        if (f.size != 1 || !f.coef || !f.pos || f.n > src_w) return 0;
        for (int i = 0; i < f.n; i++) if (f.coef[i] != 16384 || f.pos[i] != i) return 0;
        return 1;
    };
    h.hident_l = identity(desc->hLum, h.srcW);
    h.hident_c = identity(desc->hChr, h.chrSrcW);
    h.hLumC = h.hChrC = h.vLumC = h.vChrC = nullptr;
    h.hLumP = h.hChrP = h.vLumP = h.vChrP = nullptr;
    if (!h.special) {
        if (desc->hLum.n != h.dstW || desc->hChr.n != h.chrDstW || desc->vLum.n != h.dstH || desc->vChr.n != h.dstH ||
            h.hls < 1 || h.hcs < 1 || h.vls < 1 || h.vcs < 1 || !(h.th = choose_rows(desc, lines))) {
            std::fprintf(stderr, "mi355dsp: mi355_sws_create: filter banks do not fit this backend (sizes %d/%d/%d/%d)\n", h.hls, h.hcs, h.vls, h.vcs);
            delete c;
            return nullptr;
        }
        h.lum_lines = lines[0]; h.chr_lines = lines[1];
        h.hLumC = upload_bank(c, 0, desc->hLum.coef, (size_t)h.dstW * h.hls);    h.hLumP = upload_bank(c, 1, desc->hLum.pos, (size_t)h.dstW);
        h.hChrC = upload_bank(c, 2, desc->hChr.coef, (size_t)h.chrDstW * h.hcs); h.hChrP = upload_bank(c, 3, desc->hChr.pos, (size_t)h.chrDstW);
        h.vLumC = upload_bank(c, 4, desc->vLum.coef, (size_t)h.dstH * h.vls);    h.vLumP = upload_bank(c, 5, desc->vLum.pos, (size_t)h.dstH);
        h.vChrC = upload_bank(c, 6, desc->vChr.coef, (size_t)h.dstH * h.vcs);    h.vChrP = upload_bank(c, 7, desc->vChr.pos, (size_t)h.dstH);
    }
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&c->d), sizeof(SwsDev)));
    MI355_CHECK(hipMemcpy(c->d, &h, sizeof(SwsDev), hipMemcpyHostToDevice));
    MI355_CHECK(hipStreamCreate(&c->stream));
    return c;
}

extern "C" void mi355_sws_destroy(mi355_sws_ctx *c)
{
    if (!c) return;
    DeviceScope on(c->device);
    for (void *p : c->banks) if (p) MI355_CHECK(hipFree(p));
    for (uint8_t *p : c->d_src) if (p) MI355_CHECK(hipFree(p));
    if (c->d_dst) MI355_CHECK(hipFree(c->d_dst));
    if (c->d_frame) MI355_CHECK(hipFree(c->d_frame));
    if (c->d) MI355_CHECK(hipFree(c->d));
    if (c->stream) MI355_CHECK(hipStreamDestroy(c->stream));
    delete c;
}

/* waves per SIMD the three instances' register allocation aims at (developer switches; the defaults are what their LDS tiles allow) */
#ifndef MI355_SWS_WAVES_A
#define MI355_SWS_WAVES_A 8
#endif
#ifndef MI355_SWS_WAVES_B
#define MI355_SWS_WAVES_B 7
#endif
#ifndef MI355_SWS_WAVES_C
#define MI355_SWS_WAVES_C 6
#endif
extern "C" int mi355_sws_scale_frames_dev(mi355_sws_ctx *c, const mi355_sws_frame *d_frames, int nframes, void *stream)
{
    if (!c || !d_frames || nframes <= 0) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SwsDev &h = c->h;
    DeviceScope on(c->device);
    if (h.special) {
        hipLaunchKernelGGL(k_sws_c24, dim3((h.dstW + C24_COLS - 1) / C24_COLS, (h.srcH + C24_ROWS - 1) / C24_ROWS, nframes), dim3(NT), 0, s,
                           &c->d->luts, h.dstW, h.srcH, 0, d_frames);
    } else {
        /* a CU's 160 KB of LDS hold `wgs` workgroups, one wave of each per SIMD: the instance whose register budget matches */
        /* a context that does not scale: straight from the source bytes (k_sws_ident1; MI355_SWS_NO_IDENT1=1, developer switch: through the tile all the same) */
        static const bool no_ident1 = std::getenv("MI355_SWS_NO_IDENT1") != nullptr;
        if (!no_ident1 && h.hident_l && h.hident_c && h.vls == 1 && h.vcs <= 4 && !(h.dstW & 1) && h.srcW >= h.dstW && 2 * h.chrSrcW >= h.dstW) {
            const dim3 grid((h.dstW + C24_COLS - 1) / C24_COLS, (h.dstH + IDENT_ROWS - 1) / IDENT_ROWS, nframes);
            if (h.vcs <= 2) hipLaunchKernelGGL(k_sws_ident1<false>, grid, dim3(NT), 0, s, c->d, d_frames);       /* packed_mode() 1 */
            else hipLaunchKernelGGL(k_sws_ident1<true>, grid, dim3(NT), 0, s, c->d, d_frames);
            return hipGetLastError() == hipSuccess ? 0 : -2;
        }
        const dim3 grid((h.dstW + TW - 1) / TW, (h.dstH + h.th - 1) / h.th, nframes);
        static_assert(160 * 1024 / sws_lds_bytes(28, 16) >= 8 && 160 * 1024 / sws_lds_bytes(40, 20) == 7 && 160 * 1024 / sws_lds_bytes(MAXL, MAXC) == 6, "workgroups per CU of the instances");
        if (h.lum_lines <= 28 && h.chr_lines <= 16) hipLaunchKernelGGL((k_sws_generic<28, 16, MI355_SWS_WAVES_A>), grid, dim3(NT), 0, s, c->d, d_frames);
        else if (h.lum_lines <= 40 && h.chr_lines <= 20) hipLaunchKernelGGL((k_sws_generic<40, 20, MI355_SWS_WAVES_B>), grid, dim3(NT), 0, s, c->d, d_frames);
        else hipLaunchKernelGGL((k_sws_generic<MAXL, MAXC, MI355_SWS_WAVES_C>), grid, dim3(NT), 0, s, c->d, d_frames);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

/* copy a host plane into a tightly pitched device plane */
static bool plane_h2d(uint8_t *d, int dpitch, const uint8_t *h, int hstride, int wbytes, int rows, hipStream_t s)
{
    return hipMemcpy2DAsync(d, dpitch, h, hstride, wbytes, rows, hipMemcpyHostToDevice, s) == hipSuccess;
}

extern "C" int mi355_sws_scale(mi355_sws_ctx *c, const uint8_t *const src[3], const int src_stride[3], uint8_t *dst, int dst_stride)
{
    /* Every failure comes back as a negative value and leaves the context usable: the caller (contrib/libav/mi355_sws_glue.c) falls
     * back to the reference's function for that picture.  Strides must be positive and cover a line: sws_scale() itself also
     * takes negative ones (bottom-up pictures, vf_vflip) — not this entry point (-1), a 2-D copy has no negative pitch. */
    if (!c || !src || !src_stride || !dst) return -1;
    const SwsDev &h = c->h;
    const int cw = h.chrSrcW, ch = h.chrSrcH;
    const int w[3] = { h.srcW, cw, cw };
    const int out_w = h.special ? (h.dstW & ~1) * 3 : h.dstW * 3;      /* only the samples the converter writes go back: the caller's padding stays untouched */
    for (int p = 0; p < 3; p++) if (!src[p] || src_stride[p] < w[p]) return -1;
    if (dst_stride < out_w) return -1;
    DeviceScope on(c->device);
    const int pw[3] = { (h.srcW + 15) & ~15, (cw + 15) & ~15, (cw + 15) & ~15 }, ph[3] = { h.srcH, ch, ch };
    const int dpitch = (h.dstW * 3 + 15) & ~15;
    if (!c->d_dst) {
        uint8_t *ns[3] = { nullptr, nullptr, nullptr }, *nd = nullptr;
        mi355_sws_frame *nf = nullptr;
        bool ok = true;
        for (int p = 0; p < 3 && ok; p++) ok = hipMalloc(reinterpret_cast<void **>(&ns[p]), (size_t)pw[p] * ph[p] + 64) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void **>(&nd), (size_t)dpitch * (h.dstH + 1)) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void **>(&nf), sizeof(mi355_sws_frame)) == hipSuccess;
        if (ok) {
            mi355_sws_frame f;
            for (int p = 0; p < 3; p++) { f.src[p] = ns[p]; f.src_stride[p] = pw[p]; }
            f.dst = nd; f.dst_stride = dpitch;
            ok = hipMemcpy(nf, &f, sizeof(f), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (!ok) {
            for (int p = 0; p < 3; p++) if (ns[p]) (void)hipFree(ns[p]);
            if (nd) (void)hipFree(nd);
            if (nf) (void)hipFree(nf);
            (void)hipGetLastError();
            return -4;
        }
        for (int p = 0; p < 3; p++) c->d_src[p] = ns[p];
        c->d_dst = nd; c->d_frame = nf;
    }
    for (int p = 0; p < 3; p++)
        if (!plane_h2d(c->d_src[p], pw[p], src[p], src_stride[p], w[p], ph[p], c->stream)) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->stream); return -4; }
    if (mi355_sws_scale_frames_dev(c, c->d_frame, 1, c->stream) != 0) { (void)hipStreamSynchronize(c->stream); return -2; }
    if (hipMemcpy2DAsync(dst, dst_stride, c->d_dst, dpitch, out_w, h.dstH, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(c->stream); return -4; }
    if (hipStreamSynchronize(c->stream) != hipSuccess) return -4;
    return h.special ? h.srcH : h.dstH;
}

/* ---- Tier-1 line entry points ------------------------------------------------------------------------------ */
#define LAUNCH_LINE(kernel, a, ...) hipLaunchKernelGGL(kernel, dim3(1), dim3(NT), 0, (a).stream, __VA_ARGS__)

extern "C" void mi355_sws_hscale8to15(int16_t *dst, int dstW, const uint8_t *src, const int16_t *filter, const int32_t *filterPos, int filterSize)
{
    Arena &a = arena();
    int last = 0;
    for (int i = 0; i < dstW; i++) if (filterPos[i] > last) last = filterPos[i];
    const size_t nsrc = (size_t)last + filterSize;
    a.reserve(nsrc + (size_t)dstW * filterSize * 2 + (size_t)dstW * 6 + 64);
    const size_t o_src = a.take(nsrc), o_f = a.take((size_t)dstW * filterSize * 2), o_p = a.take((size_t)dstW * 4), o_d = a.take((size_t)dstW * 2);
    std::memcpy(a.h<uint8_t>(o_src), src, nsrc);
    std::memcpy(a.h<int16_t>(o_f), filter, (size_t)dstW * filterSize * 2);
    std::memcpy(a.h<int32_t>(o_p), filterPos, (size_t)dstW * 4);
    a.upload();
    LAUNCH_LINE(k_sws_line_hscale, a, a.d<int16_t>(o_d), dstW, a.d<const uint8_t>(o_src), a.d<const int16_t>(o_f), a.d<const int32_t>(o_p), filterSize);
    a.download();
    std::memcpy(dst, a.h<int16_t>(o_d), (size_t)dstW * 2);
}

static size_t pack_rows(Arena &a, const int16_t **rows, int n, int elems, int pitch)
{
    const size_t o = a.take((size_t)(n > 0 ? n : 1) * pitch * 2);
    for (int j = 0; j < n; j++) std::memcpy(a.h<int16_t>(o) + (size_t)j * pitch, rows[j], (size_t)elems * 2);
    return o;
}

static void plane_line(const int16_t *filter, int fs, const int16_t **rows, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    Arena &a = arena();
    const int n = fs ? fs : 1, pitch = (dstW + 7) & ~7;
    a.reserve((size_t)n * pitch * 2 + (size_t)n * 2 + (size_t)dstW + 128);
    const size_t o_r = pack_rows(a, rows, n, dstW, pitch), o_f = a.take((size_t)n * 2), o_di = a.take(8), o_d = a.take((size_t)dstW);
    if (fs) std::memcpy(a.h<int16_t>(o_f), filter, (size_t)fs * 2);
    std::memcpy(a.h<uint8_t>(o_di), dither, 8);
    a.upload();
    LAUNCH_LINE(k_sws_line_plane, a, a.d<const int16_t>(o_f), fs, a.d<const int16_t>(o_r), pitch, a.d<uint8_t>(o_d), dstW, a.d<const uint8_t>(o_di), offset);
    a.download();
    std::memcpy(dest, a.h<uint8_t>(o_d), (size_t)dstW);
}
extern "C" void mi355_sws_yuv2planeX_8(const int16_t *filter, int filterSize, const int16_t **src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    plane_line(filter, filterSize, src, dest, dstW, dither, offset);
}
extern "C" void mi355_sws_yuv2plane1_8(const int16_t *src, uint8_t *dest, int dstW, const uint8_t *dither, int offset)
{
    plane_line(nullptr, 0, &src, dest, dstW, dither, offset);
}

static void rgb_line(const mi355_sws_luts *luts, int mode, const int16_t *lumF, const int16_t **l, int ls, const int16_t *chrF,
                     const int16_t **u, const int16_t **v, int cs, uint8_t *dest, int dstW, int yalpha, int uvalpha)
{
    Arena &a = arena();
    const int npair = (dstW + 1) >> 1, pitch = (2 * npair + 7) & ~7;
    a.reserve(sizeof(mi355_sws_luts) + (size_t)(ls + 2 * cs + 3) * pitch * 2 + (size_t)(ls + cs) * 2 + (size_t)npair * 6 + 256);
    const size_t o_t = a.take(sizeof(mi355_sws_luts));
    std::memcpy(a.h<uint8_t>(o_t), luts, sizeof(mi355_sws_luts));
    const size_t o_l = pack_rows(a, l, ls, 2 * npair, pitch), o_u = pack_rows(a, u, cs, npair, pitch), o_v = pack_rows(a, v, cs, npair, pitch);
    const size_t o_lf = a.take((size_t)ls * 2 + 2), o_cf = a.take((size_t)cs * 2 + 2), o_d = a.take((size_t)npair * 6);
    if (lumF) std::memcpy(a.h<int16_t>(o_lf), lumF, (size_t)ls * 2);
    if (chrF) std::memcpy(a.h<int16_t>(o_cf), chrF, (size_t)cs * 2);
    a.upload();
    LAUNCH_LINE(k_sws_line_rgb, a, a.d<const mi355_sws_luts>(o_t), mode, a.d<const int16_t>(o_lf), a.d<const int16_t>(o_l), ls, a.d<const int16_t>(o_cf),
                a.d<const int16_t>(o_u), a.d<const int16_t>(o_v), cs, pitch, a.d<uint8_t>(o_d), dstW, yalpha, uvalpha);
    a.download();
    std::memcpy(dest, a.h<uint8_t>(o_d), (size_t)npair * 6);   /* like the reference: whole pairs, also for odd dstW */
}
extern "C" void mi355_sws_yuv2rgb24_X(const mi355_sws_luts *luts, const int16_t *lumFilter, const int16_t **lumSrc, int lumFilterSize,
                                      const int16_t *chrFilter, const int16_t **chrUSrc, const int16_t **chrVSrc, int chrFilterSize,
                                      uint8_t *dest, int dstW)
{
    rgb_line(luts, 0, lumFilter, lumSrc, lumFilterSize, chrFilter, chrUSrc, chrVSrc, chrFilterSize, dest, dstW, 0, 0);
}
extern "C" void mi355_sws_yuv2rgb24_2(const mi355_sws_luts *luts, const int16_t *buf[2], const int16_t *ubuf[2], const int16_t *vbuf[2],
                                      uint8_t *dest, int dstW, int yalpha, int uvalpha)
{
    rgb_line(luts, 2, nullptr, buf, 2, nullptr, ubuf, vbuf, 2, dest, dstW, yalpha, uvalpha);
}
extern "C" void mi355_sws_yuv2rgb24_1(const mi355_sws_luts *luts, const int16_t *buf0, const int16_t *ubuf[2], const int16_t *vbuf[2],
                                      uint8_t *dest, int dstW, int uvalpha)
{
    /* ubuf[1]/vbuf[1] are only looked at when uvalpha >= 2048 (output.c:1052, :1079) */
    const int cs = uvalpha < 2048 ? 1 : 2;
    rgb_line(luts, 1, nullptr, &buf0, 1, nullptr, ubuf, vbuf, cs, dest, dstW, 0, uvalpha);
}

extern "C" int mi355_sws_yuv2rgb_c_24_rgb(const mi355_sws_luts *luts, int dstW, const uint8_t *const src[3], const int srcStride[3],
                                          int srcSliceY, int srcSliceH, uint8_t *dst, int dstStride)
{
    if (!ready()) { std::fprintf(stderr, "mi355dsp: mi355_sws_yuv2rgb_c_24_rgb without mi355_init(); no CPU fallback\n"); std::abort(); }
    /* a slice may be a whole picture: device buffers per call instead of the staging arena */
    const int rows = srcSliceH, cw = dstW >> 1, crow = (rows + 1) >> 1;   /* even slices, as the reference's callers guarantee */
    const int pw[3] = { (dstW + 15) & ~15, (cw + 15) & ~15, (cw + 15) & ~15 }, ph[3] = { rows, crow, crow }, w[3] = { dstW & ~1, cw, cw };
    const int dpitch = (dstW * 3 + 15) & ~15;
    hipStream_t s = arena().stream;
    mi355_sws_frame f, *d_f;
    uint8_t *d_l, *d_dst;
    for (int p = 0; p < 3; p++) {
        uint8_t *d;
        MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&d), (size_t)pw[p] * (ph[p] + 1) + 64));
        plane_h2d(d, pw[p], src[p], srcStride[p], w[p], ph[p], s);
        f.src[p] = d; f.src_stride[p] = pw[p];
    }
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&d_dst), (size_t)dpitch * (rows + 1)));
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&d_f), sizeof(f)));
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&d_l), sizeof(mi355_sws_luts)));
    f.dst = d_dst; f.dst_stride = dpitch;
    MI355_CHECK(hipMemcpyAsync(d_f, &f, sizeof(f), hipMemcpyHostToDevice, s));
    MI355_CHECK(hipMemcpyAsync(d_l, luts, sizeof(mi355_sws_luts), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sws_c24, dim3((dstW + C24_COLS - 1) / C24_COLS, (rows + C24_ROWS - 1) / C24_ROWS, 1), dim3(NT), 0, s,
                       reinterpret_cast<const mi355_sws_luts *>(d_l), dstW, rows, 0, d_f);
    MI355_CHECK(hipMemcpy2DAsync(dst + (ptrdiff_t)srcSliceY * dstStride, dstStride, d_dst, dpitch, (dstW & ~1) * 3, rows, hipMemcpyDeviceToHost, s));
    MI355_CHECK(hipStreamSynchronize(s));
    for (int p = 0; p < 3; p++) MI355_CHECK(hipFree(const_cast<uint8_t *>(f.src[p])));
    MI355_CHECK(hipFree(d_dst)); MI355_CHECK(hipFree(d_f)); MI355_CHECK(hipFree(d_l));
    return srcSliceH;
}
