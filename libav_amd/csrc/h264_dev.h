/*
 * h264_dev.h — wave-cooperative H.264 reconstruction / deblocking building blocks.
 *
 * Execution model used by every kernel in this library: ONE 64-lane wavefront per
 * workgroup, one macroblock (or one DSP call) per wavefront, all staging in that
 * wave's own LDS.  Ordering points are MI355_WAVE_SYNC() (below): wave scope, no drain of the
 * loads and stores in flight — a kernel may keep the next macroblock's loads outstanding across them.  Every
 * function here must be called by all 64 lanes with wave-uniform arguments unless
 * it is marked "per lane".
 *
 * Arithmetic follows SURVEY.md §8(a) rows a1-a10 (reference file:line quoted at
 * each function); results are bit-exact with the reference C path.
 */
#ifndef MI355_H264_DEV_H
#define MI355_H264_DEV_H

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mi355 {

__device__ __forceinline__ int clip_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return (a + f) - 5 * (b + e) + 20 * (c + d); }
#if defined(MI355_HIP_EMU_H) || defined(MI355_PLAIN_LANE)
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
#else
/* Opaque to the optimiser at every call: values derived from the lane number are recomputed in each phase of a kernel
 * instead of being computed once and kept in registers across all of them.  +26 VALU per macroblock in k_recon_inter,
 * but 41 instead of 91 VGPRs, i.e. 8 instead of 5 waves per SIMD: 13.75 vs 15.94 ms (profiles/r02_experiments.md). */
__device__ __forceinline__ int lane_id() { int l = (int)(threadIdx.x & 63); asm volatile("" : "+v"(l)); __builtin_assume(l >= 0 && l < 64); return l; }
#endif
/* exchange inside groups of four lanes as a DPP operand modifier (quad_perm) instead of an LDS-routed shuffle */
#ifdef MI355_HIP_EMU_H
static inline int quad_xor1(int v) { return __shfl_xor(v, 1); }
static inline int quad_xor2(int v) { return __shfl_xor(v, 2); }
#else
__device__ __forceinline__ int quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }   /* quad_perm:[1,0,3,2] */
__device__ __forceinline__ int quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }   /* quad_perm:[2,3,0,1] */
#endif
/* ---- single-instruction primitives of the loop filter (device) and their plain-C meaning (emulator) ---------------
 * absdiff8: |a - b| of two values in 0..255 (v_sad_u16);  med3i: clamp x to [lo, hi], lo <= hi (v_med3_i32);
 * byte_perm: v_perm_b32 — byte i of the result is byte sel_i of {s1 (0..3), s0 (4..7)}, 0x0C gives 0x00;
 * quad_bcast<K>: the value lane K of this lane's group of four holds (DPP quad_perm, no LDS);
 * pk_absdiff_far: true when two packed (x, y) int16 vectors differ by >= 4 in a component (check_mv's
 * `abs(a - b) >= 4`, h264_loopfilter.c:442-470); the subtraction saturates, so extreme vectors cannot wrap to "near". */
#ifdef MI355_HIP_EMU_H
static inline int absdiff8(int a, int b) { return a > b ? a - b : b - a; }
static inline int med3i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline uint32_t byte_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
    const uint64_t src = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t c = (sel >> (8 * i)) & 0xFF;
        const uint32_t b = c < 8 ? (uint32_t)(src >> (8 * c)) & 0xFF : (c == 0x0C ? 0u : 0xFFu);
        r |= b << (8 * i);
    }
    return r;
}
template <int K> static inline int quad_bcast(int v) { return __shfl(v, ((int)(threadIdx.x & 63) & ~3) | K); }
static inline bool pk_absdiff_far(uint32_t a, uint32_t b, uint32_t mask = 0xFFFCFFFCu)      /* mask 0xFFFEFFFC: the vertical limit is 2 (field pictures) */
{
    const int dx = (int16_t)(a & 0xFFFF) - (int16_t)(b & 0xFFFF), dy = (int16_t)(a >> 16) - (int16_t)(b >> 16);
    return (dx < 0 ? -dx : dx) >= 4 || (dy < 0 ? -dy : dy) >= ((mask >> 16) == 0xFFFE ? 2 : 4);
}
#else
__device__ __forceinline__ int absdiff8(int a, int b) { return (int)__builtin_amdgcn_sad_u16((unsigned)a, (unsigned)b, 0u); }
__device__ __forceinline__ int med3i(int x, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ uint32_t byte_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
template <int K> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, true); }
__device__ __forceinline__ bool pk_absdiff_far(uint32_t a, uint32_t b, uint32_t mask = 0xFFFCFFFCu)
{
    typedef short v2s __attribute__((ext_vector_type(2)));
    const v2s d = __builtin_elementwise_sub_sat(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b));
    const v2s n = (v2s)((short)0) - d;
    const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(d, n));    /* |d| per half, <= 32767 (d = -32768 gives -32768: still >= 4 below) */
    return (m & mask) != 0;
}
#endif

/* a * b + c for operands known to fit 24 bits (v_mad_i32_i24: one instruction, no 32-bit multiply sequence) */
#ifdef MI355_HIP_EMU_H
static inline int mad24i(int a, int b, int c) { return a * b + c; }
#else
__device__ __forceinline__ int mad24i(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#endif

/* the value lane `l` (a literal) of the wave holds, as a scalar */
#ifdef MI355_HIP_EMU_H
static inline int lane_value(int v, int l) { return __shfl(v, l); }
#else
__device__ __forceinline__ int lane_value(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
#endif

/* a value every lane of the wave holds identically (read from this wave's LDS record): telling the
 * compiler moves everything derived from it to the scalar unit */
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* Ordering point for a workgroup that is exactly one wavefront.  __syncthreads() is a workgroup-scope
 * release/acquire: it drains every outstanding global load and store (s_waitcnt vmcnt(0)), which
 * serialises prefetches with the work they were meant to overlap.  A single wave executes its LDS and
 * its vector-memory instructions in issue order, so wavefront-scope fences (no instructions, compiler
 * ordering only) are sufficient; data dependences still get their own precise waits.
 * The SIMT emulator runs lanes as fibers and needs a real rendezvous (of the wave, like the real thing). */
#ifdef MI355_HIP_EMU_H
#define MI355_WAVE_SYNC() ((void)__shfl(0, 0))   /* a rendezvous of this wave's lanes only (a workgroup may hold several waves) */
#else
#define MI355_WAVE_SYNC()                                        \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
#endif

/* A picture plane in HBM; reads outside [0,w)x[0,h) replicate the border, which is
 * exactly what the reference's emulated_edge_mc produces (videodsp_template.c:24-96,
 * callers h264_mb.c:239-314). */
struct PlaneRef {
    const uint8_t *base;
    int stride, w, h;
};
__device__ __forceinline__ int px_clamped(const PlaneRef &p, int x, int y)
{
    x = clip3(x, 0, p.w - 1);
    y = clip3(y, 0, p.h - 1);
    return p.base[(size_t)y * p.stride + x];
}

/* developer instrumentation (-DMI355_PROF): RPROF(i) adds the shader cycles since the previous mark of this wave to
 * g_rprof[i]; only waves of the first blocks report (tools/prof_recon.sh) */
#if defined(MI355_PROF) && !defined(MI355_HIP_EMU_H)
__device__ unsigned long long g_rprof[16];
__shared__ unsigned long long rprof_last;
#define RPROF(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536u) { atomicAdd(&g_rprof[i], now_ - rprof_last); rprof_last = now_; } } while (0)
#define RPROF_START() do { if ((threadIdx.x & 63) == 0) rprof_last = __builtin_readcyclecounter(); } while (0)
#else
#define RPROF(i) do { } while (0)
#define RPROF_START() do { } while (0)
#endif

/* ---- per-wave LDS scratch -------------------------------------------------- */
/* Reference windows are staged ALIGNED TO THE BLOCK: luma window byte b of row r is picture sample
 * (ix - 4 + b, iy - 2 + r), so block column 0 sits on a dword boundary and every 4-sample segment
 * of the block reads whole dwords; chroma window byte b of row r is sample (cx + b, cy + r). */
constexpr int WY_DW = 12;               /* luma window row pitch in dwords: the window itself is 6 dwords (24 bytes = columns -4..19 of the
                                           block); the rest of a row takes the surplus of stage_windows_tiled's whole-piece writes */
constexpr int WC_DW = 4;                /* chroma window row pitch: the window is 3 dwords (columns 0..11) */
constexpr int WY_PAD = 10, WC_PAD = 4;  /* dwords in front of row 0 that those writes may reach */
struct __attribute__((aligned(8))) McScratch {
    uint32_t padY[WY_PAD];
    uint32_t winY[21 * WY_DW];          /* up to 21 rows (16 + 5) */
    uint32_t padC[WC_PAD];
    uint32_t winC[2][9 * WC_DW + WC_PAD];   /* the tail of plane 0 is the front pad of plane 1 */
    int16_t tmp[21 * 16];               /* unclipped horizontal 6-tap sums for the centre position */
};
static_assert(offsetof(McScratch, winY) % 8 == 0 && offsetof(McScratch, winC) % 8 == 0 && (WY_DW % 2) == 0, "8-byte stores into the windows");

#ifdef MI355_HIP_EMU_H
static inline uint32_t mi355_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (s & 3))); }
#else
__device__ __forceinline__ uint32_t mi355_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
#endif

/* One window's worth of loads for this lane, issued before anything waits on them.  The window starts
 * at picture column x0 (any alignment) and is `ndw` dwords wide, `wh` rows high.
 * Fast path (every dword the realignment touches lies inside the plane, rows 4-byte aligned): two
 * aligned dword loads + v_alignbyte.  Slow path: per-sample clamped reads (== emulated_edge_mc,
 * videodsp_template.c:24-96). */
/* One window's worth of loads for this lane, issued before anything waits on them.  The window starts
 * at picture column x0 (any alignment) and is `ndw` (<= 8) dwords wide, `wh` rows high.  A row takes 4 or
 * 8 lanes (some idle) so that (row, dword) of a lane are shifts and masks.
 * Fast path (every dword the realignment touches lies inside the plane, rows 4-byte aligned): two
 * aligned dword loads + v_alignbyte.  Slow path: per-sample clamped reads (== emulated_edge_mc,
 * videodsp_template.c:24-96). */
template <int MAXIT>
struct WinLoad {
    uint32_t lo[MAXIT], hi[MAXIT];
    uint32_t shift;
    bool whole;              /* loaded by the unpredicated path: commit() may write from every lane */
    __device__ __forceinline__ void issue(const PlaneRef &ref, int x0, int y0, int ndw, int wh, bool inside, int lane)
    {
        shift = (uint32_t)x0 & 3;
        const int xa = x0 & ~3, sh = ndw > 4 ? 3 : 2, dw = lane & ((1 << sh) - 1), r0 = lane >> sh;
        whole = inside;
        if (inside) {
            /* no lane is switched off: lanes past the window's last row / dword repeat the last one (same address, same
             * value, same LDS slot in commit()) — a v_min instead of an exec-mask sequence per access */
            const int dwc = dw < ndw ? dw : ndw - 1;
            const uint8_t *base = ref.base + (ptrdiff_t)y0 * ref.stride + xa + 4 * dwc;
#pragma unroll
            for (int k = 0; k < MAXIT; k++) {
                const int row = r0 + k * (64 >> sh), rc = row < wh ? row : wh - 1;
                const uint32_t *p = reinterpret_cast<const uint32_t *>(base + __mul24(rc, ref.stride));
                lo[k] = p[0];
                hi[k] = p[1];
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < MAXIT; k++) {
            const int row = r0 + k * (64 >> sh);
            lo[k] = hi[k] = 0;
            if (row >= wh || dw >= ndw) continue;
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) w |= (uint32_t)px_clamped(ref, x0 + 4 * dw + b, y0 + row) << (8 * b);
            lo[k] = w;
        }
        shift = 0;
    }
    __device__ __forceinline__ void commit(uint32_t *win, int pitch_dw, int ndw, int wh, int lane) const
    {
        const int sh = ndw > 4 ? 3 : 2, dw = lane & ((1 << sh) - 1), r0 = lane >> sh;
        if (whole) {
            const int dwc = dw < ndw ? dw : ndw - 1;
#pragma unroll
            for (int k = 0; k < MAXIT; k++) {
                const int row = r0 + k * (64 >> sh), rc = row < wh ? row : wh - 1;
                win[rc * pitch_dw + dwc] = mi355_alignbyte(hi[k], lo[k], shift);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < MAXIT; k++) {
            const int row = r0 + k * (64 >> sh);
            if (row < wh && dw < ndw) win[row * pitch_dw + dw] = mi355_alignbyte(hi[k], lo[k], shift);
        }
    }
};
__device__ __forceinline__ bool win_inside(const PlaneRef &ref, int x0, int y0, int ndw, int wh)
{
    const int xa = x0 & ~3;
    return xa >= 0 && y0 >= 0 && xa + 4 * (ndw + 1) <= ref.w && y0 + wh <= ref.h &&
           ((reinterpret_cast<uintptr_t>(ref.base) | (uintptr_t)ref.stride) & 3) == 0;
}
__device__ __forceinline__ int luma_win_dw(int bw) { return (4 + bw + 3 + 3) >> 2; }      /* columns -4..bw+2 */
__device__ __forceinline__ int chroma_win_dw(int cw) { return (cw + 1 + 3) >> 2; }          /* columns 0..cw */

/* Stage the luma window of a bw x bh block at integer position (ix,iy) and, if `cb`/`cr` are given, the
 * chroma windows of the cw x ch block at (cx,cy): all loads are in flight together, one LDS barrier
 * at the end. */
__device__ inline void stage_windows(McScratch &s, const PlaneRef *y, int ix, int iy, int bw, int bh,
                                     const PlaneRef *cb, const PlaneRef *cr, int cx, int cy, int cw, int ch)
{
    const int lane = lane_id();
    WinLoad<3> ly;          /* 21 rows x 8 lanes */
    WinLoad<1> lb, lr;      /* 9 rows x 4 lanes */
    const int ydw = luma_win_dw(bw), cdw = chroma_win_dw(cw);
    if (y) ly.issue(*y, ix - 4, iy - 2, ydw, bh + 5, win_inside(*y, ix - 4, iy - 2, ydw, bh + 5), lane);
    if (cb) {
        const bool in_c = win_inside(*cb, cx, cy, cdw, ch + 1);
        lb.issue(*cb, cx, cy, cdw, ch + 1, in_c, lane);
        lr.issue(*cr, cx, cy, cdw, ch + 1, in_c, lane);
    }
    if (y) ly.commit(s.winY, WY_DW, ydw, bh + 5, lane);
    if (cb) { lb.commit(s.winC[0], WC_DW, cdw, ch + 1, lane); lr.commit(s.winC[1], WC_DW, cdw, ch + 1, lane); }
    MI355_WAVE_SYNC();
}

/* The same for a whole 16x16 macroblock partition (the common shape) with every size a literal: the 21 x 24-byte luma
 * window goes out as 63 eight-byte pieces (one round of loads: three dwords per lane, realigned with two
 * v_alignbyte), the two 9 x 12-byte chroma windows as 36 pieces in a second round, all five loads in flight
 * before the first LDS write.  issue() only starts the loads (a caller may keep them in flight across other work:
 * k_recon_inter requests the next macroblock's windows before it computes the current one), commit() moves them into
 * the LDS windows.  Only for windows that lie inside the picture (win16_inside); others take the clamped generic path. */
struct Win16 {
    uint32_t a0, a1, a2, c0, c1, c2;
    __device__ __forceinline__ void issue(const uint8_t *yb, const uint8_t *cbb, const uint8_t *crb, int ystride, int cstride, int ix, int iy, int cx, int cy)
    {
        const int lane = lane_id();
        const int x0 = ix - 4, y0 = iy - 2;
        /* luma: lane t -> row t / 3, piece t % 3 (lane 63 repeats the last piece) */
        const int t = lane < 63 ? lane : 62;
        /* offsets in 32 bits through 24-bit multiplies: a 64-bit multiply-add per lane costs four issue slots */
        const int row = (int)(__umul24((unsigned)t, 43u) >> 7), piece = t - 3 * row;
        const uint32_t *ly = reinterpret_cast<const uint32_t *>(yb + (uint32_t)(__mul24(y0 + row, ystride) + (x0 & ~3) + 8 * piece));
        a0 = ly[0]; a1 = ly[1]; a2 = ly[2];
        /* chroma: lane u < 36 -> plane u / 18, row (u % 18) / 2, piece u & 1: piece 0 = bytes 0..7 (source dwords 0..2),
         * piece 1 = bytes 8..11 (source dwords 2, 3) */
        const int u = lane < 36 ? lane : 35;
        const int plane = u >= 18, r2 = u - 18 * plane, crow = r2 >> 1, cpiece = r2 & 1;
        const uint8_t *cbase = (plane ? crb : cbb) + (uint32_t)(__mul24(cy + crow, cstride) + (cx & ~3));
        const uint32_t *lc = reinterpret_cast<const uint32_t *>(cbase + 8 * cpiece);
        c0 = lc[0]; c1 = lc[1]; c2 = *reinterpret_cast<const uint32_t *>(cbase + 8);
    }
    __device__ __forceinline__ void commit(McScratch &s, int ix, int cx) const
    {
        const int lane = lane_id();
        const int t = lane < 63 ? lane : 62;
        const int row = (int)(__umul24((unsigned)t, 43u) >> 7), piece = t - 3 * row;
        const int u = lane < 36 ? lane : 35;
        const int plane = u >= 18, r2 = u - 18 * plane, crow = r2 >> 1, cpiece = r2 & 1;
        const uint32_t shy = (uint32_t)(ix - 4) & 3, shc = (uint32_t)cx & 3;
        typedef uint32_t u32x2 __attribute__((vector_size(8)));
        *reinterpret_cast<u32x2 *>(&s.winY[__umul24((unsigned)row, WY_DW) + 2u * (unsigned)piece]) = u32x2{ mi355_alignbyte(a1, a0, shy), mi355_alignbyte(a2, a1, shy) };
        uint32_t *wc = &s.winC[plane][__umul24((unsigned)crow, WC_DW)];
        const uint32_t v0 = mi355_alignbyte(c1, c0, shc), v1 = mi355_alignbyte(c2, c1, shc);
        if (cpiece) wc[2] = v0;
        else { wc[0] = v0; wc[1] = v1; }
    }
};
__device__ __forceinline__ bool win16_inside(const PlaneRef &y, int ix, int iy, const PlaneRef &cb, int cx, int cy)
{
    return win_inside(y, ix - 4, iy - 2, 6, 21) && win_inside(cb, cx, cy, 3, 9);
}
__device__ inline void stage_windows16(McScratch &s, const PlaneRef &y, int ix, int iy, const PlaneRef &cb, const PlaneRef &cr, int cx, int cy)
{
    if (!win16_inside(y, ix, iy, cb, cx, cy)) {
        stage_windows(s, &y, ix, iy, 16, 16, &cb, &cr, cx, cy, 8, 8);
        return;
    }
    Win16 w;
    w.issue(y.base, cb.base, cr.base, y.stride, cb.stride, ix, iy, cx, cy);
    RPROF(2);
    w.commit(s, ix, cx);
    MI355_WAVE_SYNC();
    RPROF(3);
}

/* ---- reference windows from MACROBLOCK-TILED surfaces (mi355_h264_frame.surface_layout == MI355_SURFACE_TILED) ------------
 * Luma macroblock (x, y) = 256 bytes at y * ypitch + x * 256 (16 rows of 16), chroma macroblock = 128 bytes at
 * y * cpitch + x * 128 (8 rows of 8 Cb, then 8 rows of 8 Cr).  The (bh + 5) x (bw + 5) luma window spans at most three tiles
 * in a row: it is fetched as 16-byte row pieces — lane 3 r + p piece p of row r, one global_load_dwordx4 each, the whole
 * window 6-12 cache lines instead of one (or two) per row.  Rows clamp to the picture by clamping the row number of the
 * address (emulated_edge_mc's vertical replication, videodsp_template.c:24-96, for free).
 *
 * A lane turns its piece straight into the four dwords of the block-aligned window it covers: with o = the window's first
 * column - the first column fetched (xb, a multiple of 16), byte shift o & 3, the lane funnel-shifts its four dwords and the
 * first dword of the NEXT piece — the neighbouring lane's, one DPP move — and writes them to window dwords 4 p - (o >> 2) ..
 * + 3 of its row.  The three lanes of a row write twelve consecutive dwords of which the window is six: rows are 12 dwords
 * apart (WY_DW) and the surplus lands in the gap behind the previous row's window or in front of the next one's (WY_PAD in
 * front of row 0).  No second pass, no predication.  The two chroma windows the same way: 8-byte pieces, two per row and
 * plane (lane 18 plane + 2 row + piece), rows 4 dwords apart.
 *
 * A window that reaches over the left or right picture border (columns are replicated there) takes the per-sample path with
 * clamped coordinates: the macroblocks at the picture's sides whose vectors point outwards. */
struct TiledRef {
    const uint8_t *y, *c;
    int ypitch, cpitch;          /* bytes per macroblock row of tiles */
    int mbw, mbh;
};
typedef uint32_t mi355_raw_u32x4 __attribute__((vector_size(16)));
typedef uint32_t mi355_raw_u32x2 __attribute__((vector_size(8)));
typedef uint32_t mi355_raw_u32x2a4 __attribute__((vector_size(8), aligned(4)));
/* the value the next lane holds (lane 63: unspecified) */
#ifdef MI355_HIP_EMU_H
static inline uint32_t lane_next(uint32_t v) { return (uint32_t)__shfl_down((int)v, 1); }
#else
__device__ __forceinline__ uint32_t lane_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false); }   /* wave_shl:1 */
#endif
__device__ inline void stage_windows_tiled(McScratch &s, const TiledRef &t, int ix, int iy, int bw, int bh, int cx, int cy, int cw, int ch)
{
    const int lane = lane_id();
    const int wpix = 16 * t.mbw, hpix = 16 * t.mbh, wc = 8 * t.mbw, hc = 8 * t.mbh;
    const int rows = bh + 5, crows = ch + 1;
    if (ix - 4 >= 0 && ix + bw + 2 <= wpix - 1 && cx >= 0 && cx + cw <= wc - 1) {
        /* ---- loads: everything in flight before the first LDS write ---- */
        const int xb = imin((ix - 4) & ~15, imax(0, 16 * (t.mbw - 3)));
        const int o = ix - 4 - xb;                                                     /* 0 .. 37 */
        const int L = lane < 63 ? lane : 62;
        const int r = (int)(__umul24((unsigned)L, 43u) >> 7), p = L - 3 * r;          /* L / 3, L % 3 */
        const int y = clip3(iy - 2 + imin(r, rows - 1), 0, hpix - 1);
        const int tx = imin((xb >> 4) + p, t.mbw - 1);
        const mi355_raw_u32x4 vy = *reinterpret_cast<const mi355_raw_u32x4 *>(t.y + (uint32_t)(__mul24(y >> 4, t.ypitch) + tx * 256 + (y & 15) * 16));
        const int xc = imin(cx & ~7, imax(0, 8 * (t.mbw - 2)));
        const int oc = cx - xc;                                                        /* 0 .. 13 */
        const int U = lane < 36 ? lane : 35;
        const int plane = U >= 18, rem = U - 18 * plane, crow = rem >> 1, piece = rem & 1;
        const int yc = clip3(cy + imin(crow, crows - 1), 0, hc - 1);
        const int txc = imin((xc >> 3) + piece, t.mbw - 1);
        const mi355_raw_u32x2 vc = *reinterpret_cast<const mi355_raw_u32x2 *>(t.c + (uint32_t)(__mul24(yc >> 3, t.cpitch) + txc * 128 + plane * 64 + (yc & 7) * 8));
        MI355_ISSUE_FENCE();
        /* ---- piece -> window dwords ---- */
        const uint32_t sy = (uint32_t)o & 3u, sc = (uint32_t)oc & 3u;
        const uint32_t ny = lane_next(vy[0]), nc = lane_next(vc[0]);
        uint32_t *wy = s.winY + (__mul24(r, WY_DW) + 4 * p - (o >> 2));
        *reinterpret_cast<mi355_raw_u32x2a4 *>(wy) = mi355_raw_u32x2a4{ mi355_alignbyte(vy[1], vy[0], sy), mi355_alignbyte(vy[2], vy[1], sy) };
        *reinterpret_cast<mi355_raw_u32x2a4 *>(wy + 2) = mi355_raw_u32x2a4{ mi355_alignbyte(vy[3], vy[2], sy), mi355_alignbyte(ny, vy[3], sy) };
        uint32_t *wcp = s.winC[0] + (__mul24(plane, 9 * WC_DW + WC_PAD) + crow * WC_DW + 2 * piece - (oc >> 2));
        *reinterpret_cast<mi355_raw_u32x2a4 *>(wcp) = mi355_raw_u32x2a4{ mi355_alignbyte(vc[1], vc[0], sc), mi355_alignbyte(nc, vc[1], sc) };
        MI355_WAVE_SYNC();
        return;
    }
    /* ---- per sample, coordinates clamped to the picture ---- */
    {
        const int ydw = luma_win_dw(bw), n = rows * ydw, inv = mi355_inv20(ydw);
        for (int i = lane; i < n; i += 64) {
            const int row = mi355_div20(i, inv), k = i - row * ydw;
            const int y = clip3(iy - 2 + row, 0, hpix - 1);
            const uint8_t *rp = t.y + (uint32_t)(__mul24(y >> 4, t.ypitch) + (y & 15) * 16);
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = clip3(ix - 4 + 4 * k + b, 0, wpix - 1);
                v |= (uint32_t)rp[(x >> 4) * 256 + (x & 15)] << (8 * b);
            }
            s.winY[row * WY_DW + k] = v;
        }
    }
    {
        const int cdw = chroma_win_dw(cw), per = crows * cdw, n = 2 * per, inv = mi355_inv20(cdw);
        for (int i = lane; i < n; i += 64) {
            const int pl = i >= per, j = i - pl * per, row = mi355_div20(j, inv), k = j - row * cdw;
            const int y = clip3(cy + row, 0, hc - 1);
            const uint8_t *rp = t.c + (uint32_t)(__mul24(y >> 3, t.cpitch) + pl * 64 + (y & 7) * 8);
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = clip3(cx + 4 * k + b, 0, wc - 1);
                v |= (uint32_t)rp[(x >> 3) * 128 + (x & 7)] << (8 * b);
            }
            s.winC[pl][row * WC_DW + k] = v;
        }
    }
    MI355_WAVE_SYNC();
}

__device__ __forceinline__ void bytes12(uint32_t a, uint32_t b, uint32_t c, int *v)
{
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (a >> (8 * k)) & 0xFF; v[4 + k] = (b >> (8 * k)) & 0xFF; v[8 + k] = (c >> (8 * k)) & 0xFF; }
}
/* rounded average of four packed samples: (a + b + 1) >> 1 per byte */
__device__ __forceinline__ uint32_t rnd_avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu); }

/* ---- two 16-bit lanes per register (v_pk_*_i16): the 6-tap sums of 8-bit samples stay within
 * -2550 .. 10710 and the bilinear chroma sums within 0 .. 16352, so two samples share an instruction */
#ifdef MI355_HIP_EMU_H
static inline uint32_t pk_make(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }
static inline int pk_lo(uint32_t a) { return (int16_t)(a & 0xFFFF); }
static inline int pk_hi(uint32_t a) { return (int16_t)(a >> 16); }
static inline uint32_t pk_add(uint32_t a, uint32_t b) { return pk_make(pk_lo(a) + pk_lo(b), pk_hi(a) + pk_hi(b)); }
static inline uint32_t pk_mad(uint32_t a, int k, uint32_t c) { return pk_make(pk_lo(a) * k + pk_lo(c), pk_hi(a) * k + pk_hi(c)); }
static inline uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_make(pk_lo(a) - pk_lo(b), pk_hi(a) - pk_hi(b)); }
static inline int pk_sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline uint32_t pk_adds(uint32_t a, uint32_t b) { return pk_make(pk_sat16(pk_lo(a) + pk_lo(b)), pk_sat16(pk_hi(a) + pk_hi(b))); }
static inline uint32_t pk_ashr(uint32_t a, int n) { return pk_make(pk_lo(a) >> n, pk_hi(a) >> n); }
static inline uint32_t pk_clip_u8(uint32_t a)
{
    const int l = pk_lo(a), h = pk_hi(a);
    return pk_make(l < 0 ? 0 : (l > 255 ? 255 : l), h < 0 ? 0 : (h > 255 ? 255 : h));
}
static inline uint32_t pk_clip_max(uint32_t a, int maxv)           /* both halves to 0 .. maxv */
{
    const int l = pk_lo(a), h = pk_hi(a);
    return pk_make(l < 0 ? 0 : (l > maxv ? maxv : l), h < 0 ? 0 : (h > maxv ? maxv : h));
}
static inline uint32_t pk_ashr_hi1(uint32_t a) { return pk_make(pk_lo(a), pk_hi(a) >> 1); }          /* (lo, hi >> 1) */
static inline uint32_t pk_mad2(uint32_t a, uint32_t k, uint32_t c) { return pk_make(pk_lo(a) * pk_lo(k) + pk_lo(c), pk_hi(a) * pk_hi(k) + pk_hi(c)); }
static inline int pk_dot2(uint32_t a, uint32_t b, int c) { return c + pk_lo(a) * pk_lo(b) + pk_hi(a) * pk_hi(b); }      /* v_dot2_i32_i16 */
static inline int pk_dot2k(uint32_t a, uint32_t k, int c) { return pk_dot2(a, k, c); }
static inline uint32_t bit_mask(uint32_t v, int bit) { return ((v >> bit) & 1u) ? 0xFFFFFFFFu : 0u; }                            /* v_bfe_i32, width 1 */
static inline uint32_t pk_sat_u8(uint32_t a)                                                          /* v_sat_pk_u8_i16: clip(lo) | clip(hi) << 8 */
{
    const int l = pk_lo(a), h = pk_hi(a);
    return (uint32_t)(l < 0 ? 0 : (l > 255 ? 255 : l)) | ((uint32_t)(h < 0 ? 0 : (h > 255 ? 255 : h)) << 8);
}
#else
typedef short mi355_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mi355_v2s pk_v(uint32_t a) { return __builtin_bit_cast(mi355_v2s, a); }
__device__ __forceinline__ uint32_t pk_u(mi355_v2s a) { return __builtin_bit_cast(uint32_t, a); }
__device__ __forceinline__ uint32_t pk_make(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return pk_u(pk_v(a) + pk_v(b)); }
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, int k, uint32_t c) { return pk_u(pk_v(a) * (mi355_v2s)((short)k) + pk_v(c)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_u(pk_v(a) - pk_v(b)); }
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return pk_u(__builtin_elementwise_add_sat(pk_v(a), pk_v(b))); }   /* saturating */
__device__ __forceinline__ uint32_t pk_ashr(uint32_t a, int n) { return pk_u(pk_v(a) >> (mi355_v2s)((short)n)); }
__device__ __forceinline__ uint32_t pk_clip_u8(uint32_t a)
{
    return pk_u(__builtin_elementwise_min(__builtin_elementwise_max(pk_v(a), (mi355_v2s)((short)0)), (mi355_v2s)((short)255)));
}
__device__ __forceinline__ uint32_t pk_clip_max(uint32_t a, int maxv)           /* both halves to 0 .. maxv */
{
    return pk_u(__builtin_elementwise_min(__builtin_elementwise_max(pk_v(a), (mi355_v2s)((short)0)), (mi355_v2s)((short)maxv)));
}
__device__ __forceinline__ uint32_t pk_ashr_hi1(uint32_t a) { return pk_u(pk_v(a) >> mi355_v2s{ 0, 1 }); }
__device__ __forceinline__ uint32_t pk_mad2(uint32_t a, uint32_t k, uint32_t c) { return pk_u(pk_v(a) * pk_v(k) + pk_v(c)); }
#ifdef MI355_DOT2_BUILTIN
__device__ __forceinline__ int pk_dot2(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot2(pk_v(a), pk_v(b), c, false); }
__device__ __forceinline__ int pk_dot2k(uint32_t a, uint32_t k, int c) { return pk_dot2(a, k, c); }
#else
/* the three-address form by name: for the builtin the compiler takes the accumulating two-address form (v_dot2c) and copies
 * the addend first whenever it is still needed afterwards — a v_mov per product */
__device__ __forceinline__ int pk_dot2(uint32_t a, uint32_t b, int c) { int d; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int pk_dot2k(uint32_t a, uint32_t k, int c) { int d; asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(c)); return d; }   /* k: a wave constant */
#endif
__device__ __forceinline__ uint32_t bit_mask(uint32_t v, int bit) { return (uint32_t)__builtin_amdgcn_sbfe((int)v, (uint32_t)bit, 1u); }
__device__ __forceinline__ uint32_t pk_sat_u8(uint32_t a) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(a)); return r; }
#endif
constexpr uint32_t PK_M = 0x00FF00FFu;
/* samples 0,2 and 1,3 of a dword as two packed pairs, and back */
__device__ __forceinline__ uint32_t pk_even(uint32_t w) { return w & PK_M; }
__device__ __forceinline__ uint32_t pk_odd(uint32_t w) { return (w >> 8) & PK_M; }
__device__ __forceinline__ uint32_t pk_bytes(uint32_t e, uint32_t o) { return e | (o << 8); }
/* (a + f) - 5 (b + e) + 20 (c + d) on pairs */
__device__ __forceinline__ uint32_t pk_tap6(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
    return pk_mad(pk_add(c, d), 20, pk_mad(pk_add(b, e), -5, pk_add(a, f)));
}
/* clip_u8((x + 16) >> 5) on pairs */
__device__ __forceinline__ uint32_t pk_round5(uint32_t x) { return pk_clip_u8(pk_ashr(pk_add(x, 0x00100010u), 5)); }
/* raw horizontal 6-tap sums of the four samples whose first taps are bytes 2..5 of (d0,d1,d2): pairs (0,2) and (1,3) */
__device__ __forceinline__ void pk_htaps(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t &te, uint32_t &to)
{
    const uint32_t s2 = mi355_alignbyte(d1, d0, 2), s6 = mi355_alignbyte(d2, d1, 2);
    const uint32_t e2 = pk_even(s2), o2 = pk_odd(s2), e4 = pk_even(d1), o4 = pk_odd(d1), e6 = pk_even(s6), o6 = pk_odd(s6), e8 = pk_even(d2);
    te = pk_tap6(e2, o2, e4, o4, e6, o6);
    to = pk_tap6(o2, e4, o4, e6, o6, e8);
}

/* ---- a5: quarter-pel luma MC (h264qpel_template.c:77-531) -------------------
 * Block bw x bh from the staged window, fraction (mx,my) in quarter samples.  A lane produces one
 * 4-sample row segment (whole dwords in, one dword out).  Result goes to pred[(py+y)*ppitch + px+x]
 * (LDS), either stored (avg=0) or rounded-averaged with what is there (avg=1: the reference's avg_
 * tables / second list of a bi-predicted block).  Every position is the (rounded average of the)
 * integer sample G, the horizontal half sample b, the vertical half sample h and the centre j. */
__device__ inline void mc_luma_compute(McScratch &s, int mx, int my, int bw, int bh,
                                       uint8_t *pred, int ppitch, int px, int py, int avg)
{
    const int lane = lane_id();
    const int wh = bh + 5;
    const bool use_j = (mx == 2 && my != 0) || (my == 2 && mx != 0);
    const bool use_b = mx != 0 && my != 2;
    const bool use_h = my != 0 && mx != 2;
    const bool use_g = (mx == 0 || my == 0) && ((mx | my) != 2);
    const int bdy = my == 3, hdx = mx == 3;
    const int gdx = (my == 0 && mx == 3), gdy = (mx == 0 && my == 3);
    const int nseg = bw >= 4 ? bw >> 2 : 1, lseg = nseg == 4 ? 2 : (nseg == 2 ? 1 : 0);
    if (use_j) {
        /* unclipped horizontal sums of all bh+5 rows, 4 per lane */
        for (int i = lane; i < wh * nseg; i += 64) {
            const int r = i >> lseg, sx = i & (nseg - 1);
            const uint32_t *w = &s.winY[__umul24((unsigned)r, WY_DW) + (unsigned)sx];
            uint32_t te, to;
            pk_htaps(w[0], w[1], w[2], te, to);
            /* pairs (0,2),(1,3) -> four consecutive int16 */
            uint32_t *t = reinterpret_cast<uint32_t *>(&s.tmp[r * 16 + 4 * sx]);
            t[0] = (te & 0xFFFFu) | (to << 16);
            t[1] = (te >> 16) | (to & 0xFFFF0000u);
        }
        MI355_WAVE_SYNC();
    }
    for (int i = lane; i < bh * nseg; i += 64) {
        const int y = i >> lseg, sx = i & (nseg - 1);
        uint32_t se = 0, so = 0;            /* sums of the components, samples (0,2) and (1,3) */
        if (use_g) {
            const uint32_t *w = &s.winY[__umul24((unsigned)(y + 2 + gdy), WY_DW) + (unsigned)sx + 1u];
            const uint32_t g = gdx ? mi355_alignbyte(w[1], w[0], 1) : w[0];
            se = pk_even(g); so = pk_odd(g);
        }
        if (use_b) {
            uint32_t te, to;
            if (use_j) {
                const uint32_t *t = reinterpret_cast<const uint32_t *>(&s.tmp[(y + 2 + bdy) * 16 + 4 * sx]);
                te = (t[0] & 0xFFFFu) | (t[1] << 16);
                to = (t[0] >> 16) | (t[1] & 0xFFFF0000u);
            } else {
                const uint32_t *w = &s.winY[__umul24((unsigned)(y + 2 + bdy), WY_DW) + (unsigned)sx];
                pk_htaps(w[0], w[1], w[2], te, to);
            }
            se = pk_add(se, pk_round5(te)); so = pk_add(so, pk_round5(to));
        }
        if (use_h) {
            uint32_t e[6], o[6];
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const uint32_t *w = &s.winY[__umul24((unsigned)(y + r), WY_DW) + (unsigned)sx + 1u];
                const uint32_t c = hdx ? mi355_alignbyte(w[1], w[0], 1) : w[0];
                e[r] = pk_even(c); o[r] = pk_odd(c);
            }
            se = pk_add(se, pk_round5(pk_tap6(e[0], e[1], e[2], e[3], e[4], e[5])));
            so = pk_add(so, pk_round5(pk_tap6(o[0], o[1], o[2], o[3], o[4], o[5])));
        }
        if (use_j) {
            /* second pass over the unclipped first-pass sums, two samples per instruction.  The sums reach +-430 000, but
             * (a - 5 b + 20 c + 512) >> 10 == ((((a - b) >> 2) - b + c) >> 2) + c + 32) >> 6 exactly (nested floors), and
             * with a, b, c (sums of two rows each) in -5100..21420 only the "+ c" of the inner bracket can leave int16
             * (+-33150): that addition saturates, which moves the bracket only when the final value is already beyond
             * 0..255 on the same side (c >= 21037 resp. <= -4718 there), so the clipped sample is unchanged. */
            const uint32_t *t = reinterpret_cast<const uint32_t *>(&s.tmp[y * 16 + 4 * sx]);
            uint32_t jv[2];
#pragma unroll
            for (int hlf = 0; hlf < 2; hlf++) {
                const uint32_t af = pk_add(t[hlf], t[40 + hlf]), be = pk_add(t[8 + hlf], t[32 + hlf]), cd = pk_add(t[16 + hlf], t[24 + hlf]);
                const uint32_t t1 = pk_ashr(pk_sub(af, be), 2);
                const uint32_t t2 = pk_ashr(pk_adds(pk_sub(t1, be), cd), 2);
                jv[hlf] = pk_clip_u8(pk_ashr(pk_add(pk_add(t2, cd), 0x00200020u), 6));
            }
            /* samples (0,1), (2,3) -> pairs (0,2), (1,3) */
            se = pk_add(se, byte_perm(jv[1], jv[0], 0x05040100u)); so = pk_add(so, byte_perm(jv[1], jv[0], 0x07060302u));
        }
        if ((int)use_g + (int)use_b + (int)use_h + (int)use_j == 2) {
            se = pk_ashr(pk_add(se, 0x00010001u), 1); so = pk_ashr(pk_add(so, 0x00010001u), 1);
        }
        const uint32_t v = pk_bytes(se, so);
        uint8_t *d = &pred[(py + y) * ppitch + px + 4 * sx];
        if (bw >= 4) {
            uint32_t *dw = reinterpret_cast<uint32_t *>(d);
            *dw = avg ? rnd_avg4(*dw, v) : v;
        } else {                      /* 2-sample blocks (h264qpel 2x2 tables) */
            uint16_t *d2 = reinterpret_cast<uint16_t *>(d);
            *d2 = (uint16_t)(avg ? rnd_avg4(*d2, v) : v);
        }
    }
    MI355_WAVE_SYNC();
}

/* ---- a6: 1/8-pel bilinear chroma MC (h264chroma_template.c:27-173) ----------
 * `nplanes` planes in one pass (window p -> pred[p]); a lane produces one 4-sample row segment. */
__device__ inline void mc_chroma_compute(McScratch &s, int nplanes, int fx, int fy, int bw, int bh,
                                         uint8_t *pred0, uint8_t *pred1, int ppitch, int px, int py, int avg)
{
    const int lane = lane_id();
    const int A = (8 - fx) * (8 - fy), B = fx * (8 - fy), C = (8 - fx) * fy, D = fx * fy;
    const int nseg = bw >= 4 ? bw >> 2 : 1, per_plane = bh * nseg;
    for (int i = lane; i < per_plane * nplanes; i += 64) {
        const int plane = i >= per_plane, j = i - plane * per_plane;
        const int y = nseg == 2 ? j >> 1 : j, sx = nseg == 2 ? j & 1 : 0;
        const uint32_t *w0 = &s.winC[plane][__umul24((unsigned)y, WC_DW) + (unsigned)sx], *w1 = w0 + WC_DW;
        const uint32_t a0 = w0[0], a1 = mi355_alignbyte(w0[1], w0[0], 1), b0 = w1[0], b1 = mi355_alignbyte(w1[1], w1[0], 1);
        /* (A a + B b + C c + D d + 32) >> 6 on packed pairs: the sums stay below 2^14 */
        const uint32_t ve = pk_ashr(pk_mad(pk_even(a0), A, pk_mad(pk_even(a1), B, pk_mad(pk_even(b0), C, pk_mad(pk_even(b1), D, 0x00200020u)))), 6);
        const uint32_t vo = pk_ashr(pk_mad(pk_odd(a0), A, pk_mad(pk_odd(a1), B, pk_mad(pk_odd(b0), C, pk_mad(pk_odd(b1), D, 0x00200020u)))), 6);
        const uint32_t v = pk_bytes(ve, vo);
        uint8_t *d = (plane ? pred1 : pred0) + (py + y) * ppitch + px + 4 * sx;
        if (bw >= 4) {
            uint32_t *dw = reinterpret_cast<uint32_t *>(d);
            *dw = avg ? rnd_avg4(*dw, v) : v;
        } else if (bw == 2) {
            uint16_t *d2 = reinterpret_cast<uint16_t *>(d);
            *d2 = (uint16_t)(avg ? rnd_avg4(*d2, v) : v);
        } else {
            *d = (uint8_t)(avg ? rnd_avg4(*d, v & 0xFF) : v);
        }
    }
    MI355_WAVE_SYNC();
}

/* The 8x8 chroma blocks of a 16x16 partition, both planes at once on all 64 lanes: a lane produces two neighbouring
 * samples (plane lane >> 5, row (lane >> 2) & 7, columns 2c, 2c + 1 with c = lane & 3) from the three window bytes
 * 2c .. 2c + 2 of two rows, spread to 16-bit pairs with v_perm_b32.  Same arithmetic as mc_chroma_compute. */
__device__ inline void mc_chroma16(McScratch &s, int fx, int fy, uint8_t *pred0, uint8_t *pred1, int ppitch, int avg)
{
    const int lane = lane_id();
    const int A = (8 - fx) * (8 - fy), B = fx * (8 - fy), C = (8 - fx) * fy, D = fx * fy;
    const int plane = lane >> 5, y = (lane >> 2) & 7, c = lane & 3;
    const uint32_t *w0 = &s.winC[plane][__umul24((unsigned)y, WC_DW) + (unsigned)(c >> 1)], *w1 = w0 + WC_DW;
    const uint32_t sh = 2u * (uint32_t)(c & 1);
    const uint32_t r0 = mi355_alignbyte(w0[1], w0[0], sh), r1 = mi355_alignbyte(w1[1], w1[0], sh);
    const uint32_t a0 = byte_perm(0, r0, 0x0C010C00u), a1 = byte_perm(0, r0, 0x0C020C01u);
    const uint32_t b0 = byte_perm(0, r1, 0x0C010C00u), b1 = byte_perm(0, r1, 0x0C020C01u);
    /* (A a + B b + C c + D d + 32) >> 6 on packed pairs: the sums stay below 2^14 */
    const uint32_t v = pk_ashr(pk_mad(a0, A, pk_mad(a1, B, pk_mad(b0, C, pk_mad(b1, D, 0x00200020u)))), 6);
    uint32_t two = byte_perm(0, v, 0x0C0C0200u);
    uint16_t *d = reinterpret_cast<uint16_t *>((plane ? pred1 : pred0) + y * ppitch + 2 * c);
    if (avg) two = rnd_avg4(*d, two);
    *d = (uint16_t)two;
    MI355_WAVE_SYNC();
}

/* ---- a7: explicit / implicit weighted prediction (h264dsp_template.c:30-98) -- */
__device__ inline void weight_block(uint8_t *p, int pitch, int bw, int bh, int log2_denom, int w, int o)
{
    o = (int)((unsigned)o << log2_denom);
    if (log2_denom) o += 1 << (log2_denom - 1);
    for (int i = lane_id(); i < bw * bh; i += 64) {
        int y = i / bw, x = i - y * bw;
        uint8_t *d = &p[y * pitch + x];
        *d = (uint8_t)clip_u8((*d * w + o) >> log2_denom);
    }
    MI355_WAVE_SYNC();
}
__device__ inline void biweight_block(uint8_t *dst, const uint8_t *src, int pitch, int bw, int bh,
                                      int log2_denom, int wd, int ws, int o)
{
    o = (int)((unsigned)((o + 1) | 1) << log2_denom);
    for (int i = lane_id(); i < bw * bh; i += 64) {
        int y = i / bw, x = i - y * bw;
        uint8_t *d = &dst[y * pitch + x];
        *d = (uint8_t)clip_u8((src[y * pitch + x] * ws + *d * wd + o) >> (log2_denom + 1));
    }
    MI355_WAVE_SYNC();
}

/* ---- a1: 4x4 inverse transform, 4 lanes per block (h264idct_template.c:33-67) -
 * Lane j = lane&3 holds storage COLUMN j of the (transposed) coefficient block: c[k] = block[j + 4k].
 * The reference's first loop (butterflies over k for a fixed position j, intermediates truncated to the
 * int16 block, block[0] += 32 before it) is then per lane; its second loop (within a storage row, across j)
 * runs across the four lanes with two quad exchanges.  On return lane j holds the four residuals (already
 * >> 6) of destination ROW `row` = {0,3,1,2}[j], columns 0..3 — a whole dword of samples to update.
 * Per lane, but all lanes of the quad must execute it together. */
__device__ __forceinline__ void idct4_quad(const int c[4], int j, int r[4], int &row)
{
    const int c0 = j == 0 ? (int16_t)(c[0] + 32) : c[0];
    const int e0 = c0 + c[2], e1 = c0 - c[2], e2 = (c[1] >> 1) - c[3], e3 = c[1] + (c[3] >> 1);
    const int b[4] = { (int16_t)(e0 + e3), (int16_t)(e1 + e2), (int16_t)(e1 - e2), (int16_t)(e0 - e3) };
    /* the cross-lane pass with lane constants instead of selects: lanes 0..3 form z0 = v0 + v2, z3 = v1 + (v3 >> 1),
     * z1 = v0 - v2, z2 = (v1 >> 1) - v3 from their own value and the one two lanes away (a shift by 0 / 1 and a
     * multiply-add by +-1; all operands fit 24 bits: int16 inputs, sums below 2^17), then z0 +- z3 / z1 +- z2 with the
     * neighbour */
    const int sh = j & 1, sg1 = j >= 2 ? -1 : 1, sg2 = (j & 1) ? -1 : 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int v = b[k];
        const int s = mad24i(v, sg1, quad_xor2(v) >> sh);     /* z0, z3, z1, z2 on lanes 0..3 */
        r[k] = mad24i(s, sg2, quad_xor1(s)) >> 6;
    }
    row = j == 0 ? 0 : (j == 1 ? 3 : (j == 2 ? 1 : 2));
}
/* four residuals onto four neighbouring samples (a dword when the tile row is aligned) */
/* ALIGNED: the caller's tile puts every row segment on a dword (no run-time test, no byte path) */
template <bool ALIGNED = false>
__device__ __forceinline__ void add_row4(uint8_t *p, const int *r)
{
    if (ALIGNED || (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
        /* two 16-bit lanes per register: the residuals are at most 4 * 32768 >> 6 in magnitude (second transform pass on
         * 16-bit first-pass values; (dc + 32) >> 6 for the DC-only form), sample + residual cannot wrap 16 bits */
        uint32_t *w = reinterpret_cast<uint32_t *>(p);
        const uint32_t v = *w;
        const uint32_t s0 = pk_clip_u8(pk_add(byte_perm(0u, v, 0x0C010C00u), pk_make(r[0], r[1])));
        const uint32_t s1 = pk_clip_u8(pk_add(byte_perm(0u, v, 0x0C030C02u), pk_make(r[2], r[3])));
        *w = byte_perm(s1, s0, 0x06040200u);
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = (uint8_t)clip_u8(p[k] + r[k]);
    }
}

/* one 8-point pass (h264idct_template.c:80-108) — per lane */
__device__ __forceinline__ void idct8_1d(const int in[8], int out[8])
{
    int a0 = in[0] + in[4], a2 = in[0] - in[4];
    int a4 = (in[2] >> 1) - in[6], a6 = (in[6] >> 1) + in[2];
    int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int a1 = -in[3] + in[5] - in[7] - (in[7] >> 1);
    int a3 = in[1] + in[7] - in[3] - (in[3] >> 1);
    int a5 = -in[1] + in[7] + in[5] + (in[5] >> 1);
    int a7 = in[3] + in[5] + in[1] + (in[1] >> 1);
    int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2);
    int b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    out[0] = b0 + b7; out[7] = b0 - b7;
    out[1] = b2 + b5; out[6] = b2 - b5;
    out[2] = b4 + b3; out[5] = b4 - b3;
    out[3] = b6 + b1; out[4] = b6 - b1;
}

/* 8x8 inverse transform of one block held in LDS (int16 blk[64], overwritten with the
 * intermediate); 8 lanes (i = 0..7) per block, `active` lanes only but every lane of the
 * wave must call (barriers inside).  Lane i ends with the residuals of destination
 * column i, rows 0..7.  h264idct_template.c:69-141. */
__device__ inline void idct8_lds(int16_t *blk, int i, bool active, int r[8])
{
    int in[8], out[8];
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; k++) in[k] = blk[i + 8 * k];
        if (i == 0) in[0] = (int16_t)(in[0] + 32);
        idct8_1d(in, out);
    }
    MI355_WAVE_SYNC();
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; k++) blk[i + 8 * k] = (int16_t)out[k];
    }
    MI355_WAVE_SYNC();
    if (active) {
#pragma unroll
        for (int k = 0; k < 8; k++) in[k] = blk[8 * i + k];
        idct8_1d(in, out);
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = out[k] >> 6;
    }
}

/* add n residuals down a column of an LDS picture tile, with clipping — per lane */
__device__ __forceinline__ void add_col(uint8_t *p, int pitch, const int *r, int n)
{
    for (int k = 0; k < n; k++) p[k * pitch] = (uint8_t)clip_u8(p[k * pitch] + r[k]);
}

/* ---- a3: DC transforms (h264idct_template.c:242-324) — per lane, serial ------- */
/* luma: in[16] raw DC levels (row-major), out[k] goes to the DC slot of luma block
 * dc_slot_block(k) */
__device__ inline void luma_dc_dequant(const int in[16], int qmul, int out[16])
{
    int t[16];
    for (int i = 0; i < 4; i++) {
        int s = in[4 * i] + in[4 * i + 1], d = in[4 * i] - in[4 * i + 1];
        int e = in[4 * i + 2] - in[4 * i + 3], u = in[4 * i + 2] + in[4 * i + 3];
        t[4 * i] = s + u; t[4 * i + 1] = s - u; t[4 * i + 2] = d - e; t[4 * i + 3] = d + e;
    }
    for (int i = 0; i < 4; i++) {
        int s = t[i] + t[8 + i], d = t[i] - t[8 + i];
        int e = t[4 + i] - t[12 + i], u = t[4 + i] + t[12 + i];
        out[4 * i + 0] = (int16_t)(((s + u) * qmul + 128) >> 8);
        out[4 * i + 1] = (int16_t)(((d + e) * qmul + 128) >> 8);
        out[4 * i + 2] = (int16_t)(((d - e) * qmul + 128) >> 8);
        out[4 * i + 3] = (int16_t)(((s - u) * qmul + 128) >> 8);
    }
}
/* coefficient index (in the 256-entry luma array) that out[k] of luma_dc_dequant lands on:
 * column offsets {0,2,8,10}*16 for i = k>>2, row offsets {0,1,4,5}*16 for k&3 */
__device__ __host__ __forceinline__ int luma_dc_slot(int k)
{
    const int i = k >> 2, j = k & 3;
    const int co = i == 0 ? 0 : (i == 1 ? 2 : (i == 2 ? 8 : 10));
    const int ro = j == 0 ? 0 : (j == 1 ? 1 : (j == 2 ? 4 : 5));
    return (co + ro) * 16;
}
__device__ inline void chroma_dc_dequant(int &a, int &b, int &c, int &d, int qmul)
{
    int s0 = a + b, d0 = a - b, s1 = c + d, d1 = c - d;
    a = (int16_t)(((s0 + s1) * qmul) >> 7);
    b = (int16_t)(((d0 + d1) * qmul) >> 7);
    c = (int16_t)(((s0 - s1) * qmul) >> 7);
    d = (int16_t)(((d0 - d1) * qmul) >> 7);
}

/* ---- a8: deblocking, one line across an edge — per lane -----------------------
 * h264dsp_template.c:104-150 (bS<4), :166-218 (bS==4), :233-270 / :294-318 chroma. */
/* MAXV: the largest sample value (255; (1 << bit_depth) - 1 for the 9 / 10-bit tables, whose callers scale alpha, beta
 * and tc0 as h264dsp_template.c:110-113 does) */
template <int MAXV = 255>
__device__ __forceinline__ void lf_luma_line(int &p2, int &p1, int &p0, int &q0, int &q1, int &q2,
                                             int alpha, int beta, int tc0)
{
    if (tc0 < 0) return;
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    int tc = tc0, np1 = p1, nq1 = q1;
    if (iabs(p2 - p0) < beta) {
        if (tc0) np1 = p1 + clip3(((p2 + ((p0 + q0 + 1) >> 1)) >> 1) - p1, -tc0, tc0);
        tc++;
    }
    if (iabs(q2 - q0) < beta) {
        if (tc0) nq1 = q1 + clip3(((q2 + ((p0 + q0 + 1) >> 1)) >> 1) - q1, -tc0, tc0);
        tc++;
    }
    int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    p1 = np1; q1 = nq1;
    p0 = clip3(p0 + delta, 0, MAXV);
    q0 = clip3(q0 - delta, 0, MAXV);
}
__device__ __forceinline__ void lf_luma_intra_line(int p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int q3,
                                                   int alpha, int beta)
{
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    const int P0 = p0, P1 = p1, P2 = p2, Q0 = q0, Q1 = q1, Q2 = q2;
    if (iabs(P0 - Q0) < ((alpha >> 2) + 2)) {
        if (iabs(P2 - P0) < beta) {
            p0 = (P2 + 2 * P1 + 2 * P0 + 2 * Q0 + Q1 + 4) >> 3;
            p1 = (P2 + P1 + P0 + Q0 + 2) >> 2;
            p2 = (2 * p3 + 3 * P2 + P1 + P0 + Q0 + 4) >> 3;
        } else {
            p0 = (2 * P1 + P0 + Q1 + 2) >> 2;
        }
        if (iabs(Q2 - Q0) < beta) {
            q0 = (P1 + 2 * P0 + 2 * Q0 + 2 * Q1 + Q2 + 4) >> 3;
            q1 = (P0 + Q0 + Q1 + Q2 + 2) >> 2;
            q2 = (2 * q3 + 3 * Q2 + Q1 + Q0 + P0 + 4) >> 3;
        } else {
            q0 = (2 * Q1 + Q0 + P1 + 2) >> 2;
        }
    } else {
        p0 = (2 * P1 + P0 + Q1 + 2) >> 2;
        q0 = (2 * Q1 + Q0 + P1 + 2) >> 2;
    }
}
/* tc is the value the caller passes in tc0[] (already +1 for chroma, h264_loopfilter.c:126-129) */
template <int MAXV = 255>
__device__ __forceinline__ void lf_chroma_line(int p1, int &p0, int &q0, int q1, int alpha, int beta, int tc)
{
    if (tc <= 0) return;
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
    p0 = clip3(p0 + delta, 0, MAXV);
    q0 = clip3(q0 - delta, 0, MAXV);
}
__device__ __forceinline__ void lf_chroma_intra_line(int p1, int &p0, int &q0, int q1, int alpha, int beta)
{
    if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta) return;
    const int P0 = p0, Q0 = q0;
    p0 = (2 * p1 + P0 + q1 + 2) >> 2;
    q0 = (2 * q1 + Q0 + p1 + 2) >> 2;
}

/* ---- a10: intra prediction (h264pred_template.c) — per lane ---------------------
 * Directional modes as functions of the reference vectors T[-1..2N-1] (row above,
 * T[-1] = corner sample) and L[-1..N-1] (column to the left): the standard's
 * formulation, valid for N=4 on raw edge samples and for N=8 on the pre-filtered edge.
 * T and L point at element 0 (element -1 must be addressable). */
__device__ __forceinline__ int f3(int a, int b, int c) { return (a + 2 * b + c + 2) >> 2; }
__device__ __forceinline__ int f2(int a, int b) { return (a + b + 1) >> 1; }

__device__ inline int pred_dir_px(int mode, int N, int x, int y, const uint16_t *T, const uint16_t *L)
{
    switch (mode) {
    case 0: return T[x];
    case 1: return L[y];
    case 3: /* diagonal down-left */
        return (x == N - 1 && y == N - 1) ? (T[2 * N - 2] + 3 * T[2 * N - 1] + 2) >> 2
                                          : f3(T[x + y], T[x + y + 1], T[x + y + 2]);
    case 4: /* diagonal down-right */
        if (x > y) return f3(T[x - y - 2], T[x - y - 1], T[x - y]);
        if (x < y) return f3(L[y - x - 2], L[y - x - 1], L[y - x]);
        return f3(T[0], T[-1], L[0]);
    case 5: { /* vertical-right */
        int z = 2 * x - y, i = x - (y >> 1);
        if (z >= 0 && !(z & 1)) return f2(T[i - 1], T[i]);
        if (z > 0) return f3(T[i - 2], T[i - 1], T[i]);
        if (z == -1) return f3(L[0], T[-1], T[0]);
        return f3(L[y - 2 * x - 1], L[y - 2 * x - 2], L[y - 2 * x - 3]);
    }
    case 6: { /* horizontal-down */
        int z = 2 * y - x, i = y - (x >> 1);
        if (z >= 0 && !(z & 1)) return f2(L[i - 1], L[i]);
        if (z > 0) return f3(L[i - 2], L[i - 1], L[i]);
        if (z == -1) return f3(L[0], T[-1], T[0]);
        return f3(T[x - 2 * y - 1], T[x - 2 * y - 2], T[x - 2 * y - 3]);
    }
    case 7: { /* vertical-left */
        int i = x + (y >> 1);
        return (y & 1) ? f3(T[i], T[i + 1], T[i + 2]) : f2(T[i], T[i + 1]);
    }
    case 8: { /* horizontal-up */
        int z = x + 2 * y, i = y + (x >> 1);
        if (z > 2 * N - 3) return L[N - 1];
        if (z == 2 * N - 3) return (L[N - 2] + 3 * L[N - 1] + 2) >> 2;
        return (z & 1) ? f3(L[i], L[i + 1], L[i + 2]) : f2(L[i], L[i + 1]);
    }
    }
    return 128;
}

/* which edges a 4x4 / 8x8 luma mode reads: bit0 top, bit1 left, bit2 corner, bit3 top-right */
__device__ __host__ __forceinline__ int pred_luma_needs(int mode)
{
    switch (mode) {
    case 0: case 10: return 1;
    case 1: case 9: case 8: return 2;
    case 2: return 3;
    case 3: case 7: return 1 | 8;
    case 4: case 5: case 6: return 1 | 2 | 4;
    default: return 0;
    }
}

/* DC-family value for an NxN luma block from the (already prepared) vectors */
__device__ inline int pred_dc_value(int mode, int N, const uint16_t *T, const uint16_t *L, int mid = 128)
{
    int st = 0, sl = 0;
    for (int i = 0; i < N; i++) { st += T[i]; sl += L[i]; }
    const int lg = N == 4 ? 2 : 3;
    if (mode == 2) return (st + sl + N) >> (lg + 1);
    if (mode == 9) return (sl + (N >> 1)) >> lg;
    if (mode == 10) return (st + (N >> 1)) >> lg;
    return mid;
}

/* plane prediction parameters (h264pred_template.c:434-481 for N=16, :768-802 for N=8):
 * pixel(x,y) = clip((a + x*H + y*V) >> 5).  top[-1..N-1], left[-1..N-1]. */
__device__ inline void pred_plane_params(int N, const uint16_t *top, const uint16_t *left, int &a, int &H, int &V)
{
    const int half = N >> 1;
    H = 0; V = 0;
    for (int k = 1; k <= half; k++) {
        H += k * (top[half - 1 + k] - top[half - 1 - k]);
        V += k * (left[half - 1 + k] - left[half - 1 - k]);
    }
    if (N == 16) { H = (5 * H + 32) >> 6;  V = (5 * V + 32) >> 6; }
    else         { H = (17 * H + 16) >> 5; V = (17 * V + 16) >> 5; }
    a = 16 * (left[N - 1] + top[N - 1] + 1) - (half - 1) * (V + H);
}


/* ---- wave-level intra predictor -------------------------------------------------
 * Raw edge samples are gathered by the caller into s.T / s.L (index 0 = corner
 * p[-1,-1]; T[1+i] = p[i,-1], L[1+j] = p[-1,j]); unavailable samples may hold
 * anything, they are never used for a legal (mode, availability) combination.
 * kind: 0 = 4x4 luma (T[5..8] = the caller's `topright` samples), 1 = 8x8 luma with
 * the (1,2,1) edge pre-filter (h264pred_template.c:846-874), 2 = 8x8 chroma,
 * 3 = 16x16 luma, 4 = 8x16 chroma (4:2:2, the pred8x16_* functions :502-846 that occupy the pred8x8[] slots
 * when chroma_format_idc == 2).  `mode` is the reference's table slot (h264pred.h:34-88).
 * Writes the NxN block to out[y*pitch+x] (LDS or global). */
struct PredScratch {
    uint16_t T[1 + 32];            /* unsigned: a 9 / 10-bit plane may hold any 16-bit value (transform bypass does not clip), and the reference sums its samples as unsigned */
    uint16_t L[1 + 16];
    uint16_t fT[1 + 16];
    uint16_t fL[1 + 8];
};

/* PX / BD: sample type and bit depth of the output (uint8_t / 8 for everything but the 9 / 10-bit Tier-1 tables) */
template <typename PX = uint8_t, int BD = 8>
__device__ inline void intra_pred_wave(PredScratch &s, int kind, int mode, int has_tl, int has_tr,
                                       PX *out, int pitch)
{
    constexpr int MAXV = (1 << BD) - 1, MID = 1 << (BD - 1);
    const int lane = lane_id();
    const uint16_t *T = s.T + 1, *L = s.L + 1;
    if (kind == 0 || kind == 1) {
        const int N = kind == 0 ? 4 : 8;
        const int needs = pred_luma_needs(mode);
        if (kind == 1) {
            /* pre-filter: lanes 0..15 -> fT[0..15], 16..23 -> fL[0..7], 24 -> corner */
            int v = 0;
            if (lane < 16 && (needs & 1)) {
                int x = lane;
                if (x == 0)       v = f3(has_tl ? T[-1] : T[0], T[0], T[1]);
                else if (x < 7)   v = f3(T[x - 1], T[x], T[x + 1]);
                else if (x == 7)  v = f3(has_tr ? T[8] : T[7], T[7], T[6]);
                else if (!has_tr) v = T[7];
                else if (x < 15)  v = f3(T[x - 1], T[x], T[x + 1]);
                else              v = (T[14] + 3 * T[15] + 2) >> 2;
                s.fT[1 + x] = (int16_t)v;
            } else if (lane >= 16 && lane < 24 && (needs & 2)) {
                int y = lane - 16;
                if (y == 0)      v = f3(has_tl ? L[-1] : L[0], L[0], L[1]);
                else if (y < 7)  v = f3(L[y - 1], L[y], L[y + 1]);
                else             v = (L[6] + 3 * L[7] + 2) >> 2;
                s.fL[1 + y] = (int16_t)v;
            } else if (lane == 24 && (needs & 4)) {
                v = f3(L[0], T[-1], T[0]);
                s.fT[0] = s.fL[0] = (int16_t)v;
            }
            MI355_WAVE_SYNC();
            T = s.fT + 1;
            L = s.fL + 1;
        }
        if (lane < N * N) {
            int x = lane & (N - 1), y = lane / N;
            int v;
            if (mode == 2 || mode >= 9) v = pred_dc_value(mode, N, T, L, MID);
            else v = pred_dir_px(mode, N, x, y, T, L);
            out[y * pitch + x] = (PX)v;
        }
        MI355_WAVE_SYNC();
        return;
    }
    if (kind == 3) { /* 16x16: slots DC=0 HOR=1 VERT=2 PLANE=3 LEFT_DC=4 TOP_DC=5 DC128=6 */
        int st = 0, sl = 0, a = 0, H = 0, V = 0;
        if (mode == 0 || mode == 5) for (int i = 0; i < 16; i++) st += T[i];
        if (mode == 0 || mode == 4) for (int i = 0; i < 16; i++) sl += L[i];
        if (mode == 3) pred_plane_params(16, T, L, a, H, V);
        for (int i = lane; i < 256; i += 64) {
            int x = i & 15, y = i >> 4, v;
            switch (mode) {
            case 0: v = (st + sl + 16) >> 5; break;
            case 1: v = L[y]; break;
            case 2: v = T[x]; break;
            case 3: v = clip3((a + x * H + y * V) >> 5, 0, MAXV); break;
            case 4: v = (sl + 8) >> 4; break;
            case 5: v = (st + 8) >> 4; break;
            default: v = MID; break;
            }
            out[y * pitch + x] = (PX)v;
        }
        MI355_WAVE_SYNC();
        return;
    }
    if (kind == 4) { /* 8x16 chroma: quadrant DC rules of :590-595 (left), :622-642 (top), :673-720 (both), mad-cow patches :722-766 */
        int a = 0, H = 0, V = 0;
        if (mode == 3) {     /* pred8x16_plane :804-846: 4 taps across the top, 8 down the left edge */
            for (int k = 1; k <= 4; k++) H += k * (T[3 + k] - T[3 - k]);
            for (int k = 1; k <= 8; k++) V += k * (L[7 + k] - L[7 - k]);
            H = (17 * H + 16) >> 5; V = (5 * V + 32) >> 6;
            a = 16 * (L[15] + T[7] + 1) - 7 * V - 3 * H;
        }
        for (int i = lane; i < 128; i += 64) {
            const int x = i & 7, y = i >> 3, qx = x >> 2, qy = y >> 2;
            int t = 0, l = 0, v;
            for (int k = 0; k < 4; k++) { t += T[4 * qx + k]; l += L[4 * qy + k]; }
            const int dc_left = (l + 2) >> 2, dc_top = (t + 2) >> 2;
            const int dc_full = qy == 0 ? (qx ? dc_top : (t + l + 4) >> 3) : (qx ? (t + l + 4) >> 3 : dc_left);
            switch (mode) {
            case 0: v = dc_full; break;
            case 1: v = L[y]; break;
            case 2: v = T[x]; break;
            case 3: v = clip3((a + x * H + y * V) >> 5, 0, MAXV); break;
            case 4: v = dc_left; break;
            case 5: v = dc_top; break;
            case 6: v = MID; break;
            case 7: v = (qx == 0 && qy == 0) ? (t + l + 4) >> 3 : dc_top; break;   /* L0T */
            case 8: v = (qx == 0 && qy == 0) ? dc_top : dc_full; break;              /* 0LT */
            case 9: v = qy == 1 ? MID : dc_left; break;                              /* L00: rows 4..7 only */
            default: v = qy == 0 ? MID : dc_left; break;                             /* 0L0 */
            }
            out[y * pitch + x] = (PX)v;
        }
        MI355_WAVE_SYNC();
        return;
    }
    /* kind == 2: 8x8 chroma; per-4x4-quadrant DC rules of h264pred_template.c:563-766 */
    {
        const int x = lane & 7, y = lane >> 3, qx = x >> 2, qy = y >> 2;
        int t = 0, l = 0, v, a = 0, H = 0, V = 0;
        for (int i = 0; i < 4; i++) { t += T[4 * qx + i]; l += L[4 * qy + i]; }
        if (mode == 3) pred_plane_params(8, T, L, a, H, V);
        const int dc_full = (qx == qy) ? (t + l + 4) >> 3 : (qx ? (t + 2) >> 2 : (l + 2) >> 2);
        const int dc_left = (l + 2) >> 2, dc_top = (t + 2) >> 2;
        switch (mode) {
        case 0: v = dc_full; break;
        case 1: v = L[y]; break;
        case 2: v = T[x]; break;
        case 3: v = clip3((a + x * H + y * V) >> 5, 0, MAXV); break;
        case 4: v = dc_left; break;
        case 5: v = dc_top; break;
        case 6: v = MID; break;
        case 7: v = (qx == 0 && qy == 0) ? (t + l + 4) >> 3 : dc_top; break;   /* L0T */
        case 8: v = (qx == 0 && qy == 0) ? dc_top : dc_full; break;              /* 0LT */
        case 9: v = qy ? MID : dc_left; break;                                   /* L00 */
        default: v = qy ? dc_left : MID; break;                                  /* 0L0 */
        }
        out[y * pitch + x] = (PX)v;
        MI355_WAVE_SYNC();
    }
}

/* ---- the two predictors every intra macroblock of a P / I picture uses most, without per-lane loops (intra_pred_wave above sums the edge samples
 * in EVERY lane: thirty-two LDS reads a lane for a DC mode, the plane parameters' sixteen-term sums likewise) ----
 * sums run across lanes: within groups of four (quad_xor1 / quad_xor2) and within rows of sixteen lanes (row_ror) */
#ifdef MI355_HIP_EMU_H
template <int N> static inline int row_ror(int v) { const int l = (int)(threadIdx.x & 63); return __shfl(v, (l & ~15) | ((l + N) & 15)); }
#else
template <int N> __device__ __forceinline__ int row_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xF, 0xF, true); }   /* row_ror:N */
#endif
__device__ __forceinline__ int row_sum16(int v)         /* every lane: the sum over its row of sixteen lanes */
{
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return v;
}
/* Intra16x16 (h264pred_template.c:329-486, slots DC 0, HOR 1, VERT 2, PLANE 3, LEFT_DC 4, TOP_DC 5, DC128 6), 8-bit: lane = row lane >> 2, samples 4 (lane & 3) .. + 3;
 * `out` and `pitch` on dwords.  s.T / s.L as for intra_pred_wave. */
__device__ inline void intra16_pred_fast(PredScratch &s, int mode, uint8_t *out, int pitch)
{
    const int lane = lane_id(), row = lane >> 2, seg = lane & 3;
    const uint16_t *T = s.T + 1, *L = s.L + 1;
    uint32_t v;
    if (mode == 2) {
        v = (uint32_t)T[4 * seg] | ((uint32_t)T[4 * seg + 1] << 8) | ((uint32_t)T[4 * seg + 2] << 16) | ((uint32_t)T[4 * seg + 3] << 24);
    } else if (mode == 1) {
        v = (uint32_t)L[row] * 0x01010101u;
    } else if (mode == 3) {
        /* pred16x16_plane (:434-481): H, V from eight differences each — lane k - 1 takes term k */
        const int k = (lane & 7) + 1;
        const int h = lane < 8 ? k * ((int)T[7 + k] - (int)T[7 - k]) : 0, w = lane < 8 ? k * ((int)L[7 + k] - (int)L[7 - k]) : 0;
        const int H = (5 * lane_value(row_sum16(h), 0) + 32) >> 6, V = (5 * lane_value(row_sum16(w), 0) + 32) >> 6;
        const int a = 16 * (lane_value((int)L[15], 0) + lane_value((int)T[15], 0) + 1) - 7 * (V + H);
        const int b = a + row * V + 4 * seg * H;
        /* med3i by name: written as comparisons the four clips become v_ashr_pk_u8_i32 + shifts, and the pictures came out wrong on the device
         * (tests/test_frame_gpu.py, wide_mixed / mid_* cases; the emulator has no such instruction) */
        const int p0 = med3i(b >> 5, 0, 255), p1 = med3i((b + H) >> 5, 0, 255), p2 = med3i((b + 2 * H) >> 5, 0, 255), p3 = med3i((b + 3 * H) >> 5, 0, 255);
        v = (uint32_t)p0 | ((uint32_t)p1 << 8) | ((uint32_t)p2 << 16) | ((uint32_t)p3 << 24);
    } else {
        const int st = lane_value(row_sum16(lane < 16 ? (int)T[lane & 15] : 0), 0), sl = lane_value(row_sum16(lane < 16 ? (int)L[lane & 15] : 0), 0);
        const int dc = mode == 0 ? (st + sl + 16) >> 5 : (mode == 4 ? (sl + 8) >> 4 : (mode == 5 ? (st + 8) >> 4 : 128));
        v = (uint32_t)dc * 0x01010101u;
    }
    *reinterpret_cast<uint32_t *>(out + row * pitch + 4 * seg) = v;
    MI355_WAVE_SYNC();
}
/* 8x8 chroma (h264pred_template.c:563-802, the eleven slots of intra_pred_wave's kind 2), 8-bit, one plane: a row of sixteen lanes is one 4x4 quadrant
 * (lane = 32 qy + 16 qx + 4 (y & 3) + (x & 3)), so the quadrant's top sum is a sum over the group of four and its left sum one over the row's groups */
__device__ inline void chroma8_pred_fast(PredScratch &s, int mode, uint8_t *out, int pitch)
{
    const int lane = lane_id(), qx = (lane >> 4) & 1, qy = lane >> 5, x = (lane & 3) + 4 * qx, y = ((lane >> 2) & 3) + 4 * qy;
    const uint16_t *T = s.T + 1, *L = s.L + 1;
    int v;
    if (mode == 1) v = L[y];
    else if (mode == 2) v = T[x];
    else if (mode == 3) {
        /* pred8x8_plane (:768-802): four differences each */
        const int k = (lane & 3) + 1;
        int h = lane < 4 ? k * ((int)T[3 + k] - (int)T[3 - k]) : 0, w = lane < 4 ? k * ((int)L[3 + k] - (int)L[3 - k]) : 0;
        h += quad_xor1(h); h += quad_xor2(h); w += quad_xor1(w); w += quad_xor2(w);
        const int H = (17 * lane_value(h, 0) + 16) >> 5, V = (17 * lane_value(w, 0) + 16) >> 5;
        const int a = 16 * (lane_value((int)L[7], 0) + lane_value((int)T[7], 0) + 1) - 3 * (V + H);
        v = clip_u8((a + x * H + y * V) >> 5);
    } else if (mode == 6) v = 128;
    else {
        int t = T[x], l = L[y];
        t += quad_xor1(t); t += quad_xor2(t);
        l += row_ror<4>(l) + row_ror<8>(l) + row_ror<12>(l);
        const int dc_left = (l + 2) >> 2, dc_top = (t + 2) >> 2, both = (t + l + 4) >> 3;
        const int dc_full = (qx == qy) ? both : (qx ? dc_top : dc_left);
        switch (mode) {
        case 0: v = dc_full; break;
        case 4: v = dc_left; break;
        case 5: v = dc_top; break;
        case 7: v = (qx == 0 && qy == 0) ? both : dc_top; break;      /* L0T */
        case 8: v = (qx == 0 && qy == 0) ? dc_top : dc_full; break;   /* 0LT */
        case 9: v = qy ? 128 : dc_left; break;                        /* L00 */
        default: v = qy ? dc_left : 128; break;                       /* 0L0 */
        }
    }
    out[y * pitch + x] = (uint8_t)v;
    MI355_WAVE_SYNC();
}

}  // namespace mi355
#endif
