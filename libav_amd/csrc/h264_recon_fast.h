/*
 * h264_recon_fast.h — the inter reconstruction pass on macroblock-tiled surfaces, third form (round 5): a wave walks a RUN of
 * consecutive macroblocks, and the macroblock nearly every P picture is made of — one 16x16 partition predicted from list 0
 * without weights, 4x4 transforms — takes a path of its own:
 *
 *   - what depends on nothing but the lane number (row / piece of the window fetch, operand addresses, filter matrices, the
 *     residual lanes' table) is computed ONCE per wave and kept in registers over the run;
 *   - the record's scalars (type, vector, reference slot, nnz, cbp) arrive through the scalar cache (s_load), the record itself
 *     (coefficients) by LDS-DMA — and the NEXT macroblock's record is requested as soon as this one's coefficients have been
 *     transformed into registers, so its memory round trip runs under this macroblock's prediction;
 *   - the reference windows land in LDS RAW, as the 16-byte tile rows they are in HBM (LDS-DMA, no register in flight, no
 *     re-alignment): gfx950 reads LDS at any byte address, so every tap is a plain ds_read at (row, o + column);
 *     rows beyond the picture are clamped in the fetch, columns beyond it are replicated in LDS afterwards (emulated_edge_mc,
 *     videodsp_template.c:24-96) — one path for every vector, however far outside;
 *   - the 6-tap filters (h264qpel_template.c:77-378) are v_mfma_i32_16x16x32_i8 products with a constant Toeplitz matrix: the
 *     horizontal half-sample plane straight from the window rows; vertical filters after a transposition that is itself a
 *     product (with an identity / the horizontal filter) whose result a lane holds as four ROWS of one column.  The 2-D
 *     position's intermediate (-2550 .. 10710) travels as a low and a high byte plane; everything is exact integer arithmetic;
 *   - chroma (h264chroma_template.c:27-173) is one v_dot4_u32_u8 per sample: the four neighbours gathered by v_perm_b32
 *     against the packed weights (A, B, C, D);
 *   - averages of two components are v_lerp_u8 on packed bytes.
 * Every other macroblock type in the run (two lists, weights, partitions, 8x8 transform) goes through h264_recon_dev.h's code
 * unchanged.  Reference behaviour restated: hl_decode_mb (h264_mb_template.c:41-257), hl_motion (h264_mc_template.c:64-163),
 * mc_dir_part (h264_mb.c:204-318), hl_decode_mb_idct_luma (h264_mb.c:726-795), h264idct_template.c:33-67,144-156.
 */
#ifndef MI355_H264_RECON_FAST_H
#define MI355_H264_RECON_FAST_H

#include "h264_recon_dev.h"

namespace {

/* ---- LDS of the fast path: behind py / pc (the q tiles and the motion scratch of the other paths are not live here) ---- */
/* Coefficients: where MbLds keeps them, but the 48 sixteen-byte pieces in another order — first halves of the 24 blocks (coefficients 0..7), then the
 * second halves: a lane pair reads (dword h, h + 2) of one piece per access, and with the pieces of a block 32 bytes apart the 24 blocks met in four of the
 * LDS's eight bank groups (six lanes per bank: 16 cycles a read, tools/ubench/lds_rate.hip); 16 bytes apart they spread over all eight */
/* section marks in the listing (tools/isa_sections.py): analysis builds only */
#if defined(FQ_MARKS) && !defined(MI355_HIP_EMU_H)
#define FQ_MARK(name) asm volatile("; MARK " name)
#else
#define FQ_MARK(name)
#endif

constexpr int FQ_COEF = (int)offsetof(MbCore, coef);
constexpr int FQ_RS = FQ_COEF + 768;            /* 960: the residual, int16: luma [16 rows][16], then chroma [plane][8 rows][8] — where the other paths keep their prediction tiles */
constexpr int FQ_RSC = FQ_RS + 512;
constexpr int FQ_ST = FQ_RS;                    /* ... and, once a lane holds its residual, the finished macroblock in tile order (luma 256 bytes, chroma 128 from FQ_ST + 256 — below FQ_RSC) on its way out */
constexpr int FQ_WY = FQ_RS + 768;              /* 1728: raw luma window, piece L (16 bytes) at 16 L: row L / 3 = 48 bytes = picture columns 16 t0 .. 16 t0 + 47 */
constexpr int FQ_WC = FQ_WY + 1024;             /* raw chroma window [plane][row 0..8][12 bytes], dword q at 4 q: columns (cx & ~3) .. + 11; all 64 lanes of the request write (216 bytes used of 256) */
constexpr int FQ_PL = FQ_WC + 256;              /* two transposed planes [column 0..15][24 rows]: low bytes (or the samples themselves), high bytes */
constexpr int FQ_PLANE = 16 * 24 + 8;           /* the last column's operand read runs eight bytes past its rows */
constexpr int FQ_DUMP = FQ_PL + 16 * 24;        /* where the lanes that hold no row of a product's second half write: the plane's own tail */
constexpr int FQ_WSTEP = FQ_PL + 2 * FQ_PLANE - FQ_WY;      /* the second set of windows (the NEXT macroblock's, in flight while this one is predicted) lies this far behind the first */
static_assert(FQ_RS == MB_PY_OFF && (FQ_WY % 16) == 0 && (FQ_PL % 8) == 0 && (FQ_WSTEP % 16) == 0 && FQ_WC + FQ_WSTEP + 256 <= (int)sizeof(MbLds), "fast-path regions inside MbLds");
static_assert(FQ_WY + FQ_WSTEP + 48 * 20 + 24 + 16 + 12 <= (int)sizeof(MbLds), "the second half of a transposing product reads window rows 16..20");

/* ---- primitives: one instruction each on the device, their plain meaning in the emulator ---- */
#ifdef MI355_HIP_EMU_H
static inline uint64_t fq_lds64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }     /* p on 8 bytes */
static inline uint32_t fq_lerp(uint32_t a, uint32_t b)                 /* v_lerp_u8 with 1 in every byte of the third operand: (a + b + 1) >> 1 per byte */
{
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) r |= ((((a >> (8 * i)) & 0xFF) + ((b >> (8 * i)) & 0xFF) + 1) >> 1) << (8 * i);
    return r;
}
static inline uint32_t fq_dot4(uint32_t a, uint32_t b, uint32_t c)     /* v_dot4_u32_u8 */
{
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return c;
}
/* v_mfma_i32_16x16x32_i8: D[i][j] = c + sum_k A[i][k] B[k][j] with signed bytes; lane l supplies bytes 8 (l / 16) .. + 7 of row l % 16 of A (x) and of
 * column l % 16 of B (y), and receives D[4 (l / 16) + t][l % 16], t = 0..3 */
static inline void fq_mfma(uint64_t x, uint64_t y, int c, int d[4])
{
    const int lane = (int)(threadIdx.x & 63), g = lane >> 4, j = lane & 15;
    d[0] = d[1] = d[2] = d[3] = c;
    for (int kg = 0; kg < 4; kg++) {
        const uint64_t yv = (uint64_t)(uint32_t)__shfl((int)(uint32_t)y, j + 16 * kg) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(y >> 32), j + 16 * kg) << 32);
        for (int t = 0; t < 4; t++) {
            const uint64_t xv = (uint64_t)(uint32_t)__shfl((int)(uint32_t)x, 4 * g + t + 16 * kg) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(x >> 32), 4 * g + t + 16 * kg) << 32);
            for (int b = 0; b < 8; b++) d[t] += (int)(int8_t)(xv >> (8 * b)) * (int)(int8_t)(yv >> (8 * b));
        }
    }
}
/* LDS-DMA: 16 / 4 bytes per lane from memory to LDS offset OFF + 16 / 4 x lane of the wave's MbLds */
template <int OFF> static inline void fq_dma16(const uint8_t *src, MbLds &s, int more = 0) { std::memcpy(reinterpret_cast<uint8_t *>(&s) + OFF + more + 16 * (threadIdx.x & 63), src, 16); }
template <int OFF> static inline void fq_dma4(const uint8_t *src, MbLds &s, int more = 0) { std::memcpy(reinterpret_cast<uint8_t *>(&s) + OFF + more + 4 * (threadIdx.x & 63), src, 4); }
static inline void fq_wait_vm0() {}
static inline void fq_wait_vm2() {}
static inline int fq_med3_0(int x, int hi) { return x < 0 ? 0 : (x > hi ? hi : x); }
typedef const uint32_t *fq_kptr;
static inline fq_kptr fq_konst(const void *p) { return reinterpret_cast<const uint32_t *>(p); }
struct fq_ptr2 { uint32_t w[4]; uint32_t operator[](int i) const { return w[i]; } };
static inline const fq_ptr2 *fq_konst2(const void *p) { return reinterpret_cast<const fq_ptr2 *>(p); }
#else
__device__ __forceinline__ uint64_t fq_lds64(const uint8_t *p) { return *reinterpret_cast<const uint64_t *>(p); }       /* p on 8 bytes */
__device__ __forceinline__ uint32_t fq_lerp(uint32_t a, uint32_t b) { return __builtin_amdgcn_lerp(a, b, 0x01010101u); }
__device__ __forceinline__ uint32_t fq_dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
typedef int fq_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fq_mfma(uint64_t x, uint64_t y, int c, int d[4])
{
    fq_v4i acc = { c, c, c, c };
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)x, (long)y, acc, 0, 0, 0);
    d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3];
}
/* LDS-DMA: 16 / 4 bytes per lane from memory to LDS offset OFF + 16 / 4 x lane of the wave's MbLds.  OFF goes into M0 (an offset in the instruction would
 * move the MEMORY address as well); the tile is cast to its address space first, so the sum is a literal and no generic-pointer null test is made */
typedef __attribute__((address_space(3))) uint8_t *fq_lds_ptr;
template <int OFF> __device__ __forceinline__ void fq_dma16(const uint8_t *src, MbLds &s, int more = 0)      /* more: a wave-uniform offset on top */
{
    typedef __attribute__((address_space(1))) const void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)((fq_lds_ptr)&s + OFF + more), 16, 0, 0);
}
template <int OFF> __device__ __forceinline__ void fq_dma4(const uint8_t *src, MbLds &s, int more = 0)
{
    typedef __attribute__((address_space(1))) const void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)((fq_lds_ptr)&s + OFF + more), 4, 0, 0);
}
/* vector-memory LOADS complete in issue order: "at most N outstanding" with N loads youngest means every older load has landed */
__device__ __forceinline__ void fq_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void fq_wait_vm2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
__device__ __forceinline__ int fq_med3_0(int x, int hi) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi)); return r; }      /* clamp to 0 .. hi (a wave constant) */
/* a wave-uniform address read through the scalar cache: the record's words arrive in scalar registers, no LDS read, no v_readfirstlane */
typedef const __attribute__((address_space(4))) uint32_t *fq_kptr;
__device__ __forceinline__ fq_kptr fq_konst(const void *p) { return (fq_kptr)(unsigned long long)p; }
typedef uint32_t fq_ptr2 __attribute__((ext_vector_type(4), aligned(8)));
__device__ __forceinline__ const __attribute__((address_space(4))) fq_ptr2 *fq_konst2(const void *p) { return (const __attribute__((address_space(4))) fq_ptr2 *)(unsigned long long)p; }
#endif

/* LDS answers a read that is not naturally aligned one lane per cycle (64 cycles a wave, tools/ubench/lds_rate.hip: a b64 four bytes off its alignment included):
 * bytes at any offset come from aligned dwords and v_alignbyte.  p on 4 bytes, sh = 0..3 */
__device__ __forceinline__ uint32_t fq_bytes4(const uint8_t *p, uint32_t sh)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    return mi355_alignbyte(w[1], w[0], sh);
}
__device__ __forceinline__ uint64_t fq_bytes8(const uint8_t *p, uint32_t sh)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    return (uint64_t)mi355_alignbyte(d1, d0, sh) | ((uint64_t)mi355_alignbyte(d2, d1, sh) << 32);
}

/* ---- what a lane is, for every macroblock of the run ---- */
struct FqLane {
    /* luma window fetch: piece L = min(lane, 62) is row fr = L / 3, tile L % 3 (fp256 = 256 (L % 3): the tile's bytes) */
    int fr, fp256;
    /* chroma window fetch: dword q = min(lane, 53) is plane q / 27, row crow = (q % 27) / 3, dword d = q % 3 (kq68 = 68 d: with 68 b added, b = the window's first
     * dword of its tile row, bit 7 is the tile step and bit 2 the dword of the tile row) */
    int cplane64, crow, kq68;
    /* luma: this lane's row = lane & 15 and group g = lane >> 4 (four samples 4 g .. 4 g + 3 of the row in a direct product, four rows 4 g .. of column lane & 15 in a transposing one) */
    uint32_t a1;            /* FQ_WY + 48 row + 8 g: this lane's eight bytes of a window-row operand (+ o + 2 + ...) */
    uint32_t a1b;           /* ... of window row 16 + row (the second half of a transposing product; rows past 20 do not exist: those lanes read a1 again) */
    uint32_t a2;            /* FQ_WY + 48 row + 4 g: this lane's four integer samples (+ 48 (2 + dy) + o + 4 + dx) */
    uint32_t sy;            /* 16 row + 4 g: where its four samples lie in the luma tile (the store's offset) */
    uint32_t rsy;           /* FQ_RS + 32 row + 8 g: their residual */
    uint32_t a4, a5;        /* 24 (lane & 15) + 4 g: its four rows in a transposed plane, first / second half of the window rows (g >= 2: FQ_DUMP) */
    uint32_t a6;            /* 24 (lane & 15) + 8 g: its eight rows of a transposed-plane operand */
    uint64_t t6;            /* the 6-tap filter as a product operand: byte b = tap (8 g + b) - (lane & 15) of (1, -5, 20, 20, -5, 1) */
    uint64_t i2;            /* the identity shifted by the two columns in front of the block: byte b = 1 where 8 g + b == (lane & 15) + 2 */
    /* chroma: plane lane >> 5, row (lane >> 2) & 7, samples 2 c, 2 c + 1 (c = lane & 3) */
    uint32_t c1;            /* FQ_WC + 108 plane + 12 row + 2 c */
    uint32_t sc;            /* 64 plane + 8 row + 2 c: where its two samples lie in the chroma tile */
    uint32_t rsc;           /* FQ_RSC + 128 plane + 16 row + 4 c: their residual */
    /* residual: lane 2 b + h holds columns 2 h, 2 h + 1 of block b (residual_blocks's arrangement) */
    uint32_t cwf;           /* where its first coefficient pair lies in the fast path's coefficient layout (below) */
    uint32_t wa, wb;        /* where the four residuals of its two rows go (eight bytes each) */
    uint32_t csrc;          /* coefficient fetch: the byte offset in the macroblock's 768 bytes of the piece that goes to LDS slot `lane` */
};
__device__ __forceinline__ FqLane fq_lane()
{
    FqLane k;
    const int lane = lane_id();
    const int L = lane < 63 ? lane : 62;
    k.fr = (int)(__umul24((unsigned)L, 43u) >> 7);
    k.fp256 = 256 * (L - 3 * k.fr);
    const int q = lane < 54 ? lane : 53, plane = q >= 27, rem = q - 27 * plane;
    k.cplane64 = plane * 64;
    k.crow = (int)(__umul24((unsigned)rem, 43u) >> 7);
    k.kq68 = 68 * (rem - 3 * k.crow);
    const int row = lane & 15, g = lane >> 4;
    k.a1 = (uint32_t)(FQ_WY + 48 * row + 8 * g);
    k.a1b = row < 5 ? k.a1 + 768u : k.a1;
    k.a2 = (uint32_t)(FQ_WY + 48 * row + 4 * g);
    k.sy = (uint32_t)(16 * row + 4 * g);
    k.rsy = (uint32_t)(FQ_RS + 32 * row + 8 * g);
    k.a4 = (uint32_t)(FQ_PL + 24 * row + 4 * g);
    k.a5 = g < 2 ? k.a4 + 16u : (uint32_t)FQ_DUMP;
    k.a6 = (uint32_t)(FQ_PL + 24 * row + 8 * g);
    /* tap t = 8 g + b - row: the six taps as bytes 01 FB 14 14 FB 01 sit at bit 8 (row - 8 g) of the operand */
    const int sh = 8 * (row - 8 * g);
    const uint64_t taps = 0x01FB1414FB01ull;
    k.t6 = sh >= 64 || sh <= -64 ? 0ull : (sh >= 0 ? taps << sh : taps >> -sh);
    const int s1 = sh + 16;
    k.i2 = s1 >= 64 || s1 < 0 ? 0ull : 1ull << s1;
    const int cp = lane >> 5, cy = (lane >> 2) & 7, c = lane & 3;
    k.c1 = (uint32_t)(FQ_WC + 108 * cp + 12 * cy + 2 * c);
    k.sc = (uint32_t)(64 * cp + 8 * cy + 2 * c);
    k.rsc = (uint32_t)(FQ_RSC + 128 * cp + 16 * cy + 4 * c);
    const int blk = lane < 48 ? lane >> 1 : 23;
    k.cwf = (uint32_t)(FQ_COEF + 16 * blk + 4 * (lane & 1));
    {   /* block blk: luma 4x4 block (x4, y4) in the reference's order, or chroma block blk - 16 = 4 plane + 2 y4 + x4; lane h = 0 holds rows 0 and 3, h = 1 rows 1 and 2 */
        const int h = lane & 1, ra = h ? 1 : 0, rb = h ? 2 : 3;
        if (blk < 16) {
            const int x4 = (blk & 1) + 2 * ((blk >> 2) & 1), y4 = ((blk >> 1) & 1) + 2 * (blk >> 3);
            k.wa = (uint32_t)(FQ_RS + 32 * (4 * y4 + ra) + 8 * x4);
            k.wb = (uint32_t)(FQ_RS + 32 * (4 * y4 + rb) + 8 * x4);
        } else {
            const int jj = blk & 3, pl = (blk >> 2) & 1;
            k.wa = (uint32_t)(FQ_RSC + 128 * pl + 16 * (4 * (jj >> 1) + ra) + 8 * (jj & 1));
            k.wb = (uint32_t)(FQ_RSC + 128 * pl + 16 * (4 * (jj >> 1) + rb) + 8 * (jj & 1));
        }
    }
    const int slot = lane < 48 ? lane : 47;
    k.csrc = (uint32_t)(slot < 24 ? 32 * slot : 32 * (slot - 24) + 16);
    return k;
}

/* ---- the run's macroblocks, described ONCE: lane m works out everything about macroblock m of the run that does not depend on a lane — vector, motion position,
 * window origins, flags, chroma weights — for all of them at a time (one vector instruction per step instead of a scalar one per macroblock and step: the scalar
 * unit was the kernel's narrowest place), and a macroblock's turn fetches its five words with v_readlane. ---- */
constexpr uint32_t FQA_INSIDE = 1u << 22;        /* both windows inside the picture: fq_windows_issue_inside */
constexpr uint32_t FQA_TWO = 1u << 23, FQA_VSPLIT = 1u << 24;       /* two partitions from list 0 (fq_two); ... side by side (8x16) instead of one above the other */
constexpr uint32_t FQA_PATCH_Y = 1u << 11, FQA_PATCH_C = 1u << 12, FQA_FAST = 1u << 13, FQA_INTRA = 1u << 14, FQA_CHROMA = 1u << 15, FQA_RESID = 1u << 16;
struct FqRun {
    uint32_t a;         /* bits 0-4 o + 2 | 5-6 cx & 3 | 7-10 (mx & 3) | (my & 3) << 2 | 11 / 12 luma / chroma window over a side border | 13 fast kind | 14 intra |
                           15 chroma coefficients | 16 any coefficients | 17-21 reference slot */
    uint32_t wts;       /* the chroma weights A | B << 8 | C << 16 | D << 24 (h264chroma_template.c:27-40) */
    uint32_t nnz;
    uint32_t d;         /* first luma window row iy - 2 (low half, signed) | first chroma window row cy (high half, signed) */
    uint32_t e;         /* first luma tile column t0 (low half, signed) | first chroma window column c0 (high half, signed) */
};
static_assert(offsetof(mi355_h264_mb, cbp) == 8 && offsetof(mi355_h264_mb, flags) == 11 && offsetof(mi355_h264_mb, dc_qmul) == 32 && offsetof(mi355_h264_mb, u) == 48, "the words read here");
struct FqPic {          /* what the run keeps of the picture descriptor */
    const mi355_h264_mb *mb;
    const int16_t *mv0, *coef;
    const mi355_h264_frame *desc;
    FrameHot hot;       /* recon planes and strides for the stores (store_mb_tiled) */
};
/* what one prediction's vector decides: window offsets, quarter-sample position, border flags, reference slot (word a, without the macroblock's own flags), the chroma
 * weights, the windows' first rows (d) and columns (e).  Vector code in fq_describe (lane m = macroblock m), scalar code in fq_two (a partition at a time) */
struct FqGeo {
    uint32_t a, wts, d, e;
};
__device__ __forceinline__ FqGeo fq_geometry(uint32_t mvw, int mb_x, int mb_y, int chroma_dy, uint32_t slot_byte, int mbw, int mbh)
{
    FqGeo g;
    const int mx = (int16_t)(mvw & 0xFFFF) + mb_x * 64, my = (int16_t)(mvw >> 16) + mb_y * 64;
    const int myc = my + chroma_dy;                                       /* the other-parity field offset, h264_mb.c:287-291 */
    const int ix = mx >> 2, iy = my >> 2, cx = mx >> 3, cy = myc >> 3;
    const int t0 = (ix - 4) >> 4, c0 = cx & ~3;
    const uint32_t slot = slot_byte < (uint32_t)MI355_H264_MAX_SLOTS ? slot_byte : 0u;
    static_assert(MI355_H264_MAX_SLOTS <= 32, "five bits of slot");
    /* t0 < 0 || t0 + 2 >= mbw; c0 < 0 || c0 + 11 >= 8 mbw: one unsigned comparison each (the bounds are wave constants, not below zero) */
    const bool patch_y = (uint32_t)t0 >= (uint32_t)imax(mbw - 2, 0), patch_c = (uint32_t)c0 >= (uint32_t)imax(8 * mbw - 11, 0);
    const bool inside = !patch_y && !patch_c && (uint32_t)(iy - 2) < (uint32_t)imax(16 * mbh - 20, 0) && (uint32_t)cy < (uint32_t)imax(8 * mbh - 8, 0);
    g.a = (uint32_t)(((ix - 4) & 15) + 2) | ((uint32_t)(cx & 3) << 5) | ((uint32_t)((mx & 3) | ((my & 3) << 2)) << 7) |
          (patch_y ? FQA_PATCH_Y : 0u) | (patch_c ? FQA_PATCH_C : 0u) | (inside ? FQA_INSIDE : 0u) | (slot << 17);
    const int fx = mx & 7, fy = myc & 7;
    g.wts = (uint32_t)((8 - fx) * (8 - fy)) | ((uint32_t)(fx * (8 - fy)) << 8) | ((uint32_t)((8 - fx) * fy) << 16) | ((uint32_t)(fx * fy) << 24);
    g.d = ((uint32_t)(iy - 2) & 0xFFFFu) | ((uint32_t)cy << 16);
    g.e = ((uint32_t)t0 & 0xFFFFu) | ((uint32_t)c0 << 16);
    return g;
}
__device__ __forceinline__ FqRun fq_describe(const FqPic &pic, int mb_xy0, int mb_x, int mb_y, int n_row)
{
    FqRun r;
    const int lane = lane_id(), m = lane < n_row ? lane : n_row - 1;
    const uint32_t *h = reinterpret_cast<const uint32_t *>(mi355_global_v(pic.mb + (mb_xy0 + m)));
    const uint32_t type = h[0], nnz = h[1], w2 = h[2], w12 = h[12], w14 = h[14];
    const uint32_t mvw = pic.mv0 ? reinterpret_cast<const uint32_t *>(mi355_global_v(pic.mv0 + (size_t)(mb_xy0 + m) * 32))[0] : 0u;
    /* the macroblock the fast path is for: one 16x16 partition, list 0 only, no weights, 4x4 transforms ... */
    const uint32_t want = MI355_MB_16x16 | MI355_MB_P0L0, look = MI355_MB_INTRA | MI355_MB_16x16 | MI355_MB_P0L0 | MI355_MB_P0L1 | MI355_MB_8x8DCT;
    const bool plain = !((w2 >> 24) & MI355_MBF_WEIGHTED);
    const bool fast = (type & look) == want && plain;
    /* ... and its sibling of two partitions, 16x8 or 8x16, both from list 0: the same code twice, a lane keeps the partition it lies in (fq_two) */
    const uint32_t look2 = MI355_MB_INTRA | MI355_MB_P0L0 | MI355_MB_P1L0 | MI355_MB_P0L1 | MI355_MB_P1L1 | MI355_MB_8x8DCT;
    const bool two = (type & look2) == (MI355_MB_P0L0 | MI355_MB_P1L0) && (type & (MI355_MB_16x8 | MI355_MB_8x16)) != 0 && plain && pic.mv0 != nullptr;
    const bool chroma = (w2 & 0x30u) != 0, resid = (nnz & 0xFFFFu) != 0 || chroma;
    const FqGeo g = fq_geometry(mvw, mb_x + m, mb_y, (int8_t)(w14 & 0xFFu), w12 & 0xFFu, pic.hot.mb_width, pic.hot.mb_height);
    r.a = g.a | (fast ? FQA_FAST : 0u) | (two ? FQA_TWO : 0u) | ((type & MI355_MB_8x16) ? FQA_VSPLIT : 0u) | ((type & MI355_MB_INTRA) ? FQA_INTRA : 0u) |
          (chroma ? FQA_CHROMA : 0u) | (resid ? FQA_RESID : 0u);
    r.wts = g.wts;
    r.nnz = nnz;
    r.d = g.d;
    r.e = g.e;
    return r;
}
/* the value lane `l` (wave-uniform) holds */
#ifdef MI355_HIP_EMU_H
static inline uint32_t fq_lane_word(uint32_t v, int l) { return (uint32_t)__shfl((int)v, l); }
#else
__device__ __forceinline__ uint32_t fq_lane_word(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
#endif

/* the 768 bytes of a macroblock's coefficients into the layout above: 48 lanes, 16 bytes each */
__device__ __forceinline__ void fq_coef_dma(MbLds &s, const FqLane &k, const int16_t *coef, int mb_xy)
{
    const uint8_t *cp = reinterpret_cast<const uint8_t *>(coef + (size_t)mb_xy * MI355_H264_COEFS_PER_MB);
    if (lane_id() < 48) fq_dma16<FQ_COEF>(cp + k.csrc, s);
}

/* ---- residual, first half: the 24 blocks' inverse transforms into registers (two lanes per block; residual_blocks's arithmetic, see there) ----
 * The four residuals of each of the lane's two rows go to the residual tile (FQ_RS); a block without coefficients writes zeros */
__device__ __forceinline__ void fq_idct(MbLds &s, const FqLane &k, const ResidLane &rl, uint32_t nnz, bool has_chroma, const mi355_h264_mb *rec)
{
    const int lane = lane_id();
    if (has_chroma && (nnz & (3u << MI355_NNZ_CB_DC))) {
        fq_kptr h = fq_konst(rec);                                          /* the two chroma DC dequantisers: through the scalar cache */
        const uint32_t dcq1 = h[9], dcq2 = h[10];
        if (lane < 2 && ((nnz >> (MI355_NNZ_CB_DC + lane)) & 1)) {
            int16_t *p = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(&s) + FQ_COEF + 16 * (16 + 4 * lane));      /* the DCs of blocks 16 + 4 lane .. + 3: 16 bytes apart */
            int a = p[0], b = p[8], c = p[16], d = p[24];
            chroma_dc_dequant(a, b, c, d, (int)(lane ? dcq2 : dcq1));
            p[0] = (int16_t)a; p[8] = (int16_t)b; p[16] = (int16_t)c; p[24] = (int16_t)d;
        }
        MI355_WAVE_SYNC();
    }
    const uint32_t nnz24 = nnz & (has_chroma ? 0xFFFFFFu : 0xFFFFu), hc = has_chroma ? 0xFFFFFFFFu : 0u;
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    const uint32_t keep = bit_mask(nnz24, lane >> 1);
    const uint32_t keep0 = keep | (rl.dc16 & hc);
    const int rnd = (int)(~keep & rl.rc & hc);
    const uint32_t *cw = reinterpret_cast<const uint32_t *>(base + k.cwf);
    const uint32_t c0 = pk_add(cw[0] & keep0, keep & rl.misc & 0xFFu), c1 = cw[2] & keep, c2 = cw[96] & keep, c3 = cw[98] & keep;
    const uint32_t z0 = pk_add(c0, c2), z1 = pk_sub(c0, c2), z2 = pk_sub(pk_ashr(c1, 1), c3), z3 = pk_add(c1, pk_ashr(c3, 1));
    const uint32_t w[4] = { pk_add(z0, z3), pk_add(z1, z2), pk_sub(z1, z2), pk_sub(z0, z3) };
    uint32_t ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t xh = pk_ashr_hi1((uint32_t)quad_xor1((int)w[i]));
        ra[i] = (uint32_t)pk_dot2k(xh, 0x04000400u, pk_dot2(w[i], rl.ka, rnd));          /* the residual is the upper half */
        rb[i] = (uint32_t)pk_dot2k(xh, 0xFC000400u, pk_dot2(w[i], rl.kb, rnd));
    }
    if (lane < 48) {
        *reinterpret_cast<mi355_u32x2 *>(base + k.wa) = mi355_u32x2{ byte_perm(ra[1], ra[0], 0x07060302u), byte_perm(ra[3], ra[2], 0x07060302u) };
        *reinterpret_cast<mi355_u32x2 *>(base + k.wb) = mi355_u32x2{ byte_perm(rb[1], rb[0], 0x07060302u), byte_perm(rb[3], rb[2], 0x07060302u) };
    }
}

/* ---- the windows ---- */
/* issue both fetches of a macroblock (first luma row y0 = iy - 2, first luma tile column t0, first chroma row cy, first chroma column c0 = cx & ~3) into window
 * set `woff`: rows clamped to the picture here, columns fetched from the clamped tile and replicated later (fq_windows_patch).  mc_dir_part's addressing
 * (h264_mb.c:204-318) */
__device__ __forceinline__ void fq_windows_issue(MbLds &s, const FqLane &k, const mi355_h264_frame *desc, const FrameHot &fr, int slot, int y0, int t0, int cy, int c0, int woff)
{
    fq_kptr rp = fq_konst(desc->ref[slot]);
    const uint8_t *ry = mi355_global(reinterpret_cast<const uint8_t *>((unsigned long long)rp[0] | ((unsigned long long)rp[1] << 32)));
    const uint8_t *rc = mi355_global(reinterpret_cast<const uint8_t *>((unsigned long long)rp[2] | ((unsigned long long)rp[3] << 32)));
    const int mbw = fr.mb_width, hpix = 16 * fr.mb_height, hc = 8 * fr.mb_height;
    {
        const int y = fq_med3_0(y0 + k.fr, hpix - 1);
        const int tx = fq_med3_0(t0 + (k.fp256 >> 8), mbw - 1);
        fq_dma16<FQ_WY>(ry + (uint32_t)(__mul24(y >> 4, fr.ref_stride[0]) + tx * 256 + (y & 15) * 16), s, woff);
    }
    {
        const int y = fq_med3_0(cy + k.crow, hc - 1);
        const int col = c0 + (k.kq68 >> 4), t = col >> 3;                        /* 68 d >> 4 = 4 d */
        /* a dword of a tile beyond the picture: the dword of the edge tile that holds the edge column (replicated afterwards) */
        const int tx = fq_med3_0(t, mbw - 1), within = t < 0 ? 0 : (t >= mbw ? 4 : (col & 4));
        fq_dma4<FQ_WC>(rc + (uint32_t)(__mul24(y >> 3, fr.ref_stride[1]) + tx * 128 + k.cplane64 + (y & 7) * 8 + within), s, woff);
    }
}
/* the same for windows that lie inside the picture (FQA_INSIDE: nine macroblocks of ten in a 1080p picture with vectors of +-16 samples): no clamps, the
 * window's first tile in the scalar base, twelve vector instructions for the two addresses instead of twenty-five */
__device__ __forceinline__ void fq_windows_issue_inside(MbLds &s, const FqLane &k, const mi355_h264_frame *desc, const FrameHot &fr, int slot, int y0, int t0, int cy, int c0, int woff)
{
    /* both plane pointers in ONE scalar load (as two loads the second took the first's registers and waited behind the luma request) */
    const fq_ptr2 rp = *fq_konst2(desc->ref[slot]);
    const uint8_t *ry = mi355_global(reinterpret_cast<const uint8_t *>((unsigned long long)rp[0] | ((unsigned long long)rp[1] << 32)));
    const uint8_t *rc = mi355_global(reinterpret_cast<const uint8_t *>((unsigned long long)rp[2] | ((unsigned long long)rp[3] << 32)));
    {
        const uint32_t y = (uint32_t)(y0 + k.fr);
        fq_dma16<FQ_WY>(ry + (uint32_t)(t0 * 256) + (uint32_t)(__umul24(y >> 4, (uint32_t)fr.ref_stride[0]) + (((y & 15u) << 4) + (uint32_t)k.fp256)), s, woff);
    }
    {
        const uint32_t y = (uint32_t)(cy + k.crow);
        const uint32_t x = (((uint32_t)k.kq68 + (uint32_t)((c0 >> 2) & 1) * 68u) & 0x84u) | (uint32_t)k.cplane64;
        fq_dma4<FQ_WC>(rc + (uint32_t)((c0 >> 3) * 128) + (uint32_t)(__umul24(y >> 3, (uint32_t)fr.ref_stride[1]) + (((y & 7u) << 3) + x)), s, woff);
    }
}
/* columns left / right of the picture: the edge column's sample over the whole piece.  Each lane mends the piece it fetched. */
__device__ __forceinline__ void fq_windows_patch(MbLds &s, const FqLane &k, bool patch_y, bool patch_c, int t0, int c0, int woff, int mbw)
{
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s) + woff;
    const int lane = lane_id();
    if (patch_y && lane < 63) {
        const int t = t0 + (k.fp256 >> 8);
        if (t < 0 || t >= mbw) {
            uint8_t *p = base + FQ_WY + 16 * lane;
            const uint32_t e = (uint32_t)p[t < 0 ? 0 : 15] * 0x01010101u;
            *reinterpret_cast<mi355_u32x4 *>(p) = mi355_u32x4{ e, e, e, e };
        }
    }
    if (patch_c && lane < 54) {
        const int t = (c0 + (k.kq68 >> 4)) >> 3;
        if (t < 0 || t >= mbw) {
            uint8_t *p = base + FQ_WC + 4 * lane;
            *reinterpret_cast<uint32_t *>(p) = (uint32_t)p[t < 0 ? 0 : 3] * 0x01010101u;
        }
    }
}

/* ---- luma ---- */
/* A sample enters a product as the signed byte u - 128 (u ^ 0x80).  Every product starts from zero (the matrix unit takes its addend from registers
 * or from a small literal only), so a filtered sum comes out as  sum - 128 x 32 = sum - 4096  and what each user adds back is written at its place. */
constexpr uint64_t FQ_SIGN = 0x8080808080808080ull;
/* four sums of six taps (d = sum - 4096) -> four half samples clip_u8((sum + 16) >> 5) as bytes 0..3 */
__device__ __forceinline__ uint32_t fq_half_samples(const int d[4])
{
    const uint32_t p01 = pk_ashr(pk_add(byte_perm((uint32_t)d[1], (uint32_t)d[0], 0x05040100u), 0x10101010u), 5);       /* + 4096 + 16 on both halves */
    const uint32_t p23 = pk_ashr(pk_add(byte_perm((uint32_t)d[3], (uint32_t)d[2], 0x05040100u), 0x10101010u), 5);
    return byte_perm(pk_sat_u8(p23), pk_sat_u8(p01), 0x05040100u);
}
/* byte B of four values as one dword */
template <int B>
__device__ __forceinline__ uint32_t fq_bytes(const int d[4])
{
    return byte_perm((uint32_t)d[1], (uint32_t)d[0], 0x0C0C0400u + 0x0101u * B) | byte_perm((uint32_t)d[3], (uint32_t)d[2], 0x04000C0Cu + 0x01010000u * B);
}
/* Quarter-sample prediction of the 16x16 block (h264qpel_template.c's mc00..mc33) from the raw window into the prediction tile.
 * pos = (mx & 3) | (my & 3) << 2.  Components: G integer samples, b / h horizontal / vertical half samples, j the centre. */
/* the prediction of this lane's four samples (row lane & 15, columns 4 (lane >> 4) ..) from the window set so2 lies in */
__device__ __forceinline__ uint32_t fq_luma_pred(MbLds &s, const FqLane &k, int so2, int pos)
{
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    /* b of window row `row0` + this lane's row: a direct product, the filter on the column side */
    auto hband = [&](int row0) -> uint32_t {
        int d[4];
        const uint32_t off = (uint32_t)(48 * row0 + so2);
        fq_mfma(k.t6, fq_bytes8(base + k.a1 + (off & ~3u), off & 3u) ^ FQ_SIGN, 0, d);
        return fq_half_samples(d);
    };
    auto gsamples = [&](int dy, int dx) -> uint32_t {
        const uint32_t off = (uint32_t)(48 * (2 + dy) + so2 + 2 + dx);
        return fq_bytes4(base + k.a2 + (off & ~3u), off & 3u);
    };
    /* the window transposed through a product whose result a lane holds as four rows of column (lane & 15).  With the identity: the samples of
     * columns dx .. dx + 15 as they are needed next, u ^ 0x80 (the low byte of u - 128), one byte plane.  With the filter: the unclipped horizontal sums
     * H - 4096 (-6646 .. 6614) as a low byte plane (H & 255, kept as (H & 255) ^ 0x80 for the product to come) and a high one ((H >> 8) - 16, signed) */
    auto transpose = [&](bool filtered, int dx) {
        const uint32_t off = (uint32_t)(so2 + dx);
        const uint64_t x0 = fq_bytes8(base + k.a1 + (off & ~3u), off & 3u) ^ FQ_SIGN, x1 = fq_bytes8(base + k.a1b + (off & ~3u), off & 3u) ^ FQ_SIGN;
        int d0[4], d1[4];
        if (filtered) {
            fq_mfma(x0, k.t6, 0, d0);
            fq_mfma(x1, k.t6, 0, d1);
            *reinterpret_cast<uint32_t *>(base + k.a4) = fq_bytes<0>(d0) ^ 0x80808080u;
            *reinterpret_cast<uint32_t *>(base + k.a5) = fq_bytes<0>(d1) ^ 0x80808080u;
            *reinterpret_cast<uint32_t *>(base + k.a4 + FQ_PLANE) = fq_bytes<1>(d0);
            *reinterpret_cast<uint32_t *>(base + k.a5 + FQ_PLANE) = fq_bytes<1>(d1);
        } else {
            fq_mfma(x0, k.i2, 0, d0);
            fq_mfma(x1, k.i2, 0, d1);
            *reinterpret_cast<uint32_t *>(base + k.a4) = fq_bytes<0>(d0);
            *reinterpret_cast<uint32_t *>(base + k.a5) = fq_bytes<0>(d1);
        }
        MI355_WAVE_SYNC();
    };
    /* h: the vertical filter over the transposed samples */
    auto vraw = [&]() -> uint32_t {
        int d[4];
        fq_mfma(fq_lds64(base + k.a6), k.t6, 0, d);
        return fq_half_samples(d);
    };
    /* j: the vertical filter over both planes of the horizontal sums.  With lo = sum over ((H & 255) - 128) and hi = sum over ((H >> 8) - 16):
     * J = 256 hi + lo + 256 x 16 x 32 + 128 x 32 = t + 135168, and clip((J + 512) >> 10) = clip((t / 16 + 8480) >> 6) exactly (t / 16 = 16 hi + (lo >> 4)
     * floors once more, 135168 + 512 = 16 x 8480; |t / 16 + 8480| < 2^15) */
    auto vsums = [&]() -> uint32_t {
        int lo[4], hi[4];
        uint32_t v[4];
        fq_mfma(fq_lds64(base + k.a6), k.t6, 0, lo);
        fq_mfma(fq_lds64(base + k.a6 + FQ_PLANE), k.t6, 0, hi);
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = (uint32_t)(hi[t] * 16 + (lo[t] >> 4));
        const uint32_t p01 = pk_ashr(pk_add(byte_perm(v[1], v[0], 0x05040100u), 0x21202120u), 6);
        const uint32_t p23 = pk_ashr(pk_add(byte_perm(v[3], v[2], 0x05040100u), 0x21202120u), 6);
        return byte_perm(pk_sat_u8(p23), pk_sat_u8(p01), 0x05040100u);
    };
    uint32_t v;
    FQ_MARK("luma_switch");
    switch (pos) {
    case 0: FQ_MARK("case0"); v = gsamples(0, 0); break;
    case 1: FQ_MARK("case1"); v = fq_lerp(gsamples(0, 0), hband(2)); break;
    case 2: FQ_MARK("case2"); v = hband(2); break;
    case 3: FQ_MARK("case3"); v = fq_lerp(gsamples(0, 1), hband(2)); break;
    case 4: FQ_MARK("case4"); transpose(false, 0); v = fq_lerp(gsamples(0, 0), vraw()); break;
    case 8: FQ_MARK("case8"); transpose(false, 0); v = vraw(); break;
    case 12: FQ_MARK("case12"); transpose(false, 0); v = fq_lerp(gsamples(1, 0), vraw()); break;
    case 5: FQ_MARK("case5"); { const uint32_t b = hband(2); transpose(false, 0); v = fq_lerp(b, vraw()); break; }
    case 7: FQ_MARK("case7"); { const uint32_t b = hband(2); transpose(false, 1); v = fq_lerp(b, vraw()); break; }
    case 13: FQ_MARK("case13"); { const uint32_t b = hband(3); transpose(false, 0); v = fq_lerp(b, vraw()); break; }
    case 15: FQ_MARK("case15"); { const uint32_t b = hband(3); transpose(false, 1); v = fq_lerp(b, vraw()); break; }
    case 10: FQ_MARK("case10"); transpose(true, 0); v = vsums(); break;
    case 6: FQ_MARK("case6"); { const uint32_t b = hband(2); transpose(true, 0); v = fq_lerp(b, vsums()); break; }
    case 14: FQ_MARK("case14"); { const uint32_t b = hband(3); transpose(true, 0); v = fq_lerp(b, vsums()); break; }
    case 9: FQ_MARK("case9"); { transpose(false, 0); const uint32_t h = vraw(); transpose(true, 0); v = fq_lerp(h, vsums()); break; }
    default: { FQ_MARK("case11"); transpose(false, 1); const uint32_t h = vraw(); transpose(true, 0); v = fq_lerp(h, vsums()); break; }     /* 11 */
    }
    return v;
}
/* ... the residual on top, and into the outgoing tile */
__device__ __forceinline__ void fq_luma_out(MbLds &s, const FqLane &k, uint32_t v, bool has_resid)
{
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    FQ_MARK("luma_resid");
    if (has_resid) {
        /* the four residuals of these samples on top (h264idct_template.c:54-66's "+ dst", clipped) */
        const mi355_u32x2 r = *reinterpret_cast<const mi355_u32x2 *>(base + k.rsy);
        const uint32_t s01 = pk_sat_u8(pk_add(byte_perm(0u, v, 0x0C010C00u), r[0])), s23 = pk_sat_u8(pk_add(byte_perm(0u, v, 0x0C030C02u), r[1]));
        v = byte_perm(s23, s01, 0x05040100u);
    }
    MI355_WAVE_SYNC();                                                     /* every lane has its residual: the tile may take their place */
    *reinterpret_cast<uint32_t *>(base + FQ_ST + k.sy) = v;
}
__device__ __forceinline__ void fq_luma(MbLds &s, const FqLane &k, int so2, int pos, bool has_resid) { fq_luma_out(s, k, fq_luma_pred(s, k, so2, pos), has_resid); }

/* ---- chroma: both 8x8 planes, two samples per lane ---- */
__device__ __forceinline__ uint32_t fq_chroma_pred(MbLds &s, const FqLane &k, int oc, uint32_t wts)
{
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    const uint32_t at = k.c1 + (uint32_t)oc, sh = at & 3u;                  /* FQ_WC and the 12-byte row pitch are multiples of four */
    const uint32_t r0 = fq_bytes4(base + (at & ~3u), sh), r1 = fq_bytes4(base + (at & ~3u) + 12, sh);
    const uint32_t d0 = fq_dot4(byte_perm(r1, r0, 0x05040100u), wts, 32u), d1 = fq_dot4(byte_perm(r1, r0, 0x06050201u), wts, 32u);
    return (d0 >> 6) | ((d1 >> 6) << 8);
}
__device__ __forceinline__ void fq_chroma_out(MbLds &s, const FqLane &k, uint32_t two, bool has_resid)
{
    uint8_t *const base = reinterpret_cast<uint8_t *>(&s);
    if (has_resid) two = pk_sat_u8(pk_add(byte_perm(0u, two, 0x0C010C00u), *reinterpret_cast<const uint32_t *>(base + k.rsc)));
    *reinterpret_cast<uint16_t *>(base + FQ_ST + 256 + k.sc) = (uint16_t)two;
}
__device__ __forceinline__ void fq_chroma(MbLds &s, const FqLane &k, int oc, uint32_t wts, bool has_resid) { fq_chroma_out(s, k, fq_chroma_pred(s, k, oc, wts), has_resid); }
/* the finished macroblock, 384 bytes in tile order, to the picture: whole 16-byte pieces, sixteen lanes the luma tile (two cache lines), eight the chroma tile.
 * (Four samples per lane straight to memory — sixty-four 4-byte pieces sixteen bytes apart — cost the memory pipeline sixteen address groups a store instead of
 * four and the pass 1.6 ms of 6.8: tools/gpu_r05c.sh, knock-out "nostore".) */
__device__ __forceinline__ void fq_store(MbLds &s, uint8_t *ytile, uint8_t *ctile)
{
    const uint8_t *const base = reinterpret_cast<const uint8_t *>(&s);
    const int lane = lane_id();
    if (lane < 24) {
        const mi355_u32x4 v = *reinterpret_cast<const mi355_u32x4 *>(base + FQ_ST + 16 * lane);
        if (lane < 16) *reinterpret_cast<mi355_u32x4 *>(ytile + (uint32_t)(16 * lane)) = v;
        else *reinterpret_cast<mi355_u32x4 *>(ctile + (uint32_t)(16 * lane - 256)) = v;
    }
}

/* ---- a macroblock of two partitions, 16x8 or 8x16, both predicted from list 0 (hl_motion's mc_part calls for those types, h264_mc_template.c:83-108): the plain macroblock's
 * code once per partition — each partition's vector applied to the WHOLE macroblock, its windows in a window set of its own — and a lane keeps the result of the partition it lies in.
 * Runs in the second launch (recon_inter_rest): nothing is requested ahead (both window sets are in use); the macroblock's flags and geometry are scalar code on its record's words. ---- */
__device__ __forceinline__ void fq_two(MbLds &s, const FqLane &k, const ResidLane &rl, const FqPic &pic, int mb_xy, int mb_x, int mb_y)
{
    fq_kptr h = fq_konst(pic.mb + mb_xy), mv = fq_konst(pic.mv0 + (size_t)mb_xy * 32);
    const uint32_t type = h[0], nnz = h[1], w2 = h[2], w12 = h[12], w14 = h[14];
    const bool vsplit = (type & MI355_MB_8x16) != 0, has_chroma = (w2 & 0x30u) != 0, resid = (nnz & 0xFFFFu) != 0 || has_chroma;
    const int woff = 0;
    /* the second partition: 4x4 block 8 / quadrant 2 (16x8: the lower half), block 2 / quadrant 1 (8x16: the right half) */
    const int n1 = vsplit ? 2 : 8, q1 = vsplit ? 8 : 16;
    const FqGeo g0 = fq_geometry(mv[0], mb_x, mb_y, (int8_t)(w14 & 0xFFu), w12 & 0xFFu, pic.hot.mb_width, pic.hot.mb_height);
    const FqGeo g1 = fq_geometry(mv[n1], mb_x, mb_y, (int8_t)((w14 >> q1) & 0xFFu), (w12 >> q1) & 0xFFu, pic.hot.mb_width, pic.hot.mb_height);
    const int set0 = woff, set1 = woff ^ FQ_WSTEP;
    auto issue = [&](const FqGeo &g, int set) {
        if (g.a & FQA_INSIDE) fq_windows_issue_inside(s, k, pic.desc, pic.hot, (int)((g.a >> 17) & 31u), (int16_t)(g.d & 0xFFFFu), (int16_t)(g.e & 0xFFFFu), (int)g.d >> 16, (int)g.e >> 16, set);
        else fq_windows_issue(s, k, pic.desc, pic.hot, (int)((g.a >> 17) & 31u), (int16_t)(g.d & 0xFFFFu), (int16_t)(g.e & 0xFFFFu), (int)g.d >> 16, (int)g.e >> 16, set);
    };
    issue(g0, set0);
    issue(g1, set1);
    if (resid) fq_coef_dma(s, k, pic.coef, mb_xy);
    fq_wait_vm0();
    MI355_WAVE_SYNC();
    if (resid) fq_idct(s, k, rl, nnz, has_chroma, pic.mb + mb_xy);
    if ((g0.a | g1.a) & (FQA_PATCH_Y | FQA_PATCH_C)) {
        fq_windows_patch(s, k, (g0.a & FQA_PATCH_Y) != 0, (g0.a & FQA_PATCH_C) != 0, (int16_t)(g0.e & 0xFFFFu), (int)g0.e >> 16, set0, pic.hot.mb_width);
        fq_windows_patch(s, k, (g1.a & FQA_PATCH_Y) != 0, (g1.a & FQA_PATCH_C) != 0, (int16_t)(g1.e & 0xFFFFu), (int)g1.e >> 16, set1, pic.hot.mb_width);
    }
    MI355_WAVE_SYNC();
    const int lane = lane_id();
    /* luma: lane = row (lane & 15), columns 4 (lane >> 4) ..; chroma: row (lane >> 2) & 7, columns 2 (lane & 3) .. */
    const bool second_y = vsplit ? lane >= 32 : (lane & 8) != 0, second_c = vsplit ? (lane & 2) != 0 : (lane & 16) != 0;
    uint32_t vy = 0, vc = 0;
#pragma nounroll
    for (int p = 0; p < 2; p++) {
        const uint32_t ga = p ? g1.a : g0.a, gw = p ? g1.wts : g0.wts;
        const int set = p ? set1 : set0;
        const uint32_t y = fq_luma_pred(s, k, (int)(ga & 31u) + set, (int)((ga >> 7) & 15u));
        const uint32_t c = fq_chroma_pred(s, k, (int)((ga >> 5) & 3u) + set, gw);
        if (second_y == (p != 0)) vy = y;
        if (second_c == (p != 0)) vc = c;
    }
    fq_luma_out(s, k, vy, resid);
    fq_chroma_out(s, k, vc, resid);
    MI355_WAVE_SYNC();
    fq_store(s, pic.hot.recon[0] + tile_y_off(mb_x, mb_y, pic.hot.recon_stride[0]), pic.hot.recon[1] + tile_c_off(mb_x, mb_y, pic.hot.recon_stride[1]));
    MI355_WAVE_SYNC();
}

/* ---- the run: `run` (<= 32) consecutive macroblocks of ONE row of the launch's max_w x max_h grid per wave; runs_row = ceil(max_w / run) runs to a row.
 * Wave w works on run w % runs_row of row w / runs_row of the launch (row = picture * max_h + mb_y). ---- */
__device__ __forceinline__ void recon_inter_run(MbLds &s, const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, int run, int runs_row,
                                                unsigned long long inv_runs, unsigned long long inv_h, int nwaves, int per_xcd, uint32_t *__restrict__ rest)
{
    const int wave = xcd_linear((int)blockIdx.x, per_xcd);
    if (wave >= nwaves) return;
    const int row = div_magic(wave, inv_runs);
    const int mb_x = (wave - row * runs_row) * run;
    const int f = div_magic(row, inv_h), mb_y = row - f * max_h;
    const int n = max_w - mb_x < run ? max_w - mb_x : run;
    ResidLane rl;
    resid_lane_compute(rl);             /* the transform's lane constants (signs, DC masks), from the lane number: no load whose wait would fall into the run */
    const FqLane k = fq_lane();
    auto picture = [&](int pf, FqPic &pic) -> bool {
        const mi355_h264_frame &frd = frames[pf];
        pic.hot = frame_hot(frd);
        pic.mb = pic.hot.mb; pic.mv0 = pic.hot.mv[0]; pic.coef = pic.hot.coef; pic.desc = &frd;
        return uniform(frd.surface_layout) == MI355_SURFACE_TILED && !(uniform(frd.flags) & MI355_FRAME_NO_INTER);
    };
    FqPic pic;
    const bool pic_ok = picture(f, pic);
    /* the run's macroblocks that lie inside this picture (a launch's grid may be wider / higher than a picture of its batch); every macroblock that is not of
     * the fast kind is left to the general code below */
    const int n_row = pic_ok && mb_x < pic.hot.mb_width && mb_y < pic.hot.mb_height ? (n < pic.hot.mb_width - mb_x ? n : pic.hot.mb_width - mb_x) : 0;
    uint32_t deferred = 0;
    if (n_row > 0) {
        /* The pipeline of a run of fast macroblocks.  In macroblock i's turn:
         *   the windows of macroblock i + 1 are requested (into the other window set) — they have this whole turn to arrive;
         *   ONE wait: everything requested for macroblock i in the turn before (windows, coefficients) has landed — only the two requests just made
         *   may still be out (loads complete in issue order);
         *   coefficients -> residual in registers; the coefficients of macroblock i + 1 are requested into the place just read;
         *   prediction from window set i & 1, residual on top, straight into the picture.
         * What was not requested ahead (the run's first macroblock, the one after a macroblock of another kind) is requested at the head of its own turn. */
        const int mb_xy0 = mb_y * pic.hot.mb_width + mb_x;
        const FqRun run = fq_describe(pic, mb_xy0, mb_x, mb_y, n_row);
        uint32_t ra = run.a, rw = run.wts, rn = run.nnz, rd = run.d, re = run.e;
        MI355_PIN(ra); MI355_PIN(rw); MI355_PIN(rn); MI355_PIN(rd); MI355_PIN(re);            /* its loads have landed before the first request of the run goes out */
        bool pre_w = false, pre_c = false;      /* this macroblock's windows / coefficients were requested in the turn before */
        uint32_t a = fq_lane_word(ra, 0);
        for (int i = 0; i < n_row; i++) {
            const bool has_next = i + 1 < n_row;
            const uint32_t an = has_next ? fq_lane_word(ra, i + 1) : 0u;
            const int woff = (i & 1) * FQ_WSTEP;
            bool next_w = false, next_c = false;
            if (a & FQA_FAST) {
                auto issue = [&](int m, uint32_t am, int set) {
                    const uint32_t d = fq_lane_word(rd, m), e = fq_lane_word(re, m);
                    if (am & FQA_INSIDE) fq_windows_issue_inside(s, k, pic.desc, pic.hot, (int)((am >> 17) & 31u), (int16_t)(d & 0xFFFFu), (int16_t)(e & 0xFFFFu), (int)d >> 16, (int)e >> 16, set);
                    else fq_windows_issue(s, k, pic.desc, pic.hot, (int)((am >> 17) & 31u), (int16_t)(d & 0xFFFFu), (int16_t)(e & 0xFFFFu), (int)d >> 16, (int)e >> 16, set);
                };
                FQ_MARK("issue");
                if (!pre_w) issue(i, a, woff);
                if ((a & FQA_RESID) && !pre_c) fq_coef_dma(s, k, pic.coef, mb_xy0 + i);
                next_w = (an & FQA_FAST) != 0;
                if (next_w) issue(i + 1, an, woff ^ FQ_WSTEP);
                /* LOADS complete in issue order, a store's acknowledgement may overtake them: "at most 2 outstanding" with the two window requests of
                 * macroblock i + 1 youngest means every older load has landed whatever the predecessor's store is doing (a count that allowed for the
                 * store as well returned with coefficients still in flight: found on the device, never in the emulator).  Without younger loads: all. */
                if (next_w) fq_wait_vm2(); else fq_wait_vm0();
                MI355_WAVE_SYNC();
                FQ_MARK("idct");
                if (a & FQA_RESID) fq_idct(s, k, rl, fq_lane_word(rn, i), (a & FQA_CHROMA) != 0, pic.mb + (mb_xy0 + i));
                MI355_WAVE_SYNC();                                         /* every lane has read its coefficients */
                FQ_MARK("coefnext");
                next_c = next_w && (an & FQA_RESID);
                if (next_c) fq_coef_dma(s, k, pic.coef, mb_xy0 + i + 1);
                if (a & (FQA_PATCH_Y | FQA_PATCH_C)) {
                    const uint32_t e = fq_lane_word(re, i);
                    fq_windows_patch(s, k, (a & FQA_PATCH_Y) != 0, (a & FQA_PATCH_C) != 0, (int16_t)(e & 0xFFFFu), (int)e >> 16, woff, pic.hot.mb_width);
                    MI355_WAVE_SYNC();
                }
                FQ_MARK("luma");
                fq_luma(s, k, (int)(a & 31u) + woff, (int)((a >> 7) & 15u), (a & FQA_RESID) != 0);
                FQ_MARK("chroma");
                fq_chroma(s, k, (int)((a >> 5) & 3u) + woff, fq_lane_word(rw, i), (a & FQA_RESID) != 0);
                MI355_WAVE_SYNC();
                FQ_MARK("store");
                fq_store(s, pic.hot.recon[0] + tile_y_off(mb_x + i, mb_y, pic.hot.recon_stride[0]), pic.hot.recon[1] + tile_c_off(mb_x + i, mb_y, pic.hot.recon_stride[1]));
                MI355_WAVE_SYNC();
            } else if (!(a & FQA_INTRA)) {
                deferred |= (a & FQA_TWO) ? 0x10000u << i : 1u << i;          /* upper half: two partitions from list 0 (fq_two), lower: the general code */
            }
            FQ_MARK("tail");
            pre_w = next_w; pre_c = next_c;
            a = an;
        }
    }
    /* partitions, two lists, weights, the 8x8 transform: left to recon_inter_rest, a launch of its own behind this one (the general code inside this
     * kernel — a function called once per such macroblock at the end of the run — took 17 us a macroblock where a wave of its own takes 9: its
     * registers saved and restored through scratch memory on every call, nothing of one macroblock overlapping the next) */
    if (lane_id() == 0) rest[wave] = deferred;
}

/* ---- the macroblocks the runs left: wave g looks at the words of runs FQ_REST g .. FQ_REST g + FQ_REST - 1 (one word a run: bit i = the run's macroblock i is an
 * inter macroblock of another kind than the plain one) and takes them one at a time through h264_recon_dev.h's code.  A batch of plain P macroblocks costs this
 * launch nruns / FQ_REST waves that read a few words and end (some 10 us per 2048 pictures of 1080p); a batch of partitioned macroblocks gives every wave some
 * hundred of them, a millisecond's work: the fewer runs per wave, the less the last round of waves leaves idle. ---- */
constexpr int FQ_REST = 8;
__device__ __forceinline__ void recon_inter_rest(MbLds &s, const mi355_h264_frame *__restrict__ frames, int max_w, int max_h, int run, int runs_row,
                                                 unsigned long long inv_runs, unsigned long long inv_h, int nruns, int ngroups, int per_xcd, const uint32_t *__restrict__ rest)
{
    const int g = xcd_linear((int)blockIdx.x, per_xcd);
    if (g >= ngroups) return;
    const int r = FQ_REST * g + (lane_id() & (FQ_REST - 1));
    const uint32_t mine = r < nruns ? rest[r] : 0u;
    if (!__any(mine != 0)) return;
    auto place = [&](int j, int &f, int &mb_x, int &mb_y) {
        const int w = FQ_REST * g + j;
        const int row = div_magic(w, inv_runs);
        mb_x = (w - row * runs_row) * run;
        f = div_magic(row, inv_h); mb_y = row - f * max_h;
    };
    /* first the macroblocks of two list-0 partitions: the fast path's code, its lane constants made once */
    if (__any((mine >> 16) != 0)) {
        ResidLane rl;
        resid_lane_compute(rl);
        const FqLane k = fq_lane();
        for (int j = 0; j < FQ_REST; j++) {
            uint32_t m = fq_lane_word(mine, j) >> 16;
            if (!m) continue;
            int f, mb_x, mb_y;
            place(j, f, mb_x, mb_y);
            const mi355_h264_frame &frd = frames[f];
            if (uniform(frd.surface_layout) != MI355_SURFACE_TILED) continue;
            FqPic pic;
            pic.hot = frame_hot(frd);
            pic.mb = pic.hot.mb; pic.mv0 = pic.hot.mv[0]; pic.coef = pic.hot.coef; pic.desc = &frd;
            while (m) {
                const int i = __builtin_ctz(m);
                m &= m - 1;
                fq_two(s, k, rl, pic, mb_y * pic.hot.mb_width + mb_x + i, mb_x + i, mb_y);
                fq_wait_vm0();                                  /* the macroblock's stores have left its LDS tile */
                MI355_WAVE_SYNC();
            }
        }
    }
    for (int j = 0; j < FQ_REST; j++) {
        uint32_t m = fq_lane_word(mine, j) & 0xFFFFu;
        if (!m) continue;
        int f, mb_x, mb_y;
        place(j, f, mb_x, mb_y);
        if (uniform(frames[f].surface_layout) != MI355_SURFACE_TILED) continue;
        while (m) {
            const int i = __builtin_ctz(m);
            m &= m - 1;
            recon_inter_mb<false, true>(s, frames[f], mb_x + i, mb_y);
            fq_wait_vm0();                                      /* the macroblock's stores have left its LDS tile */
            MI355_WAVE_SYNC();
        }
    }
}

}  // namespace
#endif
