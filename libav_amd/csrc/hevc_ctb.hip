/*
 * hevc_ctb.hip — mi355_hevc_recon_ctbs_dev (include/mi355_hevc_batch.h): the inter reconstruction of a coding tree block as ONE workgroup —
 * hls_coding_quadtree's prediction units (hevcdec.c:1695-1885: luma_mc / chroma_mc, put_unweighted_pred / weighted_pred) and transform units
 * (hls_transform_unit :1238-1260: idct*, add_residual) of the block, its samples kept in LDS from the first prediction to the last residual
 * and stored to the picture once, in whole lines.
 *
 * A workgroup = CTB_WAVES wavefronts; the block's prediction jobs are dealt to the waves, a workgroup barrier, its transform units likewise, a
 * barrier, the store.  A wave works on its job alone, on scratch of its own: the wave-level bodies of hevc_dev.h / hevc_batch_dev.h are
 * instantiated here with a WAVE-scope rendezvous between their LDS phases (MI355_HEVC_SYNC, a namespace of their own) — the waves of a workgroup
 * run different jobs of different lengths, a workgroup barrier inside a job would not be reached by all of them.
 * Hot shapes go through the matrix unit (hevc_ctb_fast.h); everything else through the bodies of the batch kernels, writing to the tile.
 */
#define MI355_HEVC_NS mi355_ctb
#define MI355_HEVC_SYNC() MI355_WAVE_SYNC()
#include <type_traits>
#include "mi355_rt.h"
#include "hevc_dev.h"
#include "../../include/mi355_hevc_batch.h"

using namespace mi355;
using namespace mi355_ctb;

namespace {

#include "hevc_batch_dev.h"
#include "hevc_ctb_fast.h"

/* waves per workgroup: the matrix-path kernel deals a block's jobs to more, shorter-lived waves (a 64x64 block of four 32x32 prediction units is 4 luma + 4 chroma jobs,
 * then 6 transform units: with 8 waves a round each); the general kernel keeps 4 (its per-wave scratch is larger) */
#ifndef MI355_CTB_WAVES
#define MI355_CTB_WAVES 8
#endif
constexpr int CTB_WAVES_FAST = MI355_CTB_WAVES, CTB_WAVES_GENERAL = 4;
/* Workgroups that stay and take turns (blocks blockIdx.x, + gridDim.x, ...) with the next block's records on their way while one is worked on: built and
 * measured in round 6 (profiles/r06_experiments.md) — 2.35 ms against 2.08 ms for a workgroup per block: with the records fetched a dword per lane the waves no
 * longer wait for records but for each other and for the vector unit, and the turn loop costs registers (88 against 71: two workgroups per CU instead of three
 * unless values go to scratch memory).  Kept behind this switch. */
#ifndef MI355_CTB_PERSIST
#define MI355_CTB_PERSIST 0
#endif
constexpr bool CTB_PERSIST = MI355_CTB_PERSIST != 0;
#ifndef MI355_CTB_WAVES_ATTR
#if MI355_CTB_PERSIST
#define MI355_CTB_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define MI355_CTB_WAVES_ATTR
#endif
#endif
/* the block's samples in LDS: rows 16 bytes longer than a full row, so that the rows a matrix product's sixteen lanes write (8 bytes each, one row per
 * lane) spread over the banks two by two instead of all sixteen meeting in two banks; still 16-byte aligned for the way out */
constexpr int CTB_PITCH_Y = 64 * 2 + 16, CTB_PITCH_C = 32 * 2 + 16;
struct __attribute__((aligned(16))) CtbTile {
    uint8_t y[64 * CTB_PITCH_Y];
    uint8_t c[2][32 * CTB_PITCH_C];
};
union __attribute__((aligned(16))) CtbScratch {
    HevcMcScratch mc;
    IdctScratch tu;
    CfWin win;
};

/* a / b for wave-uniform a, b with a quotient below 64 (a row inside a block): six compare steps on the scalar unit (a reciprocal would run on the vector unit) */
__device__ __forceinline__ int ctb_div(unsigned a, unsigned b)
{
#ifdef MI355_HIP_EMU_H
    return (int)(a / b);
#else
    unsigned q = 0;
#pragma unroll
    for (unsigned bit = 32; bit; bit >>= 1) if ((q + bit) * b <= a) q += bit;
    return (int)q;
#endif
}

/* the block's three planes as scalars (no array indexed at run time: that would live in scratch memory) */
struct CtbGeom {
    uint8_t *d0, *d1, *d2;
    int s0, s1, s2;
    int w, h, px;
    __device__ __forceinline__ uint8_t *dst(int pl) const { return pl == 0 ? d0 : (pl == 1 ? d1 : d2); }
    __device__ __forceinline__ int stride(int pl) const { return pl == 0 ? s0 : (pl == 1 ? s1 : s2); }
    /* which plane a pointer of a job lies in (first_plane or later), and where: the planes are separate ranges of rows */
    __device__ __forceinline__ bool locate(const uint8_t *p, int first_plane, int &pl, int &x, int &y) const
    {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const ptrdiff_t o = p - dst(q);
            const int rows = q ? h >> 1 : h, st = stride(q);
            if (q >= first_plane && o >= 0 && o < (ptrdiff_t)rows * st) {
                pl = q;
                y = ctb_div((unsigned)o, (unsigned)st);
                x = ((int)o - y * st) >> (px - 1);
                return true;
            }
        }
        return false;
    }
};
__device__ __forceinline__ uint8_t *tile_at(CtbTile &t, int pl, int x, int y, int px)
{
    return pl ? t.c[pl - 1] + y * CTB_PITCH_C + x * px : t.y + y * CTB_PITCH_Y + x * px;
}

/* the tile <-> the picture, V bytes per thread and access (rows of `rb` bytes, V divides rb, the plane's pointer and its stride) */
template <int V, bool LOAD>
__device__ __forceinline__ void tile_move(uint8_t *t, int tpitch, uint8_t *g, int gstride, int rb, int rows, int tid, int nthreads)
{
    const int per = rb / V, n = per * rows, inv = mi355_inv20(per);
    for (int i = tid; i < n; i += nthreads) {
        const int r = mi355_div20(i, inv), k = i - r * per;
        uint8_t *tp = t + r * tpitch + V * k, *gp = g + (ptrdiff_t)r * gstride + V * k;
        if (LOAD) __builtin_memcpy(tp, gp, V); else __builtin_memcpy(gp, tp, V);
    }
}
template <bool LOAD>
__device__ __forceinline__ void tile_all(CtbTile &t, const CtbGeom &G, int tid, int nthreads)
{
#pragma unroll
    for (int pl = 0; pl < 3; pl++) {
        const int rb = (pl ? G.w >> 1 : G.w) * G.px, rows = pl ? G.h >> 1 : G.h, st = G.stride(pl);
        uint8_t *tp = pl ? t.c[pl - 1] : t.y, *g = mi355_global(G.dst(pl));
        const int tpitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
        const unsigned al = (unsigned)(uintptr_t)g | (unsigned)st | (unsigned)rb;
        if ((al & 15) == 0) tile_move<16, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 7) == 0) tile_move<8, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 3) == 0) tile_move<4, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else if ((al & 1) == 0) tile_move<2, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
        else tile_move<1, LOAD>(tp, tpitch, g, st, rb, rows, tid, nthreads);
    }
}

/* what the matrix path takes (include/mi355_hevc_batch.h) */
template <bool WIDE> __device__ __forceinline__ bool mc_is_fast(const mi355_hevc_mcpred_job &j)
{
    return j.kind == MI355_HEVC_PRED_PUT && (j.width & 15) == 0 && (j.height & 15) == 0 && (j.src0_stride & (WIDE ? 15 : 7)) == 0;
}
__device__ __forceinline__ bool tu_is_fast(const mi355_hevc_tu_job &j)
{
    return j.dst && j.kind == MI355_HEVC_TU_IDCT && (j.log2_size == 4 || j.log2_size == 5) && ((uintptr_t)j.coeffs & 15) == 0;
}

/* one prediction job of the block -> the tile.  GENERAL: the bodies of the batch kernels are compiled in; without them a job the matrix path does not take
 * is reported (false) and nothing is done. */
template <bool WIDE, bool GENERAL, class Scratch>
__device__ __forceinline__ bool ctb_predict(CtbTile &tile, Scratch &s, const CtbGeom &G, mi355_hevc_mcpred_job j, int bd, int lane)
{
    int pl = 0, x = 0, y = 0, plb = 0, xb = 0, yb = 0;
    if (!G.locate(j.dst, j.chroma ? 1 : 0, pl, x, y)) return true;
    if (j.chroma == 2 && !G.locate(j.dst_b, 1, plb, xb, yb)) return true;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y, px = G.px;
    uint8_t *t0 = tile_at(tile, pl, x, y, px), *t1 = j.chroma == 2 ? tile_at(tile, plb, xb, yb, px) : nullptr;
    if (mc_is_fast<WIDE>(j)) {
#ifdef MI355_EXP_CTB_NOMC
        return true;
#endif
        const int before = j.chroma ? 1 : 3, bx = j.mx0 ? before : 0, by = j.my0 ? before : 0;
        CfPass ph, pv;
        if (j.chroma) {
            ph = j.mx0 ? cf_pass(k_epel[j.mx0], 4, bd - 8) : cf_pass_one(1, 0);
            pv = j.my0 ? cf_pass(k_epel[j.my0], 4, j.mx0 ? 6 : bd - 8) : cf_pass_one(j.mx0 ? 1 : 1 << (14 - bd), 0);
        } else {
            ph = j.mx0 ? cf_pass(k_qpel[j.mx0], 8, bd - 8) : cf_pass_one(1, 0);
            pv = j.my0 ? cf_pass(k_qpel[j.my0], 8, j.mx0 ? 6 : bd - 8) : cf_pass_one(j.mx0 ? 1 : 1 << (14 - bd), 0);
        }
        if (j.chroma == 2 && j.width == 16 && j.height == 16 && (((uintptr_t)j.src0 ^ (uintptr_t)j.src0_b) & (WIDE ? 15 : 7)) == 0) {
            /* both planes' 16x16 tiles: their windows in ONE round trip */
            MI355_PIN(lane);
            cf_mc_tile_pair<WIDE>(s.win, mi355_global(j.src0) - (ptrdiff_t)by * j.src0_stride - (ptrdiff_t)bx * px, mi355_global(j.src0_b) - (ptrdiff_t)by * j.src0_stride - (ptrdiff_t)bx * px,
                                  j.src0_stride, ph, pv, bd, t0, t1, pitch, lane);
            return true;
        }
        /* (one tile at a time: unrolled, the tiles of a job — two planes, up to four tiles a plane — would be scheduled into each other and hold twice the registers) */
#pragma unroll 1
        for (int plane = 0; plane < (j.chroma == 2 ? 2 : 1); plane++) {
            const uint8_t *src = mi355_global(plane ? j.src0_b : j.src0);
            uint8_t *tp = plane ? t1 : t0;
#pragma unroll 1
            for (int ty = 0; ty < j.height; ty += 32)
#pragma unroll 1
            for (int tx = 0; tx < j.width; tx += 32) {
                const int tw = j.width - tx < 32 ? j.width - tx : 32, th = j.height - ty < 32 ? j.height - ty : 32;
                /* the lane's number anew for every tile: the constants a tile derives from it (piece and row of the fetch, operand addresses, tap words) are loop
                 * invariants to the compiler, and taken out of the loops they hold ~50 registers for the life of the kernel */
                MI355_PIN(lane);
                cf_mc_tile<WIDE>(s.win, src + (ptrdiff_t)(ty - by) * j.src0_stride + (ptrdiff_t)(tx - bx) * px, j.src0_stride, tw, th, ph, pv, bd,
                                 tp + ty * pitch + tx * px, pitch, lane);
            }
        }
        return true;
    }
    if constexpr (GENERAL) {
        /* the body of k_hevc_mcpred_batch with the tile as its picture */
        j.dst = t0; j.dst_b = t1; j.dst_stride = pitch;
        int16_t *const keep = s.mc.tmp + HEVC_MC_BI_ROWS * HEVC_MC_TPITCH;
        switch ((j.chroma ? 4 : 0) + (j.kind & 3)) {
        case 0: hevc_mcpred_taps<8, 0, false>(j, bd, s.mc, keep); break;   case 1: hevc_mcpred_taps<8, 1, false>(j, bd, s.mc, keep); break;
        case 2: hevc_mcpred_taps<8, 2, false>(j, bd, s.mc, keep); break;   case 3: hevc_mcpred_taps<8, 3, false>(j, bd, s.mc, keep); break;
        case 4: hevc_mcpred_taps<4, 0, false>(j, bd, s.mc, keep); break;   case 5: hevc_mcpred_taps<4, 1, false>(j, bd, s.mc, keep); break;
        case 6: hevc_mcpred_taps<4, 2, false>(j, bd, s.mc, keep); break;   default: hevc_mcpred_taps<4, 3, false>(j, bd, s.mc, keep); break;
        }
        MI355_WAVE_SYNC();
        return true;
    }
    return false;
}

/* one transform unit of the block -> added to the tile; `pre`: its coefficients were requested before (cf_idct_load) and wait in `raw` */
template <bool WIDE, bool GENERAL, class Scratch>
__device__ __forceinline__ bool ctb_residual(CtbTile &tile, Scratch &s, const CtbGeom &G, mi355_hevc_tu_job j, int bd, int lane, bool pre, CfRaw &raw)
{
    int pl = 0, x = 0, y = 0;
    if (!j.dst) return !GENERAL ? false : true;
    if (!G.locate(j.dst, 0, pl, x, y)) return true;
    const int pitch = pl ? CTB_PITCH_C : CTB_PITCH_Y;
    uint8_t *tp = tile_at(tile, pl, x, y, G.px);
    if (tu_is_fast(j)) {
#ifdef MI355_EXP_CTB_NOTU
        return true;
#endif
        const uint8_t *c = reinterpret_cast<const uint8_t *>(mi355_global(j.coeffs));
        if (j.log2_size == 5) { if (!pre) cf_idct_load<5>(raw, c, j.col_limit, lane); cf_idct_run<5, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        else { if (!pre) cf_idct_load<4>(raw, c, j.col_limit, lane); cf_idct_run<4, WIDE>(raw, j.col_limit, bd, tp, pitch, lane); }
        MI355_WAVE_SYNC();
        return true;
    }
    if constexpr (GENERAL) {
        j.dst = tp; j.dst_stride = pitch;
        hevc_residual_run<false>(s.tu, j, lane < 32, lane >> 5, lane & 31, bd);
        MI355_WAVE_SYNC();
        return true;
    }
    return false;
}

struct FastScratch { CfWin win; };

/* Job records are fetched a dword per lane and read out of that register as scalars (v_readlane).  The compiler cannot know the records are read-only (the kernel
 * stores through other pointers), so reading a record field by field is a vector load per field, each waited for before the branch that depends on it: 20-odd
 * dependent memory round trips per wave before its first sample was requested — two thirds of this kernel's time (profiles/r06_experiments.md).  Now three: the
 * block's record, the wave's first prediction job and first transform unit together, then samples and coefficients. */
static_assert(sizeof(mi355_hevc_ctb_job) == 64 && sizeof(mi355_hevc_mcpred_job) == 80 && sizeof(mi355_hevc_tu_job) == 24, "k_hevc_recon_ctbs reads its records by dword index");
template <int BASE> __device__ __forceinline__ mi355_hevc_mcpred_job ctb_mc_record(int rec)
{
    uint32_t w[20];
#pragma unroll
    for (int k = 0; k < 20; k++) w[k] = (uint32_t)lane_value(rec, BASE + k);
    mi355_hevc_mcpred_job j;
    __builtin_memcpy(&j, w, sizeof(j));
    return j;
}
template <int BASE> __device__ __forceinline__ mi355_hevc_tu_job ctb_tu_record(int rec)
{
    uint32_t w[6];
#pragma unroll
    for (int k = 0; k < 6; k++) w[k] = (uint32_t)lane_value(rec, BASE + k);
    mi355_hevc_tu_job j;
    __builtin_memcpy(&j, w, sizeof(j));
    return j;
}
/* lanes 0..19: the prediction job at `mc` (when have_mc), lanes 32..37: the transform unit at `tu` (when have_tu); `safe`: any readable address */
__device__ __forceinline__ int ctb_fetch_records(const mi355_hevc_mcpred_job *mc, bool have_mc, const mi355_hevc_tu_job *tu, bool have_tu, const void *safe, int lane)
{
    const uint32_t *p = reinterpret_cast<const uint32_t *>(safe);
    if (lane < 32) { if (have_mc) p = reinterpret_cast<const uint32_t *>(mc) + (lane < 20 ? lane : 19); }
    else if (have_tu) p = reinterpret_cast<const uint32_t *>(tu) + (lane < 38 ? lane - 32 : 5);
    return (int)*mi355_global_v(p);
}

/* a block's record as scalars */
struct CtbHead {
    CtbGeom G;
    int flags, n_mc, n_tu;
    const mi355_hevc_mcpred_job *mc;
    const mi355_hevc_tu_job *tu;
};
template <bool WIDE>
__device__ __forceinline__ CtbHead ctb_head(int crec, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus)
{
    CtbHead H;
    H.G.d0 = reinterpret_cast<uint8_t *>((uintptr_t)(uint32_t)lane_value(crec, 0) | ((uintptr_t)(uint32_t)lane_value(crec, 1) << 32));
    H.G.d1 = reinterpret_cast<uint8_t *>((uintptr_t)(uint32_t)lane_value(crec, 2) | ((uintptr_t)(uint32_t)lane_value(crec, 3) << 32));
    H.G.d2 = reinterpret_cast<uint8_t *>((uintptr_t)(uint32_t)lane_value(crec, 4) | ((uintptr_t)(uint32_t)lane_value(crec, 5) << 32));
    H.G.s0 = lane_value(crec, 6); H.G.s1 = lane_value(crec, 7); H.G.s2 = lane_value(crec, 8);
    H.G.w = lane_value(crec, 9) & 0xFFFF; H.G.h = (int)((uint32_t)lane_value(crec, 9) >> 16); H.G.px = WIDE ? 2 : 1;
    H.flags = (lane_value(crec, 10) >> 8) & 0xFF;
    H.mc = mc + lane_value(crec, 11); H.n_mc = lane_value(crec, 12);
    H.tu = tus + lane_value(crec, 13); H.n_tu = lane_value(crec, 14);
    return H;
}
/* is job `i` of the block (prediction jobs first, then transform units) outside the matrix path?  What the test needs of a record, as fetched: a prediction
 * job's dwords 6 (src0_stride) and 9 (width, height, chroma, kind), a unit's dwords 0 (coeffs, low half), 2 and 3 (dst) and 5 (log2_size, col_limit, kind) */
struct CtbProbe { uint32_t a, b, c, d; int what; };      /* what: 0 nothing, 1 prediction job, 2 transform unit */
__device__ __forceinline__ CtbProbe ctb_probe(const CtbHead &H, int i)
{
    CtbProbe q;
    q.a = q.b = q.c = q.d = 0u; q.what = 0;
    if (i < H.n_mc) {
        const uint32_t *r = reinterpret_cast<const uint32_t *>(mi355_global_v(H.mc) + i);
        q.a = r[6]; q.b = r[9]; q.what = 1;
    } else if (i < H.n_mc + H.n_tu) {
        const uint32_t *r = reinterpret_cast<const uint32_t *>(mi355_global_v(H.tu) + (i - H.n_mc));
        q.a = r[0]; q.b = r[2]; q.c = r[3]; q.d = r[5]; q.what = 2;
    }
    return q;
}
template <bool WIDE> __device__ __forceinline__ bool ctb_probe_odd(const CtbProbe &q)
{
    if (q.what == 1) {
        mi355_hevc_mcpred_job j;
        j.src0_stride = (int32_t)q.a; j.width = (uint8_t)q.b; j.height = (uint8_t)(q.b >> 8); j.kind = (uint8_t)(q.b >> 24);
        return !mc_is_fast<WIDE>(j);
    }
    if (q.what == 2) {
        mi355_hevc_tu_job j;
        j.coeffs = reinterpret_cast<int16_t *>((uintptr_t)q.a); j.dst = reinterpret_cast<uint8_t *>((uintptr_t)q.b | ((uintptr_t)q.c << 32));
        j.log2_size = (uint8_t)q.d; j.kind = (uint8_t)(q.d >> 16);
        return !tu_is_fast(j);
    }
    return false;
}

/* GENERAL = false: the matrix-path kernel — it takes the blocks ALL of whose jobs are of the matrix path's shapes and leaves every other block untouched
 * (the workgroup's threads look at a job record each before anything else; nothing is stored);
 * GENERAL = true: the kernel with every body compiled in.  `only_rest`: it follows the matrix-path kernel on the same list and takes exactly the blocks that one
 * left (the same test).
 * A workgroup stays and takes blocks blockIdx.x, + gridDim.x, ...: while it works on one block the NEXT block's records are on their way (the block's record is
 * requested at the top of the turn, the wave's first prediction job and transform unit of it — and a job per thread for the shape test — once that has arrived, after
 * the predictions), so that a turn begins with the requests for its samples and coefficients instead of with three dependent round trips. */
template <bool WIDE, bool GENERAL, int NW>
__device__ __forceinline__ void ctb_turns(const mi355_hevc_ctb_job *ctbs, int n_ctbs, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus, int bd, int only_rest, uint32_t *error_word)
{
    typedef typename std::conditional<GENERAL, CtbScratch, FastScratch>::type Scratch;
    constexpr int NT = 64 * NW;
    __shared__ CtbTile tile;
    __shared__ Scratch scratch[NW];
    __shared__ int s_flag[2];
    const int tid = (int)threadIdx.x, wave = uniform(tid >> 6), lane = lane_id();
    /* which blocks are whose: the general kernel behind the matrix-path one looks at every job of a block first (a job per thread); the matrix-path kernel's
     * waves look at their own jobs as they come and drop the block before anything is stored */
    const bool probed = GENERAL && only_rest;
    if (tid < 2) s_flag[tid] = 0;
    int c = (int)blockIdx.x;
    if (c >= n_ctbs) return;
    /* workgroups go to the eight XCDs in turn: XCD k takes the k-th eighth of the list, in order — a caller's blocks run along rows, so neighbours (whose chroma rows are the
     * two halves of one line, whose windows overlap) meet in ONE L2 at about the same time instead of in eight (1.98 -> 1.96 ms; a list that does not divide: as it comes) */
    if (!CTB_PERSIST && !(n_ctbs & 7)) c = (c & 7) * (n_ctbs >> 3) + (c >> 3);
    CtbHead H = ctb_head<WIDE>((int)mi355_global_v(reinterpret_cast<const uint32_t *>(ctbs + c))[lane & 15], mc, tus);
    int rec = ctb_fetch_records(H.mc + wave, wave < H.n_mc, H.tu + wave, wave < H.n_tu, ctbs + c, lane);
    __syncthreads();
    for (int k = 0;; k++) {
        const int cn = c + (int)gridDim.x;
        const bool more = CTB_PERSIST && cn < n_ctbs;
        /* the lane's number, opaque to the compiler from here on: everything derived from it (tile addresses, operand tables, tap words) is worked out inside
         * the turn that uses it — taken out of the loop as invariants those values hold 30 registers through every phase (104 in all: four waves per SIMD
         * instead of the six the LDS allows) */
        /* the lane's number anew for each phase: what a phase derives from it (operand tables, tap words, tile addresses) is worked out inside that phase and not
         * held in registers through the others (the transform's operand words are loads from constant tables: hoisted to the top they occupy 14 registers through the
         * prediction) */
        int lane_t = lane, tid_t = tid, lane_u = lane;
        MI355_PIN(lane_t);
        if (CTB_PERSIST) MI355_PIN(tid_t);
        /* the next block's record: needed after the predictions */
        const int crec_n = CTB_PERSIST ? (int)mi355_global_v(reinterpret_cast<const uint32_t *>(ctbs + (more ? cn : c)))[lane & 15] : 0;
        bool mine = true;
        if (GENERAL && probed) {
            bool odd = false;
            for (int i = tid; i < H.n_mc + H.n_tu; i += NT) odd = odd || ctb_probe_odd<WIDE>(ctb_probe(H, i));
            if (odd) s_flag[0] = 1;
            __syncthreads();
            mine = s_flag[0] != 0;
            __syncthreads();
            if (tid == 0) s_flag[0] = 0;
        }
        mi355_hevc_mcpred_job j_mc = ctb_mc_record<0>(rec);
        mi355_hevc_tu_job j_tu = ctb_tu_record<32>(rec);
        /* the first transform unit's coefficients are requested now and arrive while the wave predicts */
        CfRaw raw;
        bool pre = false;
        if (mine && wave < H.n_tu && tu_is_fast(j_tu)) {
            const uint8_t *cf = reinterpret_cast<const uint8_t *>(mi355_global(j_tu.coeffs));
            if (j_tu.log2_size == 5) cf_idct_load<5>(raw, cf, j_tu.col_limit, lane_t); else cf_idct_load<4>(raw, cf, j_tu.col_limit, lane_t);
            pre = true;
        }
        if (mine && (H.flags & MI355_HEVC_CTB_PARTIAL)) { tile_all<true>(tile, H.G, tid_t, NT); __syncthreads(); }
        Scratch &s = scratch[wave];
        bool ok = true;
        if (mine) {
            for (int i = wave; i < H.n_mc; i += NW) {
                if (i != wave) j_mc = ctb_mc_record<0>(ctb_fetch_records(H.mc + i, true, H.tu, false, ctbs + c, lane));
                ok = ctb_predict<WIDE, GENERAL>(tile, s, H.G, j_mc, bd, lane_t) && ok;
            }
            if (!GENERAL) {
                /* a unit the matrix path does not take: known before any residual is added */
                ok = ok && (wave >= H.n_tu || tu_is_fast(j_tu));
                for (int i = wave + NW; i < H.n_tu; i += NW) ok = ok && tu_is_fast(ctb_tu_record<32>(ctb_fetch_records(H.mc, false, H.tu + i, true, ctbs + c, lane)));
                if (!ok) s_flag[k & 1] = 1;
            }
        }
        /* the next block: its record has arrived; request the wave's first jobs of it */
        CtbHead Hn = H;
        if (CTB_PERSIST) Hn = ctb_head<WIDE>(crec_n, mc, tus);
        const int rec_n = more ? ctb_fetch_records(Hn.mc + wave, wave < Hn.n_mc, Hn.tu + wave, wave < Hn.n_tu, ctbs + cn, lane) : rec;
        __syncthreads();
        if (!GENERAL && s_flag[k & 1]) {
            /* left to the general kernel; a caller that promised there are no such blocks (MI355_HEVC_RECON_UNIFORM) finds out */
            mine = false;
            if (error_word && tid == 0) atomicOr(error_word, (uint32_t)MI355_ERR_CTB_NOT_UNIFORM);
        }
        MI355_PIN(lane_u);
        if (mine)
            for (int i = wave; i < H.n_tu; i += NW) {
                if (i != wave) j_tu = ctb_tu_record<32>(ctb_fetch_records(H.mc, false, H.tu + i, true, ctbs + c, lane));
                ctb_residual<WIDE, GENERAL>(tile, s, H.G, j_tu, bd, lane_u, pre && i == wave, raw);
            }
        __syncthreads();
        if (!GENERAL && tid == 0) s_flag[k & 1] = 0;         /* everyone has read it; it is set again two turns from now at the earliest */
#ifndef MI355_EXP_CTB_NOSTORE
        if (mine) tile_all<false>(tile, H.G, tid_t, NT);
#endif
        if (!more) break;
        c = cn; H = Hn; rec = rec_n;
    }
}
/* the matrix-path kernel with registers for six waves per SIMD (three workgroups of eight waves per CU, what its LDS allows; a handful of values move to scratch
 * memory for it); the general kernel as the compiler sizes it */
template <bool WIDE, int NW>
__global__ void __launch_bounds__(64 * NW) MI355_CTB_WAVES_ATTR k_hevc_recon_ctbs(const mi355_hevc_ctb_job *ctbs, int n_ctbs, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus,
                                                                                  int bd, uint32_t *error_word)
{
    ctb_turns<WIDE, false, NW>(ctbs, n_ctbs, mc, tus, bd, 0, error_word);
}
template <bool WIDE, int NW>
__global__ void __launch_bounds__(64 * NW) k_hevc_recon_ctbs_general(const mi355_hevc_ctb_job *ctbs, int n_ctbs, const mi355_hevc_mcpred_job *mc, const mi355_hevc_tu_job *tus,
                                                                     int bd, int only_rest)
{
    ctb_turns<WIDE, true, NW>(ctbs, n_ctbs, mc, tus, bd, only_rest, nullptr);
}

}  // namespace

extern "C" int mi355_hevc_recon_ctbs_dev(const mi355_hevc_ctb_job *d_ctbs, int n_ctbs, const mi355_hevc_mcpred_job *d_mc, const mi355_hevc_tu_job *d_tus,
                                         int bit_depth, unsigned flags, void *stream)
{
    if (!bind()) { std::fprintf(stderr, "mi355dsp: HEVC batch entry point without mi355_init(); no CPU fallback\n"); std::abort(); }
    if (!d_ctbs || n_ctbs <= 0 || !(bit_depth == 8 || bit_depth == 9 || bit_depth == 10)) return -1;
    const hipStream_t st = (hipStream_t)stream;
    /* workgroups stay and take turns: as many as the device holds at once */
    static const int cus = [] { const int n = mi355_device_cus(); return n > 0 ? n : 256; }();
    auto resident = [](auto kernel, int threads) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        return per_cu;
    };
    auto grid_for = [&](int per_cu) { const long want = CTB_PERSIST ? (long)per_cu * cus : (long)n_ctbs; return dim3((unsigned)(n_ctbs < want ? n_ctbs : want)); };
    static const bool general_only = std::getenv("MI355_CTB_GENERAL_ONLY") != nullptr;      /* developer switch: every block through the general kernel */
    const bool uniform_promised = (flags & MI355_HEVC_RECON_UNIFORM) != 0;
    uint32_t *err = uniform_promised ? mi355::error_word() : nullptr;
    if (uniform_promised && !err) return -4;
    static const int waves = std::getenv("MI355_CTB_WAVES") ? std::atoi(std::getenv("MI355_CTB_WAVES")) : CTB_WAVES_FAST;      /* developer switch: 4 */
    if (!general_only) {
        if (waves == 4) {
            static const int r1 = resident(k_hevc_recon_ctbs<true, 4>, 256), r0 = resident(k_hevc_recon_ctbs<false, 4>, 256);
            if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs<true, 4>), grid_for(r1), dim3(256), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, err);
            else hipLaunchKernelGGL((k_hevc_recon_ctbs<false, 4>), grid_for(r0), dim3(256), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, err);
        } else {
            static const int r1 = resident(k_hevc_recon_ctbs<true, CTB_WAVES_FAST>, 64 * CTB_WAVES_FAST), r0 = resident(k_hevc_recon_ctbs<false, CTB_WAVES_FAST>, 64 * CTB_WAVES_FAST);
            if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs<true, CTB_WAVES_FAST>), grid_for(r1), dim3(64 * CTB_WAVES_FAST), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, err);
            else hipLaunchKernelGGL((k_hevc_recon_ctbs<false, CTB_WAVES_FAST>), grid_for(r0), dim3(64 * CTB_WAVES_FAST), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, err);
        }
    }
    if (general_only || !uniform_promised) {
        const int only_rest = general_only ? 0 : 1;
        static const int r1 = resident(k_hevc_recon_ctbs_general<true, CTB_WAVES_GENERAL>, 64 * CTB_WAVES_GENERAL), r0 = resident(k_hevc_recon_ctbs_general<false, CTB_WAVES_GENERAL>, 64 * CTB_WAVES_GENERAL);
        if (bit_depth > 8) hipLaunchKernelGGL((k_hevc_recon_ctbs_general<true, CTB_WAVES_GENERAL>), grid_for(r1), dim3(64 * CTB_WAVES_GENERAL), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, only_rest);
        else hipLaunchKernelGGL((k_hevc_recon_ctbs_general<false, CTB_WAVES_GENERAL>), grid_for(r0), dim3(64 * CTB_WAVES_GENERAL), 0, st, d_ctbs, n_ctbs, d_mc, d_tus, bit_depth, only_rest);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
