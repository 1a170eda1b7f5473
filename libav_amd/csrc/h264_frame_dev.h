/*
 * h264_frame_dev.h — small device helpers shared by the Tier-2 frame kernels (h264_frame.hip: reconstruction, surface
 * conversion; h264_deblock.hip: the loop filter): vector typedefs, macroblock-tile addressing, 16- / 8-byte moves.
 */
#ifndef MI355_H264_FRAME_DEV_H
#define MI355_H264_FRAME_DEV_H

#include "mi355_rt.h"
#include "h264_dev.h"
#include "../../include/mi355_h264_frame.h"

namespace mi355 {

typedef uint32_t mi355_u32x4 __attribute__((vector_size(16)));
typedef uint32_t mi355_u32x2 __attribute__((vector_size(8)));
typedef uint32_t mi355_u32x4u __attribute__((vector_size(16), aligned(4)));
typedef uint32_t mi355_u32x2u __attribute__((vector_size(8), aligned(4)));

__device__ __forceinline__ int blk_x4(int i) { return (i & 1) + 2 * ((i >> 2) & 1); }
__device__ __forceinline__ int blk_y4(int i) { return ((i >> 1) & 1) + 2 * (i >> 3); }
__device__ __forceinline__ int blk_index(int x4, int y4) { return (x4 & 1) + 2 * (y4 & 1) + 4 * (x4 >> 1) + 8 * (y4 >> 1); }

/* macroblock-tiled surfaces (mi355_h264_frame.h): plane 0 = 256-byte luma tiles, plane 1 = 128-byte chroma tiles (Cb, Cr) */
__device__ __forceinline__ uint32_t tile_y_off(int mb_x, int mb_y, int stride) { return (uint32_t)(__mul24(mb_y, stride) + mb_x * MI355_TILE_LUMA_BYTES); }
__device__ __forceinline__ uint32_t tile_c_off(int mb_x, int mb_y, int stride) { return (uint32_t)(__mul24(mb_y, stride) + mb_x * MI355_TILE_CHROMA_BYTES); }

/* 16 bytes of an LDS tile row (rows are 8-byte aligned: two 64-bit accesses) */
__device__ __forceinline__ uint4 lds16(const uint8_t *p)
{
    const uint2 a = reinterpret_cast<const uint2 *>(p)[0], b = reinterpret_cast<const uint2 *>(p)[1];
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void lds16(uint8_t *p, uint4 v)
{
    reinterpret_cast<mi355_u32x2 *>(p)[0] = mi355_u32x2{ v.x, v.y };
    reinterpret_cast<mi355_u32x2 *>(p)[1] = mi355_u32x2{ v.z, v.w };
}
/* 16 / 8 bytes between a picture row and an LDS tile row.  The aligned stores go through native vector types:
 * a HIP uint4 assignment is copied component by component and came out as four dword stores. */
__device__ __forceinline__ uint4 ld16(const uint8_t *p, bool al)
{
    if (al) return *reinterpret_cast<const uint4 *>(p);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void st16(uint8_t *p, uint4 v, bool al)
{
    if (al) { *reinterpret_cast<mi355_u32x4 *>(p) = mi355_u32x4{ v.x, v.y, v.z, v.w }; return; }
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
__device__ __forceinline__ uint2 ld8(const uint8_t *p, bool al)
{
    if (al) return *reinterpret_cast<const uint2 *>(p);
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    return make_uint2(w[0], w[1]);
}
__device__ __forceinline__ void st8(uint8_t *p, uint2 v, bool al)
{
    if (al) { *reinterpret_cast<mi355_u32x2 *>(p) = mi355_u32x2{ v.x, v.y }; return; }
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    w[0] = v.x; w[1] = v.y;
}

/* h264_frame_tiled.hip: launches k_recon_inter_tiled and k_recon_inter_rest (the tiled-only form of the inter reconstruction); false: no scratch words */
bool recon_inter_tiled_launch(const mi355_h264_frame *d_frames, int nframes, int max_w, int max_h, hipStream_t stream);
/* h264_frame_rest.hip: k_recon_inter_rest over the words (one per run) k_recon_inter_tiled wrote */
void recon_inter_rest_launch(const mi355_h264_frame *d_frames, int max_w, int max_h, int run, int runs_row, unsigned long long inv_runs, unsigned long long inv_h, int nruns,
                             const uint32_t *rest, hipStream_t stream);
/* h264_deblock.hip: the scratch words (ticket + progress counters) of a single-launch loop filter, one buffer per (thread, device, stream); the caller zeroes what it uses on `stream` */
uint32_t *sync_words(hipStream_t stream, size_t words);
}  // namespace mi355
namespace {
/* agent-scope accesses of the band hand-over (plain in the emulator: workgroups run one after the other there) */
#ifdef MI355_HIP_EMU_H
static inline uint32_t agent_load_u32(const uint32_t *p) { return *p; }
static inline void agent_store_u32(uint32_t *p, uint32_t v) { *p = v; }
static inline uint2 agent_load8(const uint8_t *p, bool al8) { return mi355::ld8(p, al8); }
static inline void agent_store8(uint8_t *p, uint2 v, bool al8) { mi355::st8(p, v, al8); }
static inline void agent_drain_stores() {}
static inline void agent_release() {}
static inline void agent_acquire() {}
static inline void wave_nap() { std::fprintf(stderr, "k_deblock_tiled: a band waits for a band that has not run (emulator: workgroups run in order)\n"); std::abort(); }
#else
__device__ __forceinline__ uint32_t agent_load_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint2 agent_load8(const uint8_t *p, bool al8)
{
    if (al8) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    return make_uint2(agent_load_u32(w), agent_load_u32(w + 1));
}
__device__ __forceinline__ void agent_store8(uint8_t *p, uint2 v, bool al8)
{
    if (al8) { __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    agent_store_u32(w, v.x); agent_store_u32(w + 1, v.y);
}
/* every store this wave has issued has left it (the write-through ones have reached memory) before the flag goes out; inline
 * asm: the compiler's own wait insertion may drop a wait it believes redundant (guide, G16 pitfall 12) */
__device__ __forceinline__ void agent_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void agent_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void agent_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void wave_nap() { __builtin_amdgcn_s_sleep(16); }
#endif
}  // namespace
/* sixteen bytes per lane from memory straight into LDS at lds_base + 16 * lane (lds_base wave-uniform) */
#ifdef MI355_HIP_EMU_H
template <bool AGENT> static inline void lds_dma16(const uint8_t *src, uint8_t *lds_base) { std::memcpy(lds_base + 16 * (threadIdx.x & 63), src, 16); }
#else
template <bool AGENT> __device__ __forceinline__ void lds_dma16(const uint8_t *src, uint8_t *lds_base)
{
    typedef __attribute__((address_space(1))) const void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)lds_base, 16, 0, AGENT ? 16 : 0);      /* aux 16 = sc1: served past the vector L1 */
}
#endif
/* what a wave puts between its lds_dma16 calls and its first LDS read of their data (memory loads in flight are waited for as well) */
#ifdef MI355_HIP_EMU_H
static inline void lds_dma_wait() {}
#else
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif

#endif
