/*
 * hip_emu.h — a tiny SIMT emulator for DEBUGGING the kernels in libav_amd/csrc on a
 * machine with no GPU.  TEST TOOLING ONLY: it is never linked into the product
 * library (libav_amd/libmi355dsp.so is built by hipcc for gfx950 and fails loudly
 * without a GPU).  `make -C tools/simt_emu emu` compiles the *same* .hip sources
 * with g++ against this header into tests/_emu/libmi355dsp_emu.so so that the CPU
 * test-suite (-m "not gpu") can exercise kernel indexing / LDS / shuffle logic
 * against the oracle before a GPU-minute is spent.
 *
 * Model: one workgroup at a time; every work-item is a ucontext fiber; the 64-lane
 * wavefront is honoured for the shuffle and ballot builtins; __syncthreads() and the shuffles are
 * cooperative yield points.  A barrier that not every live lane reaches aborts
 * with a message (that would hang a real GPU).  `__shared__` is `static`
 * (valid because workgroups run one after another).
 *
 * Only the slice of the HIP API our sources use is provided.
 */
#ifndef MI355_HIP_EMU_H
#define MI355_HIP_EMU_H

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace simt_emu {
struct Lane {
    dim3 tid;
};
extern Lane *cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
void barrier_block();
int shfl_exchange(int v, int src_lane_rel, int width, int mode); /* mode 0 idx,1 xor,2 up,3 down */
unsigned long long ballot(int pred);
void launch(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace simt_emu

#define threadIdx (simt_emu::cur->tid)
#define blockIdx (simt_emu::g_blockIdx)
#define blockDim (simt_emu::g_blockDim)
#define gridDim (simt_emu::g_gridDim)
static const int warpSize = 64;

static inline void __syncthreads() { simt_emu::barrier_block(); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
static inline int __shfl(int v, int lane, int w = 64) { return simt_emu::shfl_exchange(v, lane, w, 0); }
static inline int __shfl_xor(int v, int m, int w = 64) { return simt_emu::shfl_exchange(v, m, w, 1); }
static inline int __shfl_up(int v, unsigned d, int w = 64) { return simt_emu::shfl_exchange(v, (int)d, w, 2); }
static inline int __shfl_down(int v, unsigned d, int w = 64) { return simt_emu::shfl_exchange(v, (int)d, w, 3); }
static inline unsigned long long __ballot(int p) { return simt_emu::ballot(p); }
static inline int __any(int p) { return simt_emu::ballot(p) != 0; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p += v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p |= v; return o; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  /* wave-uniform by contract */
static inline void __threadfence() {}
static inline void __threadfence_block() {}

/* ---- host API subset ------------------------------------------------------ */
typedef int hipError_t;
typedef struct emu_stream *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
};
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
/* MI355_EMU_DEVICES=N: N make-believe devices (one address space); per device the emulator counts launches and allocated bytes
 * (simt_emu_device_stats), so a test can see WHERE a context's work went */
namespace simt_emu { int device_count(); int &current_device(); void note_alloc(size_t n); }
static inline hipError_t hipGetDeviceCount(int *n) { *n = simt_emu::device_count(); return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= simt_emu::device_count()) return hipErrorUnknown; simt_emu::current_device() = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = simt_emu::current_device(); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "simt-emu");
    std::strcpy(p->gcnArchName, "gfx950:emu");
    p->multiProcessorCount = 4;
    return hipSuccess;
}
/* the emulator "holds" one workgroup per CU: persistent kernels (grid = resident workgroups) take many turns each, which is what the tests want to see */
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int, size_t) { *n = 1; return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); simt_emu::note_alloc(n); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0) {
    for (size_t y = 0; y < h; y++) std::memcpy((char *)d + y * dp, (const char *)s + y * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { static char tag; *s = reinterpret_cast<hipStream_t>(&tag); return hipSuccess; }   /* non-null: callers test the handle */
#define hipStreamNonBlocking 1u
#define hipEventDefault 0u
#define hipEventBlockingSync 1u
#define hipDeviceScheduleBlockingSync 4u
static inline hipError_t hipSetDeviceFlags(unsigned) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)1 << 32; return hipSuccess; }

#include <chrono>
struct emu_event { std::chrono::steady_clock::time_point t; };
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event; return hipSuccess; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emu_event; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }                /* the emulator's launches are synchronous: always finished */
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}

template <typename K, typename... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t /*stream*/, A... args)
{
    simt_emu::launch(grid, block, [=]() { kernel(args...); });
}

#endif /* MI355_HIP_EMU_H */
