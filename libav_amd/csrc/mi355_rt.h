/*
 * mi355_rt.h — host-side runtime shared by the C-ABI entry points.
 *
 * Tier-1 (per-call, synchronous, host pointers — the reference's DSP pointer
 * tables as they are called today): every call packs the sample
 * window it touches into a pinned staging buffer, does ONE H2D copy, ONE kernel
 * launch, ONE D2H copy and unpacks the written extent.  Slow by construction
 * (tens of µs per call); it exists so the reference's own decoder and the
 * checkasm-style parity tests can run through the HIP kernels unchanged.
 * Tier-2 (batched, device pointers) lives in h264_frame.hip.
 */
#ifndef MI355_RT_H
#define MI355_RT_H

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

/* A pointer read from a job or picture record has no known address space, and accesses through it become
 * flat_* instructions, which count against the LDS counter as well as the memory one (every wait for LDS
 * then also waits for the loads in flight).  A round trip through the global address space, hidden from
 * the optimiser by an empty asm, turns them into global_* ones.  mi355_global: wave-uniform pointers
 * (kept in scalar registers), mi355_global_v: per-lane pointers. */
template <class T> __device__ __forceinline__ T *mi355_global(T *p)
{
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
    return p;
#else
    /* readfirstlane: a uniform value the compiler could not prove uniform still lands in scalar registers */
    const unsigned long long v = (unsigned long long)p;
    const unsigned long long u = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
    auto q = (__attribute__((address_space(1))) T *)u;
    asm("" : "+s"(q));
    return (T *)q;
#endif
}
template <class T> __device__ __forceinline__ T *mi355_global_v(T *p)
{
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
    return p;
#else
    auto q = (__attribute__((address_space(1))) T *)p;
    asm("" : "+v"(q));
    return (T *)q;
#endif
}

/* floor(i / n) for a wave-uniform divisor as a 24-bit multiply and a shift: mi355_div20(i, mi355_inv20(n)).  One
 * reciprocal instead of an integer division (some 35 instructions).  Exact for i * n < 2^19 — every use below divides a
 * lane index of at most a few thousand by a row length of at most a hundred — with a reciprocal that is off by an ulp
 * either way (checked exhaustively for n < 2000). */
__device__ __forceinline__ int mi355_inv20(int n)
{
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
    return (int)(1048576.0f * (1.0f / (float)n)) + 1;
#else
    return (int)(1048576.0f * __builtin_amdgcn_rcpf((float)n)) + 1;
#endif
}
__device__ __forceinline__ int mi355_div20(int i, int inv) { return (int)(__umul24((unsigned)i, (unsigned)inv) >> 20); }

/* compiler-only fence: memory operations are not moved across it (no instruction is emitted, nothing is waited for) */
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
#define MI355_ISSUE_FENCE() ((void)0)
#else
#define MI355_ISSUE_FENCE() asm volatile("" ::: "memory")
#endif
/* the instruction scheduler does not move anything across this point (a straight-line stretch in groups: registers of one group are free before the next one's loads go out) */
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
#define MI355_SCHED_BARRIER() ((void)0)
#else
#define MI355_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
/* make a just-loaded value "used" at this point, so that the wait for it is placed here and not at a later join */
#if defined(MI355_HIP_EMU_H) || !defined(__HIP_DEVICE_COMPILE__)
#define MI355_PIN(v) ((void)(v))
#else
#define MI355_PIN(v) asm volatile("" : "+v"(v))
#endif

#define MI355_CHECK(expr)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "mi355dsp: %s failed: %s (%s:%d)\n", #expr,                 \
                         hipGetErrorString(e_), __FILE__, __LINE__);                         \
            std::abort(); /* Tier 1 only: the pointer tables are void, there is no error channel */ \
        }                                                                                    \
    } while (0)
/* Tier 2 (batched entry points: they have return codes): report and hand the failure to the caller */
#define MI355_TRY(expr, rc)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "mi355dsp: %s failed: %s (%s:%d)\n", #expr,                 \
                         hipGetErrorString(e_), __FILE__, __LINE__);                         \
            return rc;                                                                       \
        }                                                                                    \
    } while (0)

namespace mi355 {

/* Thread-local staging arena: the reference calls the tables from frame/slice threads. */
struct Arena {
    hipStream_t stream = nullptr;
    uint8_t *host = nullptr;   /* pinned */
    uint8_t *dev = nullptr;
    size_t cap = 0, used = 0;
    int device = -1;           /* the device `dev` and `stream` live on */

    void ensure();
    void reset() { used = 0; }
    /* reserve n bytes (16-byte aligned) in both images; returns the offset */
    size_t take(size_t n)
    {
        size_t off = (used + 15) & ~(size_t)15;
        if (off + n > cap) { std::fprintf(stderr, "mi355dsp: staging arena overflow (an entry point did not reserve() its operands)\n"); std::abort(); }
        used = off + n;
        return off;
    }
    /* make room for `need` bytes of operands BEFORE the first take() of a call (entry points whose operand sizes are not
     * bounded by the codec: filter banks, picture lines); pointers handed out earlier do not survive a growth */
    void reserve(size_t need) { if (need + 4096 > cap) grow(need + 4096); }
    void grow(size_t need);
    template <typename T> T *h(size_t off) { return reinterpret_cast<T *>(host + off); }
    template <typename T> T *d(size_t off) { return reinterpret_cast<T *>(dev + off); }
    void upload() { MI355_CHECK(hipMemcpyAsync(dev, host, used, hipMemcpyHostToDevice, stream)); }
    void download()
    {
        MI355_CHECK(hipMemcpyAsync(host, dev, used, hipMemcpyDeviceToHost, stream));
        MI355_CHECK(hipStreamSynchronize(stream));
    }
};

Arena &arena();

/* A rectangular window of host samples mirrored in the arena at a fixed pitch. */
struct Win {
    size_t off;
    int pitch, wbytes, rows;
};
/* copy rows x wbytes from (src, stride) into the arena; rows outside [0,valid_rows) x
 * [0,valid_w) are zero-filled (used when the reference contract forbids reading them) */
Win win_pack(Arena &a, const uint8_t *src, ptrdiff_t stride, int wbytes, int rows,
             int valid_w = -1, int valid_rows = -1);
/* copy the sub-rectangle (x0,y0,w,h) of the window back to (dst, stride) where dst is the
 * host address of window sample (x0,y0) */
void win_unpack(Arena &a, const Win &w, uint8_t *dst, ptrdiff_t stride, int x0, int y0, int wbytes, int rows);

bool ready();
/* the calling thread's device's error word (include/mi355dsp.h): a device-visible pointer into pinned host memory, nullptr without a device.
 * fault_after_wait(): what a sync entry point returns once its wait is over: 0, or MI355_E_DEVICE_FAULT while bits are set (they stay set until
 * mi355_error_word_take()) */
/* the calling thread's counter buffer of `stream` (h264_deblock.hip: sync_words) is freed: a stream about to be destroyed */
void sync_words_release(hipStream_t stream);
/* h264_deblock.hip: the scratch words (ticket + progress counters) of a single-launch form, one buffer per (thread, device, stream); the caller zeroes what it uses on `stream` */
uint32_t *sync_words(hipStream_t stream, size_t words);
uint32_t *error_word();
int fault_after_wait();
bool blocking_sync();      /* waits sleep instead of spinning (MI355_BLOCKING_SYNC / mi355_prefer_blocking_sync) */
int current_device();
/* bind the calling thread to its device — mi355_set_device() of this thread, else the one chosen in mi355_init() (the reference calls the tables and the batch entry points
 * from frame / slice threads; a thread that never set a device would use device 0); false without a successful init */
bool bind();
/* inside a context's entry point: the calling thread works on the context's device until the scope ends */
struct DeviceScope {
    int prev;
    explicit DeviceScope(int device);
    ~DeviceScope();
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

}  // namespace mi355
#endif
