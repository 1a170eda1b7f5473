/* mi355_rt.hip — runtime: device selection, thread-local staging arenas. */
#include <atomic>
#include "mi355_rt.h"
#include "../../include/mi355dsp.h"

namespace mi355 {

static std::atomic<int> g_device{-1};     /* the process default: mi355_init() */
static thread_local int t_device = -1;    /* this thread's own device: mi355_set_device(); -1 = the process default */
static std::atomic<unsigned> g_checked{0};   /* bit d: device d was found to be a gfx950 */

int current_device() { return t_device >= 0 ? t_device : g_device.load(std::memory_order_acquire); }
bool ready() { return current_device() >= 0; }

/* waits that sleep instead of spinning: MI355_BLOCKING_SYNC=1 / 0 decides; without it, what the host asked for (mi355_prefer_blocking_sync) */
static std::atomic<int> g_prefer_blocking{0};
bool blocking_sync()
{
    const char *e = std::getenv("MI355_BLOCKING_SYNC");
    if (e && *e) return *e != '0';
    return g_prefer_blocking.load(std::memory_order_acquire) != 0;
}

/* 0, or mi355_init()'s error codes */
static int check_device(int device)
{
    int n = 0;
    if (device < 0 || hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device >= n) return -1;
    if (device < 32 && (g_checked.load(std::memory_order_acquire) >> device) & 1u) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -2;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "mi355dsp: device %d is %s, this library is built for gfx950 only\n", device, prop.gcnArchName);
        return -3;
    }
    if (device < 32) g_checked.fetch_or(1u << device, std::memory_order_acq_rel);
    return 0;
}

/* one word per device in pinned, device-visible host memory: a kernel's atomicOr reaches it over the fabric (rare by construction), the host
 * reads it with a plain load after a wait */
static std::atomic<uint32_t *> g_error_words[32];
uint32_t *error_word()
{
    const int d = current_device();
    if (d < 0 || d >= 32 || !bind()) return nullptr;
    uint32_t *w = g_error_words[d].load(std::memory_order_acquire);
    if (w) return w;
    uint32_t *fresh = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&fresh), 64) != hipSuccess) return nullptr;
    *fresh = 0u;
    uint32_t *expected = nullptr;
    if (!g_error_words[d].compare_exchange_strong(expected, fresh, std::memory_order_acq_rel)) { (void)hipHostFree(fresh); return expected; }
    return fresh;
}
int fault_after_wait()
{
    const int d = current_device();
    if (d < 0 || d >= 32) return 0;
    const uint32_t *w = g_error_words[d].load(std::memory_order_acquire);
    return w && __atomic_load_n(w, __ATOMIC_ACQUIRE) ? MI355_E_DEVICE_FAULT : 0;
}

bool bind()
{
    static thread_local int bound = -1;
    const int d = current_device();
    if (d < 0) return false;
    if (bound != d) {
        if (hipSetDevice(d) != hipSuccess) return false;
        /* the blocking-wait schedule is per device and per thread: every device a thread switches to gets it (once: a second call on a device with a
         * live context is refused and changes nothing) */
        if (blocking_sync()) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }      /* a refusal must not pass for a later launch's error */
        bound = d;
    }
    return true;
}

DeviceScope::DeviceScope(int device) : prev(t_device)
{
    if (device >= 0) { t_device = device; (void)bind(); }
}
DeviceScope::~DeviceScope()
{
    if (t_device != prev) { t_device = prev; (void)bind(); }
}

void Arena::ensure()
{
    if (dev && device == current_device()) return;
    if (dev) {
        /* the thread moved to another device (mi355_set_device): its staging follows */
        { DeviceScope back(device); (void)hipStreamSynchronize(stream); (void)hipHostFree(host); (void)hipFree(dev); (void)hipStreamDestroy(stream); }
        host = dev = nullptr; stream = nullptr; cap = used = 0;
    }
    if (!bind()) {
        std::fprintf(stderr, "mi355dsp: mi355_init() was not called or found no GPU; "
                             "there is no CPU fallback in this library\n");
        std::abort();
    }
    cap = 4 << 20;
    device = current_device();
    MI355_CHECK(hipStreamCreate(&stream));
    MI355_CHECK(hipHostMalloc(reinterpret_cast<void **>(&host), cap));
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), cap));
}

void Arena::grow(size_t need)
{
    size_t ncap = cap;
    while (ncap < need) ncap *= 2;
    uint8_t *nh = nullptr, *nd = nullptr;
    MI355_CHECK(hipHostMalloc(reinterpret_cast<void **>(&nh), ncap));
    MI355_CHECK(hipMalloc(reinterpret_cast<void **>(&nd), ncap));
    MI355_CHECK(hipStreamSynchronize(stream));
    MI355_CHECK(hipHostFree(host));
    MI355_CHECK(hipFree(dev));
    host = nh; dev = nd; cap = ncap;
}

Arena &arena()
{
    static thread_local Arena a;
    a.ensure();
    a.reset();
    return a;
}

Win win_pack(Arena &a, const uint8_t *src, ptrdiff_t stride, int wbytes, int rows, int valid_w, int valid_rows)
{
    Win w;
    w.wbytes = wbytes;
    w.rows = rows;
    w.pitch = (wbytes + 15) & ~15;
    w.off = a.take((size_t)w.pitch * rows);
    uint8_t *p = a.h<uint8_t>(w.off);
    if (valid_w < 0) valid_w = wbytes;
    if (valid_rows < 0) valid_rows = rows;
    for (int y = 0; y < rows; y++) {
        uint8_t *row = p + (size_t)y * w.pitch;
        if (y < valid_rows) {
            std::memcpy(row, src + (ptrdiff_t)y * stride, (size_t)valid_w);
            std::memset(row + valid_w, 0, (size_t)(w.pitch - valid_w));
        } else {
            std::memset(row, 0, (size_t)w.pitch);
        }
    }
    return w;
}

void win_unpack(Arena &a, const Win &w, uint8_t *dst, ptrdiff_t stride, int x0, int y0, int wbytes, int rows)
{
    const uint8_t *p = a.h<uint8_t>(w.off);
    for (int y = 0; y < rows; y++)
        std::memcpy(dst + (ptrdiff_t)y * stride, p + (size_t)(y0 + y) * w.pitch + x0, (size_t)wbytes);
}

}  // namespace mi355

/* A host with many threads that wait for the device (a decoder per thread) calls this with 1 BEFORE its first mi355_init(): waits then sleep
 * (hipDeviceScheduleBlockingSync, hipEventBlockingSync) instead of spinning on a core the parsing threads need.  The environment variable
 * MI355_BLOCKING_SYNC overrides either way. */
extern "C" void mi355_prefer_blocking_sync(int on) { mi355::g_prefer_blocking.store(on ? 1 : 0, std::memory_order_release); }

extern "C" int mi355_init(int device)
{
    const int rc = mi355::check_device(device);
    if (rc) return rc;
    /* MI355_BLOCKING_SYNC=1: a host thread that waits for the device sleeps instead of spinning (hipDeviceScheduleBlockingSync) — a host with
     * more waiting decoder threads than it may use cores gives the cores back to the threads that parse.  The flags belong to the calling thread's
     * CURRENT device: the device is chosen first (set before, they went to device 0 whatever `device` was). */
    if (hipSetDevice(device) != hipSuccess) return -4;
    if (mi355::blocking_sync()) { (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync); (void)hipGetLastError(); }
    mi355::g_device.store(device, std::memory_order_release);
    return 0;
}

/* One host process, several GPUs (the reference's model is one process with a thread per stream, pthread_frame.c:502-541): a
 * thread chooses ITS device; everything it allocates, copies and launches through this library from then on lives there.
 * -1 goes back to the process default.  Contexts (sessions, groups, swscale contexts) remember the device they were made on
 * and switch to it inside their entry points, whichever thread calls. */
extern "C" int mi355_set_device(int device)
{
    if (device < 0) { mi355::t_device = -1; return mi355::bind() ? 0 : -1; }
    const int rc = mi355::check_device(device);
    if (rc) return rc;
    mi355::t_device = device;
    return mi355::bind() ? 0 : -4;
}
extern "C" int mi355_get_device(void) { return mi355::current_device(); }
extern "C" int mi355_get_thread_device(void) { return mi355::t_device; }
extern "C" int mi355_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" unsigned mi355_error_word_peek(void)
{
    const int d = mi355::current_device();
    if (d < 0 || d >= 32) return 0u;
    const uint32_t *w = mi355::g_error_words[d].load(std::memory_order_acquire);
    return w ? __atomic_load_n(w, __ATOMIC_ACQUIRE) : 0u;
}
extern "C" unsigned mi355_error_word_take(void)
{
    const int d = mi355::current_device();
    if (d < 0 || d >= 32) return 0u;
    uint32_t *w = mi355::g_error_words[d].load(std::memory_order_acquire);
    return w ? __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL) : 0u;
}
namespace {
__global__ void k_error_word_inject(uint32_t *word, uint32_t bits) { atomicOr(word, bits); }
}
extern "C" int mi355_error_word_inject(unsigned bits, void *stream)
{
    uint32_t *w = mi355::error_word();
    if (!w) return -1;
    hipLaunchKernelGGL(k_error_word_inject, dim3(1), dim3(1), 0, (hipStream_t)stream, w, (uint32_t)bits);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int mi355_device_cus(void)
{
    hipDeviceProp_t prop;
    if (!mi355::ready() || hipGetDeviceProperties(&prop, mi355::current_device()) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}
