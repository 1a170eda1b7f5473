// Developer microbenchmark (round 6): what the memory system gives a copy of 64 linear 3840x2160 16-bit luma planes (7680-byte rows) depending on HOW the
// planes are walked — the question behind k_hevc_sao_ctbs: is a CTB-shaped walk (128-byte pieces of 64 rows) itself slower than a linear one?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ctb_copy tools/ubench/ctb_copy.hip && /tmp/ctb_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((vector_size(16)));
constexpr int W = 3840, H = 2160, PITCH = W * 2, PICS = 64, CX = W / 64, CY = (H + 63) / 64;
// (a) linear: 16 bytes per thread
__global__ void k_linear(const u32x4 *s, u32x4 *d, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }
// (b) a wave per CTB, STEPS rows-of-8 at a time (U loads in flight, then U stores)
template <int U>
__global__ void __launch_bounds__(64) k_ctb_wave(const uint8_t *s, uint8_t *d)
{
    const int b = blockIdx.x, pic = b / (CX * CY), c = b % (CX * CY), cy = c / CX, cx = c % CX, lane = threadIdx.x;
    const int rows = H - 64 * cy < 64 ? H - 64 * cy : 64;
    const size_t base = (size_t)pic * PITCH * H + (size_t)(64 * cy) * PITCH + 128 * cx;
    for (int r0 = 0; r0 < rows; r0 += 8 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int r = r0 + 8 * u + (lane >> 3); v[u] = *(const u32x4 *)(s + base + (size_t)(r < rows ? r : rows - 1) * PITCH + 16 * (lane & 7)); }
#pragma unroll
        for (int u = 0; u < U; u++) { const int r = r0 + 8 * u + (lane >> 3); if (r < rows) *(u32x4 *)(d + base + (size_t)r * PITCH + 16 * (lane & 7)) = v[u]; }
    }
}
// (c) a 256-thread workgroup per CTB: everything in flight at once
__global__ void __launch_bounds__(256) k_ctb_wg(const uint8_t *s, uint8_t *d)
{
    const int b = blockIdx.x, pic = b / (CX * CY), c = b % (CX * CY), cy = c / CX, cx = c % CX, t = threadIdx.x;
    const int rows = H - 64 * cy < 64 ? H - 64 * cy : 64;
    const size_t base = (size_t)pic * PITCH * H + (size_t)(64 * cy) * PITCH + 128 * cx;
    u32x4 v[2];
    for (int u = 0; u < 2; u++) { const int r = 32 * u + (t >> 3); v[u] = *(const u32x4 *)(s + base + (size_t)(r < rows ? r : rows - 1) * PITCH + 16 * (t & 7)); }
    for (int u = 0; u < 2; u++) { const int r = 32 * u + (t >> 3); if (r < rows) *(u32x4 *)(d + base + (size_t)r * PITCH + 16 * (t & 7)) = v[u]; }
}
// (d) a wave per strip of 8 rows x 1024 bytes (eight CTBs wide): whole kilobytes of a row per request
__global__ void __launch_bounds__(64) k_strip(const uint8_t *s, uint8_t *d)
{
    // strips per picture row-of-8: 7680 / 1024 = 7.5 -> 8 (the last one half)
    const int b = blockIdx.x, per_pic = 8 * (H / 8), pic = b / per_pic, q = b % per_pic, ry = q / 8, sx = q % 8, lane = threadIdx.x;
    const size_t base = (size_t)pic * PITCH * H + (size_t)(8 * ry) * PITCH + 1024 * sx;
    u32x4 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) if (1024 * sx + 16 * lane < PITCH) v[r] = *(const u32x4 *)(s + base + (size_t)r * PITCH + 16 * lane);
#pragma unroll
    for (int r = 0; r < 8; r++) if (1024 * sx + 16 * lane < PITCH) *(u32x4 *)(d + base + (size_t)r * PITCH + 16 * lane) = v[r];
}
// (e) as (b) with U = 2, but the CTBs a wave index maps to chosen so that the 8 waves the dispatcher hands to the 8 XCDs in turn are 8 DIFFERENT CTB rows and
// consecutive CTBs of a row go to ONE XCD (b' = 8 * (b / 8 ...)): the lines of a DRAM page are asked for by one L2
__global__ void __launch_bounds__(64) k_ctb_wave_xcd(const uint8_t *s, uint8_t *d)
{
    const int total = PICS * CX * CY, per = total / 8;
    const int b0 = blockIdx.x, b = (b0 & 7) * per + (b0 >> 3);       // XCD x takes the contiguous range [x * per, (x + 1) * per)
    const int pic = b / (CX * CY), c = b % (CX * CY), cy = c / CX, cx = c % CX, lane = threadIdx.x;
    const int rows = H - 64 * cy < 64 ? H - 64 * cy : 64;
    const size_t base = (size_t)pic * PITCH * H + (size_t)(64 * cy) * PITCH + 128 * cx;
    for (int r0 = 0; r0 < rows; r0 += 16) {
        u32x4 v[2];
        for (int u = 0; u < 2; u++) { const int r = r0 + 8 * u + (lane >> 3); v[u] = *(const u32x4 *)(s + base + (size_t)(r < rows ? r : rows - 1) * PITCH + 16 * (lane & 7)); }
        for (int u = 0; u < 2; u++) { const int r = r0 + 8 * u + (lane >> 3); if (r < rows) *(u32x4 *)(d + base + (size_t)r * PITCH + 16 * (lane & 7)) = v[u]; }
    }
}
int main()
{
    const size_t N = (size_t)PITCH * H * PICS;
    uint8_t *s, *d;
    hipMalloc(&s, N + 4096); hipMalloc(&d, N + 4096);
    hipMemset(s, 1, N); hipMemset(d, 0, N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-28s %.3f ms  %.2f TB/s (read + written)\n", name, best, 2.0 * N / best / 1e9);
    };
    const unsigned nctb = PICS * CX * CY;
    run("linear x4", [&] { hipLaunchKernelGGL(k_linear, dim3((unsigned)(N / 16 / 256)), dim3(256), 0, 0, (const u32x4 *)s, (u32x4 *)d, N / 16); });
    run("wave per CTB, U=1", [&] { hipLaunchKernelGGL(k_ctb_wave<1>, dim3(nctb), dim3(64), 0, 0, s, d); });
    run("wave per CTB, U=2", [&] { hipLaunchKernelGGL(k_ctb_wave<2>, dim3(nctb), dim3(64), 0, 0, s, d); });
    run("wave per CTB, U=4", [&] { hipLaunchKernelGGL(k_ctb_wave<4>, dim3(nctb), dim3(64), 0, 0, s, d); });
    run("wave per CTB, U=8", [&] { hipLaunchKernelGGL(k_ctb_wave<8>, dim3(nctb), dim3(64), 0, 0, s, d); });
    run("256 threads per CTB", [&] { hipLaunchKernelGGL(k_ctb_wg, dim3(nctb), dim3(256), 0, 0, s, d); });
    run("wave per 8 x 1024 B strip", [&] { hipLaunchKernelGGL(k_strip, dim3(PICS * 8 * (H / 8)), dim3(64), 0, 0, s, d); });
    run("wave per CTB, U=2, XCD rows", [&] { hipLaunchKernelGGL(k_ctb_wave_xcd, dim3(nctb), dim3(64), 0, 0, s, d); });
    return 0;
}
