// Developer microbenchmark: what rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts in the access shapes the
// H.264 kernels use (MI355X_MICROARCH.md: gfx950 halves wide streaming reads; other widths uncalibrated).
// Each kernel moves N bytes in and N bytes out exactly once.  Run under:
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- ./copy_calib ; rocprofv3 --pmc WRITE_SIZE ... -d out_w -- ./copy_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x2 __attribute__((vector_size(8)));
typedef uint32_t u32x3 __attribute__((vector_size(12)));
// (a) one dword per lane, coalesced
__global__ void k_copy_dword(const uint32_t *s, uint32_t *d, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }
// (b) 16 bytes per lane, coalesced
__global__ void k_copy_x4(const u32x4 *s, u32x4 *d, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) d[i] = s[i]; }
// (c) the deblocking chunk shape: 16-byte pieces, four lanes per 64-byte row piece, sixteen rows of a 1920-byte-pitch picture per wave
__global__ void k_copy_rows16(const uint8_t *s, uint8_t *d, int pitch, int rows_total)
{
    const int lane = threadIdx.x & 63, piece = lane & 3, r = lane >> 2;
    const size_t wave = blockIdx.x;                       // wave w: rows 16 * (w / 30) .., column chunk w % 30 of 64 bytes
    const size_t row = 16 * (wave / 30) + r, col = 64 * (wave % 30) + 16 * piece;
    if (row < (size_t)rows_total) *(u32x4 *)(d + row * pitch + col) = *(const u32x4 *)(s + row * pitch + col);
}
// (d) the reference-window shape: 21 rows x 3 pieces of 12 bytes at an arbitrary 4-byte aligned column (reads only; one dword written)
__global__ void k_read_windows(const uint8_t *s, uint32_t *d, int pitch, int rows_total, int cols)
{
    const int lane = threadIdx.x & 63, t = lane < 63 ? lane : 62, row = (t * 43) >> 7, piece = t - 3 * row;
    const size_t w = blockIdx.x;
    uint32_t h = (uint32_t)(w * 2654435761u);
    const size_t y0 = (h >> 8) % (size_t)(rows_total - 21), x0 = ((h >> 20) % (size_t)(cols - 32)) & ~(size_t)3;
    const u32x3 v = *(const u32x3 *)(s + (y0 + row) * pitch + x0 + 8 * piece);
    if ((v[0] ^ v[1] ^ v[2]) == 0x12345678u) d[w & 1023] = v[0];   // practically never: keeps the loads alive
}
int main()
{
    const size_t N = (size_t)1 << 30;                       // 1 GiB each way
    uint8_t *s, *d;
    hipMalloc(&s, N + 4096); hipMalloc(&d, N + 4096);
    hipMemset(s, 1, N); hipMemset(d, 0, N);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_copy_dword, dim3((unsigned)(N / 4 / 256)), dim3(256), 0, 0, (const uint32_t *)s, (uint32_t *)d, N / 4);
        hipLaunchKernelGGL(k_copy_x4, dim3((unsigned)(N / 16 / 256)), dim3(256), 0, 0, (const u32x4 *)s, (u32x4 *)d, N / 16);
        const int pitch = 1920, rows = (int)(N / pitch);
        hipLaunchKernelGGL(k_copy_rows16, dim3((unsigned)((rows / 16) * 30)), dim3(64), 0, 0, s, d, pitch, rows);
        hipLaunchKernelGGL(k_read_windows, dim3(4u << 20), dim3(64), 0, 0, s, (uint32_t *)d, pitch, rows, 1920);
    }
    hipDeviceSynchronize();
    printf("bytes moved each way: dword %zu, x4 %zu, rows16 %zu (read = written), windows: %zu useful bytes read (4 Mi windows x 21 rows x 28 bytes; sectors touched: see counters)\n",
           N, N, (size_t)(N / 1920 / 16) * 16 * 1920, (size_t)(4u << 20) * 21 * 28);
    return 0;
}
