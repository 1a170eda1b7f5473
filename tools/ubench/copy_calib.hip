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
// (e) round 6: the run kernel's LDS-DMA fetches (h264_recon_fast.h): sixteen bytes per lane straight into LDS (global_load_lds), (e1) a wave's 64 pieces contiguous
// (1 KB: a macroblock's coefficients), (e2) as window rows of a macroblock-tiled surface: 3 pieces (48 bytes) of each of 21 tile rows 16 bytes apart inside 256-byte tiles,
// i.e. 63 lanes over three neighbouring tiles' rows (the raw luma window of one macroblock: 1008 bytes requested, lines touched: see the counter)
__global__ void k_lds_dma_linear(const uint8_t *s, uint32_t *d, size_t n16)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    const size_t i = blockIdx.x * (size_t)64 + threadIdx.x;
    typedef __attribute__((address_space(1))) const void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    if (i < n16) __builtin_amdgcn_global_load_lds((gptr)(s + 16 * i), (lptr)buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (buf[threadIdx.x * 16] == 0x7B && buf[1] == 0x11) d[blockIdx.x & 1023] = 1;      // practically never: keeps the loads alive
}
__global__ void k_lds_dma_windows(const uint8_t *s, uint32_t *d, int tiles_x, int tiles_y)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    const int lane = threadIdx.x, t = lane < 63 ? lane : 62, row = t / 3, piece = t - 3 * row;
    // window w: macroblock (mx, my) of a surface of 256-byte tiles in raster order; rows -2 .. 18 of the macroblock, pieces at tile columns -1, 0, +1 (clamped)
    const size_t w = blockIdx.x;
    const int mx = (int)(w % (size_t)tiles_x), my = (int)((w / (size_t)tiles_x) % (size_t)tiles_y);
    int ty = my, r = row - 2;
    if (r < 0) { ty = my > 0 ? my - 1 : my; r = my > 0 ? r + 16 : 0; }
    if (r > 15) { ty = my + 1 < tiles_y ? my + 1 : my; r = my + 1 < tiles_y ? r - 16 : 15; }
    const int tx = mx + piece - 1 < 0 ? 0 : (mx + piece - 1 >= tiles_x ? tiles_x - 1 : mx + piece - 1);
    const uint8_t *p = s + ((size_t)ty * tiles_x + tx) * 256 + 16 * r;
    typedef __attribute__((address_space(1))) const void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
    __builtin_amdgcn_global_load_lds((gptr)p, (lptr)buf, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (buf[threadIdx.x * 16] == 0x7B && buf[1] == 0x11) d[blockIdx.x & 1023] = 1;
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
        hipLaunchKernelGGL(k_lds_dma_linear, dim3((unsigned)(N / 1024)), dim3(64), 0, 0, s, (uint32_t *)d, N / 16);
        hipLaunchKernelGGL(k_lds_dma_windows, dim3(120u * 68u * 512u), dim3(64), 0, 0, s, (uint32_t *)d, 120, 68 * 512);        // 512 pictures' worth of 1080p macroblock tiles (1 GiB)
    }
    hipDeviceSynchronize();
    printf("bytes moved each way: dword %zu, x4 %zu, rows16 %zu (read = written), windows: %zu useful bytes read (4 Mi windows x 21 rows x 28 bytes; sectors touched: see counters)\n",
           N, N, (size_t)(N / 1920 / 16) * 16 * 1920, (size_t)(4u << 20) * 21 * 28);
    printf("lds_dma_linear: %zu bytes read; lds_dma_windows: %zu windows x 63 pieces x 16 bytes = %zu bytes requested over 1 GiB of tiles (each tile row is asked for by three windows)\n", N, (size_t)120 * 68 * 512, (size_t)120 * 68 * 512 * 63 * 16);
    return 0;
}
