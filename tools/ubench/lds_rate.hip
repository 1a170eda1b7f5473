// Developer microbenchmark: LDS cycles per wave-instruction (per CU) of the access shapes the inter kernel uses, wave64 on gfx950:
// aligned / byte-unaligned ds_read_b64 and ds_read_b32 at the window pitch (48 bytes), the transposed planes (24 bytes), b128 rows, the
// coefficient pattern.  8 waves per SIMD on every CU; each wave issues ITER x 8 reads.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip ; run: ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 1024
enum { B64, B32, B128, W32, R2B32 };
template <int OP> __device__ __forceinline__ void op8(unsigned a, unsigned long long &acc)
{
    unsigned long long r0, r1, r2, r3;
    if (OP == B64) {
        asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:768\n ds_read_b64 %2, %4 offset:48\n ds_read_b64 %3, %4 offset:816\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a));
        acc += r0 ^ r1 ^ r2 ^ r3;
        asm volatile("ds_read_b64 %0, %4 offset:96\n ds_read_b64 %1, %4 offset:864\n ds_read_b64 %2, %4 offset:144\n ds_read_b64 %3, %4 offset:912\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a));
        acc += r0 ^ r1 ^ r2 ^ r3;
    } else if (OP == B32 || OP == W32) {
        unsigned s0, s1, s2, s3;
        for (int h = 0; h < 2; h++) {
            if (OP == B32)
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:768\n ds_read_b32 %2, %4 offset:48\n ds_read_b32 %3, %4 offset:816\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(a));
            else {
                s0 = s1 = s2 = s3 = (unsigned)acc;
                asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:768\n ds_write_b32 %0, %3 offset:48\n ds_write_b32 %0, %1 offset:816\n s_waitcnt lgkmcnt(0)"
                             : : "v"(a), "v"(s0), "v"(s1), "v"(s2) : "memory");
            }
            acc += s0 ^ s1 ^ s2 ^ s3;
        }
    } else if (OP == R2B32) {
        for (int h = 0; h < 2; h++) {
            asm volatile("ds_read2_b32 %0, %4 offset1:2\n ds_read2_b32 %1, %4 offset0:4 offset1:6\n ds_read2_b32 %2, %4 offset0:8 offset1:10\n ds_read2_b32 %3, %4 offset0:12 offset1:14\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a));
            acc += r0 ^ r1 ^ r2 ^ r3;
        }
    } else {
        typedef unsigned v4 __attribute__((ext_vector_type(4)));
        v4 q0, q1, q2, q3;
        for (int h = 0; h < 2; h++) {
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a));
            acc += q0[0] ^ q1[1] ^ q2[2] ^ q3[3];
        }
    }
}
// pattern: 0 window operand 48 (l & 15) + 8 (l >> 4) + sh; 1 window samples 48 (l & 15) + 4 (l >> 4) + sh; 2 plane operand 24 (l & 15) + 8 (l >> 4) + sh;
// 3 plane rows 24 (l & 15) + 4 (l >> 4); 4 rows 16 l (first 24 lanes: l < 24 ? 16 l : 0); 5 coefficient pairs (l >> 1) * 32 + (l & 1) * 4; 6 linear 8 l + sh; 7 linear 4 l + sh
template <int OP> __global__ void __launch_bounds__(64) k(unsigned *out, int pattern, int sh)
{
    __shared__ unsigned char lds[5120];
    const unsigned l = threadIdx.x;
    for (int i = l; i < 1280; i += 64) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned a;
    switch (pattern) {
    case 0: a = 48 * (l & 15) + 8 * (l >> 4) + sh; break;
    case 1: a = 48 * (l & 15) + 4 * (l >> 4) + sh; break;
    case 2: a = 24 * (l & 15) + 8 * (l >> 4) + sh; break;
    case 3: a = 24 * (l & 15) + 4 * (l >> 4) + sh; break;
    case 4: a = l < 24 ? 16 * l : 0; break;
    case 5: a = (l >> 1) * 32 + (l & 1) * 4; break;
    case 6: a = 8 * l + sh; break;
    default: a = 4 * l + sh; break;
    }
    a += (unsigned)(size_t)lds & 0xFFFF;
    unsigned long long acc = 0;
    for (int i = 0; i < ITER; i++) op8<OP>(a, acc);
    out[blockIdx.x * 64 + l] = (unsigned)acc ^ (unsigned)(acc >> 32);
}
template <class K> void run(const char *name, K kern, unsigned *d, int pattern, int sh)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 32;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, pattern, sh);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, pattern, sh);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = 2.0 * 32 * ITER * 8.0;
    printf("%-22s pattern %d shift %d: %8.3f ms -> %.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, pattern, sh, ms, ms * 1e-3 * 2.4e9 / per_cu);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 32 * 64 * 4);
    for (int sh = 0; sh < 9; sh++) run("ds_read_b64 window", k<B64>, d, 0, sh);
    for (int sh = 0; sh < 5; sh++) run("ds_read_b32 window", k<B32>, d, 1, sh);
    for (int sh = 0; sh < 2; sh++) run("ds_read_b64 plane", k<B64>, d, 2, sh * 4);
    run("ds_write_b32 plane", k<W32>, d, 3, 0);
    run("ds_read_b32 plane", k<B32>, d, 3, 0);
    run("ds_read_b128 rows24", k<B128>, d, 4, 0);
    run("ds_read2_b32 coefs", k<R2B32>, d, 5, 0);
    for (int sh = 0; sh < 3; sh++) run("ds_read_b64 linear", k<B64>, d, 6, sh);
    for (int sh = 0; sh < 3; sh++) run("ds_read_b32 linear", k<B32>, d, 7, sh);
    return 0;
}
