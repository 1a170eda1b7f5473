// Developer microbenchmark: issue cost of the integer VALU instructions the H.264 kernels lean on, wave64 on gfx950.
// Each kernel runs ITER x 8 instructions of one kind (four independent chains) on 8 waves per SIMD of every CU.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
#define OPS(body) \
    for (int i = 0; i < ITER; i++) { asm volatile(body body : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e2) : "v"(x), "v"(y), "s"(sx)); }
#define KERNEL(name, body)                                                                          \
    __global__ void __launch_bounds__(64) name(unsigned *out, unsigned x, unsigned y, unsigned sx)   \
    {                                                                                                \
        unsigned a = threadIdx.x, b = a + x, c = b + y, d = c + 1;                                   \
        unsigned long long e2 = a;                                                                   \
        x += threadIdx.x; y ^= threadIdx.x;                                                          \
        OPS(body)                                                                                    \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (unsigned)e2;                           \
    }
KERNEL(k_add, "v_add_u32 %0, %0, %5\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %6\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %5\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %5\n v_mul_lo_u32 %3, %3, %6\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %5\n v_mul_u32_u24 %1, %1, %6\n v_mul_u32_u24 %2, %2, %5\n v_mul_u32_u24 %3, %3, %6\n")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %5, %6\n v_mad_u32_u24 %1, %1, %6, %5\n v_mad_u32_u24 %2, %2, %5, %6\n v_mad_u32_u24 %3, %3, %6, %5\n")
KERNEL(k_mad64, "v_mad_u64_u32 %4, vcc, %0, %5, %4\n v_mad_u64_u32 %4, vcc, %1, %6, %4\n v_mad_u64_u32 %4, vcc, %2, %5, %4\n v_mad_u64_u32 %4, vcc, %3, %6, %4\n")
KERNEL(k_lshladd64, "v_lshl_add_u64 %4, %4, 2, %4\n v_lshl_add_u64 %4, %4, 1, %4\n v_lshl_add_u64 %4, %4, 2, %4\n v_lshl_add_u64 %4, %4, 1, %4\n")
KERNEL(k_pkmad, "v_pk_mad_i16 %0, %0, %5, %6\n v_pk_mad_i16 %1, %1, %6, %5\n v_pk_mad_i16 %2, %2, %5, %6\n v_pk_mad_i16 %3, %3, %6, %5\n")
KERNEL(k_pkadd, "v_pk_add_i16 %0, %0, %5\n v_pk_add_i16 %1, %1, %6\n v_pk_add_i16 %2, %2, %5\n v_pk_add_i16 %3, %3, %6\n")
KERNEL(k_perm, "v_perm_b32 %0, %0, %5, %6\n v_perm_b32 %1, %1, %6, %5\n v_perm_b32 %2, %2, %5, %6\n v_perm_b32 %3, %3, %6, %5\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %5, %7\n v_alignbyte_b32 %1, %1, %6, %7\n v_alignbyte_b32 %2, %2, %5, %7\n v_alignbyte_b32 %3, %3, %6, %7\n")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_sad, "v_sad_u16 %0, %0, %5, %6\n v_sad_u16 %1, %1, %6, %5\n v_sad_u16 %2, %2, %5, %6\n v_sad_u16 %3, %3, %6, %5\n")
KERNEL(k_med3, "v_med3_i32 %0, %0, %5, %6\n v_med3_i32 %1, %1, %6, %5\n v_med3_i32 %2, %2, %5, %6\n v_med3_i32 %3, %3, %6, %5\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %5, 8, 8\n v_bfe_u32 %2, %6, 16, 8\n v_bfe_u32 %3, %3, 8, 8\n")
KERNEL(k_sdwa, "v_add_u32_sdwa %0, %0, %5 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_add_u32_sdwa %1, %1, %6 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_add_u32_sdwa %2, %2, %5 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2\n v_add_u32_sdwa %3, %3, %6 dst_sel:DWORD src0_sel:BYTE_1 src1_sel:BYTE_2\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %6, vcc\n")
KERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %5\n v_cmp_lt_u32 vcc, %1, %6\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %6\n")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n")
KERNEL(k_salu, "s_add_u32 s20, s20, %7\n s_add_u32 s21, s21, %7\n s_add_u32 s22, s22, %7\n s_add_u32 s23, s23, %7\n")
KERNEL(k_mixed, "v_add_u32 %0, %0, %5\n s_add_u32 s20, s20, %7\n v_add_u32 %2, %2, %5\n s_add_u32 s21, s21, %7\n")
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %5, %6\n v_dot4_u32_u8 %1, %1, %6, %5\n v_dot4_u32_u8 %2, %2, %5, %6\n v_dot4_u32_u8 %3, %3, %6, %5\n")
KERNEL(k_lerp, "v_lerp_u8 %0, %0, %5, %6\n v_lerp_u8 %1, %1, %6, %5\n v_lerp_u8 %2, %2, %5, %6\n v_lerp_u8 %3, %3, %6, %5\n")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cndmask_b32_e64 %1, %1, %6, s[20:21]\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cndmask_b32_e64 %3, %3, %6, s[20:21]\n")
KERNEL(k_cmpcnd, "v_cmp_lt_u32 vcc, %0, %5\n v_cndmask_b32 %1, %1, %6, vcc\n v_cmp_lt_u32 vcc, %2, %5\n v_cndmask_b32 %3, %3, %6, vcc\n")
KERNEL(k_cmpcnd64, "v_cmp_lt_u32_e64 s[20:21], %0, %5\n v_cndmask_b32_e64 %1, %1, %6, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %2, %5\n v_cndmask_b32_e64 %3, %3, %6, s[22:23]\n")
KERNEL(k_minmax, "v_min_i32 %0, %0, %5\n v_max_i32 %1, %1, %6\n v_min_u32 %2, %2, %5\n v_max_u32 %3, %3, %6\n")
KERNEL(k_logic, "v_and_b32 %0, %0, %5\n v_or_b32 %1, %1, %6\n v_xor_b32 %2, %2, %5\n v_lshlrev_b32 %3, 3, %3\n")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 8, %5\n v_lshl_add_u32 %1, %1, 2, %6\n v_and_or_b32 %2, %2, %5, %6\n v_bfi_b32 %3, %3, %6, %5\n")
KERNEL(k_sub, "v_sub_u32 %0, %0, %5\n v_ashrrev_i32 %1, 1, %1\n v_subrev_u32 %2, %2, %5\n v_lshrrev_b32 %3, 1, %3\n")
KERNEL(k_pkminmax, "v_pk_max_i16 %0, %0, %5\n v_pk_min_i16 %1, %1, %6\n v_pk_ashrrev_i16 %2, 2, %2\n v_pk_sub_i16 %3, %3, %6\n")
KERNEL(k_mov, "v_mov_b32 %0, %5\n v_mov_b32 %1, %6\n v_mov_b32 %2, %5\n v_mov_b32 %3, %6\n")
KERNEL(k_bperm, "ds_bpermute_b32 %0, %5, %0\n ds_bpermute_b32 %1, %6, %1\n ds_bpermute_b32 %2, %5, %2\n ds_bpermute_b32 %3, %6, %3\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_add3, "v_add3_u32 %0, %0, %5, %6\n v_add3_u32 %1, %1, %6, %5\n v_add3_u32 %2, %2, %5, %6\n v_add3_u32 %3, %3, %6, %5\n")

template <class K> void run(const char *name, K k, unsigned *d, int waves_per_simd)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 1u);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 4; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 3u, 5u, 1u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 4.0 * waves_per_simd * ITER * 8.0;
    printf("%-12s %d waves/SIMD: %7.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
#define R(k) run(#k, k, d, 8);
    R(k_add) R(k_add) R(k_mul_lo) R(k_mul24) R(k_mad24) R(k_mad64) R(k_lshladd64) R(k_pkmad) R(k_pkadd) R(k_perm) R(k_alignbyte) R(k_dpp) R(k_sad) R(k_med3)
    R(k_bfe) R(k_sdwa) R(k_cndmask) R(k_cmp) R(k_readlane) R(k_salu) R(k_mixed) R(k_dot4) R(k_lerp) R(k_add3) R(k_cnd_e64) R(k_cmpcnd) R(k_cmpcnd64) R(k_minmax) R(k_logic) R(k_lshlor) R(k_sub) R(k_pkminmax) R(k_mov) R(k_bperm)
    run("k_add", k_add, d, 1); run("k_add", k_add, d, 2); run("k_mixed", k_mixed, d, 2); run("k_salu", k_salu, d, 1);
    return 0;
}
