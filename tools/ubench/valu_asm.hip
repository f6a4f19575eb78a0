// Micro-benchmark with inline assembly: sustained issue cost of single VALU instructions on gfx950 (cycles per
// wave64 instruction per SIMD at the reported clock), 8 independent chains per lane, 5 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_asm valu_asm.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n)
{
    float r[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { r[i] = threadIdx.x * 1e-3f + a + i; d[i] = r[i]; }
    asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(r[0]), "v"(a) : "vcc");
    for (int i = 0; i < n; ++i) {
#define S0(K) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[K]) : "v"(a), "v"(b));
#define S1(K) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[K]) : "v"(a));
#define S2(K) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[K]) : "v"(a), "v"(b));
#define S3(K) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[K]));
#define S4(K) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[K]));
#define S5(K) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(r[K]));
#define S6(K) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(r[K]), "v"(a) : "vcc");
#define S7(K) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(r[K]) : "v"(a), "v"(b));
#define S8(K) asm volatile("v_bfe_i32 %0, %0, 8, 12" : "+v"(r[K]));
#define S9(K) asm volatile("v_floor_f32 %0, %0" : "+v"(r[K]));
#define S10(K) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[K]));
#define S11(K) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[K]) : "v"(a), "v"(b));
#define S12(K) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[K]) : "v"(a), "v"(b));
#define S13(K) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[K]) : "v"(a), "v"(b));
#define S14(K) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[K]));
#define S15(K) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(d[K]));
#define S16(K) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r[K]));
#define S17(K) asm volatile("v_mov_b32 %0, %1" : "=v"(r[K]) : "v"(a));
#define S18(K) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[K]));
#define S19(K) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[K]));
#define S20(K) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[K]) : "v"(a));
#define S21(K) asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(d[K]));
#define S22(K) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(r[K]), "v"(a) : "vcc");
#define S23(K) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[K]) : "v"(a));
#define S24(K) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r[K]));
#define S25(K) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[K]));
#define S26(K) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" :: "v"(r[K]), "v"(a) : "s20", "s21");
#define S27(K) asm volatile("v_sub_f32 %0, 1.0, %0" : "+v"(r[K]));
#define S28(K) asm volatile("v_ashrrev_i32 %0, 20, %0" : "+v"(r[K]));
#define S29(K) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(r[K]) : "v"(a), "v"(b));
#define S30(K) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[K]));
#define S31(K) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(r[K]) :: "s20");
#define S32(K) asm volatile("v_fma_f32 %0, s20, %1, %0" : "+v"(r[K]) : "v"(b));
#define S33(K) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[K]) : "v"(b));
#define S34(K) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[K]) : "v"(b));
#define S35(K) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[K]) : "v"(b));
#define CASE(N) if (OP == N) { R8(S##N) R8(S##N) }
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
        CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24)
        CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31) CASE(32) CASE(33) CASE(34) CASE(35)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const char* names[] = {"v_fma_f32", "v_mul_f32", "v_fmac_f32", "v_cvt_f32_i32", "v_cvt_f32_f16", "v_cvt_f32_ubyte0", "v_cmp_le_f32 vcc",
                           "v_cndmask_b32 (vcc)", "v_bfe_i32", "v_floor_f32", "v_rcp_f32", "v_med3_f32", "v_perm_b32", "v_and_or_b32",
                           "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_u32_f32", "v_mov_b32", "v_add_f64", "v_lshlrev_b32", "v_and_b32",
                           "v_lshl_add_u64", "v_cmp_le_u32 vcc", "v_add_f32", "v_cvt_f32_f16 sdwa WORD_1", "v_fma_f64",
                           "v_cmp_le_f32 sgpr pair", "v_sub_f32 const", "v_ashrrev_i32", "v_fma_f32 neg mod (VOP3)", "v_cvt_f32_u32",
                           "v_mul_f32 sgpr operand", "v_fma_f32 sgpr operand", "v_cvt_pkrtz_f16_f32", "v_mul_u32_u24", "v_mad_u32_u24"};
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 5;                        // 5 workgroups of 256 threads per CU = 5 waves per SIMD
    float* d;
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%d CUs, clock %.0f MHz (reported), %d blocks, 16 instructions x %d iterations per wave\n", cus, prop.clockRate / 1e3, blocks, ITERS);
    for (int op = 0; op <= 35; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            switch (op) {
#define L(N) case N: hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, ITERS); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20)
                L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) L(32) L(33) L(34) L(35)
            }
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                const double ops_per_simd = 5.0 * ITERS * 16;
                const double cyc = ms * 1e-3 * prop.clockRate * 1e3;
                printf("  %-28s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD\n", names[op], ms, cyc / ops_per_simd);
            }
        }
    }
    return 0;
}
