// Micro-benchmark: sustained issue cost (cycles per wave-instruction per SIMD) of the VALU instruction classes the
// reduction kernel uses, at the occupancy it runs at.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n)
{
    float r0 = threadIdx.x * 1e-3f + a, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    for (int i = 0; i < n; ++i) {
#define APPLY(F) r0 = F(r0); r1 = F(r1); r2 = F(r2); r3 = F(r3); r4 = F(r4); r5 = F(r5); r6 = F(r6); r7 = F(r7);
        if (OP == 0) { APPLY([&](float x) { return __builtin_fmaf(x, a, b); }) }
        if (OP == 1) { APPLY([&](float x) { return x * a; }) }
        if (OP == 2) { APPLY([&](float x) { return x + b; }) }
        if (OP == 3) { APPLY([&](float x) { return __builtin_amdgcn_rcpf(x); }) }
        if (OP == 4) { APPLY([&](float x) { return (float)(int)x; }) }                       // cvt_i32_f32 + cvt_f32_i32
        if (OP == 5) { APPLY([&](float x) { return a / x; }) }                               // IEEE division sequence
        if (OP == 6) { APPLY([&](float x) { return x > b ? a : x; }) }                       // cmp + cndmask
        if (OP == 7) { APPLY([&](float x) { return __builtin_floorf(x) + b; }) }             // floor + add
        if (OP == 8) { APPLY([&](float x) { return __int_as_float((__float_as_int(x) >> 3) & 0x3fffffff); }) }   // shift + and
        if (OP == 9) { APPLY([&](float x) { return (float)((double)x * 1.0000001); }) }      // cvt f64, mul f64, cvt f32
        if (OP == 10) { APPLY([&](float x) { return __int_as_float(__float_as_int(x) * 3 + 1); }) }   // v_mul_lo_u32 / mad
        if (OP == 11) { APPLY([&](float x) { return (float)__float_as_int(x); }) }                     // v_cvt_f32_i32
        if (OP == 12) { APPLY([&](float x) { return (float)(__float_as_uint(x) & 0xffu); }) }
        if (OP == 13) { APPLY([&](float x) { return (float)__builtin_bit_cast(_Float16, (unsigned short)__float_as_uint(x)); }) }          // v_cvt_f32_f16
        if (OP == 14) { APPLY([&](float x) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(__float_as_uint(x) >> 16)); }) }  // ... src0_sel:WORD_1
        if (OP == 15) { APPLY([&](float x) { return __int_as_float((__float_as_int(x) << 12) >> 20); }) }                                   // v_bfe_i32
        if (OP == 16) { APPLY([&](float x) { return (float)((__float_as_uint(x) >> 16) & 0xffu); }) }
        if (OP == 17) { APPLY([&](float x) { return __builtin_floorf(x); }) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

int main()
{
    const char* names[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_rcp_f32", "cvt f32->i32->f32 (2 instr)", "IEEE div f32 (~10 instr)",
                           "v_cmp + v_cndmask (2)", "v_floor + v_add (2)", "v_ashr + v_and (2)", "cvt_f64 + mul_f64 + cvt_f32 (3)", "int mul/mad 32-bit",
                           "v_cvt_f32_i32", "v_cvt_f32_ubyte0", "v_cvt_f32_f16", "v_cvt_f32_f16 sdwa WORD_1", "v_bfe_i32", "v_cvt_f32_ubyte2", "v_floor_f32"};
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 5;                        // 5 workgroups of 256 threads per CU = 5 waves per SIMD
    float* d;
    CHK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%d CUs, clock %.0f MHz (reported), %d blocks\n", cus, prop.clockRate / 1e3, blocks);
    for (int op = 0; op <= 17; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipEventRecord(e0));
            switch (op) {
#define L(N) case N: hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, ITERS); break;
                L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17)
            }
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 1) {
                // per SIMD: 5 waves x ITERS x 8 "ops"; cycles at the reported clock
                const double ops_per_simd = 5.0 * ITERS * 8;
                const double cyc = ms * 1e-3 * prop.clockRate * 1e3;
                printf("  %-36s %8.3f ms  -> %.2f cycles per wave-op per SIMD\n", names[op], ms, cyc / ops_per_simd);
            }
        }
    }
    return 0;
}
