// VALU issue rate against occupancy on gfx950: cycles per wave64 instruction per SIMD, in REAL shader cycles (s_memtime against
// s_memrealtime gives the clock the chip holds under this load), for a fast-class (v_fma_f32) and a slow-class (v_cvt_f32_i32)
// instruction, 16 independent instructions per loop iteration, 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_occupancy valu_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, float a, float b, int n)
{
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 1e-3f + a + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#define S0(K) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[K]) : "v"(a), "v"(b));
#define S1(K) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[K]));
        if (OP == 0) { R8(S0) R8(S0) } else { R8(S1) R8(S1) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 1 << 15;
    float* d;
    unsigned long long* st;
    CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
    CHK(hipMalloc(&st, (size_t)cus * 8 * 16));
    std::vector<unsigned long long> h((size_t)cus * 8 * 2);
    printf("%d CUs, nominal clock %.0f MHz; %d iterations x 16 instructions per wave; waves/SIMD = workgroups of 256 threads per CU\n", cus, prop.clockRate / 1e3, iters);
    for (int op = 0; op < 2; ++op)
        for (int w = 1; w <= 8; ++w) {
            const int blocks = cus * w;
            hipEvent_t e0, e1;
            CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0));
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, st, 1.0001f, 0.5f, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, st, 1.0001f, 0.5f, iters);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                CHK(hipEventElapsedTime(&ms, e0, e1));
            }
            CHK(hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost));
            std::vector<double> clk, cyc;
            for (int b = 0; b < blocks; ++b) {
                clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
                cyc.push_back((double)h[2 * b] * 1.0 / ((double)iters * 16.0 * w));      // the w waves of a SIMD share it for the whole run
            }
            std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            printf("  %-14s %d waves/SIMD: %8.3f ms, clock %4.0f MHz (median workgroup), %.2f real cycles per wave-instruction per SIMD\n",
                   op == 0 ? "v_fma_f32" : "v_cvt_f32_i32", w, ms, clk[clk.size() / 2], cyc[cyc.size() / 2]);
        }
    return 0;
}
