// Micro-benchmark: does the VGPR bank of a VALU instruction's source operands change its issue cost on gfx950?
// 16 independent v_fma_f32 per iteration with explicit registers (accumulators v0..v15), 4 or 8 waves per SIMD (one or two 1024-thread workgroups per CU), cycles by s_memtime (the
// shader clock itself, not wall time x nominal clock).  Cases: sources in three different banks / two in one bank / three in one bank
// (bank = register number mod 4), and the same for v_mul_f32 / v_add_f32 (two sources).
// Build: hipcc --offload-arch=gfx950 -O3 -o vgpr_banks vgpr_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v40","v41","v42","v43","v44","v45","v48","v52"

// X(dst, a, b): one instruction
#define FMA(d, a, b) "v_fma_f32 v" #d ", v" #a ", v" #b ", v" #d "\n"
#define MUL(d, a, b) "v_mul_f32 v" #d ", v" #a ", v" #d "\n"
#define ADD(d, a, b) "v_add_f32 v" #d ", v" #a ", v" #d "\n"
#define CVT(d, a, b) "v_cvt_f32_i32 v" #d ", v" #d "\n"
#define FMK(d, a, b) "v_fmac_f32 v" #d ", v" #a ", v" #b "\n"
// the same operations in the 8-byte VOP3 encoding (the compiler emits the 4-byte VOP2 form whenever the operands allow it)
#define MUL64(d, a, b) "v_mul_f32_e64 v" #d ", v" #a ", v" #d "\n"
#define ADD64(d, a, b) "v_add_f32_e64 v" #d ", v" #a ", v" #d "\n"
#define FMK64(d, a, b) "v_fmac_f32_e64 v" #d ", v" #a ", v" #b "\n"
#define MULF(d, a, b) "v_fma_f32 v" #d ", v" #a ", v" #d ", 0\n"
#define CVT64(d, a, b) "v_cvt_f32_i32_e64 v" #d ", v" #d "\n"
// out of place: destination differs from the sources (dst d, sources a and d+16)
#define MULO(d, a, b) "v_mul_f32 v" #d ", v" #a ", v41\n"
// the VOP2 FMA with its first multiplicand routed through the DPP network (identity permutation)
#define FMKD(d, a, b) "v_fmac_f32_dpp v" #d ", v" #a ", v" #b " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
// 16 instructions, accumulators 0..15 in order (banks 0,1,2,3,0,...)
#define SEQ_ALL(X, a, b) X(0,a,b) X(1,a,b) X(2,a,b) X(3,a,b) X(4,a,b) X(5,a,b) X(6,a,b) X(7,a,b) X(8,a,b) X(9,a,b) X(10,a,b) X(11,a,b) X(12,a,b) X(13,a,b) X(14,a,b) X(15,a,b)
// 16 instructions, accumulators of bank 0 only (0,4,8,12 four times)
#define SEQ_B0(X, a, b) X(0,a,b) X(4,a,b) X(8,a,b) X(12,a,b) X(0,a,b) X(4,a,b) X(8,a,b) X(12,a,b) X(0,a,b) X(4,a,b) X(8,a,b) X(12,a,b) X(0,a,b) X(4,a,b) X(8,a,b) X(12,a,b)
// accumulators of banks 2 and 3 only: never the bank of v40 / v44 (0) or v41 / v45 (1)
#define SEQ_B23(X, a, b) X(2,a,b) X(3,a,b) X(6,a,b) X(7,a,b) X(10,a,b) X(11,a,b) X(14,a,b) X(15,a,b) X(2,a,b) X(3,a,b) X(6,a,b) X(7,a,b) X(10,a,b) X(11,a,b) X(14,a,b) X(15,a,b)

template <int CASE>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, float* sink, int n)
{
    asm volatile("v_mov_b32 v40, 1.0\n v_mov_b32 v41, 0.5\n v_mov_b32 v44, 0.25\n v_mov_b32 v45, 2.0\n v_mov_b32 v48, 1.0\n v_mov_b32 v52, 0.5\n"
                 "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"
                 "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n" ::: CLOB);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (CASE == 0) asm volatile(SEQ_B23(FMA, 40, 41) ::: CLOB);       // fma: src banks 0, 1, acc 2 / 3: three different banks
        if (CASE == 1) asm volatile(SEQ_B23(FMA, 40, 44) ::: CLOB);       // fma: src0, src1 both bank 0
        if (CASE == 2) asm volatile(SEQ_B0(FMA, 40, 41) ::: CLOB);        // fma: src0 and acc bank 0
        if (CASE == 3) asm volatile(SEQ_B0(FMA, 40, 44) ::: CLOB);        // fma: all three bank 0
        if (CASE == 4) asm volatile(SEQ_ALL(FMA, 40, 41) ::: CLOB);       // fma: accumulators in all banks (what a compiler produces)
        if (CASE == 5) asm volatile(SEQ_B23(MUL, 40, 41) ::: CLOB);       // mul: different banks
        if (CASE == 6) asm volatile(SEQ_B0(MUL, 40, 41) ::: CLOB);        // mul: same bank
        if (CASE == 7) asm volatile(SEQ_B23(ADD, 40, 41) ::: CLOB);       // add: different banks
        if (CASE == 8) asm volatile(SEQ_B0(ADD, 40, 41) ::: CLOB);        // add: same bank
        if (CASE == 9) asm volatile(SEQ_ALL(CVT, 40, 41) ::: CLOB);       // cvt (one source)
        if (CASE == 10) asm volatile(SEQ_B23(FMK, 40, 41) ::: CLOB);      // fmac (VOP2 encoding): different banks
        if (CASE == 11) asm volatile(SEQ_B0(FMK, 40, 44) ::: CLOB);       // fmac: all three bank 0
        if (CASE == 12) asm volatile(SEQ_B23(MUL64, 40, 41) ::: CLOB);
        if (CASE == 13) asm volatile(SEQ_B23(ADD64, 40, 41) ::: CLOB);
        if (CASE == 14) asm volatile(SEQ_B23(FMK64, 40, 41) ::: CLOB);
        if (CASE == 15) asm volatile(SEQ_B23(MULF, 40, 41) ::: CLOB);
        if (CASE == 16) asm volatile(SEQ_ALL(CVT64, 40, 41) ::: CLOB);
        if (CASE == 17) asm volatile(SEQ_B23(MULO, 40, 41) ::: CLOB);
        if (CASE == 18) asm volatile(SEQ_B23(FMA, 40, 40) ::: CLOB);      // a square: both multiplicands the SAME register
        if (CASE == 19) asm volatile(SEQ_B23(FMKD, 40, 41) ::: CLOB);     // dpp fmac, different banks
        if (CASE == 20) asm volatile(SEQ_B23(FMKD, 40, 40) ::: CLOB);     // dpp fmac, a square
        if (CASE == 21) asm volatile(SEQ_B23(FMKD, 40, 44) ::: CLOB);     // dpp fmac, multiplicands in one bank
        if (CASE == 22) asm volatile(SEQ_B23(FMK, 40, 40) ::: CLOB);      // VOP2 fmac, a square
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s;
    asm volatile("v_add_f32 %0, v0, v1\n v_add_f32 %0, %0, v2\n v_add_f32 %0, %0, v3\n v_add_f32 %0, %0, v4\n v_add_f32 %0, %0, v8\n v_add_f32 %0, %0, v12" : "=v"(s) :: CLOB);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int CASE>
static int run(const char* name, int cus, int waves, unsigned long long* d_cyc, float* d_sink, int n)
{
    const int blocks = cus * (waves / 4);            // 1024-thread workgroups (4 waves per SIMD each): one or two per CU, evenly placed
    hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(1024), 0, 0, d_cyc, d_sink, 16);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(1024), 0, 0, d_cyc, d_sink, n);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long h[8192];
    CHK(hipMemcpy(h, d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)h[i];
    mean /= blocks;
    // s_memtime ticks at 100 MHz on gfx950 (constant clock), so cycles come from wall time x the clock the run reports
    const double per_instr_ticks = mean / ((double)n * 16 * waves);
    const double ns_per_instr = 1e6 * ms / ((double)n * 16 * waves);
    printf("%-58s %d waves/SIMD: %.3f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz; memtime ticks per instr %.4f)\n", name, waves, ns_per_instr,
           ns_per_instr * 2.4, per_instr_ticks);
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_cyc;
    float* d_sink;
    CHK(hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 8));
    CHK(hipMalloc(&d_sink, sizeof(float) * cus * 8 * 1024));
    const int n = 20000;
    for (int waves : {4, 8, 4, 8}) {
        if (run<0>("v_fma_f32 sources in three banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<1>("v_fma_f32 src0, src1 in one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<2>("v_fma_f32 src0 and accumulator in one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<3>("v_fma_f32 all three sources in one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<4>("v_fma_f32 accumulators over all banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<5>("v_mul_f32 two banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<6>("v_mul_f32 one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<7>("v_add_f32 two banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<8>("v_add_f32 one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<9>("v_cvt_f32_i32", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<10>("v_fmac_f32 (VOP2) three banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<11>("v_fmac_f32 (VOP2) one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<12>("v_mul_f32_e64 (VOP3 encoding)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<13>("v_add_f32_e64 (VOP3 encoding)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<14>("v_fmac_f32_e64 (VOP3 encoding) three banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<15>("v_fma_f32 d, a, d, 0 (a multiply as an FMA)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<16>("v_cvt_f32_i32_e64 (VOP3 encoding)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<17>("v_mul_f32 d, a, b (VOP2, destination not a source)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<18>("v_fma_f32 d, a, a, d (both multiplicands one register)", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<19>("v_fmac_f32_dpp identity, different banks", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<20>("v_fmac_f32_dpp identity, a square", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<21>("v_fmac_f32_dpp identity, multiplicands in one bank", cus, waves, d_cyc, d_sink, n)) return 1;
        if (run<22>("v_fmac_f32 (VOP2), a square", cus, waves, d_cyc, d_sink, n)) return 1;
    }
    return 0;
}
