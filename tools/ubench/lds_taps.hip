// What a bilinear tap costs when the texels sit in an LDS window: per-lane LDS reads at tap-like addresses (lane L reads the texels at
// x0 + L of a pseudo-random row of a 32 KB window) by instruction form and alignment, 4 reads in flight per lane, 5 workgroups of 256
// threads per CU.  Reports ns per wave-instruction per CU and checks the values of the unaligned forms.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_taps lds_taps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned u2 __attribute__((ext_vector_type(2)));

// MODE 0: ds_read2_b32 offset1:1 at 4-byte alignment (two neighbouring 4-byte texels)
// MODE 1: ds_read_b64 at 4-byte alignment (same bytes, one 8-byte access off its natural alignment)
// MODE 2: ds_read_b32 at 4-byte alignment
// MODE 3: ds_read_u8 (1-byte texels)
// MODE 4: ds_read_u16 at 1-byte alignment (two neighbouring 1-byte texels)
// MODE 5: ds_read_b32 at 2-byte alignment (two neighbouring 2-byte texels)
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned* vals)
{
    __shared__ __attribute__((aligned(16))) unsigned win[8192];             // 32 KB
    for (int i = threadIdx.x; i < 8192; i += 256) win[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    constexpr unsigned ES = (MODE <= 2) ? 4u : (MODE == 5 ? 2u : 1u);          // texel size in bytes
    unsigned h = wave * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned a[4], r0[4], r1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            a[j] = ((h >> 8) % (32768u - 64u * ES - 16u)) / ES * ES + lane * ES;     // texel-aligned start, lanes = neighbouring texels
            r1[j] = 0;
        }
        if (MODE == 0) {
            u2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v[j]) : "v"(a[j]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) { r0[j] = v[j].x; r1[j] = v[j].y; }
        } else if (MODE == 1) {
            u2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(v[j]) : "v"(a[j]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) { r0[j] = v[j].x; r1[j] = v[j].y; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 2 || MODE == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(r0[j]) : "v"(a[j]));
                if (MODE == 3) asm volatile("ds_read_u8 %0, %1" : "=v"(r0[j]) : "v"(a[j]));
                if (MODE == 4) asm volatile("ds_read_u16 %0, %1" : "=v"(r0[j]) : "v"(a[j]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]));
        }
        if (vals != nullptr && i == 0 && blockIdx.x == 0 && threadIdx.x < 64) {
            // expected bytes straight from the window
            const unsigned char* wb = reinterpret_cast<const unsigned char*>(win);
            unsigned bad = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned e0 = 0, e1 = 0;
                const unsigned nb = (MODE <= 1) ? 8u : (MODE == 2 || MODE == 5) ? 4u : (MODE == 4 ? 2u : 1u);
                for (unsigned b = 0; b < nb && b < 4; ++b) e0 |= (unsigned)wb[a[j] + b] << (8 * b);
                for (unsigned b = 4; b < nb; ++b) e1 |= (unsigned)wb[a[j] + b] << (8 * (b - 4));
                bad += (r0[j] != e0) + ((MODE <= 1) && (r1[j] != e1));
            }
            vals[threadIdx.x] = bad;
        }
        acc += (r0[0] ^ r0[1] ^ r0[2] ^ r0[3]) + (r1[0] ^ r1[1] ^ r1[2] ^ r1[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
int run(const char* name, unsigned* out, unsigned* vals, int cus)
{
    const int iters = 2048, blocks = cus * 5;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    CHK(hipMemset(vals, 0, 64 * 4));
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, vals);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned hv[64]; unsigned bad = 0;
    CHK(hipMemcpy(hv, vals, sizeof hv, hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; ++i) bad += hv[i];
    const double insts = (double)blocks * 4 * iters * 4;
    printf("  %-52s %7.3f ms  %6.2f ns per wave-instruction per CU   values %s\n", name, ms, ms * 1e6 / (insts / cus), bad ? "WRONG" : "ok");
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *out, *vals;
    CHK(hipMalloc(&out, (size_t)cus * 5 * 256 * 4));
    CHK(hipMalloc(&vals, 64 * 4));
    if (run<0>("ds_read2_b32 offset1:1, 4-byte aligned", out, vals, cus)) return 1;
    if (run<1>("ds_read_b64, 4-byte aligned", out, vals, cus)) return 1;
    if (run<2>("ds_read_b32, 4-byte aligned", out, vals, cus)) return 1;
    if (run<3>("ds_read_u8", out, vals, cus)) return 1;
    if (run<4>("ds_read_u16, 1-byte aligned", out, vals, cus)) return 1;
    if (run<5>("ds_read_b32, 2-byte aligned", out, vals, cus)) return 1;
    return 0;
}
