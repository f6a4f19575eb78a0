// Cost of a per-lane gather by element size and alignment on gfx950: every lane of a wave reads `BYTES` bytes at
// base + row * pitch + (x0 + lane) * STRIDE, x0 and row pseudo-random per wave (neighbouring lanes = neighbouring texels, as the
// tap rows of the reduction kernel), from a table of 256 MB (HBM / Infinity Cache) or 2 MB (L2).  Reports ns per wave-instruction
// per CU and GB/s of useful bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_align gather_align.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct __attribute__((packed, aligned(4))) a4x2 { uint32_t a, b; };
struct __attribute__((packed, aligned(4))) a4x3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(2))) a2x3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) a1x3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(8))) a8x4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) a1x1 { uint16_t a; };

template <class T, int STRIDE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ tab, size_t rows, size_t pitch, uint32_t* out, int iters)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned h = wave * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        uint32_t part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                        // 4 independent gathers in flight, like the 4 records per lane
            h = h * 1664525u + 1013904223u;
            const size_t row = (h >> 8) % rows;
            const size_t x0 = (h >> 3) % 512;
            const T v = *reinterpret_cast<const T*>(tab + row * pitch + (x0 + lane) * STRIDE);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
            uint32_t s = 0;
            for (unsigned q = 0; q < sizeof(T) / 4; ++q) s ^= w[q];
            if (sizeof(T) < 4) s = *reinterpret_cast<const uint16_t*>(&v);
            part[j] = s;
        }
        acc += part[0] ^ part[1] ^ part[2] ^ part[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class T, int STRIDE>
int run(const char* name, const uint8_t* tab, size_t bytes, uint32_t* out, int cus)
{
    const size_t pitch = 8192, rows = bytes / pitch - 1;
    const int iters = 512, blocks = cus * 8;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<T, STRIDE>), dim3(blocks), dim3(256), 0, 0, tab, rows, pitch, out, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double insts = (double)blocks * 4 * iters * 4;                 // wave-instructions
    printf("  %-34s table %4zu MB: %7.3f ms  %6.1f ns per wave-instruction per CU  %7.1f GB/s useful\n", name, bytes >> 20, ms,
           ms * 1e6 / (insts / cus), insts * 64 * sizeof(T) / (ms * 1e6));
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t* tab; uint32_t* out;
    const size_t big = 256u << 20, small = 2u << 20;
    CHK(hipMalloc(&tab, big + 65536));
    CHK(hipMemset(tab, 1, big + 65536));
    CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    for (size_t bytes : {small, big}) {
        if (run<a4x2, 4>("8 B at 4-byte alignment (today)", tab, bytes, out, cus)) return 1;
        if (run<a4x3, 4>("12 B at 4-byte alignment", tab, bytes, out, cus)) return 1;
        if (run<a2x3, 6>("12 B at 2-byte alignment (6 B texels)", tab, bytes, out, cus)) return 1;
        if (run<a1x3, 5>("12 B at 1-byte alignment (5 B texels)", tab, bytes, out, cus)) return 1;
        if (run<a8x4, 8>("16 B at 8-byte alignment (8 B texels)", tab, bytes, out, cus)) return 1;
        if (run<a1x1, 1>("2 B at 1-byte alignment", tab, bytes, out, cus)) return 1;
    }
    return 0;
}
