// What a per-lane tap gather costs when the texture unit does the unpacking: typed buffer loads (MUBUF buffer_load_format_*:
// the resource descriptor names a data / numeric format and the texture data unit returns float32 per channel) against the
// untyped 8-byte gather of the reduction kernel.  Same access pattern as gather_align.hip: every lane of a wave reads the two
// texels at base + row * pitch + (x0 + lane) * 4, x0 and row pseudo-random per wave, 4 gathers in flight per lane.
// Also checks the values: element = two 4-byte texels {f16 gx, f16 gy} at a 4-byte-aligned (not 8-byte-aligned) address.
// Build: hipcc --offload-arch=gfx950 -O3 -o typed_gather typed_gather.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ f4 llvm_load_format_v4f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ f2 llvm_load_format_v2f32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2f32");
__device__ u2 llvm_load_v2i32(i4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");

// word 3 of a buffer resource (gfx9): dst_sel x,y,z,w = 4,5,6,7; num_format << 12; data_format << 15
constexpr int W3(int nfmt, int dfmt) { return 4 | (5 << 3) | (6 << 6) | (7 << 9) | (nfmt << 12) | (dfmt << 15); }
enum { NF_USCALED = 2, NF_SSCALED = 3, NF_FLOAT = 7, DF_8_8 = 3, DF_16_16 = 5, DF_8_8_8_8 = 10, DF_16_16_16_16 = 12 };

__device__ __forceinline__ i4 make_rsrc(const void* p, unsigned bytes, int w3)
{
    const uint64_t a = (uint64_t)p;
    i4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = w3;
    return r;
}

// MODE 0: global_load_dwordx2 + unpack of two {f16,f16} texels with v_cvt_f32_f16 (4 conversions)
// MODE 1: raw buffer_load_dwordx2 (same bytes through the buffer path), same unpack
// MODE 2: buffer_load_format_xyzw 16_16_16_16 FLOAT (4 floats, no VALU)
// MODE 3: buffer_load_format_xyzw 16_16_16_16 SSCALED (int16 -> float)
// MODE 4: buffer_load_format_xy 8_8 USCALED at 1-byte alignment (two neighbouring intensity bytes)
// MODE 5: global_load_ushort at 1-byte alignment + 2 v_cvt_f32_ubyte
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ tab, unsigned bytes, size_t rows, size_t pitch, float* out, int iters)
{
    const unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned h = wave * 2654435761u + 12345u;
    const i4 rs = make_rsrc(tab, bytes, MODE == 2 ? W3(NF_FLOAT, DF_16_16_16_16) : MODE == 3 ? W3(NF_SSCALED, DF_16_16_16_16)
                                        : MODE == 4 ? W3(NF_USCALED, DF_8_8) : 0x00020000);
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        float part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) % (unsigned)rows;
            const unsigned x0 = (h >> 3) % 512;
            const unsigned off = row * (unsigned)pitch + (x0 + lane) * ((MODE >= 4) ? 1u : 4u);
            float s;
            if (MODE == 0) {
                struct __attribute__((packed, aligned(4))) a4x2 { uint32_t a, b; };
                const a4x2 v = *reinterpret_cast<const a4x2*>(tab + off);
                s = (__half2float(__ushort_as_half((unsigned short)(v.a & 0xffffu))) + __half2float(__ushort_as_half((unsigned short)(v.a >> 16)))) +
                    (__half2float(__ushort_as_half((unsigned short)(v.b & 0xffffu))) + __half2float(__ushort_as_half((unsigned short)(v.b >> 16))));
            } else if (MODE == 1) {
                const u2 v = llvm_load_v2i32(rs, (int)off, 0, 0);
                s = (__half2float(__ushort_as_half((unsigned short)(v.x & 0xffffu))) + __half2float(__ushort_as_half((unsigned short)(v.x >> 16)))) +
                    (__half2float(__ushort_as_half((unsigned short)(v.y & 0xffffu))) + __half2float(__ushort_as_half((unsigned short)(v.y >> 16))));
            } else if (MODE == 2 || MODE == 3) {
                const f4 v = llvm_load_format_v4f32(rs, (int)off, 0, 0);
                s = (v.x + v.y) + (v.z + v.w);
            } else if (MODE == 4) {
                const f2 v = llvm_load_format_v2f32(rs, (int)off, 0, 0);
                s = v.x + v.y;
            } else {
                struct __attribute__((packed, aligned(1))) a1 { uint16_t a; };
                const unsigned v = reinterpret_cast<const a1*>(tab + off)->a;
                s = (float)(v & 0xffu) + (float)(v >> 8);
            }
            part[j] = s;
        }
        acc += (part[0] + part[1]) + (part[2] + part[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// values: out[lane][0..3] = the four channels of the element at byte offset offs[lane]
__global__ void k_check(const uint8_t* tab, unsigned bytes, const unsigned* offs, float* out, int w3, int n)
{
    const i4 rs = make_rsrc(tab, bytes, w3);
    const int t = threadIdx.x;
    if (t >= n) return;
    const f4 v = llvm_load_format_v4f32(rs, (int)offs[t], 0, 0);
    out[t * 4 + 0] = v.x; out[t * 4 + 1] = v.y; out[t * 4 + 2] = v.z; out[t * 4 + 3] = v.w;
}

template <int MODE>
int run(const char* name, const uint8_t* tab, size_t bytes, float* out, int cus)
{
    const size_t pitch = 8192, rows = bytes / pitch - 1;
    const int iters = 512, blocks = cus * 8;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, tab, (unsigned)bytes, rows, pitch, out, iters);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double insts = (double)blocks * 4 * iters * 4;
    printf("  %-58s table %4zu MB: %7.3f ms  %6.1f ns per wave-instruction per CU\n", name, bytes >> 20, ms, ms * 1e6 / (insts / cus));
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint8_t* tab; float* out;
    const size_t big = 256u << 20, small = 2u << 20;
    CHK(hipMalloc(&tab, big + 65536));
    // texels {f16 gx, f16 gy} with integer values: texel i holds gx = (i % 2041) - 1020, gy = 1020 - (i % 2039)
    {
        std::vector<uint32_t> h((big + 65536) / 4);
        for (size_t i = 0; i < h.size(); ++i) {
            const __half a = __float2half((float)((int)(i % 2041) - 1020)), b = __float2half((float)(1020 - (int)(i % 2039)));
            h[i] = (uint32_t)__half_as_ushort(a) | ((uint32_t)__half_as_ushort(b) << 16);
        }
        CHK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    // --- value check: elements at 8-byte-aligned and at 4-mod-8 offsets, FLOAT format ---
    {
        unsigned hoffs[64]; unsigned* doffs; float* dv; float hv[256];
        for (int i = 0; i < 64; ++i) hoffs[i] = 4u * (unsigned)(1000 + 37 * i);          // odd and even texel indices
        CHK(hipMalloc(&doffs, sizeof hoffs)); CHK(hipMalloc(&dv, sizeof hv));
        CHK(hipMemcpy(doffs, hoffs, sizeof hoffs, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, tab, (unsigned)small, doffs, dv, W3(NF_FLOAT, DF_16_16_16_16), 64);
        CHK(hipMemcpy(hv, dv, sizeof hv, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 64; ++i) {
            const size_t t = hoffs[i] / 4;
            const float e[4] = {(float)((int)(t % 2041) - 1020), (float)(1020 - (int)(t % 2039)), (float)((int)((t + 1) % 2041) - 1020), (float)(1020 - (int)((t + 1) % 2039))};
            for (int c = 0; c < 4; ++c) if (hv[i * 4 + c] != e[c]) { if (bad < 8) printf("  MISMATCH off %u ch %d: got %g want %g\n", hoffs[i], c, hv[i * 4 + c], e[c]); ++bad; }
        }
        printf("typed 16_16_16_16 FLOAT, 8-byte elements at 4-byte alignment: %s (%d mismatches of 256)\n", bad ? "WRONG" : "values exact", bad);
    }
    for (size_t bytes : {small, big}) {
        if (run<0>("global_load_dwordx2 + 4 cvt_f32_f16 (untyped)", tab, bytes, out, cus)) return 1;
        if (run<1>("buffer_load_dwordx2 + 4 cvt_f32_f16 (untyped, buffer path)", tab, bytes, out, cus)) return 1;
        if (run<2>("buffer_load_format_xyzw 16_16_16_16 FLOAT", tab, bytes, out, cus)) return 1;
        if (run<3>("buffer_load_format_xyzw 16_16_16_16 SSCALED", tab, bytes, out, cus)) return 1;
        if (run<4>("buffer_load_format_xy 8_8 USCALED, 1-byte aligned", tab, bytes, out, cus)) return 1;
        if (run<5>("global_load_ushort 1-byte aligned + 2 cvt_f32_ubyte", tab, bytes, out, cus)) return 1;
    }
    return 0;
}
