// HIP kernels (gfx950 / CDNA4, wave64) of the dense RGB-D direct-alignment core.
//
// Built with -ffp-contract=off: the coordinate path (back-projection, SE(3) transform, projection) must
// round exactly like oracle/dvo_oracle.py (IEEE float32, one rounding per operation, no FMA) so that the
// in-bounds decision of every pixel is bit-identical; everywhere an FMA is wanted it is written fmaf().
//
// Kernels
//   k_batch_median_down K1 3x3 median (replicated border) + keep even rows/cols, u8+u16, 4 px/lane (utils/image_pyramid.py:19-21)
//   k_batch_sobel_pack K2  Sobel 3x3 BORDER_REFLECT, packed with the intensity, 4 px/lane (utils/jacobian.py:47-73)
//   k_bgr_ingest       K0  BGR->gray fixed point + depth clamp                     (core/base_dense_visual_odometry.py:58-59)
//   k_reduce<...>      K3  residual + Jacobian + weight + 30-term reduction        (cpu_robust...:134-200 + base_robust_dvo.py:168-188)
//                          (scale_pass form = K3s, weighter/t_weighter.py:36-47)
//   k_sum_partials     K4  deterministic fixed-order sum of the per-workgroup partials
//   k_dense<...>       K5  dense r / J / valid / uv writer (return contract of compute_residuals_and_jacobian)
//   k_init_state, k_level_begin, k_solve, k_finish   device-side Gauss-Newton control (base_robust_dvo.py:152-232)
//   k_gn_small_256 / _512 / _1024<...>  every iteration of the coarse levels of one pair in one persistent workgroup (reduction +
//                          control; the narrow workgroups call the control step out of line: gn_control_small)
//   k_ingest_level_fused<TW>  one pass per level: planes, Sobel + pack, next level's median pyramid, valid-pixel counts
//   k_list_scan, k_list_build  valid-point lists (records {z, x | y | I} of the pixels with depth != 0, raster order)
//   k_batch_bgr_ingest     batched colour conversion + depth clamp in front of the level-0 pass
//
// Template parameter WMODE of k_reduce / k_gn_small: 0 no weight, 1 Student-t with the lambda of wparam, 2 Huber; k_reduce only, the
// reference's Student-t in one pass per iteration: 4 moment form (second set of sums), 5 weights under a predicted lambda.
// Template parameter VARIANT: 0 = V1 (reference CPU default), 1 = V2 (approximate_image2_gradient), 2 = V3 (numba.cuda kernel
// semantics), 3 = GN (build-defined: V1 with the gradient normalised and J_w at the warped point).
//
// References are relative to /root/reference/src/dense_visual_odometry/.
#include "dvo_internal.h"
#include <math.h>

// ================================================================================================
// K1: median 3x3 + decimate
// ================================================================================================
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c) { return min(min(a, b), c); }     // v_min3_u32
__device__ __forceinline__ unsigned umax3(unsigned a, unsigned b, unsigned c) { return max(max(a, b), c); }     // v_max3_u32

// Median of nine as 13 three-operand instructions: sort each triple (min3 / med3 / max3), then the median of
// {largest of the minima, median of the medians, smallest of the maxima}.
__device__ __forceinline__ unsigned median9(unsigned p0, unsigned p1, unsigned p2, unsigned p3, unsigned p4,
                                            unsigned p5, unsigned p6, unsigned p7, unsigned p8)
{
    const unsigned lo = umax3(umin3(p0, p1, p2), umin3(p3, p4, p5), umin3(p6, p7, p8));
    const unsigned mi = umed3(umed3(p0, p1, p2), umed3(p3, p4, p5), umed3(p6, p7, p8));
    const unsigned hi = umin3(umax3(p0, p1, p2), umax3(p3, p4, p5), umax3(p6, p7, p8));
    return umed3(lo, mi, hi);
}

__global__ __launch_bounds__(DVO_BLOCK) void k_unpack(const uint32_t* __restrict__ packed, int w, int h, int pitch,
                                                      int16_t* __restrict__ gx, int16_t* __restrict__ gy)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t v = packed[(size_t)y * pitch + x];
    if (gx) gx[(size_t)y * w + x] = (int16_t)(((int32_t)(v << 12)) >> 20);
    if (gy) gy[(size_t)y * w + x] = (int16_t)(((int32_t)v) >> 20);
}

hipError_t dvo_launch_unpack(const uint32_t* packed, int w, int h, int pitch, int16_t* gx, int16_t* gy, hipStream_t s)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_unpack, grid, dim3(DVO_BLOCK), 0, s, packed, w, h, pitch, gx, gy);
    return hipGetLastError();
}

// ================================================================================================
// K0: colour ingest
// ================================================================================================
__global__ __launch_bounds__(DVO_BLOCK) void k_bgr_ingest(const uint8_t* __restrict__ bgr,
                                                          const uint16_t* __restrict__ depth_in, int n,
                                                          uint8_t* __restrict__ gray, uint16_t* __restrict__ depth_out,
                                                          double depth_scale, double max_distance)
{
    int i = blockIdx.x * DVO_BLOCK + threadIdx.x;
    if (i >= n) return;
    int b = bgr[3 * (size_t)i], g = bgr[3 * (size_t)i + 1], r = bgr[3 * (size_t)i + 2];
    gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    uint16_t d = depth_in[i];
    // depth_image[(depth_image * depth_scale) > max_distance] = 0   (float64 product, as NumPy computes it)
    depth_out[i] = ((double)d * depth_scale > max_distance) ? (uint16_t)0 : d;
}

hipError_t dvo_launch_bgr_ingest(const uint8_t* bgr, const uint16_t* depth_in, int w, int h, uint8_t* gray,
                                 uint16_t* depth_out, double depth_scale, double max_distance, hipStream_t s)
{
    int n = w * h;
    hipLaunchKernelGGL(k_bgr_ingest, dim3((n + DVO_BLOCK - 1) / DVO_BLOCK), dim3(DVO_BLOCK), 0, s, bgr, depth_in, n,
                       gray, depth_out, depth_scale, max_distance);
    return hipGetLastError();
}

// Batched colour ingest: n frames of BGR u8 [n][h][w][3] + depth u16 [n][h][w] (device) -> gray / clamped depth written straight
// into the level-0 planes of the listed slots (blockIdx.z = frame), 4 pixels per lane: 12 + 8 bytes in, 4 + 8 bytes out.
// The depth clamp `depth * depth_scale > max_distance -> 0` (core/base_dense_visual_odometry.py:59, a float64 product) is
// monotonic in the integer depth, so the host turns it into the exact integer threshold d_max (largest depth that is kept).
__global__ __launch_bounds__(DVO_BLOCK) void k_batch_bgr_ingest(const DvoBatchIngestParams p, const uint8_t* __restrict__ bgr,
                                                                const uint16_t* __restrict__ depth_in, unsigned d_max)
{
    const int f = blockIdx.z;
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    const size_t n = (size_t)p.g[0].npix;
    const size_t i = ((size_t)blockIdx.x * DVO_BLOCK + threadIdx.x) * 4;
    if (i >= n) return;
    const uint8_t* sb = bgr + ((size_t)f * n + i) * 3;
    const uint16_t* sd = depth_in + (size_t)f * n + i;
    uint8_t* dg = slot + p.g[0].off_gray + i;
    uint16_t* dd = reinterpret_cast<uint16_t*>(slot + p.g[0].off_depth) + i;
    unsigned c[12], d[4];
    const bool full = (i + 4 <= n);
    if (full && ((reinterpret_cast<uintptr_t>(sb) & 3u) == 0) && ((reinterpret_cast<uintptr_t>(sd) & 7u) == 0)) {
        const uint32_t a0 = reinterpret_cast<const uint32_t*>(sb)[0], a1 = reinterpret_cast<const uint32_t*>(sb)[1],
                       a2 = reinterpret_cast<const uint32_t*>(sb)[2];
        const uint2 dv = *reinterpret_cast<const uint2*>(sd);
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = (a0 >> (8 * k)) & 0xffu; c[4 + k] = (a1 >> (8 * k)) & 0xffu; c[8 + k] = (a2 >> (8 * k)) & 0xffu; }
        d[0] = dv.x & 0xffffu; d[1] = dv.x >> 16; d[2] = dv.y & 0xffffu; d[3] = dv.y >> 16;
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = (i + k / 3 < n) ? sb[k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = (i + k < n) ? sd[k] : 0u;
    }
    unsigned g4 = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned gv = (c[3 * k] * 3735u + c[3 * k + 1] * 19235u + c[3 * k + 2] * 9798u + (1u << 14)) >> 15;     // as k_bgr_ingest
        g4 |= gv << (8 * k);
        d[k] = (d[k] > d_max) ? 0u : d[k];
    }
    if (full && ((reinterpret_cast<uintptr_t>(dg) & 3u) == 0) && ((reinterpret_cast<uintptr_t>(dd) & 7u) == 0)) {
        *reinterpret_cast<uint32_t*>(dg) = g4;
        *reinterpret_cast<uint2*>(dd) = make_uint2(d[0] | (d[1] << 16), d[2] | (d[3] << 16));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k < n) { dg[k] = (uint8_t)(g4 >> (8 * k)); dd[k] = (uint16_t)d[k]; }
    }
}

hipError_t dvo_launch_batch_bgr_ingest(const DvoBatchIngestParams& p, const uint8_t* bgr, const uint16_t* depth_in, unsigned d_max,
                                       hipStream_t s)
{
    const unsigned nb = (unsigned)(((size_t)p.g[0].npix + DVO_BLOCK * 4 - 1) / (DVO_BLOCK * 4));
    hipLaunchKernelGGL(k_batch_bgr_ingest, dim3(nb, 1, (unsigned)p.n_frames), dim3(DVO_BLOCK), 0, s, p, bgr, depth_in, d_max);
    return hipGetLastError();
}

// ================================================================================================
// Batched frame ingest: the same K1/K2 arithmetic for n frames per launch (blockIdx.z = frame), so a
// whole batch of frames costs 3 launches per pyramid level instead of 3 per level per frame.
// ================================================================================================
__global__ __launch_bounds__(DVO_BLOCK) void k_batch_copy_level0(const DvoBatchIngestParams p)
{
    const int f = blockIdx.z;
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    const size_t n = (size_t)p.g[0].npix;
    const size_t i = ((size_t)blockIdx.x * DVO_BLOCK + threadIdx.x) * 16;
    if (i >= n) return;
    const uint8_t* sg = p.src_gray + (size_t)f * n;
    const uint16_t* sd = p.src_depth + (size_t)f * n;
    uint8_t* dg = slot + p.g[0].off_gray;
    uint16_t* dd = reinterpret_cast<uint16_t*>(slot + p.g[0].off_depth);
    if (i + 16 <= n && (((size_t)f * n) % 16 == 0)) {
        *reinterpret_cast<uint4*>(dg + i) = *reinterpret_cast<const uint4*>(sg + i);
        *reinterpret_cast<uint4*>(dd + i) = *reinterpret_cast<const uint4*>(sd + i);
        *reinterpret_cast<uint4*>(dd + i + 8) = *reinterpret_cast<const uint4*>(sd + i + 8);
    } else {
        for (size_t k = i; k < n && k < i + 16; ++k) { dg[k] = sg[k]; dd[k] = sd[k]; }
    }
}

// 4 output pixels per lane; gray (u8) and depth (u16) of one level in the same launch.  Fast path: one 8-byte
// (u8) / 16-byte (u16) load per source row + the left neighbour; any odd geometry falls back to clamped scalar loads.
template <typename T>
__device__ __forceinline__ void median_down4(const T* __restrict__ src, int sw, int sh, T* __restrict__ dst, int dw,
                                             int x4, int y)
{
    const int cy = 2 * y, ym = max(cy - 1, 0), yp = min(cy + 1, sh - 1);
    const T* rows[3] = {src + (size_t)ym * sw, src + (size_t)cy * sw, src + (size_t)yp * sw};
    unsigned c[3][9];
    const int sx = 2 * x4;
    const bool fast = (sw % 8 == 0) && (sx + 8 <= sw);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        c[r][0] = rows[r][max(sx - 1, 0)];
        if (fast) {
            if (sizeof(T) == 1) {
                const uint2 v = *reinterpret_cast<const uint2*>(rows[r] + sx);
#pragma unroll
                for (int k = 0; k < 4; ++k) { c[r][1 + k] = (v.x >> (8 * k)) & 0xffu; c[r][5 + k] = (v.y >> (8 * k)) & 0xffu; }
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(rows[r] + sx);
                c[r][1] = v.x & 0xffffu; c[r][2] = v.x >> 16; c[r][3] = v.y & 0xffffu; c[r][4] = v.y >> 16;
                c[r][5] = v.z & 0xffffu; c[r][6] = v.z >> 16; c[r][7] = v.w & 0xffffu; c[r][8] = v.w >> 16;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[r][1 + k] = rows[r][min(sx + k, sw - 1)];
        }
    }
    unsigned m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        m[i] = median9(c[0][2 * i], c[0][2 * i + 1], c[0][2 * i + 2], c[1][2 * i], c[1][2 * i + 1], c[1][2 * i + 2],
                       c[2][2 * i], c[2][2 * i + 1], c[2][2 * i + 2]);
    T* out = dst + (size_t)y * dw + x4;
    if ((dw % 4 == 0) && (x4 + 4 <= dw)) {
        if (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(out) = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
        else *reinterpret_cast<uint2*>(out) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (x4 + i < dw) out[i] = (T)m[i];
    }
}

__global__ __launch_bounds__(DVO_BLOCK) void k_batch_median_down(const DvoBatchIngestParams p, int level)
{
    const int f = blockIdx.z;
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    const DvoLevelGeom& gs = p.g[level - 1];
    const DvoLevelGeom& gd = p.g[level];
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= gd.w || y >= gd.h) return;
    median_down4<uint8_t>(slot + gs.off_gray, gs.w, gs.h, slot + gd.off_gray, gd.w, x4, y);
    median_down4<uint16_t>(reinterpret_cast<const uint16_t*>(slot + gs.off_depth), gs.w, gs.h,
                           reinterpret_cast<uint16_t*>(slot + gd.off_depth), gd.w, x4, y);
}

__global__ __launch_bounds__(DVO_BLOCK) void k_batch_sobel_pack(const DvoBatchIngestParams p, int level)
{
    const int f = blockIdx.z;
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    const DvoLevelGeom& g = p.g[level];
    const uint8_t* gray = slot + g.off_gray;
    uint32_t* packed = reinterpret_cast<uint32_t*>(slot + g.off_packed);
    const int w = g.w, h = g.h;
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= w || y >= h) return;
    // BORDER_REFLECT with radius 1 == index clamp
    const int ym = max(y - 1, 0), yp = min(y + 1, h - 1);
    const uint8_t* rows[3] = {gray + (size_t)ym * w, gray + (size_t)y * w, gray + (size_t)yp * w};
    int c[3][6];
    const bool fast = (w % 4 == 0) && (x4 + 4 <= w);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        c[r][0] = rows[r][max(x4 - 1, 0)];
        if (fast) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(rows[r] + x4);
            c[r][1] = v & 0xffu; c[r][2] = (v >> 8) & 0xffu; c[r][3] = (v >> 16) & 0xffu; c[r][4] = v >> 24;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) c[r][1 + k] = rows[r][min(x4 + k, w - 1)];
        }
        c[r][5] = rows[r][min(x4 + 4, w - 1)];
    }
    uint32_t out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gx = (c[0][i + 2] + 2 * c[1][i + 2] + c[2][i + 2]) - (c[0][i] + 2 * c[1][i] + c[2][i]);
        const int gy = (c[2][i] + 2 * c[2][i + 1] + c[2][i + 2]) - (c[0][i] + 2 * c[0][i + 1] + c[0][i + 2]);
        out[i] = (uint32_t)c[1][i + 1] | (((uint32_t)gx & 0xfffu) << 8) | (((uint32_t)gy & 0xfffu) << 20);
    }
    uint32_t* dst = packed + (size_t)y * g.pitch + x4;
    if (x4 + 4 <= w) *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (x4 + i < w) dst[i] = out[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused per-level ingest (levels whose width is a multiple of 16 and whose height is even): one pass over level L does
// what k_batch_copy_level0 (L = 0) + k_batch_sobel_pack(L) + k_batch_median_down(L + 1) + the valid-pixel count of
// level L do in four.  A workgroup stages a 256 x 8 pixel tile of intensity and depth (plus a one-pixel halo, clamped
// at the image border: cv2's BORDER_REFLECT of radius 1 and the median's replicate border are both an index clamp)
// in LDS.
// ------------------------------------------------------------------------------------------------
// Tile: ING_TW pixels wide (256, 128, 64 or 32: the launch picks the width that wastes the fewest lanes on the level's
// row length), 2048 / ING_TW rows; 64 4-pixel column groups x row pairs = 256 threads either way.

template <int ING_TW>
__global__ __launch_bounds__(DVO_BLOCK) void k_ingest_level_fused(const DvoBatchIngestParams p, int segs_per_frame, int level,
                                                                  int seg_base)
{
    constexpr int ING_TH = 2048 / ING_TW;
    constexpr int ING_PAD = 4;                      // pad words left of a tile row (only the last is a halo): rows stay 16-byte aligned
    constexpr int ING_GP = ING_TW / 4 + 2 * ING_PAD;    // gray tile pitch in words (4 px per word)
    constexpr int ING_DP = ING_TW / 2 + 2 * ING_PAD;    // depth tile pitch in words (2 px per word)
    constexpr int CGN = ING_TW / 4;                 // 4-pixel column groups per tile row
    constexpr int ING_NSEG = 4 * (ING_TH - 1) + 2;  // list segments one tile can touch: (rows - 1) * 4096 / 1024 + 2 (<= 254)
    constexpr int ROWS = ING_TH + 2;
    constexpr int GQ = ING_TW / 16, DQ = ING_TW / 8;            // 16-byte pieces per tile row (intensity, depth)
    constexpr int NQ = ROWS * (GQ + DQ);
    constexpr int NI = (NQ + DVO_BLOCK - 1) / DVO_BLOCK;        // interior load rounds
    constexpr int NH = (4 * ROWS + DVO_BLOCK - 1) / DVO_BLOCK;  // halo load rounds (2 planes x 2 sides x rows)
    __shared__ __attribute__((aligned(16))) uint32_t gs[ROWS * ING_GP];
    __shared__ __attribute__((aligned(16))) uint32_t ds[ROWS * ING_DP];
    __shared__ int segc[ING_NSEG];
    const int f = blockIdx.z, t = threadIdx.x;
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    const DvoLevelGeom& g = p.g[level];
    const int w = g.w, h = g.h;
    const size_t n = (size_t)g.npix;
    const bool copy = (level == 0) && (p.src_gray != nullptr);     // else the level's planes are already in the slot
    const uint8_t* sg = copy ? p.src_gray + (size_t)f * n : slot + g.off_gray;
    const uint16_t* sd = copy ? p.src_depth + (size_t)f * n : reinterpret_cast<const uint16_t*>(slot + g.off_depth);
    const int x0 = blockIdx.x * ING_TW, y0 = blockIdx.y * ING_TH;
    if (t < ING_NSEG) segc[t] = 0;
    // Tile load: all global loads of a thread are issued before its first LDS store, 16 bytes per lane for the interior
    // (w % 16 == 0, so a piece is entirely inside the image or entirely right of it) — enough bytes in flight per CU to
    // stream from HBM.
    {
        uint4 v[NI];
        uint32_t hv[NH];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = t + k * DVO_BLOCK;
            v[k] = make_uint4(0u, 0u, 0u, 0u);
            if (i < ROWS * GQ) {
                const int r = i / GQ, c = i - r * GQ;
                const uint8_t* row = sg + (size_t)min(max(y0 - 1 + r, 0), h - 1) * w;
                const int gx = x0 + 16 * c;
                if (gx < w) v[k] = *reinterpret_cast<const uint4*>(row + gx);
                else { const uint32_t e = (uint32_t)row[w - 1] * 0x01010101u; v[k] = make_uint4(e, e, e, e); }
            } else if (i < NQ) {
                const int j = i - ROWS * GQ;
                const int r = j / DQ, c = j - r * DQ;
                const uint16_t* row = sd + (size_t)min(max(y0 - 1 + r, 0), h - 1) * w;
                const int gx = x0 + 8 * c;
                if (gx < w) v[k] = *reinterpret_cast<const uint4*>(row + gx);
                else { const uint32_t e = (uint32_t)row[w - 1] * 0x00010001u; v[k] = make_uint4(e, e, e, e); }
            }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int i = t + k * DVO_BLOCK;
            hv[k] = 0u;
            if (i < 4 * ROWS) {
                const int r = i >> 2, side = i & 1, plane = (i >> 1) & 1;
                const size_t ro = (size_t)min(max(y0 - 1 + r, 0), h - 1) * w;
                const int gx = side ? x0 + ING_TW : x0 - (plane ? 2 : 4);       // first pixel of the halo word
                if (plane == 0) {
                    const uint8_t* row = sg + ro;
                    hv[k] = (gx < 0) ? (uint32_t)row[0] * 0x01010101u : (gx >= w) ? (uint32_t)row[w - 1] * 0x01010101u
                                                                                  : *reinterpret_cast<const uint32_t*>(row + gx);
                } else {
                    const uint16_t* row = sd + ro;
                    hv[k] = (gx < 0) ? (uint32_t)row[0] * 0x00010001u : (gx >= w) ? (uint32_t)row[w - 1] * 0x00010001u
                                                                                  : *reinterpret_cast<const uint32_t*>(row + gx);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = t + k * DVO_BLOCK;
            if (i < ROWS * GQ) {
                const int r = i / GQ, c = i - r * GQ;
                *reinterpret_cast<uint4*>(gs + r * ING_GP + ING_PAD + 4 * c) = v[k];
            } else if (i < NQ) {
                const int j = i - ROWS * GQ;
                const int r = j / DQ, c = j - r * DQ;
                *reinterpret_cast<uint4*>(ds + r * ING_DP + ING_PAD + 4 * c) = v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) {
            const int i = t + k * DVO_BLOCK;
            if (i < 4 * ROWS) {
                const int r = i >> 2, side = i & 1, plane = (i >> 1) & 1;
                if (plane == 0) gs[r * ING_GP + (side ? ING_PAD + ING_TW / 4 : ING_PAD - 1)] = hv[k];
                else ds[r * ING_DP + (side ? ING_PAD + ING_TW / 2 : ING_PAD - 1)] = hv[k];
            }
        }
    }
    __syncthreads();
    // --- this level: planes (when they come from outside), Sobel + pack, valid-pixel counts ------------------
    {
        const int cg = t % CGN, rp = t / CGN;
        const int x = x0 + 4 * cg;
        uint8_t* dg = slot + g.off_gray;
        uint16_t* dd = reinterpret_cast<uint16_t*>(slot + g.off_depth);
        uint32_t* packed = reinterpret_cast<uint32_t*>(slot + g.off_packed);
        const int seg_first = (y0 * w + x0) >> 10;
#pragma unroll
        for (int dr = 0; dr < 2; ++dr) {
            const int y = y0 + 2 * rp + dr, tr = 2 * rp + dr + 1;
            if (x < w && y < h) {
                int c[3][6];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const uint32_t* q = gs + (tr - 1 + r) * ING_GP + ING_PAD - 1 + cg;
                    const uint32_t a = q[0], b = q[1], d = q[2];
                    c[r][0] = a >> 24;
                    c[r][1] = b & 0xffu; c[r][2] = (b >> 8) & 0xffu; c[r][3] = (b >> 16) & 0xffu; c[r][4] = b >> 24;
                    c[r][5] = d & 0xffu;
                }
                uint32_t out[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gx = (c[0][i + 2] + 2 * c[1][i + 2] + c[2][i + 2]) - (c[0][i] + 2 * c[1][i] + c[2][i]);
                    const int gy = (c[2][i] + 2 * c[2][i + 1] + c[2][i + 2]) - (c[0][i] + 2 * c[0][i + 1] + c[0][i + 2]);
                    out[i] = (uint32_t)c[1][i + 1] | (((uint32_t)gx & 0xfffu) << 8) | (((uint32_t)gy & 0xfffu) << 20);
                }
                {   // non-temporal: the packed plane is next read by the reduction launches (it would only displace their data)
                    uint32_t* pd = packed + (size_t)y * g.pitch + x;
                    __builtin_nontemporal_store(out[0], pd); __builtin_nontemporal_store(out[1], pd + 1);
                    __builtin_nontemporal_store(out[2], pd + 2); __builtin_nontemporal_store(out[3], pd + 3);
                }
                const uint32_t da = ds[tr * ING_DP + ING_PAD + 2 * cg], db = ds[tr * ING_DP + ING_PAD + 2 * cg + 1];
                if (copy) {
                    *reinterpret_cast<uint32_t*>(dg + (size_t)y * w + x) = gs[tr * ING_GP + ING_PAD + cg];
                    *reinterpret_cast<uint2*>(dd + (size_t)y * w + x) = make_uint2(da, db);
                }
                const int cnt = ((da & 0xffffu) != 0u) + ((da >> 16) != 0u) + ((db & 0xffffu) != 0u) + ((db >> 16) != 0u);
                if (cnt) atomicAdd(&segc[((y * w + x) >> 10) - seg_first], cnt);
            }
        }
        __syncthreads();
        if (t < ING_NSEG && segc[t] != 0) atomicAdd(p.seg_counts + (size_t)f * segs_per_frame + seg_base + seg_first + t, segc[t]);
    }
    // --- next level: 3x3 median at the even pixels (pyrDownMedianSmooth, utils/image_pyramid.py:19-54) --------
    if (level + 1 < p.levels) {
        const DvoLevelGeom& g1 = p.g[level + 1];
        uint8_t* g1p = slot + g1.off_gray;
        uint16_t* d1p = reinterpret_cast<uint16_t*>(slot + g1.off_depth);
        const int col = t % (ING_TW / 2);
        const int x1 = x0 / 2 + col;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r1 = (t / (ING_TW / 2)) * 2 + k;
            const int y1 = y0 / 2 + r1;
            if (x1 < g1.w && y1 < g1.h) {
                unsigned a[9], b[9];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int tr = 2 * r1 + dy;                          // tile row of L0 row 2*y1 - 1 + dy
                        const int pg = 2 * col + dx - 1 + 4 * ING_PAD;       // gray tile pixel of column 2*x1 - 1 + dx
                        const int pd = 2 * col + dx - 1 + 2 * ING_PAD;       // depth tile pixel
                        a[dy * 3 + dx] = (gs[tr * ING_GP + (pg >> 2)] >> ((pg & 3) * 8)) & 0xffu;
                        b[dy * 3 + dx] = (ds[tr * ING_DP + (pd >> 1)] >> ((pd & 1) * 16)) & 0xffffu;
                    }
                g1p[(size_t)y1 * g1.w + x1] = (uint8_t)median9(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]);
                d1p[(size_t)y1 * g1.w + x1] = (uint16_t)median9(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Valid-point lists.  Per frame and level: the pixels with depth != 0, in raster order, one DvoPointRec each.  Three
// launches for the whole batch and all levels: per-segment counts, exclusive scan per (frame, level), ordered scatter.
// A segment is DVO_LIST_SEG = 1024 consecutive pixels handled by one wave, lane L taking the 16 consecutive pixels
// 16 L .. 16 L + 15 (two 16-byte depth loads, one 16-byte intensity load).  The scatter compacts the records of a
// segment in LDS (lane-major order == raster order) and writes them out as whole cache lines.  Deterministic.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int list_level_of_segment(const DvoBatchIngestParams& p, int seg, int& seg_in_level)
{
    int l = 0;
    while (l + 1 < p.levels && seg >= p.g[l].nseg) { seg -= p.g[l].nseg; ++l; }
    seg_in_level = seg;
    return l;
}

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// What one wave holds of one segment between its loads and its scatter.
struct ListSegData {
    unsigned dw[8];      // 16 depths, two per word
    unsigned gw[4];      // 16 intensities
    int l, sl;           // level and segment within the level
    bool live;           // the segment exists
};

template <bool SCATTER>
__device__ __forceinline__ void list_seg_load(const DvoBatchIngestParams& p, const uint8_t* slot, int seg, int seg_end, int lane,
                                              ListSegData& d)
{
    d.live = seg < seg_end;
#pragma unroll
    for (int k = 0; k < 8; ++k) d.dw[k] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) d.gw[k] = 0u;
    d.l = 0; d.sl = 0;
    if (!d.live) return;                                                   // wave-uniform
    d.l = list_level_of_segment(p, seg, d.sl);
    const DvoLevelGeom& g = p.g[d.l];
    const uint16_t* depth = reinterpret_cast<const uint16_t*>(slot + g.off_depth);
    const uint8_t* gray = slot + g.off_gray;
    const int idx0 = d.sl * DVO_LIST_SEG + lane * 16;
    if (idx0 + 16 <= g.npix) {
        const uint4 a = *reinterpret_cast<const uint4*>(depth + idx0), b = *reinterpret_cast<const uint4*>(depth + idx0 + 8);
        d.dw[0] = a.x; d.dw[1] = a.y; d.dw[2] = a.z; d.dw[3] = a.w; d.dw[4] = b.x; d.dw[5] = b.y; d.dw[6] = b.z; d.dw[7] = b.w;
        if (SCATTER) {
            const uint4 c = *reinterpret_cast<const uint4*>(gray + idx0);
            d.gw[0] = c.x; d.gw[1] = c.y; d.gw[2] = c.z; d.gw[3] = c.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i0 = idx0 + 2 * k, i1 = i0 + 1;
            d.dw[k] = ((i0 < g.npix) ? (unsigned)depth[i0] : 0u) | (((i1 < g.npix) ? (unsigned)depth[i1] : 0u) << 16);
        }
        if (SCATTER) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) if (idx0 + 4 * k + b < g.npix) d.gw[k] |= (unsigned)gray[idx0 + 4 * k + b] << (8 * b);
        }
    }
}

template <bool SCATTER>
__device__ __forceinline__ void list_seg_emit(const DvoBatchIngestParams& p, uint8_t* slot, int* segc, int lane, const ListSegData& d,
                                              DvoPointRec* st)
{
    if (!d.live) return;                                                   // wave-uniform
    const DvoLevelGeom& g = p.g[d.l];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt += ((d.dw[k] & 0xffffu) != 0u) + ((d.dw[k] >> 16) != 0u);
    if (!SCATTER) {
        const int tot = wave_sum_i(cnt);
        if (lane == 0) *segc = tot;
        return;
    }
    // exclusive prefix of the per-lane counts across the wave
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    const int pos = incl - cnt;
    const int idx0 = d.sl * DVO_LIST_SEG + lane * 16;
    int y = idx0 / g.w, x = idx0 - y * g.w;
    DvoPointRec* list = reinterpret_cast<DvoPointRec*>(slot + g.off_list) + *segc;     // exclusive offset (k_list_scan)
    // The segment is compacted in two rounds through a half-size stage (lanes 0..31, then lanes 32..63): 4 KB of LDS per wave lets
    // 8 workgroups share a CU instead of 5 (the scatter is bound by bytes in flight: 197 -> 181 us per 257 VGA frames).
    const int total_a = __shfl(incl, 31, 64);
    for (int half = 0; half < 2; ++half) {
        const int base = half ? total_a : 0, cnt_h = half ? total - total_a : total_a;
        if ((lane >> 5) == half) {
            int q = pos - base;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const unsigned dd = (d.dw[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                if (dd != 0u) {
                    DvoPointRec rec;
                    rec.z = (float)((double)dd * p.depth_scale);            // camera_model.py:200-201
                    rec.pix = (unsigned)x | ((unsigned)y << 12) | (((d.gw[k >> 2] >> ((k & 3) * 8)) & 0xffu) << 24);
                    st[q++] = rec;
                }
                if (++x == g.w) { x = 0; ++y; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // non-temporal stores: the records are next read by the reduction launches, long after the caches have turned over
        for (int i = lane; i < cnt_h; i += 64) {
            const DvoPointRec r = st[i];
            __builtin_nontemporal_store(((unsigned long long)r.pix << 32) | (unsigned long long)__float_as_uint(r.z),
                                        reinterpret_cast<unsigned long long*>(list + base + i));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (d.sl == g.nseg - 1) {
        // tail padding: 256 records with z = NaN behind the last real one (inside DVO_LIST_SLACK).  A wave of k_reduce reads whole
        // 256-record groups; a NaN depth makes every warped coordinate NaN, which fails the in-bounds test like any other invalid
        // point — so the reduction needs no per-record `index < count` test (2 instructions per point).
        DvoPointRec pad;
        pad.z = __builtin_nanf("");
        pad.pix = 0u;
        for (int i = lane; i < 64 * DVO_PX_PER_THREAD; i += 64) list[total + i] = pad;
    }
}

// One wave takes DVO_LIST_SEGS_PER_WAVE consecutive segments, all loads (depth and intensity) issued before the first compaction.
// Measured per 257 VGA frames: 1 / 2 / 4 segments per wave 176 / 188 / 210 us — more independent waves beat longer ones.
#ifndef DVO_LIST_SEGS_PER_WAVE
#define DVO_LIST_SEGS_PER_WAVE 1
#endif
template <bool SCATTER>
__global__ __launch_bounds__(DVO_BLOCK) void k_list_build(const DvoBatchIngestParams p, int segs_per_frame, int seg_begin,
                                                          int seg_end)
{
    __shared__ DvoPointRec stage[SCATTER ? DVO_BLOCK / 64 : 1][SCATTER ? DVO_LIST_SEG / 2 : 1];    // half a segment per wave: 16 KB per workgroup
    const int f = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg0 = seg_begin + ((int)blockIdx.x * (DVO_BLOCK / 64) + wave) * DVO_LIST_SEGS_PER_WAVE;
    if (seg0 >= seg_end) return;                                           // wave-uniform
    uint8_t* slot = p.slots + (size_t)p.slot_ids[f] * p.slot_stride;
    ListSegData d[DVO_LIST_SEGS_PER_WAVE];
#pragma unroll
    for (int k = 0; k < DVO_LIST_SEGS_PER_WAVE; ++k) list_seg_load<SCATTER>(p, slot, seg0 + k, seg_end, lane, d[k]);
#pragma unroll
    for (int k = 0; k < DVO_LIST_SEGS_PER_WAVE; ++k)
        list_seg_emit<SCATTER>(p, slot, p.seg_counts + (size_t)f * segs_per_frame + seg0 + k, lane, d[k], stage[wave]);
}

// exclusive scan of the per-segment counts of one (frame, level); total -> list_count[slot][level]
__global__ __launch_bounds__(DVO_BLOCK) void k_list_scan(const DvoBatchIngestParams p, int segs_per_frame)
{
    __shared__ int part[DVO_BLOCK];
    const int f = blockIdx.y, l = blockIdx.x;
    int first = 0;
    for (int k = 0; k < l; ++k) first += p.g[k].nseg;
    int* c = p.seg_counts + (size_t)f * segs_per_frame + first;
    const int n = p.g[l].nseg;
    const int per = (n + DVO_BLOCK - 1) / DVO_BLOCK;
    const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += c[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < DVO_BLOCK; off <<= 1) {             // inclusive Hillis-Steele scan of the 256 partial sums
        const int v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) { const int v = c[i]; c[i] = run; run += v; }
    if (threadIdx.x == DVO_BLOCK - 1) p.list_count[p.slot_ids[f] * DVO_LIST_LEVELS + l] = part[DVO_BLOCK - 1];
}

hipError_t dvo_launch_batch_ingest(const DvoBatchIngestParams& p, hipStream_t s)
{
    const unsigned nf = (unsigned)p.n_frames;
    int segs = 0;
    for (int l = 0; l < p.levels; ++l) segs += p.g[l].nseg;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.src_gray) | reinterpret_cast<uintptr_t>(p.src_depth)) & 15u) == 0;
    const bool allow_fused = p.allow_fused != 0;
    hipError_t e = hipMemsetAsync(p.seg_counts, 0, (size_t)nf * segs * sizeof(int), s);    // fused levels accumulate their counts
    if (e != hipSuccess) return e;
    bool have_planes = (p.src_gray == nullptr);          // planes of the current level already in the slots?
    int seg_base = 0;
    for (int l = 0; l < p.levels; ++l) {
        const DvoLevelGeom& g = p.g[l];
        const bool fused = allow_fused && (g.w % 16 == 0) && (g.h % 2 == 0) && (l > 0 || aligned);
        dim3 grid((g.w + 255) / 256, (g.h + 3) / 4, nf);
        if (fused) {
            // tile width with the best lane utilisation on this row length (ties: the wider tile)
            int tw = 256;
            double best = 0.0;
            for (int c = 256; c >= 32; c >>= 1) {
                const double u = (double)g.w / (double)(((g.w + c - 1) / c) * c);
                if (u > best + 1e-9) { best = u; tw = c; }
            }
            const int th = 2048 / tw;
            dim3 fg((g.w + tw - 1) / tw, (g.h + th - 1) / th, nf);
            if (tw == 256) hipLaunchKernelGGL(k_ingest_level_fused<256>, fg, dim3(DVO_BLOCK), 0, s, p, segs, l, seg_base);
            else if (tw == 128) hipLaunchKernelGGL(k_ingest_level_fused<128>, fg, dim3(DVO_BLOCK), 0, s, p, segs, l, seg_base);
            else if (tw == 64) hipLaunchKernelGGL(k_ingest_level_fused<64>, fg, dim3(DVO_BLOCK), 0, s, p, segs, l, seg_base);
            else hipLaunchKernelGGL(k_ingest_level_fused<32>, fg, dim3(DVO_BLOCK), 0, s, p, segs, l, seg_base);
        } else {
            if (l == 0 && !have_planes) {
                dim3 cg((unsigned)((g.npix + DVO_BLOCK * 16 - 1) / (DVO_BLOCK * 16)), 1, nf);
                hipLaunchKernelGGL(k_batch_copy_level0, cg, dim3(DVO_BLOCK), 0, s, p);
            }
            hipLaunchKernelGGL(k_batch_sobel_pack, grid, dim3(DVO_BLOCK), 0, s, p, l);
            const unsigned cb = (unsigned)((g.nseg + (DVO_BLOCK / 64) * DVO_LIST_SEGS_PER_WAVE - 1) / ((DVO_BLOCK / 64) * DVO_LIST_SEGS_PER_WAVE));
            hipLaunchKernelGGL(k_list_build<false>, dim3(cb, 1, nf), dim3(DVO_BLOCK), 0, s, p, segs, seg_base, seg_base + g.nseg);
            if (l + 1 < p.levels) {
                const DvoLevelGeom& gn = p.g[l + 1];
                dim3 mg((gn.w + 255) / 256, (gn.h + 3) / 4, nf);
                hipLaunchKernelGGL(k_batch_median_down, mg, dim3(DVO_BLOCK), 0, s, p, l + 1);
            }
        }
        have_planes = true;
        seg_base += g.nseg;
    }
    hipLaunchKernelGGL(k_list_scan, dim3((unsigned)p.levels, nf), dim3(DVO_BLOCK), 0, s, p, segs);
    const unsigned lb = (unsigned)((segs + (DVO_BLOCK / 64) * DVO_LIST_SEGS_PER_WAVE - 1) / ((DVO_BLOCK / 64) * DVO_LIST_SEGS_PER_WAVE));
    hipLaunchKernelGGL(k_list_build<true>, dim3(lb, 1, nf), dim3(DVO_BLOCK), 0, s, p, segs, 0, segs);
    return hipGetLastError();
}

// ================================================================================================
// the per-pixel pipeline shared by K3 and K5
// ================================================================================================
// Valid-point records are read once per launch: a non-temporal load (`nt`) keeps them from displacing the texel lines and the
// coarser levels' data in the L2 / Infinity Cache.  Which launches do so is a launch parameter (DvoReduceParams::nt_records:
// level 0), tested once per record group by a wave-uniform branch around the group's four loads.
template <bool NT>
__device__ __forceinline__ uint2 dvo_rec_load(const uint2* p)
{
    if (NT) {
        const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
        return make_uint2((unsigned)v, (unsigned)(v >> 32));
    }
    return *p;
}

template <bool NT>
__device__ __forceinline__ unsigned long long dvo_rec_load64(const unsigned long long* p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}

struct PixelOut {
    float r;
    float J[6];
};

// Phase A of a pixel: geometry up to the in-bounds decision and the tap address (no dependent loads).
struct PixelGeo {
    float x, y, z;       // back-projected point (un-warped) as (x/z, y/z, z): x/z, y/z are the normalised pixel coordinates
                         // (variant GN: the warped point as (x'/z', y'/z', 1/z'))
    float uw, vw;        // warped pixel coordinates
    float ax, ay;        // their fractional parts uw - floor(uw), vw - floor(vw) (exact)
    unsigned off;        // offset of tap (y0, x0) in the packed plane (0 when !ok, so the load is harmless)
    bool ok;             // depth != 0 && warped point inside the closed image rectangle
    int n_ok;            // ... how many lanes of the wave have it (dvo_geo only; wave-uniform)
};

struct __attribute__((packed, aligned(4))) u32x2_a4 { uint32_t a, b; };
struct PixelTaps { u32x2_a4 t0, t1; uint32_t own; };   // (y0,x0)(y0,x1) / (y1,x0)(y1,x1) / prev-frame pixel (V2)

// v_rcp_f32 with a scalar instruction on either side.  A VALU instruction issued right behind a transcendental one stalls the
// SIMD for ~16 cycles on gfx950 - measured with the issue-rate probe (tools/ubench/probe_orders.py, profiles/r04_ubench_probe_orders.txt,
// 5 waves per SIMD): one v_rcp_f32 amid simple VALU code costs 13.1 ns, followed by an s_nop 6.2 ns, between two s_nops 4.2 ns
// (3.4 ns in a run of reciprocals); any scalar instruction in the slot does it, a v_nop does not.  The compiler knows no such rule,
// hence the three instructions as one asm block (not volatile: an unused reciprocal still disappears).
#ifndef DVO_RCP_PLAIN
__device__ __forceinline__ float dvo_rcp(float x)
{
    float r;
#if defined(DVO_RCP_AFTER_ONLY)
    asm("v_rcp_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(x));
#elif defined(DVO_RCP_NOP3)
    asm("s_nop 0\n\tv_rcp_f32 %0, %1\n\ts_nop 3" : "=v"(r) : "v"(x));
#else
    asm("s_nop 0\n\tv_rcp_f32 %0, %1\n\ts_nop 0" : "=v"(r) : "v"(x));
#endif
    return r;
}
#else
__device__ __forceinline__ float dvo_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

__device__ __forceinline__ float fast_rcp(float z)
{
    // v_rcp_f32 (1 ulp) + one Newton step; only used where the oracle tolerance is relative 1e-6, never
    // on the coordinate path.
    float r = dvo_rcp(z);
    return fmaf(fmaf(-z, r, 1.0f), r, r);
}

__device__ __forceinline__ void unpack_tap(uint32_t p, float& I, float& gx, float& gy)
{
    I = (float)(p & 0xffu);
    gx = (float)(((int32_t)(p << 12)) >> 20);
    gy = (float)(((int32_t)p) >> 20);
}

// a / b, correctly rounded, from the refined reciprocal rb = v_rcp_f32(b) + one Newton step: one residual correction of the
// product (Markstein's sequence), which is correctly rounded for EVERY numerator when rb is the float nearest to 1 / b - and on
// gfx950 it is, for every one of the 2^23 mantissas of b (decided exactly on the device: tools/ubench/quotient3.hip,
// profiles/r04_ubench_quotient3.txt: 0 exceptions; the raw v_rcp_f32 is not the nearest float for 897 675 of them; repeated in the
// GPU suite through dvo_debug_divide with a == NULL).  The same tool compared the quotient itself with the IEEE division over
// 3.4e10 operand pairs and every mantissa of b x 192 numerators: no mismatch.  DVO_DIV_TWO_STEPS restores the two-correction form.
__device__ __forceinline__ float dvo_div_with_rcp(float a, float b, float rb)
{
    const float q0 = a * rb;
    const float e0 = fmaf(-b, q0, a);
    const float q1 = fmaf(e0, rb, q0);
#ifdef DVO_DIV_TWO_STEPS
    const float e1 = fmaf(-b, q1, a);
    return fmaf(e1, rb, q1);
#else
    return q1;
#endif
}

// SKEW = false: the camera matrix has no skew term (K[0][1] == 0), so the zero products of the reference's matrix
// expressions are skipped: x + (+-0) == x, exact.
// WARPED (variant GN): the point handed to the Jacobian is the WARPED one, (x'/z', y'/z', z').
template <bool SKEW, bool WARPED = false>
__device__ __forceinline__ void dvo_geo(const DvoLevelGeom& g, const DvoLevelCam& cam, const float* __restrict__ T,
                                        float u, float v, float z, bool live, PixelGeo& o)
{
    // `live` = the lane holds a depth-valid pixel (mask = depth != 0, camera_model.py:195); dead lanes compute on
    // whatever they hold and are masked by `ok`.  z = (float)((double)depth * depth_scale) comes from the caller.
    // --- deproject (camera_model.py:195-221) -------------------------------------------------------
    const float xn = SKEW ? (((cam.ia * u) + (cam.ib * v)) + cam.ic) : ((cam.ia * u) + cam.ic);
    const float yn = (cam.id * v) + cam.ie;
    const float x = xn * z;
    const float y = yn * z;
    // --- transform (cpu_robust...:173) and project (camera_model.py:247-252) -------------------------
    const float xp = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
    const float yp = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
    const float zp = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
    const float un = SKEW ? (((cam.fx * xp) + (cam.s * yp)) + (cam.cx * zp)) : ((cam.fx * xp) + (cam.cx * zp));
    const float vn = (cam.fy * yp) + (cam.cy * zp);
    // Correctly rounded quotients: v_rcp_f32 refined by one Newton step (shared by both quotients), then per quotient the
    // product and one residual correction (dvo_div_with_rcp) -- the arithmetic core of an IEEE float32 division without the
    // v_div_scale / v_div_fixup wrappers, which only rescale when the quotient or an intermediate leaves the normal range
    // (|a/b| beyond 2^+-96, denormals) and patch up inf / NaN / 0: for pixel coordinates (|a/b| < 1e5, |b| ~ 1) they are
    // identities.  A non-finite or zero zp gives NaN here instead of +-inf: invalid either way.  Checked bit for bit
    // against `/` in tests (dvo_debug_divide) and by tools/ubench/quotient3.hip.
    const float rz0 = dvo_rcp(zp);
    const float rz = fmaf(fmaf(-zp, rz0, 1.0f), rz0, rz0);
    const float uw = dvo_div_with_rcp(un, zp, rz);
    const float vw = dvo_div_with_rcp(vn, zp, rz);
    // closed-boundary rule: valid <=> 0 <= x <= W-1 and 0 <= y <= H-1 ; NaN compares false
    // Non-negative floats order like their bit patterns and every negative float or NaN has a larger pattern than any
    // bound, so one unsigned compare per coordinate decides 0 <= c <= max.  (-0.0 would be rejected where the float
    // compare accepts it, but the division above cannot return -0.0 for a non-zero divisor.)
    const bool ok = (int)live & (int)(__float_as_uint(uw) <= __float_as_uint(g.wmax_f)) & (int)(__float_as_uint(vw) <= __float_as_uint(g.hmax_f));
    // v_fract_f32 = c - floor(c), exact; floor(c) = c - fract(c), exact as well
    const float ax = __builtin_amdgcn_fractf(uw), ay = __builtin_amdgcn_fractf(vw);
    if (WARPED) { o.x = xp * rz; o.y = yp * rz; o.z = rz; }      // z slot = 1 / z' (refined reciprocal), used as such by dvo_finish<3>
    else { o.x = xn; o.y = yn; o.z = z; }
    o.uw = uw; o.vw = vw; o.ax = ax; o.ay = ay; o.ok = ok;
    // (s_bcnt1 of the compares' own result masks; a ballot of the combined flag would have it materialised in a VGPR first)
    o.n_ok = __builtin_popcountll(__builtin_amdgcn_ballot_w64(__float_as_uint(uw) <= __float_as_uint(g.wmax_f)) &
                                  __builtin_amdgcn_ballot_w64(__float_as_uint(vw) <= __float_as_uint(g.hmax_f)) & __builtin_amdgcn_ballot_w64(live));
    // Tap BYTE offset 4 (y0 pitch + x0), plus g.tap_bias.  Levels with 4 (h + 1) pitch < 2^23 (every level up to 1920 x 1080): formed
    // in float32 with 2^23 added - every integer in [2^23, 2^24) is a float32, so both FMAs are exact and the BIT PATTERN of the sum is
    // 0x4B000000 + the offset: no float -> integer conversion and no left shift (both half rate on gfx950, like every v_cvt / v_bfe /
    // v_lshlrev; v_fma / v_sub / v_lshrrev / v_ashrrev / v_and / v_or are full rate: tools/ubench/probe_orders.py,
    // profiles/r04_ubench_instruction_rates.txt); dvo_load_taps takes the bias off the (wave-uniform) base address.  Larger levels:
    // (and any camera with skew) the offset in float32 (exact: < 2^24), converted and shifted.
    // The fast instantiations (SKEW = false) are launched only for levels that have the bias (level_camera in dvo_abi.hip).
    if constexpr (!SKEW) {
        const float c = fmaf(uw - ax, 4.0f, 8388608.0f);
        o.off = ok ? __float_as_uint(fmaf(vw - ay, g.pitch4_f, c)) : 0x4B000000u;
    } else {
        o.off = (ok ? ((unsigned)fmaf(vw - ay, g.pitch_f, uw - ax)) << 2 : 0u) + g.tap_bias;
    }
}

// index (pixels) of the first tap in the packed plane, from the biased byte offset
__device__ __forceinline__ unsigned dvo_tap_index(const DvoLevelGeom& g, const PixelGeo& q) { return (q.off - g.tap_bias) >> 2; }

__device__ __forceinline__ void dvo_load_taps(const DvoLevelGeom& g, const uint32_t* __restrict__ cur,
                                              const PixelGeo& q, PixelTaps& t)
{
    // 32-bit byte offset from two wave-uniform row bases: the loads take the scalar base + vector offset form, no
    // 64-bit address arithmetic per lane ((h+1)*pitch < 2^24 is checked at dvo_create, so the offset fits)
    unsigned boff = q.off;                // byte offset + g.tap_bias (dvo_geo)
    asm("" : "+v"(boff));                 // keep the offset 32-bit up to the load (else the zero-extension is hoisted)
    const char* base0 = reinterpret_cast<const char*>(cur) - g.tap_bias;
    const char* base1 = base0 + 4 * (size_t)g.pitch;
    t.t0 = *reinterpret_cast<const u32x2_a4*>(base0 + boff);
    t.t1 = *reinterpret_cast<const u32x2_a4*>(base1 + boff);
}

// Variant V3 (the reference's numba.cuda kernel, cuda/residuals_kernel.py:46-109): phase A with the arithmetic types
// numba infers for that source — `tx - cx` is int32 - float32 = float64, so back-projection, rigid transform, the six
// Jacobian terms and the projection are float64 expressions (rounded to float32 where the kernel stores / passes
// float32); z = depth * depth_scale and the clamped central-difference gradient of I1 are float32; strict bilinear
// bounds (x1 < W, y1 < H, :30-31).  The Jacobian needs no tap of I2, so it is finished here.
__device__ __forceinline__ void dvo_geo_v3(const DvoLevelGeom& g, const DvoLevelCam& cam, double depth_scale,
                                           const float* __restrict__ T, const uint8_t* __restrict__ gray_prev,
                                           int ux, int vy, unsigned d, PixelGeo& o, float* __restrict__ J)
{
    o.ok = false;
    o.off = g.tap_bias;
    o.x = o.y = o.z = 0.0f;
    o.uw = o.vw = __builtin_nanf("");
    o.ax = o.ay = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) J[k] = 0.0f;
    if (d == 0u) return;                                                          // :60-65
    const double z = (double)((float)d * (float)depth_scale);                     // :67 uint16 * float32 -> float32
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const double x = ((double)ux - cx) * z / fx;                                  // :69-70
    const double y = ((double)vy - cy) * z / fy;
    const double x1 = (((double)T[0] * x + (double)T[1] * y) + (double)T[2] * z) + (double)T[3];     // :73-75
    const double y1 = (((double)T[4] * x + (double)T[5] * y) + (double)T[6] * z) + (double)T[7];
    const double z1 = (((double)T[8] * x + (double)T[9] * y) + (double)T[10] * z) + (double)T[11];
    const uint8_t* row = gray_prev + (size_t)vy * g.w;
    const double gx = (double)(0.5f * ((float)row[min(ux + 1, g.w - 1)] - (float)row[max(ux - 1, 0)]));            // :8-19
    const double gy = (double)(0.5f * ((float)gray_prev[(size_t)min(vy + 1, g.h - 1) * g.w + ux] -
                                       (float)gray_prev[(size_t)max(vy - 1, 0) * g.w + ux]));
    const double z2 = z1 * z1;
    J[0] = (float)(fx * gx / z1);                                                 // :87-92
    J[1] = (float)(fy * gy / z1);
    J[2] = (float)((-fx) * gx * x1 / z2 - fy * gy * y1 / z2);
    J[3] = (float)((-fx) * gx * x1 * y1 / z2 - fy * gy * ((y1 * y1) / z2 + 1.0));
    J[4] = (float)(fx * gx * ((x1 * x1) / z2 + 1.0) + fy * gy * x1 * y1 / z2);
    J[5] = (float)((-fx) * gx * y1 / z1 + fy * gy * x1 / z1);
    const float wx = (float)(fx * x1 / z1 + cx);                                  // :95-96, float32 argument of :99
    const float wy = (float)(fy * y1 / z1 + cy);
    const float x0 = floorf(wx), y0 = floorf(wy);
    const bool ok = (x0 >= 0.0f) && (y0 >= 0.0f) && (x0 + 1.0f < (float)g.w) && (y0 + 1.0f < (float)g.h);
    o.uw = wx; o.vw = wy; o.ax = wx - x0; o.ay = wy - y0; o.ok = ok;
    o.off = (ok ? ((unsigned)fmaf(y0, (float)g.pitch, x0)) << 2 : 0u) + g.tap_bias;
}

// Phase B: bilinear photometric lookup, residual, 1x6 Jacobian row.
// VARIANT 0 (V1): gradient of I2 sampled at the warped point; 1 (V2): Sobel gradient of I1 at the pixel;
// 2 (V3): Jacobian row already finished by dvo_geo_v3, only the residual is computed here;
// 3 (GN, build-defined): V1's sampling with the Sobel response normalised (1/8: a unit derivative) and J_w at the warped
// point (dvo_geo<.., true>), i.e. the true Gauss-Newton row of the forward-compositional photometric error.
// The taps as float32 values, in the order (y0,x0) (y0,x1) (y1,x0) (y1,x1); own = gradient of I1 at the pixel (V2).
struct PixelTapsF {
    float I[4], gx[4], gy[4];
    float own_gx, own_gy;
};

template <int VARIANT, bool NEED_J, typename I1>
__device__ __forceinline__ void dvo_finish_f(const DvoLevelCam& cam, const PixelGeo& q, const PixelTapsF& t,
                                             I1 i1, PixelOut& o)                 // i1: the previous frame's intensity, integer or already a float
{
    // --- bilinear weights (cpu_robust...:202-254) ------------------------------------------------------
    const float ax = q.ax, ay = q.ay;                      // uw - floor(uw), vw - floor(vw): exact
    const float bx = 1.0f - ax, by = 1.0f - ay;            // == (x0+1) - x up to one rounding when x < 1
    const float w00 = bx * by, w01 = bx * ay, w10 = ax * by, w11 = ax * ay;
    const float i2 = fmaf(w11, t.I[3], fmaf(w10, t.I[1], fmaf(w01, t.I[2], w00 * t.I[0])));
    o.r = i2 - (float)i1;                                                       // :191-194
    if (!NEED_J || VARIANT == 2) return;        // V3: the Jacobian row was finished in phase A (dvo_geo_v3)
    float gxw, gyw;
    if (VARIANT == 0 || VARIANT == 3) {
        gxw = fmaf(w11, t.gx[3], fmaf(w10, t.gx[1], fmaf(w01, t.gx[2], w00 * t.gx[0])));    // :123-128
        gyw = fmaf(w11, t.gy[3], fmaf(w10, t.gy[1], fmaf(w01, t.gy[2], w00 * t.gy[0])));
    } else {
        gxw = t.own_gx;                                                         // :69-75 (Sobel of I1 at the pixel)
        gyw = t.own_gy;
    }
    // --- J_w at the UN-warped point (utils/jacobian.py:39-42, quirk Q2) ------------------------------
    // x/z and y/z are the normalised coordinates the point was built from (x = xn * z), and 1/z is v_rcp_f32 (1 ulp):
    // both within 2e-7 relative of the reference's float32 quotients, far inside the 1e-5 bound on the increments.
    const float iz = (VARIANT == 3) ? q.z : dvo_rcp(q.z);
    const float a = q.x, b = q.y;
    const float jfx = (VARIANT == 3) ? cam.fx * 0.125f : cam.fx, jfy = (VARIANT == 3) ? cam.fy * 0.125f : cam.fy;   // wave-uniform
    const float p = jfx * gxw, qq = jfy * gyw;
    // J = gx * Jw[0,:] + gy * Jw[1,:]   (_fill_jacobian, cpu_robust...:79-90), with the common factor m = p a + q b taken out:
    //   -(p ab + q (1 + b^2)) = -(b m + q),   p (1 + a^2) + q ab = a m + p        (11 instructions instead of 16)
    const float m = fmaf(p, a, qq * b);
    o.J[0] = p * iz;
    o.J[1] = qq * iz;
    o.J[2] = -(m * iz);
    o.J[3] = -fmaf(b, m, qq);
    o.J[4] = fmaf(a, m, p);
    o.J[5] = fmaf(qq, a, -(p * b));
}

template <int VARIANT, bool NEED_J, typename I1>
__device__ __forceinline__ void dvo_finish(const DvoLevelCam& cam, const PixelGeo& q, const PixelTaps& t,
                                           I1 i1, PixelOut& o)
{
    PixelTapsF f;
    unpack_tap(t.t0.a, f.I[0], f.gx[0], f.gy[0]);
    unpack_tap(t.t0.b, f.I[1], f.gx[1], f.gy[1]);
    unpack_tap(t.t1.a, f.I[2], f.gx[2], f.gy[2]);
    unpack_tap(t.t1.b, f.I[3], f.gx[3], f.gy[3]);
    f.own_gx = f.own_gy = 0.0f;
    if (NEED_J && VARIANT == 1) { float Ip; unpack_tap(t.own, Ip, f.own_gx, f.own_gy); }
    dvo_finish_f<VARIANT, NEED_J, I1>(cam, q, f, i1, o);
}

// Test hook: out[i] = the kernels' quotient a[i]/b[i]; ref[i] = the compiler's IEEE division.  a == NULL: out[i] = the refined
// reciprocal of b[i] itself (what dvo_geo shares between its two quotients), ref[i] = 1 / b[i] by the IEEE division.
__global__ void k_debug_divide(const float* a, const float* b, float* out, float* ref, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rz0 = dvo_rcp(b[i]);
    const float rz = fmaf(fmaf(-b[i], rz0, 1.0f), rz0, rz0);
    if (a == nullptr) { out[i] = rz; ref[i] = 1.0f / b[i]; return; }
    out[i] = dvo_div_with_rcp(a[i], b[i], rz);
    ref[i] = a[i] / b[i];
}

hipError_t dvo_launch_debug_divide(const float* a, const float* b, float* out, float* ref, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_debug_divide, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, ref, n);
    return hipGetLastError();
}

// ================================================================================================
// K3: fused residual/Jacobian/weight + normal-equation reduction
// ================================================================================================
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Workgroup (256 threads) reduction of the DVO_NACC per-lane float32 sums into out[0..31] (float64; global or LDS).
// The sums are transposed through LDS instead of 30 x 6 cross-lane butterflies: every lane stores its sums
// ([term][thread], padded so stores and loads are bank-conflict free), then 8 lanes per term each add 32 of the 256
// values in float64 (fixed order => reproducible), and a 3-step shuffle joins the 8 segments.  Two rounds of 15
// terms keep the LDS footprint at 16.5 KB per workgroup.
// DVO_CLOCK_PROBE (diagnostic build, tools/clock_probe.py; never in the product library): the two reserved slots of the
// record carry, summed over the workgroups of a pair, the shader-clock ticks (s_memtime) and the constant 100 MHz ticks
// (s_memrealtime) each workgroup's first wave spent in the kernel body: their ratio is the clock the chip really holds.
#define DVO_REDUCE_TBUF_FLOATS (16 * 8 * 33)             // LDS floats block_reduce_terms_buf needs (16.5 KB)
template <int NTERMS = DVO_NACC>
__device__ __forceinline__ void block_reduce_terms_buf(const float* acc, double* out, float* tbuf /* LDS [DVO_REDUCE_TBUF_FLOATS] */,
                                                       unsigned long long t0 = 0, unsigned long long r0 = 0)
{
    static_assert(NTERMS == DVO_NACC || NTERMS == DVO_STRIDE || NTERMS == 6, "30 sums, all 32 slots of the record, or 6 extra sums");
    constexpr int ROW = 8 * 33;                          // 8 segments of 32 values + 1 pad each
    constexpr int HALF = NTERMS / 2;
    const int t = threadIdx.x;
    const int wpos = (t >> 5) * 33 + (t & 31);
    const int term = t >> 3, segi = t & 7;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (round) __syncthreads();                      // previous round's loads are done
#pragma unroll
        for (int i = 0; i < HALF; ++i) tbuf[i * ROW + wpos] = acc[round * HALF + i];
        __syncthreads();
        double sgm = 0.0;
        if (t < HALF * 8) {
            const float* src = tbuf + term * ROW + segi * 33;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) sgm += (double)src[k];
        }
        sgm += __shfl_xor(sgm, 1, 64);
        sgm += __shfl_xor(sgm, 2, 64);
        sgm += __shfl_xor(sgm, 4, 64);
        if (t < HALF * 8 && segi == 0) out[round * HALF + term] = sgm;
    }
    if (NTERMS == DVO_STRIDE || NTERMS == 6) return;
#ifdef DVO_CLOCK_PROBE
    if (t == 30) out[t] = (double)(__builtin_amdgcn_s_memtime() - t0);
    if (t == 31) out[t] = (double)(__builtin_amdgcn_s_memrealtime() - r0);
#else
    if (t >= 30 && t < DVO_STRIDE) out[t] = 0.0;
#endif
}

template <int NTERMS = DVO_NACC>
__device__ __forceinline__ void block_reduce_terms(const float* acc, double* out, unsigned long long t0 = 0, unsigned long long r0 = 0)
{
    __shared__ float tbuf[DVO_REDUCE_TBUF_FLOATS];
    block_reduce_terms_buf<NTERMS>(acc, out, tbuf, t0, r0);
}

#ifndef DVO_REDUCE_WAVES_PER_SIMD
#define DVO_REDUCE_WAVES_PER_SIMD 4
#endif
// Student-t IRLS of one iteration (weighter/t_weighter.py:24-32) from the per-workgroup sums of the two scale passes:
// lambda1 = 1 / S1; converged (|lambda1 - lambda0| < 1e-3) keeps it, otherwise lambda2 = 1 / S2.  with_second = false: the
// lambda the second pass itself runs under (lambda1).  S is summed in exactly the order sum_partials_block + k_solve use
// (eight interleaved running sums, then those in order: lanes 0-7 for S1, 8-15 for S2), so lambda is bit-identical to the
// one the control kernel derives; the workgroup shares the result.
__device__ __forceinline__ double student_t_lambda(const DvoReduceParams& p, int pair, double lambda0, bool with_second)
{
    __shared__ double sg[16];
    __shared__ double lam_sh;
    const int t = threadIdx.x;
    if (t < (with_second ? 16 : 8)) {
        const double* x = ((t < 8) ? p.scale1 : p.scale2) + (size_t)pair * p.blocks_per_pair;
        double s = 0.0;
        for (int b = t & 7; b < p.blocks_per_pair; b += 8) s += x[b];
        sg[t] = s;
    }
    __syncthreads();
    if (t == 0) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 8; ++k) s1 += sg[k];
        double lam = 1.0 / s1;
        if (with_second && !(fabs(lam - lambda0) < 1e-3)) {
            for (int k = 0; k < 8; ++k) s2 += sg[8 + k];
            lam = 1.0 / s2;
        }
        lam_sh = lam;
    }
    __syncthreads();
    return lam_sh;
}

template <int VARIANT, int WMODE, bool SCALE_PASS, bool SKEW>
__global__ __launch_bounds__(DVO_BLOCK, (VARIANT == 2) ? 2 : DVO_REDUCE_WAVES_PER_SIMD) void k_reduce(const DvoReduceParams p)
{
    constexpr int NACC = SCALE_PASS ? 2 : (WMODE == 5 ? DVO_STRIDE : DVO_NACC);       // scale pass: sum of the weighted r^2, number of residuals
    // WMODE 4 / 5 (internal weight modes 4 / 5 of dvo_launch_reduce): the reference's Student-t evaluated in ONE pass per
    // iteration, see student_t_moment_combine / student_t_predicted_combine
    // XCD-aware block order (speed only): workgroups are dealt round-robin to the 8 XCDs, so give XCD k the
    // k-th contiguous eighth of the logical grid; the workgroups of one frame pair (which share tap rows of the
    // packed plane) then run behind one L2.  Bijective for any grid size.
    int lb = blockIdx.x;
    if (p.xcd_remap) {
        const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = lb & 7;
        lb = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lb >> 3);
    }
    const int pair = lb / p.blocks_per_pair;
    const int blk = lb - pair * p.blocks_per_pair;
    if (p.active != nullptr && p.active[pair] == 0) return;        // block-uniform
    if constexpr (SCALE_PASS) {
        if (p.irls_done != nullptr && p.irls_done[pair] != 0) return;      // this pair's IRLS has converged: nothing to sum
    }
#ifdef DVO_CLOCK_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#else
    const unsigned long long probe_t0 = 0, probe_r0 = 0;
#endif

    const uint8_t* prev = p.slots + (size_t)p.prev_slot[pair] * p.slot_stride;
    const uint8_t* curs = p.slots + (size_t)p.cur_slot[pair] * p.slot_stride;
    const uint8_t* gray_prev = prev + p.g.off_gray;
    const uint16_t* depth_prev = reinterpret_cast<const uint16_t*>(prev + p.g.off_depth);
    const uint32_t* packed_prev = reinterpret_cast<const uint32_t*>(prev + p.g.off_packed);
    const uint32_t* packed_cur = reinterpret_cast<const uint32_t*>(curs + p.g.off_packed);

    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = p.T[pair * 12 + i];
    const DvoLevelCam& cam = p.cam;
    float wpar = 0.0f;
    if (WMODE != 0) {
        double lam = p.wparam[pair];
        if constexpr (WMODE == 1 && !SCALE_PASS) {
            if (p.scale1 != nullptr) lam = student_t_lambda(p, pair, lam, true);
        }
        wpar = (float)lam;
    }
    // WMODE 4 / 5: the first scale pass runs under the CONSTANT lambda0 (p.lambda0); wparam[pair] (wpar) is the GUESSED lambda the
    // weights of this iteration are evaluated under (4: 0 at the first level, a quarter of the level before's last lambda at the
    // first iteration of a level, then the last / the linearly extrapolated lambda; 5: the geometric extrapolation of the last three)
    const float lam0 = (WMODE == 4 || WMODE == 5) ? p.lambda0 : 0.0f;

    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
    // WMODE 4 (Student-t of the reference in ONE pass, "moment form", see k_solve): a second set of sums, r^2-weighted
    float acc2[(WMODE == 4) ? DVO_NACC : 1];
#pragma unroll
    for (int i = 0; i < ((WMODE == 4) ? DVO_NACC : 1); ++i) acc2[i] = 0.0f;
    // WMODE 5: b1 = sum w^2 r^3 J (see below); lands in slots 21..26 of the second record, where WMODE 4 puts its b1
    float acc3[(WMODE == 5) ? 6 : 1];
#pragma unroll
    for (int i = 0; i < ((WMODE == 5) ? 6 : 1); ++i) acc3[i] = 0.0f;

    // Work items are the records of the previous frame's valid-point list (depth != 0, raster order, built once at
    // ingest): no validity test, no pixel-index arithmetic and no empty lanes inside the Gauss-Newton loop.  A wave
    // owns 256 consecutive records per group and lane L takes records L, L+64, L+128, L+192 of them, so every load
    // instruction of the wave (the 8-byte records and, because neighbouring records are neighbouring pixels, the two
    // 8-byte tap rows of the packed plane) covers whole cache lines.
    // (the wave index through readfirstlane: the compiler then knows that the loop's bounds and the prefetch condition are
    // wave-uniform - scalar branches, no exec-mask bookkeeping and no register copies to keep "inactive lanes'" records)
    const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const DvoPointRec* list = reinterpret_cast<const DvoPointRec*>(prev + p.g.off_list);
    const int count = p.list_count[p.prev_slot[pair] * DVO_LIST_LEVELS + p.level];
    // The G = ceil(count / 1024) record groups of the pair are split into contiguous, count-balanced ranges: workgroup blk takes
    // groups [blk G / n, (blk + 1) G / n).  Balanced whatever the list length (the grid is sized for the pixel count, the list is
    // ~30 % shorter: a split by pixel index leaves the last workgroups empty), and contiguous, so that neighbouring workgroups
    // do not both fetch the tap rows a round-robin deal of single groups makes them share (HBM traffic 886 -> ~760 MB per launch).
    const int ngroups = (count + DVO_PX_PER_BLOCK - 1) / DVO_PX_PER_BLOCK;
    const int g_lo = (blk * ngroups) / p.blocks_per_pair, g_hi = ((blk + 1) * ngroups) / p.blocks_per_pair;
    const int wave_off = wave_ * (64 * DVO_PX_PER_THREAD);                            // wave-uniform
    const int rec0 = g_lo * DVO_PX_PER_BLOCK + wave_off;
    // The records of a group are requested during the group before it (behind that group's tap gathers, see the loop), the first ones here.
    // Held as 64-bit values: the loop-carried registers are the loads' own destination pairs.
    unsigned long long rn[DVO_PX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rn[j] = 0ull;
    const bool nt_records = p.nt_records != 0;                            // launch-uniform
    if (g_lo < g_hi && rec0 < count) {
        if (nt_records) {
#pragma unroll
            for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rn[j] = dvo_rec_load64<true>(reinterpret_cast<const unsigned long long*>(list + rec0 + j * 64 + lane_));
        } else {
#pragma unroll
            for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rn[j] = dvo_rec_load64<false>(reinterpret_cast<const unsigned long long*>(list + rec0 + j * 64 + lane_));
        }
    }
    // the number of valid points is counted per WAVE from the in-bounds lane masks (s_bcnt1 on the compare's result: scalar unit) instead of
    // one v_add per lane and point; lane 0 hands it to the reduction after the loop
    int n_ok = 0;
    for (int grp = g_lo; grp < g_hi; ++grp) {
        const int seg = grp * DVO_PX_PER_BLOCK + wave_off;
        if (seg >= count) break;                                          // wave-uniform
        // phase A for all pixels of the lane, then all tap loads back to back (memory-level parallelism: the
        // dependent gathers of 4 pixels are in flight together), then the request for the next group's records, then phase B.
        // Phase A takes everything it needs out of the records (the previous frame's intensity as a float), so the next
        // group's records are loaded over them: no current / next pair of register sets to rotate (that cost 8 v_mov_b64 per trip;
        // 4 are left, the allocator's own split of the range).
        PixelGeo geo[DVO_PX_PER_THREAD];
        PixelTaps taps[DVO_PX_PER_THREAD];
        float i1f[DVO_PX_PER_THREAD];
        float jv3[(VARIANT == 2) ? DVO_PX_PER_THREAD : 1][6];
#pragma unroll
        for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
            // V1 / V2: records past `count` are the list's NaN padding (k_list_build) and fail the bounds test by themselves
            const bool live = (VARIANT == 2) ? ((seg + j * 64 + lane_) < count) : true;
            const unsigned pix = (unsigned)(rn[j] >> 32);
            i1f[j] = (float)(pix >> 24);
            const unsigned xi = pix & 0xfffu, yi = (pix >> 12) & 0xfffu;
            if constexpr (VARIANT == 2) {
                // the numba.cuda kernel converts depth in float32 (dvo_geo_v3): it starts from the raw depth again
                const unsigned d = live ? (unsigned)depth_prev[yi * (unsigned)p.g.w + xi] : 0u;
                dvo_geo_v3(p.g, cam, p.depth_scale, T, gray_prev, (int)xi, (int)yi, d, geo[j], jv3[j]);
            } else {
                dvo_geo<SKEW, VARIANT == 3>(p.g, cam, T, (float)xi, (float)yi, __uint_as_float((unsigned)rn[j]), live, geo[j]);
            }
            dvo_load_taps(p.g, packed_cur, geo[j], taps[j]);
            if (VARIANT == 1) taps[j].own = geo[j].ok ? packed_prev[yi * (unsigned)p.g.pitch + xi] : 0u;
        }
        {
            // the records of the next group: behind the tap gathers in the memory queue (loads return in order, so phase B waits
            // for the taps only), their latency covered by phase B
            // UNCONDITIONAL (after the wave's last group: that group once more, an address that is valid by construction and whose
            // data is not used): a load under a condition keeps the old registers alive for the lanes / trips that do not load, which
            // costs the copies this arrangement is there to remove
            __builtin_amdgcn_sched_barrier(0);          // (not hoisted into phase A, where the current records are still live)
            const int segn0 = seg + DVO_PX_PER_BLOCK;
            const int segn = (grp + 1 < g_hi && segn0 < count) ? segn0 : seg;     // wave-uniform  (reads past `count` stay inside DVO_LIST_SLACK)
            if (nt_records) {
#pragma unroll
                for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rn[j] = dvo_rec_load64<true>(reinterpret_cast<const unsigned long long*>(list + segn + j * 64 + lane_));
            } else {
#pragma unroll
                for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rn[j] = dvo_rec_load64<false>(reinterpret_cast<const unsigned long long*>(list + segn + j * 64 + lane_));
            }
        }
        float rkeep[SCALE_PASS ? DVO_PX_PER_THREAD : 1];
        if constexpr (SCALE_PASS) {
#pragma unroll
            for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rkeep[j] = __builtin_nanf("");
        }
#pragma unroll
        for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
            if constexpr (VARIANT == 2) n_ok += __builtin_popcountll(__builtin_amdgcn_ballot_w64(geo[j].ok));     // (dvo_geo_v3 does not count)
            else n_ok += geo[j].n_ok;
            if (geo[j].ok) {                                  // depth != 0 (camera_model.py:195) and in bounds
                PixelOut o;
                dvo_finish<VARIANT, !SCALE_PASS>(cam, geo[j], taps[j], i1f[j], o);
                if constexpr (VARIANT == 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) o.J[k] = jv3[j][k];
                }
                const float r = o.r;
                const float r2 = r * r;
                if constexpr (SCALE_PASS) {
                    // r^2 * (dof+1)/(dof + r^2*lambda), dof = 5 (weighter/t_weighter.py:43-45)
                    acc[0] += r2 * (6.0f * fast_rcp(fmaf(r2, wpar, 5.0f)));
                    rkeep[j] = r;
                } else {
                    float wt = 1.0f;
                    if (WMODE == 1 || WMODE == 4 || WMODE == 5) wt = 6.0f * fast_rcp(fmaf(r2, wpar, 5.0f));   // t_weighter.py:34 (4 / 5: under the guessed lambda)
                    if (WMODE == 2) { const float ar = fabsf(r); wt = (ar <= wpar) ? 1.0f : wpar * fast_rcp(ar); }
                    float wj[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) wj[k] = (WMODE == 0) ? o.J[k] : wt * o.J[k];
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = a; b < 6; ++b) { acc[q] = fmaf(wj[a], o.J[b], acc[q]); ++q; }
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[21 + k] = fmaf(wj[k], r, acc[21 + k]);
                    if (WMODE != 5) acc[27] = fmaf((WMODE == 0 || WMODE == 4) ? r : wt * r, r, acc[27]);   // (4: the plain moment sum r^2)
                    if (WMODE == 1 || WMODE == 2 || WMODE == 5) acc[29] += r2;             // WMODE 0: copied from acc[27] after the loop
                    if constexpr (WMODE == 5) {
                        // beside the system weighted with the predicted lambda: the first scale pass under the constant lambda0
                        // (exact per point), the moments r^4, r^6 from which the control step rebuilds the second pass, and
                        // b1 = sum w^2 r^3 J, the derivative of J^T W r with respect to lambda (x -6): the control step corrects the
                        // right-hand side to first order in (lambda2 - predicted lambda)
                        acc[27] += r2 * (6.0f * fast_rcp(fmaf(r2, lam0, 5.0f)));
                        const float r4 = r2 * r2;
                        acc[30] += r4;
                        acc[31] = fmaf(r4, r2, acc[31]);
                        const float u = (wt * r2) * r;
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc3[k] = fmaf(wj[k], u, acc3[k]);
                    }
                    if constexpr (WMODE == 4) {
                        // first scale pass under the constant lambda0 (t_weighter.py:23, :43-45), exact per point
                        acc[29] += r2 * (6.0f * fast_rcp(fmaf(r2, lam0, 5.0f)));
                        // derivatives with respect to lambda (x -6) of the weighted sums above: H1 = sum w^2 r^2 J J^T,
                        // b1 = sum w^2 r^3 J; and the even moments r^4, r^6, r^8
                        const float wr = wt * r;
                        float rj[6];
#pragma unroll
                        for (int k = 0; k < 6; ++k) rj[k] = wr * o.J[k];
                        int q2 = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a)
#pragma unroll
                            for (int b = a; b < 6; ++b) { acc2[q2] = fmaf(rj[a], rj[b], acc2[q2]); ++q2; }
                        const float wr2 = wt * r2;
#pragma unroll
                        for (int k = 0; k < 6; ++k) acc2[21 + k] = fmaf(rj[k], wr2, acc2[21 + k]);
                        const float r4 = r2 * r2;
                        acc2[27] += r4;
                        acc2[28] = fmaf(r4, r2, acc2[28]);
                        acc2[29] = fmaf(r4, r4, acc2[29]);
                    }
                }
            }
        }
        if constexpr (SCALE_PASS) {
            // the residuals do not change between the scale passes of one pose: keep them for the next pass
            if (p.rcache != nullptr) {
                float* rc_out = p.rcache + (size_t)pair * p.g.npix + seg + lane_;
#pragma unroll
                for (int j = 0; j < DVO_PX_PER_THREAD; ++j)
                    if (seg + j * 64 + lane_ < count) rc_out[j * 64] = rkeep[j];
            }
        }
    }

    acc[SCALE_PASS ? 1 : 28] = (lane_ == 0) ? (float)n_ok : 0.0f;               // (a wave counts at most 2^24 points: exact)
    if constexpr (!SCALE_PASS && WMODE == 0) acc[NACC - 1] = acc[NACC - 3];     // sum r^2 == sum w r^2 when w == 1
    // Workgroup reduction.  float32 only inside one lane; everything across lanes / waves / workgroups is float64.
    // The 30 per-lane sums are transposed through LDS instead of 30 x 6 cross-lane butterflies: every lane stores
    // its sums ([term][thread], padded so stores and loads are bank-conflict free), then 8 lanes per term each add
    // 32 of the 256 values in float64 (fixed order => reproducible), and a 3-step shuffle joins the 8 segments.
    // Two rounds of 15 terms keep the LDS footprint at 15.5 KB per workgroup.
    if constexpr (SCALE_PASS) {
        __shared__ double lds1[2][DVO_BLOCK / 64];
        const double sw = wave_sum((double)acc[0]), cw = wave_sum((double)acc[1]);
        if ((threadIdx.x & 63) == 0) { lds1[0][threadIdx.x >> 6] = sw; lds1[1][threadIdx.x >> 6] = cw; }
        __syncthreads();
        if (p.scale_out != nullptr) {
            if (threadIdx.x == 0) p.scale_out[(size_t)pair * p.blocks_per_pair + blk] = (lds1[0][0] + lds1[0][1]) + (lds1[0][2] + lds1[0][3]);
        } else if (threadIdx.x < DVO_STRIDE) {
            double tot = 0.0;
            if (threadIdx.x < 2) tot = (lds1[threadIdx.x][0] + lds1[threadIdx.x][1]) + (lds1[threadIdx.x][2] + lds1[threadIdx.x][3]);
            p.partials[((size_t)pair * p.blocks_per_pair + blk) * DVO_STRIDE + threadIdx.x] = tot;
        }
    } else {
        double* out = p.partials + ((size_t)pair * p.blocks_per_pair + blk) * DVO_STRIDE;
        __shared__ float tbuf[DVO_REDUCE_TBUF_FLOATS];     // one transpose buffer for every reduction of the launch
        block_reduce_terms_buf<NACC>(acc, out, tbuf, probe_t0, probe_r0);
        if constexpr (WMODE == 4) {
            __syncthreads();                               // the transpose buffer is reused
            block_reduce_terms_buf(acc2, p.partials2 + ((size_t)pair * p.blocks_per_pair + blk) * DVO_STRIDE, tbuf);
        }
        if constexpr (WMODE == 5) {
            __syncthreads();
            block_reduce_terms_buf<6>(acc3, p.partials2 + ((size_t)pair * p.blocks_per_pair + blk) * DVO_STRIDE + 21, tbuf);
        }
    }
}

// Second (and any later) Student-t scale pass of one pose: the same sum as k_reduce<..., SCALE_PASS> over the residuals
// the first pass kept, with the same record->lane map, the same per-lane order and the same cross-lane combination, so
// the result is bit-identical to recomputing the residuals.
__global__ __launch_bounds__(DVO_BLOCK) void k_scale_cached(const DvoReduceParams p)
{
    const int lb = blockIdx.x;
    const int pair = lb / p.blocks_per_pair;
    const int blk = lb - pair * p.blocks_per_pair;
    if (p.active != nullptr && p.active[pair] == 0) return;
    if (p.irls_done != nullptr && p.irls_done[pair] != 0) return;
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int count = p.list_count[p.prev_slot[pair] * DVO_LIST_LEVELS + p.level];
    const int ngroups = (count + DVO_PX_PER_BLOCK - 1) / DVO_PX_PER_BLOCK;           // the same split as k_reduce
    const int g_lo = (blk * ngroups) / p.blocks_per_pair, g_hi = ((blk + 1) * ngroups) / p.blocks_per_pair;
    const int wave_off = wave_ * (64 * DVO_PX_PER_THREAD);
    const float wpar = (float)((p.scale1 != nullptr) ? student_t_lambda(p, pair, 0.0, false) : p.wparam[pair]);
    const float* rc_in = p.rcache + (size_t)pair * p.g.npix;
    float acc = 0.0f, cnt = 0.0f;
    for (int grp = g_lo; grp < g_hi; ++grp) {
        const int seg = grp * DVO_PX_PER_BLOCK + wave_off;
        if (seg >= count) break;
        float r[DVO_PX_PER_THREAD];
#pragma unroll
        for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
            const int idx = seg + j * 64 + lane_;
            r[j] = (idx < count) ? rc_in[idx] : __builtin_nanf("");
        }
#pragma unroll
        for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
            if (r[j] == r[j]) {
                const float r2 = r[j] * r[j];
                acc += r2 * (6.0f * fast_rcp(fmaf(r2, wpar, 5.0f)));
                cnt += 1.0f;
            }
        }
    }
    __shared__ double lds1[2][DVO_BLOCK / 64];
    const double sw = wave_sum((double)acc), cw = wave_sum((double)cnt);
    if ((threadIdx.x & 63) == 0) { lds1[0][threadIdx.x >> 6] = sw; lds1[1][threadIdx.x >> 6] = cw; }
    __syncthreads();
    if (p.scale_out != nullptr) {
        if (threadIdx.x == 0) p.scale_out[(size_t)pair * p.blocks_per_pair + blk] = (lds1[0][0] + lds1[0][1]) + (lds1[0][2] + lds1[0][3]);
    } else if (threadIdx.x < DVO_STRIDE) {
        double tot = 0.0;
        if (threadIdx.x < 2) tot = (lds1[threadIdx.x][0] + lds1[threadIdx.x][1]) + (lds1[threadIdx.x][2] + lds1[threadIdx.x][3]);
        p.partials[((size_t)pair * p.blocks_per_pair + blk) * DVO_STRIDE + threadIdx.x] = tot;
    }
}

hipError_t dvo_launch_scale_cached(const DvoReduceParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_scale_cached, dim3((unsigned)(p.n_pairs * p.blocks_per_pair)), dim3(DVO_BLOCK), 0, s, p);
    return hipGetLastError();
}

// Which instantiations exist (the library's kernel inventory is deliberate, see include/dvo_hip.h "build"):
//   * the GENERAL kernels (SKEW = true: every term of the reference's matrix expressions, exact for any camera, and for a
//     camera without skew bit-identical to the fast ones since x + (+-0) == x) for every variant and weighting;
//   * FAST kernels (SKEW = false: the zero-skew terms dropped, 4 instructions per point) for the paths that are measured:
//     every weighting of the reference default V1 (incl. its one-pass Student-t forms 4 / 5) and the unweighted V2 / GN;
//   * the one-pass Student-t forms (weight modes 4 / 5) for V1 only: the other variants run the passes as written.
template <int VARIANT, bool SKEW>
static hipError_t launch_reduce_v(const DvoReduceParams& p, int weight_mode, int scale_pass, hipStream_t s)
{
    dim3 grid((unsigned)(p.n_pairs * p.blocks_per_pair)), block(DVO_BLOCK);
    if (scale_pass) {
        hipLaunchKernelGGL((k_reduce<VARIANT, 1, true, SKEW>), grid, block, 0, s, p);
    } else if (weight_mode == 0) {
        hipLaunchKernelGGL((k_reduce<VARIANT, 0, false, SKEW>), grid, block, 0, s, p);
    } else if (weight_mode == 1 || weight_mode == 3) {          // both Student-t forms weight with 6 / (5 + r^2 lambda)
        hipLaunchKernelGGL((k_reduce<VARIANT, 1, false, SKEW>), grid, block, 0, s, p);
    } else if (weight_mode == 4) {                              // internal: the reference's Student-t in one pass (moment form)
        if constexpr (VARIANT != 0) return hipErrorInvalidValue;
        else hipLaunchKernelGGL((k_reduce<VARIANT, 4, false, SKEW>), grid, block, 0, s, p);
    } else if (weight_mode == 5) {                              // internal: ... weighted with the predicted lambda
        if constexpr (VARIANT != 0) return hipErrorInvalidValue;
        else hipLaunchKernelGGL((k_reduce<VARIANT, 5, false, SKEW>), grid, block, 0, s, p);
    } else {
        hipLaunchKernelGGL((k_reduce<VARIANT, 2, false, SKEW>), grid, block, 0, s, p);
    }
    return hipGetLastError();
}

hipError_t dvo_launch_reduce(const DvoReduceParams& p, int variant, int weight_mode, int scale_pass, hipStream_t s)
{
    if (variant == 2) return launch_reduce_v<2, true>(p, weight_mode, scale_pass, s);          // V3: float64 geometry of its own (SKEW unused)
    const bool fast = p.cam.skew0 != 0;
    if (variant == 0) return fast ? launch_reduce_v<0, false>(p, weight_mode, scale_pass, s) : launch_reduce_v<0, true>(p, weight_mode, scale_pass, s);
    const bool plain = (weight_mode == 0) && !scale_pass;
    if (variant == 1) {
        if (fast && plain) { hipLaunchKernelGGL((k_reduce<1, 0, false, false>), dim3((unsigned)(p.n_pairs * p.blocks_per_pair)), dim3(DVO_BLOCK), 0, s, p); return hipGetLastError(); }
        return launch_reduce_v<1, true>(p, weight_mode, scale_pass, s);
    }
    if (variant == 3) {
        if (fast && plain) { hipLaunchKernelGGL((k_reduce<3, 0, false, false>), dim3((unsigned)(p.n_pairs * p.blocks_per_pair)), dim3(DVO_BLOCK), 0, s, p); return hipGetLastError(); }
        return launch_reduce_v<3, true>(p, weight_mode, scale_pass, s);
    }
    return hipErrorInvalidValue;
}

// ================================================================================================
// K4: fixed-order sum of the per-workgroup partials -> one record per pair
// ================================================================================================
__device__ __forceinline__ void sum_partials_block(const double* __restrict__ part, int nblk, double* rec /*LDS [32]*/)
{
    // 256 threads: column c = tid & 31, row group g = tid >> 5 (8 groups); fixed order => reproducible sums
    __shared__ double tmp[8][DVO_STRIDE];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0.0;
    for (int b = g; b < nblk; b += 8) s += part[(size_t)b * DVO_STRIDE + c];
    tmp[g][c] = s;
    __syncthreads();
    if (threadIdx.x < DVO_STRIDE) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += tmp[k][threadIdx.x];
        rec[threadIdx.x] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(DVO_BLOCK) void k_sum_partials(const double* __restrict__ partials, int nblk,
                                                            double* __restrict__ out)
{
    __shared__ double rec[DVO_STRIDE];
    sum_partials_block(partials + (size_t)blockIdx.x * nblk * DVO_STRIDE, nblk, rec);
    if (threadIdx.x < DVO_STRIDE) out[(size_t)blockIdx.x * DVO_STRIDE + threadIdx.x] = rec[threadIdx.x];
}

hipError_t dvo_launch_sum_partials(const double* partials, int blocks_per_pair, int n_pairs, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_sum_partials, dim3(n_pairs), dim3(DVO_BLOCK), 0, s, partials, blocks_per_pair, out);
    return hipGetLastError();
}

// ================================================================================================
// K5: dense writer
// ================================================================================================
template <int VARIANT>
__global__ __launch_bounds__(DVO_BLOCK) void k_dense(const DvoDenseParams p)
{
    const int idx = blockIdx.x * DVO_BLOCK + threadIdx.x;
    if (idx >= p.g.npix) return;
    const uint8_t* prev = p.slots + (size_t)p.prev_slot * p.slot_stride;
    const uint8_t* curs = p.slots + (size_t)p.cur_slot * p.slot_stride;
    const uint8_t* gray_prev = prev + p.g.off_gray;
    const uint16_t* depth_prev = reinterpret_cast<const uint16_t*>(prev + p.g.off_depth);
    const uint32_t* packed_prev = reinterpret_cast<const uint32_t*>(prev + p.g.off_packed);
    const uint32_t* packed_cur = reinterpret_cast<const uint32_t*>(curs + p.g.off_packed);
    const int vy = idx / p.g.w, ux = idx - vy * p.g.w;
    const unsigned d = depth_prev[idx];
    PixelOut o;
    PixelGeo q;
    q.ok = false;
    q.x = q.y = q.z = 0.0f;
    q.uw = q.vw = __builtin_nanf("");
    q.ax = q.ay = 0.0f;
    o.r = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.J[k] = 0.0f;
    if (d != 0u) {
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = p.T[i];
        float jpre[6];
        if constexpr (VARIANT == 2) dvo_geo_v3(p.g, p.cam, p.depth_scale, T, gray_prev, ux, vy, d, q, jpre);
        else {
            const float z = (float)((double)d * p.depth_scale);               // camera_model.py:200-201
            if (p.cam.skew0) dvo_geo<false, VARIANT == 3>(p.g, p.cam, T, (float)ux, (float)vy, z, true, q);
            else dvo_geo<true, VARIANT == 3>(p.g, p.cam, T, (float)ux, (float)vy, z, true, q);
        }
        if (q.ok) {
            PixelTaps t;
            dvo_load_taps(p.g, packed_cur, q, t);
            t.own = (VARIANT == 1) ? packed_prev[(size_t)vy * p.g.pitch + ux] : 0u;
            dvo_finish<VARIANT, true>(p.cam, q, t, gray_prev[idx], o);
            if constexpr (VARIANT == 2) {
#pragma unroll
                for (int k = 0; k < 6; ++k) o.J[k] = jpre[k];
            }
        }
    }
    if (p.r) p.r[idx] = o.r;
    if (p.valid) p.valid[idx] = q.ok ? 1 : 0;
    if (p.J) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.J[(size_t)idx * 6 + k] = o.J[k];
    }
    if (p.uv) { p.uv[(size_t)idx * 2] = q.uw; p.uv[(size_t)idx * 2 + 1] = q.vw; }
}

hipError_t dvo_launch_dense(const DvoDenseParams& p, int variant, hipStream_t s)
{
    dim3 grid((p.g.npix + DVO_BLOCK - 1) / DVO_BLOCK), block(DVO_BLOCK);
    if (variant == 0) hipLaunchKernelGGL(k_dense<0>, grid, block, 0, s, p);
    else if (variant == 1) hipLaunchKernelGGL(k_dense<1>, grid, block, 0, s, p);
    else if (variant == 3) hipLaunchKernelGGL(k_dense<3>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_dense<2>, grid, block, 0, s, p);
    return hipGetLastError();
}

// ================================================================================================
// Device-side Gauss-Newton control (float64): 6x6 solve, SE(3) update, stop rules
// ================================================================================================
#define DVO_LIE_EPS 1e-6          // utils/lie_algebra/base_special_group.py:8
#define DVO_FLT_MAX 3.4028234663852886e+38

__device__ inline double wrap_angle_d(double a)
{   // common.py:32-46 ; Python's % is a floored modulo
    const double two_pi = 6.283185307179586;
    double x = a + 3.141592653589793;
    double m = x - floor(x / two_pi) * two_pi;
    return m - 3.141592653589793;
}

__device__ inline void quat_to_R(const double* q, double* R)
{   // So3.exp, special_orthogonal_group.py:158-189
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 2 * (w * w + x * x) - 1; R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 2 * (w * w + y * y) - 1; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 2 * (w * w + z * z) - 1;
}

__device__ inline void quat_log(const double* q, double* phi)
{   // So3.log, special_orthogonal_group.py:191-209
    const double n = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < DVO_LIE_EPS) { phi[0] = phi[1] = phi[2] = 0.0; return; }
    const double theta = wrap_angle_d(2.0 * atan2(n, q[0]) / n);
    phi[0] = theta * q[1]; phi[1] = theta * q[2]; phi[2] = theta * q[3];
}

// 1 / x for a float64 x in the normal range: v_rcp_f64 (about 26 good bits) + two Newton steps, about 1 ulp.  The control
// step is ONE lane's serial float64 program and its time is its instruction count (a lone wave issues one instruction
// every few cycles): the compiler's IEEE division is ~15 instructions and a float64 sqrt ~30, this is 5.
__device__ __forceinline__ double rcp64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(e, r, r);
    e = fma(-x, r, 1.0);
    return fma(e, r, r);
}

// sqrt(x) for a float64 x >= 0 in the normal range (or 0): v_rsq_f64 + two Newton steps on the reciprocal root, then x * rsqrt(x)
// with one correction: <= 1 ulp, ~12 instructions instead of the ~30 of the IEEE sqrt expansion (control step = instruction count).
__device__ __forceinline__ double sqrt64_fast(double x)
{
    if (!(x > 1e-300)) return (x == 0.0) ? 0.0 : sqrt(x);
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    const double s = x * y;
    return fma(fma(-s, s, x), 0.5 * y, s);
}

// sin(a), cos(a) of the half rotation angle a = theta / 2.  Between consecutive frames |a| is ~1e-3: the Taylor series to a^15 /
// a^14 is exact to float64 rounding for |a| <= 0.25 (next term a^17 / 17! < 2e-25) in 2 x 7 FMAs; larger angles take the library
// routine (with its range reduction: ~150 instructions).
__device__ __forceinline__ void sincos_half_angle(double a, double* s, double* c)
{
    if (fabs(a) <= 0.25) {
        const double z = a * a;
        double ps = fma(z, -1.0 / 1307674368000.0, 1.0 / 6227020800.0);
        ps = fma(z, ps, -1.0 / 39916800.0);
        ps = fma(z, ps, 1.0 / 362880.0);
        ps = fma(z, ps, -1.0 / 5040.0);
        ps = fma(z, ps, 1.0 / 120.0);
        ps = fma(z, ps, -1.0 / 6.0);
        *s = fma(a * z, ps, a);
        double pc = fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
        pc = fma(z, pc, -1.0 / 3628800.0);
        pc = fma(z, pc, 1.0 / 40320.0);
        pc = fma(z, pc, -1.0 / 720.0);
        pc = fma(z, pc, 1.0 / 24.0);
        pc = fma(z, pc, -0.5);
        *c = fma(z, pc, 1.0);
        return;
    }
    sincos(a, s, c);
}

// Se3.from_se3 (special_euclidean_group.py:105-123): xi = (upsilon, omega) -> (q, t = V upsilon)
__device__ inline void se3_from_xi(const double* xi, double* q, double* t)
{
    const double ox = xi[3], oy = xi[4], oz = xi[5];
    const double th2 = ox * ox + oy * oy + oz * oz;
    double theta = sqrt64_fast(th2);
    if (theta < DVO_LIE_EPS) {
        q[0] = 1; q[1] = q[2] = q[3] = 0;
        t[0] = xi[0]; t[1] = xi[1]; t[2] = xi[2];
        return;
    }
    double px = ox, py = oy, pz = oz, th = theta;
    if (!(theta < 3.0)) {
        // angles near or beyond pi: So3.__init__ wraps theta (:46-50) and rebuilds phi from the unit axis.  Below 3 rad the
        // wrap is the identity up to the rounding of (theta + pi) - pi, and phi = omega.
        const double ax = ox / theta, ay = oy / theta, az = oz / theta;
        theta = wrap_angle_d(theta);
        px = theta * ax; py = theta * ay; pz = theta * az;
        th = sqrt(px * px + py * py + pz * pz);
        if (th < DVO_LIE_EPS) {                                    // wrapped to ~0: So3.log would return 0
            q[0] = 1; q[1] = q[2] = q[3] = 0;
            t[0] = xi[0]; t[1] = xi[1]; t[2] = xi[2];
            return;
        }
    }
    double s2, c2;
    sincos_half_angle(th * 0.5, &s2, &c2);
    const double rth = rcp64(th);
    const double k = s2 * rth;
    q[0] = c2; q[1] = k * px; q[2] = k * py; q[3] = k * pz;
    // V = I + (1-cos)/th^2 [phi]x + (th - sin)/th^3 [phi]x^2, with 1 - cos th = 2 sin^2(th/2), sin th = 2 sin(th/2) cos(th/2)
    const double rth2 = rth * rth;
    const double A = 2.0 * s2 * s2 * rth2, B = (th - 2.0 * s2 * c2) * (rth2 * rth);
    const double ux = xi[0], uy = xi[1], uz = xi[2];
    const double c1x = py * uz - pz * uy, c1y = pz * ux - px * uz, c1z = px * uy - py * ux;          // phi x u
    const double c2x = py * c1z - pz * c1y, c2y = pz * c1x - px * c1z, c2z = px * c1y - py * c1x;    // phi x (phi x u)
    t[0] = ux + A * c1x + B * c2x;
    t[1] = uy + A * c1y + B * c2y;
    t[2] = uz + A * c1z + B * c2z;
}

// Se3.__mul__ (special_euclidean_group.py:91-96) with quat_mult (common.py:51-73): (qa,ta) * (qb,tb)
__device__ inline void se3_mul(const double* qa, const double* ta, const double* qb, const double* tb, double* q, double* t)
{
    double R[9];
    quat_to_R(qa, R);
    const double w = qa[0] * qb[0] - (qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3]);
    const double x = qa[0] * qb[1] + qb[0] * qa[1] + (qa[2] * qb[3] - qa[3] * qb[2]);
    const double y = qa[0] * qb[2] + qb[0] * qa[2] + (qa[3] * qb[1] - qa[1] * qb[3]);
    const double z = qa[0] * qb[3] + qb[0] * qa[3] + (qa[1] * qb[2] - qa[2] * qb[1]);
    const double t0 = ta[0] + R[0] * tb[0] + R[1] * tb[1] + R[2] * tb[2];
    const double t1 = ta[1] + R[3] * tb[0] + R[4] * tb[1] + R[5] * tb[2];
    const double t2 = ta[2] + R[6] * tb[0] + R[7] * tb[1] + R[8] * tb[2];
    q[0] = w; q[1] = x; q[2] = y; q[3] = z;
    t[0] = t0; t[1] = t1; t[2] = t2;
}

// Se3.exp (special_euclidean_group.py:35-52): rotation block is the identity when |log| < eps
__device__ inline void se3_write_T(const double* q, const double* t, float* T)
{
    double R[9];
    bool ident;
    if (q[0] > 0.5) {
        // rotation below 120 degrees (always, between consecutive frames): |log q| = 2 asin(n) >= 2 n with n = |q_xyz|, and
        // So3.log returns 0 for n < eps — so |log| < eps <=> n < eps, without the atan2 of So3.log
        ident = (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) < DVO_LIE_EPS * DVO_LIE_EPS;
    } else {
        double phi[3];
        quat_log(q, phi);
        ident = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]) < DVO_LIE_EPS;
    }
    if (ident) { R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0; }
    else quat_to_R(q, R);
    T[0] = (float)R[0]; T[1] = (float)R[1]; T[2] = (float)R[2];  T[3] = (float)t[0];
    T[4] = (float)R[3]; T[5] = (float)R[4]; T[6] = (float)R[5];  T[7] = (float)t[1];
    T[8] = (float)R[6]; T[9] = (float)R[7]; T[10] = (float)R[8]; T[11] = (float)t[2];
}

// Se3.log (special_euclidean_group.py:54-77)
__device__ inline void se3_log(const double* q, const double* t, double* xi)
{
    double phi[3];
    quat_log(q, phi);
    const double theta = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    if (theta < DVO_LIE_EPS) { xi[0] = t[0]; xi[1] = t[1]; xi[2] = t[2]; xi[3] = xi[4] = xi[5] = 0; return; }
    const double ax = phi[0] / theta, ay = phi[1] / theta, az = phi[2] / theta;
    const double t2 = theta * 0.5;
    const double A = t2 * cos(t2) / sin(t2);
    // V^-1 = A I + (1-A) a a^T - (theta/2) [a]x
    const double adt = ax * t[0] + ay * t[1] + az * t[2];
    const double cx = ay * t[2] - az * t[1], cy = az * t[0] - ax * t[2], cz = ax * t[1] - ay * t[0];
    xi[0] = A * t[0] + (1 - A) * ax * adt - t2 * cx;
    xi[1] = A * t[1] + (1 - A) * ay * adt - t2 * cy;
    xi[2] = A * t[2] + (1 - A) * az * adt - t2 * cz;
    xi[3] = phi[0]; xi[4] = phi[1]; xi[5] = phi[2];
}

// The reference solves with scipy lstsq / gelsy on a FLOAT32 H (base_robust_dvo.py:196-198): LAPACK's rank decision drops every
// direction whose singular value is below rcond = float32 eps of the largest, and returns the minimum-norm solution of the rest.  For
// the symmetric positive semi-definite H singular values are eigenvalues: directions with eigenvalue <= 1.19e-7 lambda_max carry no
// increment (tests/golden/rank_cases.npz: reference runs with cond(H) = inf, 1.5e8 and 1.6e3).
#define DVO_RANK_RCOND 1.1920928955078125e-07

// Minimum-norm solution of a numerically rank-deficient symmetric 6x6 system through a cyclic Jacobi
// eigen-decomposition (what LAPACK gelsy, base_robust_dvo.py:196-198, returns for a rank-deficient system).
// Rare path: kept out of line so its runtime-indexed arrays do not push the common path into scratch.
__device__ __noinline__ void solve6_pinv(const double* hu, const double* b, double dmax, double* x, double* ws /* [72] LDS */)
{
    // every loop is kept rolled (#pragma unroll 1) and the two matrices live in a workspace the caller provides in LDS (one lane
    // per workgroup runs the control step): the function then needs a dozen registers and no scratch; unrolled, the compiler
    // promotes both matrices to registers (128 VGPRs + spills), and a callee's registers count against every kernel that can reach it
    double (*A)[6] = reinterpret_cast<double (*)[6]>(ws);
    double (*V)[6] = reinterpret_cast<double (*)[6]>(ws + 36);
    {
        int q = 0;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {
#pragma unroll 1
            for (int j = i; j < 6; ++j) { A[i][j] = hu[q]; A[j][i] = hu[q]; ++q; }
        }
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
#pragma unroll 1
        for (int j = 0; j < 6; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {
#pragma unroll 1
            for (int j = i + 1; j < 6; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-60 * dmax * dmax || off == 0.0) break;
#pragma unroll 1
        for (int p = 0; p < 5; ++p) {
#pragma unroll 1
            for (int q = p + 1; q < 6; ++q) {
                if (A[p][q] == 0.0) continue;
                const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
#pragma unroll 1
                for (int k = 0; k < 6; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
#pragma unroll 1
                for (int k = 0; k < 6; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
#pragma unroll 1
                for (int k = 0; k < 6; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
            }
        }
    }
    double emax = 0.0;
#pragma unroll 1
    for (int i = 0; i < 6; ++i) emax = fmax(emax, fabs(A[i][i]));
#pragma unroll 1
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
#pragma unroll 1
    for (int k = 0; k < 6; ++k) {
        const double e = A[k][k];
        if (!(e > DVO_RANK_RCOND * emax)) continue;           // the reference's rank rule (see solve6)
        double vb = 0.0;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) vb += V[i][k] * b[i];
        vb /= e;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) x[i] += V[i][k] * vb;
    }
}

// Solves H x = b for the symmetric positive semi-definite 6x6 H (upper triangle hu[21], row-major).
// Fast path: fully unrolled LDL^T IN PLACE on the 21 packed entries (unit-lower L overwrites the triangle, d the diagonal):
// no square roots, one Newton reciprocal per pivot, substitutions by multiplication — ~150 float64 instructions instead of
// the ~800 of a Cholesky with IEEE sqrt and divisions (the pivots are the same numbers: d_j = l_jj^2).  Falls back to
// solve6_pinv when a pivot is not safely positive (the minimum-norm answer gelsy gives, base_robust_dvo.py:196-198).
// lo6(i, j), i >= j: position of element (i, j) of the lower triangle = position of (j, i) in the row-major upper triangle.
__device__ __forceinline__ constexpr int lo6(int i, int j) { return j * 6 - (j * (j - 1)) / 2 + (i - j); }

__device__ __forceinline__ void solve6(const double* hu, const double* b, double* x, double* ws /* [72] LDS, rare path */)
{
    double a[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) a[q] = hu[q];
    // every entry finite <=> the sum of their magnitudes is (27 adds and one class test instead of 27 class tests; a sum that
    // overflows to infinity although every term is finite would need entries beyond 1e306)
    double dmax = 0.0, mag = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        dmax = fmax(dmax, fabs(a[lo6(i, i)]));
        mag += fabs(b[i]);
    }
#pragma unroll
    for (int q = 0; q < 21; ++q) mag += fabs(a[q]);
    const bool finite = isfinite(mag);
    if (!finite) {
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = __builtin_nan("");
        return;
    }
    if (dmax == 0.0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = 0.0;
        return;
    }
    bool ok = true;
    double rd[6];                                            // 1 / d_j
    const double dmin = 1e-13 * dmax;
    double trace = 0.0, det = 1.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) trace += a[lo6(j, j)];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v[6];                                         // v_k = L_jk d_k
        double d = a[lo6(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = a[lo6(j, k)] * a[lo6(k, k)];
            d = fma(-a[lo6(j, k)], v[k], d);
        }
        ok = ok && (d > dmin);
        d = ok ? d : 1.0;
        det *= d * (5.0 / trace);                            // (scaled per pivot so that the product stays in range)
        a[lo6(j, j)] = d;
        rd[j] = rcp64(d);
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double sacc = a[lo6(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc = fma(-a[lo6(i, k)], v[k], sacc);
            a[lo6(i, j)] = sacc * rd[j];
        }
    }
    // Full rank under the reference's rule?  lambda_min = det / (product of the other five eigenvalues) >= det (5 / trace)^5 (their
    // sum is below the trace: AM-GM) and lambda_max <= trace, so lambda_min / lambda_max >= [prod_j d_j (5 / trace)] / 5: when that
    // exceeds rcond no direction is dropped and the LDL^T solve below IS the reference's answer (fixture frames: 3e-5 against
    // 1.2e-7).  Otherwise the eigen-decomposition decides exactly (rare: eigenvalue spread beyond ~2e4).
    ok = ok && (det * 0.2 > DVO_RANK_RCOND);
    if (ok) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {                       // L z = b
            double sacc = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) sacc = fma(-a[lo6(i, k)], x[k], sacc);
            x[i] = sacc;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] *= rd[i];           // D w = z
#pragma unroll
        for (int i = 5; i >= 0; --i) {                      // L^T x = w
            double sacc = x[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) sacc = fma(-a[lo6(k, i)], x[k], sacc);
            x[i] = sacc;
        }
        return;
    }
    solve6_pinv(hu, b, dmax, x, ws);
}

// What a perturbation |db_j| <= dbound[j] of the right-hand side can move the solution of H x = b by, component by component:
// ebound[k] = sum_j |H^-1|_kj dbound[j] (rigorous: the triangle inequality on x = H^-1 b), from the explicit inverse (six unit-vector
// solves on the LDL^T factors).  Also hdiag[k] = (H^-1)_kk and, when h1 != null, *trace1 = trace(H^-1 H1).  Returns false when H is
// not safely positive definite.  Only the one-pass Student-t iterations call it (one lane; ~300 float64 operations); out of line so
// that the common control path does not carry its registers.
__device__ __noinline__ bool solve6_inverse_bound(const double* hu, const double* dbound, const double* h1 /* 21 packed or null */,
                                                  double* ebound, double* hdiag, double* trace1)
{
    double a[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) a[q] = hu[q];
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) dmax = fmax(dmax, fabs(a[lo6(i, i)]));
    if (!(dmax > 0.0) || !isfinite(dmax)) return false;
    bool ok = true;
    double rd[6];
    const double dmin = 1e-13 * dmax;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v[6];
        double d = a[lo6(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = a[lo6(j, k)] * a[lo6(k, k)];
            d = fma(-a[lo6(j, k)], v[k], d);
        }
        ok = ok && (d > dmin);
        d = ok ? d : 1.0;
        a[lo6(j, j)] = d;
        rd[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double sacc = a[lo6(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc = fma(-a[lo6(i, k)], v[k], sacc);
            a[lo6(i, j)] = sacc * rd[j];
        }
    }
    if (!ok) return false;
    double tr = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) ebound[k] = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {                           // column j of H^-1
        double x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {                       // L z = e_j (z_i = 0 above j)
            double sacc = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) sacc = fma(-a[lo6(i, k)], x[k], sacc);
            x[i] = (i < j) ? 0.0 : sacc;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] *= rd[i];
#pragma unroll
        for (int i = 5; i >= 0; --i) {                      // L^T x = w
            double sacc = x[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) sacc = fma(-a[lo6(k, i)], x[k], sacc);
            x[i] = sacc;
        }
        hdiag[j] = x[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) ebound[k] = fma(fabs(x[k]), dbound[j], ebound[k]);
        if (h1 != nullptr) {
#pragma unroll
            for (int k = 0; k < 6; ++k) tr = fma(x[k], h1[(k >= j) ? lo6(k, j) : lo6(j, k)], tr);
        }
    }
    *trace1 = tr;
    return true;
}

__global__ void k_init_state(DvoPairState* st, float* T, const float* T_init, int n_pairs, double lambda0,
                             const double* prior_qt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    DvoPairState s;
    s.q[0] = 1; s.q[1] = s.q[2] = s.q[3] = 0;
    s.t[0] = s.t[1] = s.t[2] = 0;
    if (T_init != nullptr) {
        // rotation matrix -> quaternion (So3._SE3_to_quat, special_orthogonal_group.py:86-127)
        const float* M = T_init + (size_t)i * 12;
        const double R[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
        double tr = R[0] + R[4] + R[8];
        if (tr > 0) {
            double t = sqrt(1 + tr);
            s.q[0] = 0.5 * t; t = 0.5 / t;
            s.q[1] = (R[7] - R[5]) * t; s.q[2] = (R[2] - R[6]) * t; s.q[3] = (R[3] - R[1]) * t;
        } else {
            int a = 0;
            if (R[4] > R[0]) a = 1;
            if (R[8] > R[a * 4]) a = 2;
            const int b = (a + 1) % 3, c = (b + 1) % 3;
            double t = sqrt(R[a * 4] - R[b * 4] - R[c * 4] + 1);
            s.q[1 + a] = 0.5 * t; t = 0.5 / t;
            s.q[0] = (R[c * 3 + b] - R[b * 3 + c]) * t;
            s.q[1 + b] = (R[b * 3 + a] + R[a * 3 + b]) * t;
            s.q[1 + c] = (R[c * 3 + a] + R[a * 3 + c]) * t;
        }
        s.t[0] = M[3]; s.t[1] = M[7]; s.t[2] = M[11];
    }
    // motion prior (base_robust_dvo.py:154): `old` restarts at every level from the previous pair's transform
    s.lq[0] = 1; s.lq[1] = s.lq[2] = s.lq[3] = 0; s.lt[0] = s.lt[1] = s.lt[2] = 0;
    if (prior_qt != nullptr) {
        for (int k = 0; k < 4; ++k) s.lq[k] = prior_qt[(size_t)i * 7 + k];
        for (int k = 0; k < 3; ++k) s.lt[k] = prior_qt[(size_t)i * 7 + 4 + k];
    }
    for (int k = 0; k < 4; ++k) s.oq[k] = s.lq[k];
    for (int k = 0; k < 3; ++k) s.ot[k] = s.lt[k];
    s.err_prev = DVO_FLT_MAX;
    s.err_last = __builtin_nan("");
    s.lambda = -1.0;
    s.increased = 0;
    s.active = 1;
    s.iters = 0;
    s.pad = 0;
    s.lam_hist[0] = s.lam_hist[1] = s.lam_hist[2] = 0.0;
    s.lam_n = 0;
    s.pad2 = 0;
    s.lam_err = 1.0;
    st[i] = s;
    se3_write_T(s.q, s.t, T + (size_t)i * 12);
    (void)lambda0;
}

hipError_t dvo_launch_init_state(DvoPairState* st, float* T, const float* T_init, int n_pairs, double lambda0,
                                 const double* prior_qt, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_state, dim3((n_pairs + 63) / 64), dim3(64), 0, s, st, T, T_init, n_pairs, lambda0, prior_qt);
    return hipGetLastError();
}

// start of a pyramid level: err_prev = FLT_MAX, err_increased_count = 0 (base_robust_dvo.py:158-159)
// one_pass: the level starts with the corrected one-pass Student-t form, whose weights are evaluated under a guessed lambda (wparam):
// a quarter of the last lambda of the level before (four times the points: lambda = 1 / sum w r^2), or 0 when there is none (the plain
// moment form).  Any guess is legal - the control step bounds what it costs - a closer one just leaves a smaller remainder.
__global__ void k_level_begin(DvoPairState* st, int* active, double* wparam, int n_pairs, int weight_mode, double lambda0, int one_pass)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    st[i].err_prev = DVO_FLT_MAX;
    st[i].increased = 0;
    st[i].active = 1;
    const double guess = (st[i].lam_n > 0 && st[i].lam_hist[0] > 0.0) ? 0.25 * st[i].lam_hist[0] : 0.0;
    st[i].lam_n = 0;
    st[i].lam_err = 1.0;
    active[i] = 1;
    for (int k = 0; k < 4; ++k) st[i].oq[k] = st[i].lq[k];          // old = last_estimated_transform.copy()  (:154)
    for (int k = 0; k < 3; ++k) st[i].ot[k] = st[i].lt[k];
    if (weight_mode == 1 || weight_mode == 3) wparam[i] = one_pass ? guess : lambda0;   // t_weighter.py:23
    if (weight_mode == 2) wparam[i] = (st[i].lambda > 0.0) ? st[i].lambda : DVO_FLT_MAX;   // no scale yet: w = 1
}

hipError_t dvo_launch_level_begin(DvoPairState* st, int* active, double* wparam, int n_pairs, int weight_mode,
                                  double lambda0, int one_pass, hipStream_t s)
{
    hipLaunchKernelGGL(k_level_begin, dim3((n_pairs + 63) / 64), dim3(64), 0, s, st, active, wparam, n_pairs,
                       weight_mode, lambda0, one_pass);
    return hipGetLastError();
}

// Scalar control of one Gauss-Newton iteration of one pair (run by ONE thread): base_robust_dvo.py:166-232.
// `rec` is the summed normal-equation record (modified in place when the motion prior is on).
// P supplies the scalars (tolerance, max_increased, weight_mode, weight_param, inv_sigma, trace*): DvoSolveParams (k_solve:
// state / T / active / wparam in global memory) or DvoSmallParams (k_gn_small: the same objects held in LDS).
//
// Student-t of the reference in ONE pass over the points (weighter/t_weighter.py:21-47 + base_robust_dvo.py:171-188).
// The reference's IRLS restarts every Gauss-Newton iteration from lambda0 = 1/25, takes lambda1 = 1 / S(lambda0), then
// lambda2 = 1 / S(lambda1) and stops (|lambda2 - lambda1| < 1e-3): because its scale S is a SUM over the image (quirk Q3), lambda1 and
// lambda2 are ~1e-9..1e-7 and x = lambda r^2 / 5 is tiny for every point.  One pass yields
//   S(lambda0)  summed exactly per point (lambda0 is a constant),
//   S(lambda1)  = 1.2 (M1 - x1 M2 + x1^2 M3 [- x1^3 M4]),   Mk = sum r^(2k): moments of the SAME pass      => the exact lambda2,
// and the weighted sums under a GUESSED lambda_g (w_g = 6 / (5 + r^2 lambda_g)), corrected to first order in d = lambda2 - lambda_g
// with their derivatives, which are sums of the same pass too.  The identity behind it, per point:
//   w(lambda2) = w_g - (d / 6) r^2 w_g^2 + rho,     rho = (d / 6)^2 r^4 w_g^2 w(lambda2)   (>= 0, exact)
// "corrected form" (first iterations of a level; lambda_g = 0 - i.e. w_g = 1.2, the plain moment form - at the first level, a
// quarter of the level before's last lambda at the first iteration of a level, then the last / the linearly extrapolated one):
//   J^T W J = Hg - (d / 6) H1,   J^T W r = bg - (d / 6) b1,     H1 = sum w_g^2 r^2 J J^T,  b1 = sum w_g^2 r^3 J  (second record)
// "predicted form" (from the fourth iteration on; lambda_g = the geometric extrapolation of the last three lambdas, good to
// 1e-4..1e-3 relative): only the right-hand side is corrected (b1: six extra sums), the matrix is kept as it is.
// What each form drops is bounded per iteration ON THE INCREMENT (gn_control): rho in the right-hand side through Cauchy-Schwarz
// (dbound), the matrix through eps >= || H^-1/2 dH H^-1/2 ||.  Return value: 0 the form applies; 1 x is too large for the series of
// S(lambda1) (the pair is solved again with the passes as written, which check the shape of the loop exactly); 2 the series holds
// and the reference's loop would not take exactly two passes here (solved again by the loop itself).
//
// Corrected form.  In: rec = {Hg, bg, M1 [27], N [28], S(lambda0) [29]}, rec2 = {H1, b1, M2 [27], M3 [28], M4 [29]}.  Out: rec as a
// weighted record under lambda2 ([27] = sum w r^2, [29] = sum r^2).
//   |db_k| <= sum rho |J_k r| <= 1.2 (d/6)^2 sqrt(sum w_g^2 r^8 . sum w_g^2 r^2 J_k^2) <= 0.04 d^2 sqrt(M4 H1_kk)
//   dH = sum rho J J^T <= 1.2 (d/6)^2 r2max H1  (positive semi-definite order; r2max = M4^(1/4) >= every r^2)
//      => eps = eps_fac trace(H^-1 H1),  eps_fac = d^2 r2max / 30   (the caller has H^-1)
__device__ __forceinline__ int student_t_moment_combine(double* rec, const double* rec2, double lambda0, double lam_g, double* dbound,
                                                        double* lam_out, double* eps_fac)
{
    const double S0 = rec[29], M1 = rec[27], M2 = rec2[27], M3 = rec2[28], M4 = rec2[29];
    const double lam1 = 1.0 / S0;
    bool ok = (S0 > 0.0) && isfinite(lam1) && !(fabs(lam1 - lambda0) < 1e-3);        // (converged after one pass: lambda stays ~lambda0)
    const double x1 = lam1 * 0.2;
    const double S1 = 1.2 * (M1 - x1 * (M2 - x1 * (M3 - x1 * M4)));
    const double lam2 = 1.0 / S1;
    ok = ok && (S1 > 0.0) && isfinite(lam2) && (fabs(lam2 - lam1) < 1e-3);             // the reference stops after the second pass
    const double x2 = lam2 * 0.2;
    // r^2 <= M4^(1/4): every point's x is below xmax; the series of S(lambda1) and of sum w r^2 need xmax << 1
    const double r2max = sqrt(sqrt(M4));
    const double xmax = fmax(fmax(x1, x2), lam_g * 0.2) * r2max;
    const bool series = (xmax < 1e-2) && (lam_g >= 0.0);
    const double d = lam2 - lam_g, c = d * (1.0 / 6.0);
#pragma unroll
    for (int q = 0; q < 27; ++q) rec[q] = fma(-c, rec2[q], rec[q]);
    rec[27] = 1.2 * (M1 - x2 * (M2 - x2 * M3));
    rec[29] = M1;
    const int dg[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
    for (int k = 0; k < 6; ++k) dbound[k] = 0.04 * d * d * sqrt(M4 * fmax(rec2[dg[k]], 0.0));
    *eps_fac = d * d * r2max * (1.0 / 30.0);
    *lam_out = lam2;
    return series ? (ok ? 0 : 2) : 1;        // (without the series lambda2 is not known well enough to judge the shape: the passes as written do)
}

// Predicted form.  In: rec = {Hp, bp under lam_p, S(lambda0) [27], N [28], M1 [29], M2 [30], M3 [31]}, b1[6] = sum w_p^2 r^3 J.  Out:
// rec as a weighted record under lambda2 ([27] = sum w r^2 from the moments, [29] = sum r^2).
//   |db_k| <= sum rho |J_k r| <= 1.2 (d/6)^2 r2max sqrt(sum w_p^2 r^6 . sum w_p^2 J_k^2) <= (d^2 / 30) r2max sqrt(1.728 M3 H_kk)
//   |dH| <= sum |w(lambda2) - w_p| J J^T <= 0.2 |d| r2max (1 + xmax) H   => eps      (r2max = M3^(1/3) >= every r^2)
__device__ __forceinline__ int student_t_predicted_combine(double* rec, const double* b1, double lambda0, double lam_p, double* dbound,
                                                           double* lam_out, double* eps)
{
    const double S0 = rec[27], M1 = rec[29], M2 = rec[30], M3 = rec[31];
    const double lam1 = 1.0 / S0;
    bool ok = (S0 > 0.0) && isfinite(lam1) && !(fabs(lam1 - lambda0) < 1e-3);
    const double x1 = lam1 * 0.2;
    const double S1 = 1.2 * (M1 - x1 * (M2 - x1 * M3));
    const double lam2 = 1.0 / S1;
    ok = ok && (S1 > 0.0) && isfinite(lam2) && (fabs(lam2 - lam1) < 1e-3);
    const double x2 = lam2 * 0.2;
    // r^2 <= M3^(1/3) for every point: with xmax < 1 the two series (S(lambda1) above, sum w r^2 below) alternate with decreasing
    // terms, so stopping after the M3 term is off by less than the next one, x^3 M4 <= x^3 M3^(4/3): kept below 1e-6 of the sum
    const double c3 = cbrt(M3);
    const double xmax = fmax(fmax(x1, x2), lam_p * 0.2) * c3;
    const bool series = (xmax < 0.5) && (xmax * xmax * xmax * c3 <= 1e-6 * M1) && (lam_p >= 0.0);
    const double d = lam2 - lam_p, c = d * (1.0 / 6.0);
    const int dg[6] = {0, 6, 11, 15, 18, 20};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        dbound[k] = d * d * (1.0 / 30.0) * c3 * sqrt(1.728 * M3 * fmax(rec[dg[k]], 0.0));
        rec[21 + k] = fma(-c, b1[k], rec[21 + k]);
    }
    *eps = 0.2 * fabs(d) * c3 * (1.0 + xmax);
    rec[27] = 1.2 * (M1 - x2 * (M2 - x2 * M3));
    rec[30] = 0.0;
    rec[31] = 0.0;
    *lam_out = lam2;
    return series ? (ok ? 0 : 2) : 1;        // (without the series lambda2 is not known well enough to judge the shape: the passes as written do)
}

// The scalars gn_control reads, for k_gn_small: a copy in LDS filled once per launch.  Read from the kernel arguments they are
// loop invariants, and so is every float64 value derived from them (1 / inv_sigma, the trace row address, ...): the compiler
// computes those once, before the level loop, in vector registers (float64 has no scalar form) and keeps them across every
// record loop of the launch, which at the kernel's register cap means ~100 spilled registers.  LDS reads stay where they are.
struct DvoControlParams {
    double tolerance, weight_param, inv_sigma;
    double* trace;
    int* irls_done;
    const int* trace_row;
    int* pair_flags;             // always null here: the persistent kernel repeats an unsafe iteration itself
    double st_tol;
    int max_increased, weight_mode, trace_stride, trace_cap;
};

template <class P>
__device__ __forceinline__ void gn_control(const P& p, int phase, int pair, DvoPairState& st, float* T, int* active,
                                           double* wparam, double* rec, double* ws /* [72] LDS workspace */,
                                           const double* rec2 = nullptr, int* inexact = nullptr, double lambda0 = 0.0,
                                           int predicted = 0, double lam_known = -1.0, const double* b1 = nullptr)
{
    // rec2: the second record of the corrected form (H1, b1, moments), or null; b1: the six sums the predicted form corrects its
    // right-hand side with (predicted != 0)
    // predicted: 1 = the record was weighted under the predicted lambda in *wparam (k_solve); unsafe iterations are counted in
    // inexact[0..3] and their pair is flagged (the host solves it again).  lam_known > 0: the lambda the IRLS loop of this
    // iteration ended on (k_gn_small), recorded for the guess of the next level's first one-pass iteration.
    if (phase != 0) {
        // Student-t scale pass k (weighter/t_weighter.py:24-32): lambda <- 1 / scale, stop when lambda has settled.
        //   weight_mode 1 (the reference): scale = SUM of the weighted r^2 (quirk Q3), |d lambda| < 1e-3 absolute
        //   weight_mode 3 (proper):        scale = MEAN of the weighted r^2, |d lambda| <= 1e-3 lambda
        const double last = *wparam;
        if (st.pad == 0) {                       // pad: 1 once the IRLS loop has converged for this iteration
            const double cur = (p.weight_mode == 3) ? rec[1] / rec[0] : 1.0 / rec[0];
            const bool conv = (p.weight_mode == 3) ? (fabs(cur - last) <= 1e-3 * cur) : (fabs(cur - last) < 1e-3);
            if (conv) { st.pad = 1; if (p.irls_done != nullptr) p.irls_done[pair] = 1; }
            *wparam = cur;
        }
        return;
    }

    double dbound[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const double lam_pred_in = *wparam;     // Student-t: the lambda predicted for this iteration at the end of the one before
    int form = 0;                           // 0: the one-pass form applies; 1: series not applicable; 2: IRLS not of the two-pass shape
    const bool one_pass = (rec2 != nullptr) || (predicted != 0);
    double lam2 = 0.0, eps_fac = 0.0;
    if (rec2 != nullptr) form = student_t_moment_combine(rec, rec2, lambda0, *wparam, dbound, &lam2, &eps_fac);
    else if (predicted) form = student_t_predicted_combine(rec, b1, lambda0, *wparam, dbound, &lam2, &eps_fac);
    const bool moment_ok = (form == 0);
    const double n = rec[28];
    double err = (n > 0.0) ? (double)(float)(rec[27] / n) : __builtin_nan("");      // np.mean(...) float32, :177/:184
    double bneg[6], inc[6];
    for (int i = 0; i < 6; ++i) bneg[i] = -rec[21 + i];                                // b = -J^T W r, :188
    if (p.inv_sigma > 0.0) {
        // H += I/sigma ; b += (I/sigma) log(old) ; err += 0.5 sigma |log(old)|        (:190-194)
        double xo[6];
        se3_log(st.oq, st.ot, xo);
        const int dg[6] = {0, 6, 11, 15, 18, 20};
        double nrm = 0.0;
        for (int i = 0; i < 6; ++i) { rec[dg[i]] += p.inv_sigma; bneg[i] += p.inv_sigma * xo[i]; nrm += xo[i] * xo[i]; }
        err += 0.5 * (1.0 / p.inv_sigma) * sqrt(nrm);
    }
    // (compiler fences around the solve: without them the loads of the pose state, of the trace pointers and of the record are
    // all hoisted to the top of the control step and stay live across it — the step is one lane's serial program and its
    // register peak, not its latency, is what matters inside k_gn_small)
    asm volatile("" ::: "memory");
    solve6(rec, bneg, inc, ws);
    asm volatile("" ::: "memory");
    if (one_pass && inexact != nullptr) {
        // What the one-pass form changes, bounded on the increment itself (not on the right-hand side alone):
        //   inc_true - inc = (H + dH)^-1 (db - dH inc),   |db_k| <= dbound_k (Cauchy-Schwarz bounds of the combine functions),
        //   || H^-1/2 dH H^-1/2 || <= eps (above)
        //   => |inc_true - inc|_k <= ( sum_j |H^-1|_kj dbound_j  +  eps sqrt((H^-1)_kk inc^T b) ) / (1 - eps)
        // (first term: triangle inequality with the explicit inverse; second: |e_k| <= sqrt((H^-1)_kk) ||e||_H and
        // ||(H + dH)^-1 dH inc||_H <= eps / (1 - eps) ||inc||_H with ||inc||_H^2 = inc^T b).  The iteration is kept when the 2-norm
        // of that bound is within `tol` of the increment's, half the parity tolerance of 1e-5; else it is run again as written
        // (inside the launch: k_gn_small; per pair, after the batch: k_solve, see DvoSolveParams::pair_flags).
        double eb[6], hd[6], tr1 = 0.0;
        const bool inv_ok = solve6_inverse_bound(rec, dbound, rec2, eb, hd, &tr1);
        const double eps = (rec2 != nullptr) ? eps_fac * fmax(tr1, 0.0) : eps_fac;
        double e2 = 0.0, i2 = 0.0, ib = 0.0, eb2 = 0.0;
        for (int k = 0; k < 6; ++k) ib = fma(inc[k], bneg[k], ib);
        for (int k = 0; k < 6; ++k) {
            const double bk = eb[k] + eps * sqrt(fmax(hd[k] * ib, 0.0));
            e2 = fma(bk, bk, e2);
            eb2 = fma(eb[k], eb[k], eb2);
            i2 = fma(inc[k], inc[k], i2);
        }
        const bool bound_ok = inv_ok && (eps < 0.5) && (e2 <= (p.st_tol * p.st_tol) * (1.0 - eps) * (1.0 - eps) * i2);
        rec[30] = (i2 > 0.0) ? sqrt(e2 / i2) / (1.0 - eps) : 0.0;       // (trace / diagnostics: the bound relative to the increment, and eps)
#ifdef DVO_ST_PROBE_EB
        rec[31] = (i2 > 0.0) ? sqrt(eb2 / i2) : 0.0;
#else
        rec[31] = eps;
#endif
        if (n > 0.0) {
            // inexact[0]: iterations that failed; [1]: the IRLS would not take exactly two passes / series not applicable;
            // [2]: bound of the moment form; [3]: bound of the predicted-lambda form
            if (!moment_ok) atomicAdd(inexact + 1, 1);
            else if (!bound_ok) atomicAdd(inexact + (predicted ? 3 : 2), 1);
            if (!(moment_ok && bound_ok)) {
                atomicAdd(inexact, 1);
                if (p.pair_flags != nullptr) p.pair_flags[pair] |= (form == 2) ? 2 : 1;
            }
        }
    }
    st.iters += 1;
    st.err_last = err;
    if (p.trace != nullptr && st.iters <= p.trace_cap) {
        const int row = (p.trace_row != nullptr) ? p.trace_row[pair] : pair;
        double* tr = p.trace + ((size_t)row * p.trace_cap + (st.iters - 1)) * p.trace_stride;
        for (int i = 0; i < DVO_STRIDE; ++i) tr[i] = rec[i];
        for (int i = 0; i < 6; ++i) tr[DVO_STRIDE + i] = inc[i];
        for (int i = 0; i < 12; ++i) tr[DVO_STRIDE + 6 + i] = (double)T[i];
    }
    asm volatile("" ::: "memory");
    const double diff = err - st.err_prev;
    bool stop = false;
    if (fabs(diff) < p.tolerance) {                                                     // :207
        stop = true;
    } else {
        if (diff < 0.0) {                                                               // :212-221
            double qi[4], ti[3], qn[4], tn[3];
            se3_from_xi(inc, qi, ti);
            se3_mul(qi, ti, st.q, st.t, qn, tn);                                        // estimate = inc * estimate
            for (int i = 0; i < 4; ++i) st.q[i] = qn[i];
            for (int i = 0; i < 3; ++i) st.t[i] = tn[i];
            st.err_prev = err;
            st.increased = 0;
            se3_write_T(st.q, st.t, T);
            if (p.inv_sigma > 0.0) {                                                    // old = inc.inverse() * old  (:217-218)
                double qv[4] = {qi[0], -qi[1], -qi[2], -qi[3]}, Ri[9], tv[3], qo[4], to[3];
                quat_to_R(qv, Ri);
                for (int a = 0; a < 3; ++a) tv[a] = -(Ri[3 * a] * ti[0] + Ri[3 * a + 1] * ti[1] + Ri[3 * a + 2] * ti[2]);
                se3_mul(qv, tv, st.oq, st.ot, qo, to);
                for (int a = 0; a < 4; ++a) st.oq[a] = qo[a];
                for (int a = 0; a < 3; ++a) st.ot[a] = to[a];
            }
        } else {
            st.increased += 1;                                                          // :222 (no rollback, quirk Q8)
        }
        if (st.increased > p.max_increased) stop = true;                                // :224
    }
    if (stop) { st.active = 0; *active = 0; }
    // (control launch after every scale pass, DVO_ST_SOLVE_PHASES: the two passes enqueued did not end the reference's loop)
    if (p.weight_mode == 1 && p.irls_done != nullptr && p.pair_flags != nullptr && st.pad == 0 && n > 0.0) p.pair_flags[pair] |= 2;
    if (p.weight_mode == 1) { *wparam = 1.0 / 25.0; st.pad = 0; }                       // next iteration restarts IRLS (t_weighter.py:23)
    if (!one_pass && lam_known > 0.0) lam2 = lam_known;
    if (one_pass || lam_known > 0.0) {
        // how good the prediction of THIS iteration's lambda was (*wparam still holds it when at least two lambdas of this level
        // were known when it was made): the persistent kernel tries the predicted form only after a prediction within 2e-4
        st.lam_err = (st.lam_n >= 2 && lam2 > 0.0) ? fabs(lam_pred_in - lam2) / lam2 : 1.0;
        // lambda of the next iteration's weights (the reduction weighted with a predicted lambda reads it from wparam)
        // (lambda settles geometrically with the pose: the next step of the sequence is the last step times the ratio of the last
        // two, clamped to [0, 1]; on the reference's fixtures that predicts lambda to 6e-5..5e-4 relative, ten times better than a
        // linear extrapolation, and the increments then differ from those under the exact lambda by <= 3e-7 relative)
        st.lam_hist[2] = st.lam_hist[1];
        st.lam_hist[1] = st.lam_hist[0];
        st.lam_hist[0] = lam2;
        st.lam_n = (st.lam_n < 3) ? st.lam_n + 1 : 3;
        double next = lam2;
        if (st.lam_n >= 2) {
            const double d1 = st.lam_hist[0] - st.lam_hist[1];
            double q = 1.0;
            if (st.lam_n >= 3) {
                const double d0 = st.lam_hist[1] - st.lam_hist[2];
                q = (d0 != 0.0) ? d1 / d0 : 0.0;
                q = fmin(fmax(q, 0.0), 1.0);
            }
            next = st.lam_hist[0] + d1 * q;
        }
        *wparam = next;
    }
    if (p.weight_mode == 3) st.pad = 0;                                                  // ... warm-started from this iteration's lambda
    if ((p.weight_mode == 1 || p.weight_mode == 3) && p.irls_done != nullptr) p.irls_done[pair] = 0;
    if (p.weight_mode == 2) {
        const double sig = (n > 0.0) ? sqrt(rec[29] / n) : -1.0;
        st.lambda = (sig > 0.0) ? p.weight_param * sig : -1.0;
        *wparam = (st.lambda > 0.0) ? st.lambda : DVO_FLT_MAX;
    }
}

// The control step of k_gn_small, OUT OF LINE.  Inlined, it sits inside the level and iteration loops of the kernel, and the
// compiler moves everything loop-invariant in it out of them: the float64 literals of its polynomials (sincos, atan2: ~35
// register pairs; float64 literals cannot be instruction operands) and what it derives from the parameters - all held in vector
// registers across every record loop of the launch, i.e. spilled to scratch at the kernel's register cap and loaded back one by
// one on every iteration.  As a function it has no enclosing loop: the literals are built where they are used; what its ~165 live
// registers cost under the kernel's 128-register cap is spilled inside its own frame, by one lane.
__device__ __noinline__ void gn_control_small(const DvoControlParams* ctl, int pair, DvoPairState* st, float* T, int* active,
                                              double* wparam, double* rec, double* ws, double lambda0, double lam_known)
{
    gn_control(*ctl, 0, pair, *st, T, active, wparam, rec, ws, nullptr, nullptr, lambda0, 0, lam_known);
}

// rec[0..31] (LDS) = the pair's per-workgroup partials summed by one wave in sum_partials_block's order (see k_solve)
__device__ __forceinline__ void solve_sum_partials(const double* __restrict__ part, int nblk, double* rec, double (*tmp)[DVO_STRIDE])
{
    const int c = threadIdx.x & 31, h = threadIdx.x >> 5;
    // lane (c, h) keeps the running sums g = h, h+2, h+4, h+6 of column c; the loads of two rounds (8 per lane) are
    // issued before the first add, each sum still adds its terms in increasing b
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b0 = 0; b0 < nblk; b0 += 16) {
        double x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = b0 + (q >> 2) * 8 + h + 2 * (q & 3);
            x[q] = (b < nblk) ? part[(size_t)b * DVO_STRIDE + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = b0 + (q >> 2) * 8 + h + 2 * (q & 3);
            if (b < nblk) s4[q & 3] += x[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) tmp[h + 2 * q][c] = s4[q];
    __syncthreads();
    if (threadIdx.x < DVO_STRIDE) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += tmp[k][threadIdx.x];
        rec[threadIdx.x] = t;
    }
    __syncthreads();
}

// One WAVE per pair: the 64 lanes sum the pair's per-workgroup partials (same fixed order as sum_partials_block: eight
// interleaved running sums per column, then those in order, so the record is bit-identical to k_sum_partials'), then
// lane 0 runs the scalar control of one GN iteration.  A 64-thread workgroup occupies one SIMD slot instead of a whole CU
// (the control code is a one-lane float64 program at 1 wave/SIMD): the reductions of other streams keep running beside it.
__global__ __launch_bounds__(64) void k_solve(const DvoSolveParams p)
{
    const int pair = blockIdx.x;
    if (p.active[pair] == 0) return;
    // this wave is a serial chain on its stream's critical path, sharing its SIMD with four or five reduction waves of other
    // streams: its instructions go first whenever they are ready (the SIMD arbitrates by priority, then age)
    __builtin_amdgcn_s_setprio(3);
    __shared__ double rec[DVO_STRIDE];
    __shared__ double rec2[DVO_STRIDE];
    __shared__ double tmp[8][DVO_STRIDE];
    // the one-pass forms leave a second record: the corrected form all of it, the predicted form b1 in its slots 21..26 only
    const bool second = (p.partials2 != nullptr) && (p.phase == 0);
    const bool moments = second && !p.predicted;
    solve_sum_partials(p.partials + (size_t)pair * p.blocks_per_pair * DVO_STRIDE, p.blocks_per_pair, rec, tmp);
    if (second) solve_sum_partials(p.partials2 + (size_t)pair * p.blocks_per_pair * DVO_STRIDE, p.blocks_per_pair, rec2, tmp);
    if (p.scale1 != nullptr && p.phase == 0 && p.pair_flags != nullptr) {
        // Passes as written without a control launch between them (two scale passes, then the weighted pass): that IS the
        // reference's loop (weighter/t_weighter.py:24-32) when its first pass does not converge and its second does, or when the
        // first converges (student_t_lambda keeps lambda1 then).  Any other shape - a third pass - is flagged: the pair is solved
        // again by the persistent kernel, which runs the loop itself.  Sums in student_t_lambda's order, so the same lambdas.
        __shared__ double sg[16];
        if (threadIdx.x < 16) {
            const double* x = ((threadIdx.x < 8) ? p.scale1 : p.scale2) + (size_t)pair * p.blocks_per_pair;
            double s = 0.0;
            for (int b = threadIdx.x & 7; b < p.blocks_per_pair; b += 8) s += x[b];
            sg[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < 8; ++k) s1 += sg[k];
            for (int k = 0; k < 8; ++k) s2 += sg[8 + k];
            const double lam1 = 1.0 / s1, lamb2 = 1.0 / s2;
            if (!(fabs(lam1 - p.lambda0) < 1e-3) && !(fabs(lamb2 - lam1) < 1e-3) && rec[28] > 0.0) p.pair_flags[pair] |= 2;
        }
    }
    if (threadIdx.x != 0) return;
    gn_control(p, p.phase, pair, p.state[pair], p.T + (size_t)pair * 12, p.active + pair, p.wparam + pair, rec, &tmp[0][0],
               moments ? rec2 : nullptr, p.inexact, p.lambda0, (p.phase == 0) ? p.predicted : 0, -1.0, rec2 + 21);
}

hipError_t dvo_launch_solve(const DvoSolveParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(k_solve, dim3(p.n_pairs), dim3(64), 0, s, p);
    return hipGetLastError();
}

// ================================================================================================
// k_gn_small: persistent per-pair Gauss-Newton kernel for the small pyramid levels
// ================================================================================================
// One workgroup per frame pair runs EVERY iteration of up to DVO_SMALL_LEVELS consecutive coarse levels: per iteration
// the same per-record pipeline as k_reduce over the pair's valid-point list, a workgroup reduction, and the control step
// of base_robust_dvo.py:166-232 (gn_control, one lane) — no launch per iteration, no partial sums through HBM, no
// host-enqueued launches for pairs that have already stopped.  At 160x120 and below a level holds fewer records than one
// launch of k_reduce needs workgroups to fill the chip, and its iterations are a chain of ~12 us launches plus a control
// launch each: here an iteration costs the records' arithmetic plus two barriers.
// The Student-t IRLS (weighter/t_weighter.py:21-34) runs as the reference's loop (until |d lambda| < 1e-3, at most 50
// passes), the residuals of a pose kept between its passes.
#define DVO_SMALL_MAX_THREADS 1024
#define DVO_SMALL_MAX_WAVES (DVO_SMALL_MAX_THREADS / 64)

// Wave totals (float64) of the NT per-lane float32 sums -> out[0..31] (LDS).  Transposed through a wave-private LDS
// region eight terms at a time: lane l adds eight of the 64 values of term l & 7 in float64 (conflict-free reads), three
// shuffles join the eight segments.  Wave-synchronous: no workgroup barrier.
template <int NT>
__device__ __forceinline__ void wave_reduce_terms(const float* acc, float* wbuf /* [8][65] */, double* out)
{
    static_assert(NT == DVO_NACC, "the 30 sums of a normal-equation record");
    const int lane = threadIdx.x & 63;
    const int term = lane & 7, seg = lane >> 3;
#pragma unroll
    for (int c = 0; c < (NT + 7) / 8; ++c) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                    // loads of the previous chunk are done
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (c * 8 + i < NT) wbuf[i * 65 + lane] = acc[c * 8 + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double sgm = 0.0;
        const float* src = wbuf + term * 65 + seg * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) sgm += (double)src[k];
        sgm += __shfl_xor(sgm, 8, 64);
        sgm += __shfl_xor(sgm, 16, 64);
        sgm += __shfl_xor(sgm, 32, 64);
        if (seg == 0 && c * 8 + term < NT) out[c * 8 + term] = sgm;
    }
    if (NT < DVO_STRIDE && lane >= NT && lane < DVO_STRIDE) out[lane] = 0.0;
}

// ONE kernel for the three launched workgroup sizes (256 / 512 / 1024 threads; blockDim.x is read at run time): all of them are
// compiled for 4 waves per SIMD (128 VGPRs: what a 1024-thread workgroup allows anyway), so the code is the same and only the
// loop strides and the number of wave totals differ.  The per-wave transpose buffers (the one LDS array whose size depends on
// the shape: 2080 bytes per wave) are dynamic shared memory sized by dvo_launch_gn_small; the rest is sized for 16 waves (4.7 KB).
template <int VARIANT, int WMODE, bool SKEW>
__device__ __forceinline__ void gn_small_body(const DvoSmallParams& p)
{
    static_assert(VARIANT == 0 || VARIANT == 1 || VARIANT == 3, "V3 keeps the per-iteration launches");
    constexpr int NWAVES = DVO_SMALL_MAX_WAVES;
    __shared__ DvoPairState st;
    __shared__ float T_sh[12];
    __shared__ int active_sh;
    __shared__ DvoControlParams ctl;
    __shared__ double wpar_sh;
    __shared__ double rec_sh[DVO_STRIDE];
    __shared__ double wsum[NWAVES][DVO_STRIDE];      // (also the 72-double workspace of solve6_pinv)
    __shared__ double scal[2][NWAVES];
    __shared__ double scnt_sh[2][NWAVES];
    extern __shared__ float wbuf_dyn[];              // [waves launched][8 * 65]
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int lane_ = tid & 63, wave_ = tid >> 6;
    const int nwaves = (int)(blockDim.x >> 6);
    float* wbuf_mine = wbuf_dyn + wave_ * (8 * 65);
    if (tid == 0) {
        st = p.state[pair]; active_sh = 1; wpar_sh = p.wparam[pair];
        ctl.tolerance = p.tolerance; ctl.weight_param = p.weight_param; ctl.inv_sigma = p.inv_sigma;
        ctl.trace = p.trace; ctl.irls_done = nullptr; ctl.trace_row = p.trace_row; ctl.pair_flags = nullptr; ctl.st_tol = 0.0;
        ctl.max_increased = p.max_increased; ctl.weight_mode = p.weight_mode;
        ctl.trace_stride = p.trace_stride; ctl.trace_cap = p.trace_cap;
    }
    if (tid < 12) T_sh[tid] = p.T[(size_t)pair * 12 + tid];
    const int prev_id = p.prev_slot[pair];
    const uint8_t* prev = p.slots + (size_t)prev_id * p.slot_stride;
    const uint8_t* curs = p.slots + (size_t)p.cur_slot[pair] * p.slot_stride;
    const int gstride = (int)blockDim.x * DVO_PX_PER_THREAD;                      // records per group of the workgroup
    const int rec0 = wave_ * (64 * DVO_PX_PER_THREAD);
    // diagnostic stamps (experiment builds only, -DDVO_EXPERIMENTS + DVO_SMALL_STAMPS: workgroup-uniform branches around s_memtime)
#ifdef DVO_EXPERIMENTS
    const bool stamping = p.stamps != nullptr;
#else
    constexpr bool stamping = false;
#endif
    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_last = 0, t_begin = 0;
    auto stamp = [&](int k) {
        if (stamping) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tk[k] += now - t_last;
            t_last = now;
        }
    };
    if (stamping) { t_begin = __builtin_amdgcn_s_memtime(); t_last = t_begin; }

    for (int li = 0; li < p.n_levels; ++li) {
        const DvoLevelGeom& g = p.g[li];
        const DvoLevelCam& cam = p.cam[li];
        const uint32_t* packed_prev = reinterpret_cast<const uint32_t*>(prev + g.off_packed);
        const uint32_t* packed_cur = reinterpret_cast<const uint32_t*>(curs + g.off_packed);
        const DvoPointRec* list = reinterpret_cast<const DvoPointRec*>(prev + g.off_list);
        const int count = p.list_count[prev_id * DVO_LIST_LEVELS + p.level_id[li]];
        float* rc_mine = (p.rcache != nullptr) ? p.rcache + (size_t)pair * p.rcache_stride : nullptr;
        if (tid == 0) {
            // start of a pyramid level (k_level_begin): err_prev = FLT_MAX, err_increased_count = 0 (base_robust_dvo.py:158-159),
            // old = last_estimated_transform.copy() (:154)
            st.err_prev = DVO_FLT_MAX;
            st.increased = 0;
            st.active = 1;
            st.lam_n = 0;
            st.lam_err = 1.0;
            active_sh = 1;
            for (int k = 0; k < 4; ++k) st.oq[k] = st.lq[k];
            for (int k = 0; k < 3; ++k) st.ot[k] = st.lt[k];
            if (WMODE == 1) wpar_sh = p.lambda0;                                           // t_weighter.py:23 (both Student-t forms)
            if (WMODE == 2) wpar_sh = (st.lambda > 0.0) ? st.lambda : DVO_FLT_MAX;        // no scale yet: w = 1
        }
        __syncthreads();
        for (int it = 0; it < p.level_iters[li]; ++it) {
            if (active_sh == 0) break;                    // workgroup-uniform: written before the barrier that ended the last iteration
            // the pose is workgroup-uniform: keep it in SGPRs like k_reduce does (LDS loads land in VGPRs)
            float T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(T_sh[i])));
            float wpar = 0.0f;
            if constexpr (WMODE == 2) wpar = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)wpar_sh)));
            double lam_exact = -1.0;
            {
                if constexpr (WMODE == 1) {
                    {
                        // TDistributionWeighter.weight (weighter/t_weighter.py:21-34): lambda <- 1 / scale(lambda) until |d lambda| < 1e-3
                        // student_t_mean (weight_mode 3, "proper"): scale = mean, relative tolerance, at most 4 passes, warm-started from the
                        // previous iteration's lambda; otherwise the reference's loop (sum, |d lambda| < 1e-3, at most 50 passes, from 1/25)
                        const bool mean = p.student_t_mean != 0;
                        const int max_pass = mean ? 4 : 50;
                        double last = mean ? wpar_sh : p.lambda0, cur = last;
                        for (int pass = 0; pass < max_pass; ++pass) {
                            const float lamf = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)last)));
                            float sacc = 0.0f, scnt = 0.0f;
                            for (int seg = rec0; seg < count; seg += gstride) {          // wave-uniform bounds
                                float r[DVO_PX_PER_THREAD];
                                if (pass == 0 || rc_mine == nullptr) {
                                    uint2 rc[DVO_PX_PER_THREAD];
                                    PixelGeo geo[DVO_PX_PER_THREAD];
                                    PixelTaps taps[DVO_PX_PER_THREAD];
        #pragma unroll
                                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rc[j] = *reinterpret_cast<const uint2*>(list + seg + j * 64 + lane_);
        #pragma unroll
                                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                                        const bool live = (seg + j * 64 + lane_) < count;
                                        const unsigned pix = rc[j].y;
                                        dvo_geo<SKEW, VARIANT == 3>(g, cam, T, (float)(pix & 0xfffu), (float)((pix >> 12) & 0xfffu), __uint_as_float(rc[j].x), live, geo[j]);
                                        dvo_load_taps(g, packed_cur, geo[j], taps[j]);
                                        taps[j].own = 0u;
                                    }
        #pragma unroll
                                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                                        r[j] = __builtin_nanf("");
                                        if (geo[j].ok) {
                                            PixelOut o;
                                            dvo_finish<VARIANT, false>(cam, geo[j], taps[j], rc[j].y >> 24, o);
                                            r[j] = o.r;
                                        }
                                    }
                                    if (rc_mine != nullptr) {
        #pragma unroll
                                        for (int j = 0; j < DVO_PX_PER_THREAD; ++j)
                                            if (seg + j * 64 + lane_ < count) rc_mine[seg + j * 64 + lane_] = r[j];
                                    }
                                } else {
        #pragma unroll
                                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                                        const int idx = seg + j * 64 + lane_;
                                        r[j] = (idx < count) ? rc_mine[idx] : __builtin_nanf("");
                                    }
                                }
        #pragma unroll
                                for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                                    if (r[j] == r[j]) {
                                        const float r2 = r[j] * r[j];
                                        sacc += r2 * (6.0f * fast_rcp(fmaf(r2, lamf, 5.0f)));       // t_weighter.py:43-45, dof = 5
                                        scnt += 1.0f;
                                    }
                                }
                            }
                            const double sw = wave_sum((double)sacc), cw = wave_sum((double)scnt);
                            if (lane_ == 0) { scal[pass & 1][wave_] = sw; scnt_sh[pass & 1][wave_] = cw; }
                            __syncthreads();
                            double S = 0.0, N = 0.0;
                            for (int w = 0; w < nwaves; ++w) { S += scal[pass & 1][w]; N += scnt_sh[pass & 1][w]; }   // every thread, same order
                            cur = mean ? N / S : 1.0 / S;
                            if (mean ? (fabs(cur - last) <= 1e-3 * cur) : (fabs(cur - last) < 1e-3)) break;      // uniform
                            last = cur;
                        }
                        wpar = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)cur)));
                        if (mean && tid == 0) wpar_sh = cur;          // (every thread read wpar_sh before the first barrier of the loop)
                        if (!mean) lam_exact = cur;
                        stamp(4);
                    }
                }

                constexpr int NT = DVO_NACC;
                float acc[NT];
                // (an opaque zero: from the literal the compiler keeps a block of NT zeroed registers alive through the whole
                // iteration loop to copy from, and stores / reloads it around the control call)
                float zero = 0.0f;
                asm volatile("" : "+v"(zero));
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i] = zero;
                for (int seg = rec0; seg < count; seg += gstride) {                     // wave-uniform bounds
                    uint2 rc[DVO_PX_PER_THREAD];
                    PixelGeo geo[DVO_PX_PER_THREAD];
                    PixelTaps taps[DVO_PX_PER_THREAD];
#pragma unroll
                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) rc[j] = *reinterpret_cast<const uint2*>(list + seg + j * 64 + lane_);
#pragma unroll
                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                        const bool live = (seg + j * 64 + lane_) < count;
                        const unsigned pix = rc[j].y;
                        const unsigned xi = pix & 0xfffu, yi = (pix >> 12) & 0xfffu;
                        dvo_geo<SKEW, VARIANT == 3>(g, cam, T, (float)xi, (float)yi, __uint_as_float(rc[j].x), live, geo[j]);
                        dvo_load_taps(g, packed_cur, geo[j], taps[j]);
                        taps[j].own = 0u;
                        if (VARIANT == 1) taps[j].own = geo[j].ok ? packed_prev[yi * (unsigned)g.pitch + xi] : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < DVO_PX_PER_THREAD; ++j) {
                        if (geo[j].ok) {
                            PixelOut o;
                            dvo_finish<VARIANT, true>(cam, geo[j], taps[j], rc[j].y >> 24, o);
                            const float r = o.r;
                            const float r2 = r * r;
                            float wt = 1.0f;
                            if (WMODE == 1) wt = 6.0f * fast_rcp(fmaf(r2, wpar, 5.0f));              // t_weighter.py:34
                            if (WMODE == 2) { const float ar = fabsf(r); wt = (ar <= wpar) ? 1.0f : wpar * fast_rcp(ar); }
                            float wj[6];
#pragma unroll
                            for (int k = 0; k < 6; ++k) wj[k] = (WMODE == 0) ? o.J[k] : wt * o.J[k];
                            int q = 0;
#pragma unroll
                            for (int a = 0; a < 6; ++a)
#pragma unroll
                                for (int b = a; b < 6; ++b) { acc[q] = fmaf(wj[a], o.J[b], acc[q]); ++q; }
#pragma unroll
                            for (int k = 0; k < 6; ++k) acc[21 + k] = fmaf(wj[k], r, acc[21 + k]);
                            acc[27] = fmaf((WMODE == 0) ? r : wt * r, r, acc[27]);
                            acc[28] += 1.0f;
                            if (WMODE != 0) acc[29] += r2;
                        }
                    }
                }
                if constexpr (WMODE == 0) acc[DVO_NACC - 1] = acc[DVO_NACC - 3];
                stamp(0);
                wave_reduce_terms<NT>(acc, wbuf_mine, wsum[wave_]);
                __syncthreads();
                stamp(1);
                if (tid < DVO_STRIDE) {
                    double t = 0.0;
                    for (int w = 0; w < nwaves; ++w) t += wsum[w][tid];                  // fixed order
                    rec_sh[tid] = t;
                }
                __syncthreads();
                stamp(2);
                if (tid == 0) {
                    __builtin_amdgcn_s_setprio(3);          // the one-lane control chain goes first on its SIMD (see k_solve)
                    // (lam_exact: the lambda this iteration's IRLS loop ended on - the control step keeps the last ones of the level, from
                    // which the first per-iteration level above takes the guess its one-pass form starts from, see k_level_begin)
                    gn_control_small(&ctl, pair, &st, T_sh, &active_sh, &wpar_sh, rec_sh, &wsum[0][0], p.lambda0, lam_exact);
                    __builtin_amdgcn_s_setprio(0);
                }
                __syncthreads();
                stamp(3);
            }
        }
        __syncthreads();
    }
    if (tid == 0) { p.state[pair] = st; p.active[pair] = active_sh; p.wparam[pair] = wpar_sh; }
    if (tid < 12) p.T[(size_t)pair * 12 + tid] = T_sh[tid];
    if (stamping && tid == 0) {
        tk[5] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int k = 0; k < 6; ++k) p.stamps[(size_t)pair * 8 + k] += tk[k];
    }
}

// Compiled for 4 waves per SIMD (128 VGPRs).  The record loops and reductions need ~100 registers; the one-lane float64 control
// step, which needs ~165, is a function (gn_control_small) and spills inside its own frame.  For the narrow workgroups the budget
// is a choice, and a job-level one: beside a 128-register wave of this kernel a SIMD holds FOUR 96-register waves of another
// stream's reduction, beside a 168-register one three, and the reduction is saturated from four waves per SIMD up (168 -> 128:
// isolated launch unchanged, four-stream job +1.5 %; 253 registers: isolated launch -11 %, job -1 %).
template <int VARIANT, int WMODE, bool SKEW>
__global__ __launch_bounds__(DVO_SMALL_MAX_THREADS) void k_gn_small(const DvoSmallParams p)
{
    gn_small_body<VARIANT, WMODE, SKEW>(p);
}

// Instantiations: the general kernels (SKEW = true) for every variant and weighting, fast ones (SKEW = false) for every
// weighting of the reference default V1 and for the unweighted V2 / GN (as dvo_launch_reduce).
hipError_t dvo_launch_gn_small(const DvoSmallParams& p, int variant, int threads, hipStream_t s)
{
    if (threads != 256 && threads != 512 && threads != DVO_SMALL_MAX_THREADS) return hipErrorInvalidValue;
    const dim3 grid((unsigned)p.n_pairs), block((unsigned)threads);
    const size_t lds = (size_t)(threads / 64) * 8 * 65 * sizeof(float);
    const bool fast = p.cam[0].skew0 != 0;
    const int wm = (p.weight_mode == 0) ? 0 : (p.weight_mode == 1 || p.weight_mode == 3) ? 1 : 2;
#define DVO_SMALL_LAUNCH(V, W, S) hipLaunchKernelGGL((k_gn_small<V, W, S>), grid, block, lds, s, p)
#define DVO_SMALL_GENERAL(V)                                   \
    do {                                                       \
        if (wm == 0) DVO_SMALL_LAUNCH(V, 0, true);             \
        else if (wm == 1) DVO_SMALL_LAUNCH(V, 1, true);        \
        else DVO_SMALL_LAUNCH(V, 2, true);                     \
    } while (0)
    if (variant == 0) {
        if (fast) {
            if (wm == 0) DVO_SMALL_LAUNCH(0, 0, false);
            else if (wm == 1) DVO_SMALL_LAUNCH(0, 1, false);
            else DVO_SMALL_LAUNCH(0, 2, false);
        } else DVO_SMALL_GENERAL(0);
    } else if (variant == 1) {
        if (fast && wm == 0) DVO_SMALL_LAUNCH(1, 0, false);
        else DVO_SMALL_GENERAL(1);
    } else if (variant == 3) {
        if (fast && wm == 0) DVO_SMALL_LAUNCH(3, 0, false);
        else DVO_SMALL_GENERAL(3);
    } else return hipErrorInvalidValue;
#undef DVO_SMALL_GENERAL
#undef DVO_SMALL_LAUNCH
    return hipGetLastError();
}

__global__ void k_finish(const DvoPairState* st, int n_pairs, double* xi_out, int* iters_out, double* err_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    double xi[6];
    se3_log(st[i].q, st[i].t, xi);
    for (int k = 0; k < 6; ++k) xi_out[(size_t)i * 6 + k] = xi[k];
    iters_out[i] = st[i].iters;
    err_out[i] = st[i].err_prev;
}

hipError_t dvo_launch_finish(const DvoPairState* st, int n_pairs, double* xi_out, int* iters_out, double* err_out,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_finish, dim3((n_pairs + 63) / 64), dim3(64), 0, s, st, n_pairs, xi_out, iters_out, err_out);
    return hipGetLastError();
}

#ifdef DVO_EXPERIMENTS
// rejected formulations of the tap fetch (round 3), kept for A/B builds only: `make EXTRA=-DDVO_EXPERIMENTS OUT=...`
#include "dvo_exp_taps.inc"
#endif

// ================================================================================================
// Issue-rate probe (dvo_debug_valu_rate): what a VALU wave-instruction costs on THIS chip, now
// ================================================================================================
// The level-0 reduction is bound by VALU issue: its time is its instruction count times the cost of a
// wave-instruction at its occupancy.  This kernel measures that cost in the running process, so that bench.py can put the
// prediction beside the measured launch (roofline.valu_issue).  ALU only, explicit registers, values that stay finite, straight-line
// code (one backward branch per ~500 instructions).  Four instruction classes:
//   fast      v_fmac / v_mul / v_add / v_sub / v_and (and right shifts, v_or, v_mov_b32, v_add_u32: all full rate), operands in different VGPR banks
//   conflict  v_fmac whose two multiplicands sit in ONE bank (register number mod 4): +40-60 % (tools/ubench/vgpr_banks.hip); the
//             compiler does not allocate by bank, so a real kernel has some (tools/isa_count.py counts them)
//   slow      half rate: conversions, bit-field extract, compare, select, v_fract, LEFT shift, 64-bit move, min / max (tools/isa_count.py: SLOW)
//   rcp       v_rcp_f32 (quarter rate)
// mode 0 issues the composition of ONE TRIP of k_reduce<V1, unweighted>'s record loop as tools/isa_count.py reads it off the built
// library - DVO_PROBE_FAST / _CONFLICT / _SLOW / _RCP below; dvo_debug_valu_rate reports the composition it ran, bench.py compares it
// with the kernel that was built, tests/test_cabi_symbols.py with profiles/kernel_isa.json - modes 1..4 one class alone, mode 5 the
// composition again in another order (the order matters by ~8 %: tools/ubench/probe_orders.py).
// The launch places `waves_per_simd` 256-thread workgroups on every CU (dynamic LDS sized so that no CU takes more).
#define DVO_PROBE_FAST 360
#define DVO_PROBE_CONFLICT 60
#define DVO_PROBE_SLOW 109
#define DVO_PROBE_RCP 8
#define VR_CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","vcc"
// fast class: accumulators (a x 1 + 0: the values never move), multiplicands v24 = 1.0 (bank 0), v25 = 0.0 (bank 1), accumulators in
// banks 2 / 3 so that no source pair shares a bank
#define VR_F0(d) "v_fmac_f32 v" #d ", v24, v25\n"
#define VR_F1(d) "v_mul_f32 v" #d ", v24, v" #d "\n"
#define VR_F2(d) "v_add_f32 v" #d ", v25, v" #d "\n"
#define VR_F3(d) "v_sub_f32 v" #d ", v" #d ", v25\n"
#define VR_F4(d) "v_and_b32 v" #d ", v27, v" #d "\n"
#define VR_FAST8A VR_F0(2) VR_F0(3) VR_F1(6) VR_F0(7) VR_F0(10) VR_F2(11) VR_F0(14) VR_F0(15)
#define VR_FAST8B VR_F0(2) VR_F1(3) VR_F0(6) VR_F0(7) VR_F3(10) VR_F0(11) VR_F4(14) VR_F0(15)
// conflict class: both multiplicands in bank 0 (v24, v28)
#define VR_C(d) "v_fmac_f32 v" #d ", v24, v28\n"
#define VR_CONF4 VR_C(2) VR_C(3) VR_C(6) VR_C(7)
// slow class: destinations v16..v23, sources the constants v26 (an integer) / v24
#define VR_SLOW4A "v_cvt_f32_i32 v16, v26\n v_bfe_i32 v17, v26, 8, 12\n v_cvt_f32_ubyte0 v18, v26\n v_fract_f32 v19, v24\n"
#define VR_SLOW4B "v_cmp_le_u32 vcc, v26, v27\n v_cndmask_b32 v20, v26, v27, vcc\n v_lshlrev_b32 v21, 2, v26\n v_cvt_f32_i32 v22, v26\n"
#define VR_RCP "s_nop 0\n v_rcp_f32 v23, v24\n s_nop 0\n"            // as dvo_rcp: a scalar instruction on either side
#define VR_RCP0 "v_rcp_f32 v23, v24\n"                                     // (the bare instruction: placement experiments)
#define VR_R2(x) x x
#define VR_R4(x) VR_R2(x) VR_R2(x)
#define VR_R8(x) VR_R4(x) VR_R4(x)
#define VR_R16(x) VR_R8(x) VR_R8(x)
// Mode 0 arranges the composition the way the compiled loop is arranged (tools/isa_count.py prints the class sequence): per point a
// cluster of conversion-class instructions (the texel unpack), then the FMAs with the bank-conflicted ones among them, the reciprocals
// in between.  One point: 27 slow, 90 fast, 15 conflict, 2 rcp; a trip = four points (+ one more slow-class instruction: tools/isa_count.py
// counts the loop's strongly connected component, which holds the record registers' hand-over - four 64-bit moves - twice, once on the
// path a wave takes only when it runs out of records) = 360 fast, 60 conflict, 109 slow, 8 rcp.
#define VR_SLOW27 VR_R4(VR_SLOW4A) VR_SLOW4B VR_SLOW4A "v_cvt_f32_i32 v16, v26\n v_cvt_f32_i32 v17, v26\n v_cvt_f32_i32 v18, v26\n"
#define VR_FAST80 VR_R4(VR_FAST8A VR_FAST8B) VR_FAST8A VR_FAST8B
#define VR_CONF15 VR_CONF4 VR_CONF4 VR_CONF4 VR_C(2) VR_C(3) VR_C(6)
#define VR_CONF16 VR_R4(VR_CONF4)
#define VR_POINT_A VR_SLOW27 VR_FAST80 VR_RCP VR_F0(2) VR_F0(3) VR_F0(6) VR_F1(7) VR_F0(10) VR_F0(11) VR_F0(14) VR_F0(15) VR_F1(2) VR_F2(3) VR_CONF15 VR_RCP
#define VR_POINT_C "v_cvt_f32_i32 v16, v26\n" VR_POINT_A
#define VR_TRIP VR_POINT_A VR_POINT_A VR_POINT_A VR_POINT_C
static_assert(4 * 90 == DVO_PROBE_FAST && 4 * 15 == DVO_PROBE_CONFLICT && 4 * 27 + 1 == DVO_PROBE_SLOW && 4 * 2 == DVO_PROBE_RCP,
              "VR_TRIP must issue the composition the DVO_PROBE_* constants name");
// (mode 5, the same composition in small interleaved blocks: 16 fast + 8 slow + 4 conflict = 28;   16 fast + 1 rcp = 17)
#define VR_X1 VR_FAST8A VR_SLOW4A VR_FAST8B VR_SLOW4B VR_CONF4
#define VR_X2 VR_FAST8A VR_FAST8B VR_RCP
#define VR_TRIP_BLOCKS VR_R8(VR_X1) VR_R4(VR_X1) VR_X1 VR_CONF4 VR_CONF4 VR_SLOW4A "v_cvt_f32_i32 v16, v26\n" VR_R4(VR_X2) VR_X2 VR_R4(VR_FAST8A VR_FAST8B) VR_FAST8A VR_RCP VR_RCP VR_RCP
__global__ __launch_bounds__(DVO_BLOCK) void k_valu_rate(int mode, int iterations, float* sink)
{
    extern __shared__ float vr_lds[];                    // only sized: caps the workgroups per CU
    asm volatile("v_mov_b32 v24, 1.0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0x12345\n v_mov_b32 v27, -1\n v_mov_b32 v28, 0\n"
                 "v_mov_b32 v0, 1.0\n v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" ::: VR_CLOB);
    if (mode == 0) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_TRIP ::: VR_CLOB);
    } else if (mode == 1) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R2(VR_FAST8A VR_FAST8B)) ::: VR_CLOB);          // 512 fast
    } else if (mode == 2) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8(VR_CONF4)) ::: VR_CLOB);                     // 512 conflict
    } else if (mode == 3) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R4(VR_SLOW4A VR_SLOW4B)) ::: VR_CLOB);          // 512 slow
    } else if (mode == 5) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_TRIP_BLOCKS ::: VR_CLOB);      // the composition in small interleaved blocks
#ifdef DVO_EXPERIMENTS
    } else if (mode == 6) {
        // ... grouped per trip: 136 slow, 340 fast, 64 conflict, 8 rcp
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R2(VR_SLOW4A)) VR_SLOW4B VR_SLOW4B VR_R16(VR_FAST8A VR_FAST8B) VR_R4(VR_FAST8A VR_FAST8B) VR_FAST8A VR_FAST8B VR_F0(2) VR_F0(3) VR_F0(6) VR_F1(7) VR_R16(VR_CONF4) VR_R8(VR_RCP) ::: VR_CLOB);
    } else if (mode == 7) {
        // alternating single instructions: fast, slow, fast, fast, slow ... (the finest interleave)
#define VR_ALT VR_F0(2) "v_cvt_f32_i32 v16, v26\n" VR_F0(3) VR_F1(6) "v_bfe_i32 v17, v26, 8, 12\n" VR_F0(7) VR_F0(10) "v_cvt_f32_ubyte0 v18, v26\n" VR_F2(11) VR_F0(14) "v_fract_f32 v19, v24\n" VR_F0(15)
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R2(VR_ALT)) VR_R16(VR_CONF4) VR_R8(VR_RCP) ::: VR_CLOB);       // 256 fast + 128 slow + 64 conf + 8 rcp = 456
    } else if (mode == 22) {
        // per point, the conversion-class instructions spread between the FMAs (4 : 10) instead of one cluster of 34
#define VR_POINT_I VR_R8(VR_SLOW4A VR_FAST8A VR_F0(2) VR_F1(3)) "v_cvt_f32_i32 v16, v26\n v_cvt_f32_i32 v17, v26\n" VR_RCP VR_F0(2) VR_F0(3) VR_F0(6) VR_F1(7) VR_F0(10) VR_R4(VR_CONF4) VR_RCP
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R4(VR_POINT_I) ::: VR_CLOB);
    } else if (mode == 23) {
        // ... and the bank-conflicted FMAs spread too (2 per block)
#define VR_POINT_J VR_R8(VR_SLOW4A VR_FAST8A VR_F0(2) VR_F1(3) VR_C(6) VR_C(7)) "v_cvt_f32_i32 v16, v26\n v_cvt_f32_i32 v17, v26\n" VR_RCP VR_F0(2) VR_F0(3) VR_F0(6) VR_F1(7) VR_F0(10) VR_RCP
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R4(VR_POINT_J) ::: VR_CLOB);
    } else if (mode == 8) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B "v_rcp_f32 v23, v24\n") ::: VR_CLOB);                   // 256 fast, 16 rcp each followed by fast
    } else if (mode == 9) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B) VR_R16(VR_RCP0) ::: VR_CLOB);           // 256 fast, 16 rcp in a row
    } else if (mode == 10) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 VR_SLOW4A) ::: VR_CLOB);         // rcp followed by slow
    } else if (mode == 11) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_SLOW4A) VR_R16(VR_RCP0) ::: VR_CLOB);
    } else if (mode == 12) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 "s_nop 0\n") ::: VR_CLOB);       // rcp, one idle issue slot, fast
    } else if (mode == 13) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_SLOW4A) ::: VR_CLOB);                // reference for 10 / 11: no rcp
    } else if (mode == 14) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 "s_nop 1\n") ::: VR_CLOB);
    } else if (mode == 15) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 "s_nop 3\n") ::: VR_CLOB);
    } else if (mode == 16) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 "s_nop 7\n") ::: VR_CLOB);
    } else if (mode == 17) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B VR_RCP0 "s_mov_b32 s20, 0\n") ::: "s20", VR_CLOB);     // a scalar instruction after it
    } else if (mode == 18) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B "s_nop 0\n" VR_RCP0 "s_nop 0\n") ::: VR_CLOB);        // nop before and after
    } else if (mode == 19) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R8(VR_FAST8A VR_FAST8B VR_FAST8A VR_FAST8B VR_RCP0 VR_RCP0 "s_nop 0\n") ::: VR_CLOB);   // pairs of rcp + nop
    } else if (mode == 20) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B "v_rcp_f32 v23, v24\n v_nop\n") ::: VR_CLOB);         // v_nop after it
    } else if (mode == 21) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_FAST8A VR_FAST8B "v_rcp_f32 v23, v24\n v_cvt_f32_i32 v16, v26\n") ::: VR_CLOB);   // ONE slow op after it
    // 24..32: single instructions of the conversion class, 512 each (which of them are half rate?)
#define VR_ONE512(x) for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8(x x x x)) ::: VR_CLOB)
    } else if (mode == 24) { VR_ONE512("v_cvt_f32_i32 v16, v26\n");
    } else if (mode == 25) { VR_ONE512("v_bfe_i32 v17, v26, 8, 12\n");
    } else if (mode == 26) { VR_ONE512("v_ashrrev_i32 v18, 20, v26\n");
    } else if (mode == 27) { VR_ONE512("v_cvt_f32_ubyte0 v19, v26\n");
    } else if (mode == 28) { VR_ONE512("v_and_or_b32 v20, v26, v27, v24\n");
    } else if (mode == 29) { VR_ONE512("v_bfi_b32 v21, v27, v26, v24\n");
    } else if (mode == 30) { VR_ONE512("v_lshlrev_b32 v22, 2, v26\n");
    } else if (mode == 31) { VR_ONE512("v_fract_f32 v19, v24\n");
    } else if (mode == 32) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8("v_cmp_le_u32 vcc, v26, v27\n v_cndmask_b32 v20, v26, v27, vcc\n v_cmp_le_u32 vcc, v26, v27\n v_cndmask_b32 v21, v26, v27, vcc\n")) ::: VR_CLOB);
    // 40..: more single instructions, destinations rotating over v16..v19 (sources the constants v24 = 1.0, v26 = 0x12345, v27 = -1)
#define VR_ROT512(op, tail) for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8(op " v16, " tail "\n" op " v17, " tail "\n" op " v18, " tail "\n" op " v19, " tail "\n")) ::: VR_CLOB)
    } else if (mode == 40) { VR_ROT512("v_lshlrev_b32", "2, v26");
    } else if (mode == 41) { VR_ROT512("v_lshrrev_b32", "20, v26");
    } else if (mode == 42) { VR_ROT512("v_ashrrev_i32", "20, v26");
    } else if (mode == 43) { VR_ROT512("v_and_b32", "0xfff, v26");
    } else if (mode == 44) { VR_ROT512("v_and_b32", "v27, v26");
    } else if (mode == 45) { VR_ROT512("v_or_b32", "v27, v26");
    } else if (mode == 46) { VR_ROT512("v_mov_b32", "v26");
    } else if (mode == 47) {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8("v_mov_b64 v[16:17], v[26:27]\n v_mov_b64 v[18:19], v[26:27]\n v_mov_b64 v[20:21], v[26:27]\n v_mov_b64 v[22:23], v[26:27]\n")) ::: VR_CLOB);
    } else if (mode == 48) { VR_ROT512("v_add_u32", "v27, v26");
    } else if (mode == 49) { VR_ROT512("v_sub_f32", "v24, v25");
    } else if (mode == 50) { VR_ROT512("v_mul_f32", "v24, v25");
    } else if (mode == 51) { VR_ROT512("v_cvt_f32_u32", "v26");
    } else if (mode == 52) { VR_ROT512("v_cvt_u32_f32", "v24");
    } else if (mode == 53) { VR_ROT512("v_cvt_f32_ubyte3", "v26");
    } else if (mode == 54) { VR_ROT512("v_mad_u32_u24", "v26, v26, v27");
    } else if (mode == 55) { VR_ROT512("v_perm_b32", "v26, v27, v26");
    } else if (mode == 56) { VR_ROT512("v_max_f32", "v24, v25");
    } else if (mode == 57) { VR_ROT512("v_med3_f32", "v24, v25, v24");
    } else if (mode == 58) { VR_ROT512("v_fma_f32", "v24, v25, v24");
    } else if (mode == 59) { VR_ROT512("v_bfe_u32", "v26, 12, 12");
    } else if (mode == 60) { VR_ROT512("v_cvt_f32_i32", "v26");
    } else if (mode == 61) { VR_ROT512("v_fract_f32", "v24");
    } else if (mode == 62) { VR_ROT512("v_cvt_f32_f16", "v26");
    } else if (mode == 63) { VR_ROT512("v_cvt_f32_ubyte1", "v26");
    // 64..: squares and same-bank multiplicand pairs in other encodings (accumulators rotate over v2, v3, v6, v7)
#define VR_ACC512(op, tail) for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8(op " v2, " tail "\n" op " v3, " tail "\n" op " v6, " tail "\n" op " v7, " tail "\n")) ::: VR_CLOB)
    } else if (mode == 64) { VR_ACC512("v_fmac_f32", "v24, v24");
    } else if (mode == 65) { VR_ACC512("v_fmac_f32_dpp", "v24, v24 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf");
    } else if (mode == 66) { VR_ROT512("v_mul_f32", "v24, v24");
    } else if (mode == 67) { VR_ACC512("v_fmac_f32", "v24, v28");
    } else if (mode == 68) { VR_ACC512("v_fmac_f32_dpp", "v24, v28 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf");
    } else if (mode == 69) { VR_ACC512("v_fmac_f32", "v24, v25");
    } else if (mode == 70) { VR_ACC512("v_fmac_f32_dpp", "v24, v25 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf");
    } else if (mode == 71) { VR_ROT512("v_mul_f32_sdwa", "v24, v24 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD");
    } else if (mode == 33) {
        // the trip with the texel unpack done by mask-and-merge into a float (v_and_or_b32 + v_sub_f32, simple class) instead of
        // bit-field extract + conversion: per point 16 conversion-class instructions become 16 simple ones
#define VR_UNPACK_T "v_cvt_f32_ubyte0 v16, v26\n v_and_or_b32 v2, v26, v27, v24\n v_sub_f32 v3, v3, v25\n v_and_or_b32 v6, v26, v27, v24\n v_sub_f32 v7, v7, v25\n"
#define VR_POINT_M VR_R4(VR_UNPACK_T) VR_R2(VR_SLOW4A) "v_cvt_f32_i32 v16, v26\n v_bfe_i32 v17, v26, 8, 12\n v_cvt_f32_ubyte0 v18, v26\n v_fract_f32 v19, v24\n" VR_FAST80 VR_RCP VR_F0(2) VR_F0(3) VR_F0(6) VR_F1(7) VR_F0(10) VR_CONF16 VR_RCP
        for (int i = 0; i < iterations; ++i) asm volatile(VR_POINT_M VR_POINT_M VR_POINT_M "v_cvt_f32_i32 v16, v26\n" VR_POINT_M ::: VR_CLOB);
#endif
    } else {
        for (int i = 0; i < iterations; ++i) asm volatile(VR_R16(VR_R8(VR_RCP)) ::: VR_CLOB);                       // 128 rcp
    }
    float s;
    asm volatile("v_add_f32 %0, v2, v3\n v_add_f32 %0, %0, v6\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v23" : "=v"(s) :: VR_CLOB);
    if (sink != nullptr && s == 12345.678f) sink[blockIdx.x * DVO_BLOCK + threadIdx.x] = s + vr_lds[0];     // (never true: keeps the work alive)
}

// instructions one iteration of `mode` issues; composition of mode 0
int dvo_valu_rate_instructions(int mode, int* comp4)
{
    if (comp4) { comp4[0] = DVO_PROBE_FAST; comp4[1] = DVO_PROBE_CONFLICT; comp4[2] = DVO_PROBE_SLOW; comp4[3] = DVO_PROBE_RCP; }
    if (mode == 6 || mode == 22 || mode == 23) return 548;
    if (mode == 33) return 541;
    if (mode == 7) return 456;
    if (mode == 8 || mode == 9 || mode == 12 || (mode >= 14 && mode <= 19)) return 272;
    if (mode == 20 || mode == 21) return 288;
    if (mode == 10 || mode == 11) return 336;
    if (mode == 13) return 320;
    return (mode == 0 || mode == 5) ? DVO_PROBE_FAST + DVO_PROBE_CONFLICT + DVO_PROBE_SLOW + DVO_PROBE_RCP : (mode == 4) ? 128 : 512;
}

hipError_t dvo_launch_valu_rate(int mode, int iterations, int n_cus, int waves_per_simd, float* sink, hipStream_t s)
{
    // 160 KB of LDS per CU: a workgroup that asks for 160 KB / waves_per_simd (minus the allocation granule) leaves room for
    // exactly waves_per_simd of them
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 512;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_valu_rate), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_valu_rate, dim3((unsigned)(n_cus * waves_per_simd)), dim3(DVO_BLOCK), lds, s, mode, iterations, sink);
    return hipGetLastError();
}
