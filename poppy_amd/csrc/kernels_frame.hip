// kernels_frame.hip — per-frame HIP kernels of the morph hot path for gfx950 (CDNA4, wave64).
//
// Every kernel here is HBM/latency bound byte work: no MFMA anywhere (there is no dense float
// contraction on this path).  Arithmetic is bit-compatible with the reference's SSE3-baseline build:
//   * this file is compiled with -ffp-contract=off: every float multiply and add rounds on its own;
//   * divisions are IEEE (__fdiv_rn); float->int is round-half-even with the x86 "integer indefinite"
//     result for out-of-range inputs;
//   * where the reference's 4-lane SIMD body and scalar tail associate a sum differently, the split
//     point is reproduced per element (pyramids.cpp:380-402,503-521,848-855,897).
// Reference routines are cited per kernel (OCV = third/opencv-4.6.0/modules).
#include "kernels.h"
#include "pyramid_device.h"
#include <atomic>
#include "warp_device.h"
#include <hip/hip_ext.h>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace poppy_hip {

// ------------------------------------------------------------------------------------------------
// m2 = 1 - (0.114 B + 0.587 G + 0.299 R)      color_rgb.simd.hpp:594-643, matrix_expressions.cpp:1332
// ------------------------------------------------------------------------------------------------
__global__ void k_gray_inv(const float* __restrict__ g, float* __restrict__ m2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float b = g[3 * (size_t)i], gg = g[3 * (size_t)i + 1], r = g[3 * (size_t)i + 2];
    float gray = b * 0.114f + gg * 0.587f + r * 0.299f;
    m2[i] = 1.f - gray;
}
void launch_gray_inv(const float* gabor2, float* m2, int n_px, hipStream_t s) {
    hipLaunchKernelGGL(k_gray_inv, dim3((n_px + 255) / 256), dim3(256), 0, s, gabor2, m2, n_px);
}

// ------------------------------------------------------------------------------------------------
// Triangle-id raster.  One wave per triangle.
//   outline: Line(LINE_8) = clipLine + 8-connected Bresenham, x increasing   drawing.cpp:80-297
//   fill   : FillConvexPoly 16.16 edge walk, spans [(xl+.5)>>16, (xr+.5)>>16] drawing.cpp:1093-1255
// fillConvexPoly paints triangles one after another, so a pixel ends with the id of the LAST triangle
// that touches it; ids grow with the index, hence atomicMax.
// ------------------------------------------------------------------------------------------------
__device__ bool clip_segment(long long W, long long H, long long& x1, long long& y1, long long& x2, long long& y2) {
    long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a; c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a; c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// lanes walk the Bresenham steps of one segment in parallel: the minor-axis offset after k steps
// is the number of steps j < k whose running error was negative, which has the closed form below.
__device__ void outline_segment(int32_t* map, int W, int H, int ax, int ay, int bx, int by, int value, int lane) {
    if ((unsigned)ax >= (unsigned)W || (unsigned)bx >= (unsigned)W || (unsigned)ay >= (unsigned)H || (unsigned)by >= (unsigned)H) {
        long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
        if (!clip_segment(W, H, x1, y1, x2, y2)) return;
        ax = (int)x1; ay = (int)y1; bx = (int)x2; by = (int)y2;
    }
    int dx = bx - ax, dy = by - ay;
    int x0 = ax, y0 = ay;
    if (dx < 0) { dx = -dx; dy = -dy; x0 = bx; y0 = by; }
    int sMinor = 1;
    if (dy < 0) { dy = -dy; sMinor = -1; }
    const bool steep = dy > dx;
    int major = steep ? dy : dx, minor = steep ? dx : dy;
    // err_0 = major - 2*minor; step j adds -2*minor and, when err_j < 0, +2*major and a minor step.
    // minor offset after k steps: m_k = max(0, ceil((2*minor*k - major) / (2*major)))  (major > 0)
    // after clipping |coordinates| < 2^15, so 2*minor*k < 2^31: 32-bit arithmetic, and the quotient (< 2^15) comes from
    // a float reciprocal with one correction step instead of an integer division
    const int count = major + 1;
    const int den = 2 * major;
    const float inv = den > 0 ? 1.f / (float)den : 0.f;
    for (int k = lane; k < count; k += 64) {
        int m = 0;
        if (major > 0) {
            const int num = 2 * minor * k - major;
            if (num > 0) {
                const int a = num + den - 1;                      // ceil(num / den) = floor(a / den)
                m = (int)((float)a * inv);
                const int r = a - m * den;
                m += (r >= den) - (r < 0);
            }
        }
        int x, y;
        if (!steep) { x = x0 + k; y = y0 + sMinor * m; }
        else        { y = y0 + sMinor * k; x = x0 + m; }
        atomicMax(&map[(size_t)y * W + x], value);
    }
}

// work item = (triangle, chunk of kRasterChunkRows rows): 4 waves take rows round-robin, lanes split the span; or
// (triangle, -1): the triangle's outline, one segment per wave.  Outlines are items of their own (listed first) so
// that their longer dependent chain runs beside the fills instead of in front of a chunk's rows.  The fill edges come from the host plan
// (RasterTri, frame_plan.h): FillConvexPoly's two-chain walk (drawing.cpp:1164-1252) is decided there once per
// triangle, a row here is one 64-bit multiply-add per chain.
struct RasterTriDev {
    int ymin, ystop, n0, n1;
    int ybeg[4];
    long long ex[4], edx[4];
};

__global__ void __launch_bounds__(256) k_raster(const int* __restrict__ tri_xy, const RasterTriDev* __restrict__ edges,
                                                const int2* __restrict__ work, int n_work, int32_t* __restrict__ map, int W, int H, uint32_t id_base) {
    if ((int)blockIdx.x >= n_work) return;
    const int2 item = work[blockIdx.x];
    const int t = item.x, chunk = item.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int value = (int)id_base + t + 1;

    if (chunk < 0) {                   // outline item: (v2->v0), (v0->v1), (v1->v2), one segment per wave
        if (wave < 3) {
            int vx[3], vy[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { vx[i] = tri_xy[t * 6 + 2 * i]; vy[i] = tri_xy[t * 6 + 2 * i + 1]; }
            const int a = wave == 0 ? 2 : wave - 1, b = wave == 0 ? 0 : wave;
            outline_segment(map, W, H, vx[a], vy[a], vx[b], vy[b], value, lane);
        }
        return;
    }

    const RasterTriDev& r = edges[t];
    const int ymin = r.ymin, ystop = r.ystop;
    const bool two0 = r.n0 > 1, two1 = r.n1 > 1;
    const int yb00 = r.ybeg[0], yb01 = r.ybeg[1], yb10 = r.ybeg[2], yb11 = r.ybeg[3];
    const long long ex00 = r.ex[0], ex01 = r.ex[1], ex10 = r.ex[2], ex11 = r.ex[3];
    const long long dx00 = r.edx[0], dx01 = r.edx[1], dx10 = r.edx[2], dx11 = r.edx[3];

    const int y0 = ymin + chunk * kRasterChunkRows;
    const int y1 = min(y0 + kRasterChunkRows, ystop);
    for (int y = y0 + wave; y < y1; y += 4) {
        if (y < 0) continue;
        const bool s0 = two0 && y >= yb01, s1 = two1 && y >= yb11;
        const long long a = (s0 ? ex01 : ex00) + (long long)(y - (s0 ? yb01 : yb00)) * (s0 ? dx01 : dx00);
        const long long b = (s1 ? ex11 : ex10) + (long long)(y - (s1 ? yb11 : yb10)) * (s1 ? dx11 : dx10);
        const long long xl = a > b ? b : a, xr = a > b ? a : b;
        int xx1 = (int)((xl + 32768) >> 16), xx2 = (int)((xr + 32768) >> 16);
        if (xx2 >= 0 && xx1 < W) {
            if (xx1 < 0) xx1 = 0;
            if (xx2 >= W) xx2 = W - 1;
            int32_t* row = map + (size_t)y * W;
            for (int x = xx1 + lane; x <= xx2; x += 64) atomicMax(&row[x], value);
        }
    }
}
void launch_raster(const int* tri_xy, const void* edges, const int* work, int n_work, int32_t* triMap, int w, int h, uint32_t id_base, hipStream_t s) {
    if (n_work > 0)
        hipLaunchKernelGGL(k_raster, dim3(n_work), dim3(256), 0, s, tri_xy, (const RasterTriDev*)edges, (const int2*)work, n_work, triMap, w, h, id_base);
}

// ------------------------------------------------------------------------------------------------
// Fused create_map + remap (+ both sources).  algo.cpp:146-176, imgwarp.cpp:1197-1234,721-731,808-852
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_warp(int32_t* __restrict__ triMap, const float* __restrict__ inv1, const float* __restrict__ inv2,
                                              const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                              uint8_t* __restrict__ tr1, uint8_t* __restrict__ tr2, int W, int H, WarpExtras ex) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= W) return;
    size_t p = (size_t)y * W + x;
    int idx = decode_id((uint32_t)triMap[p], ex.id_base) - 1;
    const size_t po = (size_t)y * (ex.out_pitch > 0 ? ex.out_pitch : W) + x;      // the outputs' rows may be padded
    if (ex.m2) ex.mask[po] = mask_value(ex.m2[p], ex.alpha, ex.beta);
    float mx1 = (float)x, my1 = (float)y, mx2 = mx1, my2 = my1;
    if (idx >= 0) {
        map_point(inv1 + (size_t)idx * 9, x, y, mx1, my1);
        map_point(inv2 + (size_t)idx * 9, x, y, mx2, my2);
    }
    uint8_t o1[3], o2[3];
    sample3(c1, W, H, mx1, my1, o1);
    sample3(c2, W, H, mx2, my2, o2);
    tr1[po * 3] = o1[0]; tr1[po * 3 + 1] = o1[1]; tr1[po * 3 + 2] = o1[2];
    tr2[po * 3] = o2[0]; tr2[po * 3 + 1] = o2[1]; tr2[po * 3 + 2] = o2[2];
}
// --- 4 pixels per thread (W % 4 == 0) ---------------------------------------------------------------
// One 16-byte load of the id map, the two inverse matrices fetched once per run of equal ids, each 2x2
// footprint fetched as two unaligned 8-byte loads (6 useful bytes: two BGR pixels) when it lies inside the
// image, and 12 contiguous output bytes per image written as three dwords.  Source buffers carry 16 bytes of
// tail padding so the 8-byte loads of the last footprint stay inside the allocation.
__device__ __forceinline__ uint64_t ld_u64_unaligned(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

struct Tap { int w00, w01, w10, w11; int ix, iy; bool inside; };

__device__ __forceinline__ Tap make_tap(float mx, float my, int W, int H) {
    Tap t;
    int sx = cv_round_x86(mx * 32.f), sy = cv_round_x86(my * 32.f);
    bilinear_weights(sx & 31, sy & 31, t.w00, t.w01, t.w10, t.w11);
    t.ix = max(-32768, min(32767, sx >> 5)); t.iy = max(-32768, min(32767, sy >> 5));
    t.inside = t.ix >= 0 && t.ix < W - 1 && t.iy >= 0 && t.iy < H - 1;
    return t;
}

__device__ __forceinline__ uint32_t blend_tap(const Tap& t, uint64_t a, uint64_t b) {
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int v00 = (int)((a >> (8 * k)) & 255), v01 = (int)((a >> (24 + 8 * k)) & 255);
        int v10 = (int)((b >> (8 * k)) & 255), v11 = (int)((b >> (24 + 8 * k)) & 255);
        int acc = __mul24(v00, t.w00) + __mul24(v01, t.w01) + __mul24(v10, t.w10) + __mul24(v11, t.w11);
        out |= (uint32_t)sat_u8((acc + (1 << 14)) >> 15) << (8 * k);
    }
    return out;
}

// The kernel is written as straight-line phases (ids -> matrices -> coordinates -> ALL footprint loads -> blends)
// so that a thread has its 16 footprint loads in flight at once; the rare footprints that touch the image border
// are recomputed by the byte-wise path afterwards.  Block = 64 x 4: the four waves of a block work on four
// consecutive rows, which share source rows in the CU's L1.
__global__ void __launch_bounds__(256) k_warp4(int4* __restrict__ triMap4, const float* __restrict__ inv1, const float* __restrict__ inv2,
                                               const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                               uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H, WarpExtras ex) {
    const int W4 = W >> 2;
    const int q = blockIdx.x * 64 + threadIdx.x;               // index of the 4-pixel group inside the row
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (q >= W4 || y >= H) return;
    const size_t g = (size_t)y * W4 + q;
    const int4 ids = triMap4[g];
    if (ex.m2) {                                               // lbmask of the same four pixels
        const float4 m = ((const float4*)ex.m2)[g];
        ((float4*)ex.mask)[g] = make_float4(mask_value(m.x, ex.alpha, ex.beta), mask_value(m.y, ex.alpha, ex.beta),
                                            mask_value(m.z, ex.alpha, ex.beta), mask_value(m.w, ex.alpha, ex.beta));
    }
    const int id[4] = {decode_id((uint32_t)ids.x, ex.id_base) - 1, decode_id((uint32_t)ids.y, ex.id_base) - 1,
                       decode_id((uint32_t)ids.z, ex.id_base) - 1, decode_id((uint32_t)ids.w, ex.id_base) - 1};
    float mx[2][4], my[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = q * 4 + k;
        const size_t m = (size_t)max(id[k], 0) * 9;
        float h1[9], h2[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) { h1[e] = inv1[m + e]; h2[e] = inv2[m + e]; }
        float ax, ay, bx, by;
        map_point(h1, x, y, ax, ay);
        map_point(h2, x, y, bx, by);
        const bool mapped = id[k] >= 0;
        mx[0][k] = mapped ? ax : (float)x; my[0][k] = mapped ? ay : (float)y;
        mx[1][k] = mapped ? bx : (float)x; my[1][k] = mapped ? by : (float)y;
    }
    Tap t[2][4];
    uint64_t ra[2][4], rb[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[0][k] = make_tap(mx[0][k], my[0][k], W, H);
        t[1][k] = make_tap(mx[1][k], my[1][k], W, H);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const uint8_t* src = im ? c2 : c1;
            const size_t o = t[im][k].inside ? ((size_t)t[im][k].iy * W + t[im][k].ix) * 3 : 0;
            ra[im][k] = ld_u64_unaligned(src + o);
            rb[im][k] = ld_u64_unaligned(src + o + (size_t)W * 3);
        }
    }
    uint32_t p1[4], p2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p1[k] = blend_tap(t[0][k], ra[0][k], rb[0][k]);
        p2[k] = blend_tap(t[1][k], ra[1][k], rb[1][k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                               // footprints crossing the border: exact byte-wise path
        if (!t[0][k].inside) { uint8_t o[3]; sample3(c1, W, H, mx[0][k], my[0][k], o); p1[k] = o[0] | (o[1] << 8) | (o[2] << 16); }
        if (!t[1][k].inside) { uint8_t o[3]; sample3(c2, W, H, mx[1][k], my[1][k], o); p2[k] = o[0] | (o[1] << 8) | (o[2] << 16); }
    }
    // 4 BGR pixels -> 3 dwords
    const size_t o = ((size_t)y * W4 + q) * 3;
    tr1[o] = p1[0] | (p1[1] << 24);            tr2[o] = p2[0] | (p2[1] << 24);
    tr1[o + 1] = (p1[1] >> 8) | (p1[2] << 16); tr2[o + 1] = (p2[1] >> 8) | (p2[2] << 16);
    tr1[o + 2] = (p1[2] >> 16) | (p1[3] << 8); tr2[o + 2] = (p2[2] >> 16) | (p2[3] << 8);
}

// t0 / t1 (optional): events attached to the dispatch itself, i.e. the kernel's own begin and end timestamps — what a
// profiler's kernel trace reports — rather than markers queued around it.
void launch_warp(int32_t* triMap, const float* inv1, const float* inv2, const uint8_t* c1, const uint8_t* c2,
                 uint8_t* tr1, uint8_t* tr2, int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    if ((w & 3) == 0 && w >= 8 && h >= 2 && (ex.out_pitch == 0 || ex.out_pitch == w))
        hipExtLaunchKernelGGL(k_warp4, dim3((w / 4 + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, t0, t1, 0, (int4*)triMap, inv1, inv2,
                              c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, w, h, ex);
    else
        hipExtLaunchKernelGGL(k_warp, dim3((w + 255) / 256, h), dim3(256), 0, s, t0, t1, 0, triMap, inv1, inv2, c1, c2, tr1, tr2, w, h, ex);
}

// ------------------------------------------------------------------------------------------------
// Gaussian pyramid primitives, evaluated per output element.
//   pyrDown_: pyramids.cpp:745-900; float SIMD bodies :344-402 (H) and :503-521 (V)
//   pyrUp_  : pyramids.cpp:903-1005 (H scalar only; V SIMD == scalar bitwise)
// ------------------------------------------------------------------------------------------------

// kDownStrip vertically adjacent outputs of one element column per thread: the horizontal sums of the
// 2*kDownStrip + 3 source rows they share are computed once (the per-element form recomputes 5 rows per output).
// The H association depends only on the element column and the V association only on the element column too,
// so every output is the same expression tree as pyrdown_elem.  Strips that touch a border fall back to it.
template <bool U8, int V>
__device__ __forceinline__ void pyrdown_strip(const void* src, const DownGeom& g, int y0, int xe, float* __restrict__ dst) {
    const int cn = g.cn, dwe = g.dp * cn;                      // (row stride of the destination buffer)
    const int px = xe / cn, c = xe - px * cn;
    const bool interior = px >= 1 && 2 * px + 2 <= g.sw - 1 && y0 >= 1 && 2 * (y0 + V - 1) + 2 <= g.sh - 1;
    if (!interior) {
        for (int k = 0; k < V; ++k)
            if (y0 + k < g.dh) dst[(size_t)(y0 + k) * dwe + xe] = pyrdown_elem<U8>(src, g, y0 + k, xe);
        return;
    }
    const bool hBody = (xe >= cn) && (xe < g.hBodyEnd);
    const bool vBody = xe < g.vBodyEnd;
    const size_t rowlen = (size_t)g.sp * cn;
    const size_t col = (size_t)(2 * px - 2) * cn + c;
    float r[2 * V + 3];
#pragma unroll
    for (int k = 0; k < 2 * V + 3; ++k) {
        const size_t base = (size_t)(2 * y0 - 2 + k) * rowlen + col;
        float t0 = ld<U8>(src, base), t1 = ld<U8>(src, base + cn), t2 = ld<U8>(src, base + 2 * cn);
        float t3 = ld<U8>(src, base + 3 * cn), t4 = ld<U8>(src, base + 4 * cn);
        r[k] = hBody ? t2 * 6.f + ((t1 + t3) * 4.f + (t0 + t4))
                     : t2 * 6.f + (t1 + t3) * 4.f + t0 + t4;
    }
    const float s = 1.f / 256;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float *q = r + 2 * k;
        dst[(size_t)(y0 + k) * dwe + xe] = vBody ? ((q[1] + q[3] + q[2]) * 4.f + (q[0] + q[4] + (q[2] + q[2]))) * s
                                                : (q[2] * 6.f + (q[1] + q[3]) * 4.f + q[0] + q[4]) * s;
    }
}

// --- one reduction step, L / R / mask selected by blockIdx.z -------------------------------------
template <bool U8, int V>
__global__ void __launch_bounds__(256) k_pyrdown(const void* __restrict__ srcL, const void* __restrict__ srcR, const float* __restrict__ srcM,
                                                 float* __restrict__ dstL, float* __restrict__ dstR, float* __restrict__ dstM,
                                                 DownGeom g3, DownGeom g1) {
    const int which = blockIdx.z;
    const int xe = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = blockIdx.y * V;
    if (which == 2) {
        if (xe >= g1.dw) return;
        pyrdown_strip<false, V>(srcM, g1, y0, xe, dstM);
    } else {
        if (xe >= g3.dw * 3) return;
        if (which) pyrdown_strip<U8, V>(srcR, g3, y0, xe, dstR);
        else       pyrdown_strip<U8, V>(srcL, g3, y0, xe, dstL);
    }
}
template <bool U8, int V>
static void launch_pyrdown_v(const void* srcL, const void* srcR, const float* srcM, float* dstL, float* dstR, float* dstM,
                             const DownGeom& g3, const DownGeom& g1, hipStream_t s) {
    dim3 grid((g3.dw * 3 + 255) / 256, (g3.dh + V - 1) / V, 3);
    hipLaunchKernelGGL((k_pyrdown<U8, V>), grid, dim3(256), 0, s, srcL, srcR, srcM, dstL, dstR, dstM, g3, g1);
}
void launch_pyrdown(const void* srcL, const void* srcR, const float* srcM, bool src_u8,
                    float* dstL, float* dstR, float* dstM, int sw, int sh, hipStream_t s, const double* mask_ab, int sp, int mp, int dp) {
    if (launch_pyrdown_vec(srcL, srcR, srcM, src_u8, dstL, dstR, dstM, sw, sh, s, mask_ab, sp, mp, dp)) return;
    if (mask_ab) abort();                      // the caller asks for the m2 form only where pyr_level0_vec_ok holds
    DownGeom g3 = make_down_geom(sw, sh, 3, sp, dp), g1 = make_down_geom(sw, sh, 1, mp, dp);
    // taller strips share more row sums but leave fewer threads: only worth it when the level is large
    const size_t outputs = (size_t)g3.dw * g3.dh * 3;
    const int V = outputs >= (size_t)5000000 ? 4 : outputs >= (size_t)1000000 ? 2 : 1;
    if (src_u8) {
        if (V == 4) launch_pyrdown_v<true, 4>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
        else if (V == 2) launch_pyrdown_v<true, 2>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
        else launch_pyrdown_v<true, 1>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
    } else {
        if (V == 4) launch_pyrdown_v<false, 4>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
        else if (V == 2) launch_pyrdown_v<false, 2>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
        else launch_pyrdown_v<false, 1>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, s);
    }
}

// --- one collapse step ----------------------------------------------------------------------------

// Thread = one low-resolution element (sy, sxe = sx*3 + c) -> the 2x2 output quad above it, one channel.
// The 3x3 low-resolution neighbourhood of each of the three upsampled images (L, R, blended) is loaded once
// and serves all four outputs (the per-element form reloads it four times).  Interior quads take this path;
// quads on the first/last low-resolution row or column, where pyrUp uses its edge formulas, fall back to
// collapse_elem.  The arithmetic per output is the same expression tree as pyrup_elem + mix_lr.
struct UpQuad { float ee, eo, oe, oo; };     // (even row, even col), (even, odd), (odd, even), (odd, odd)

__device__ __forceinline__ UpQuad pyrup_quad(const float* __restrict__ p, int stride) {
    // p -> centre element of the 3x3 neighbourhood; +-3 = neighbouring pixel, +-stride = neighbouring row
    float he[3], ho[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* q = p + (r - 1) * stride;
        float a = q[-3], b = q[0], c = q[3];
        he[r] = a + b * 6.f + c;
        ho[r] = (b + c) * 4.f;
    }
    const float s = 1.f / 64;
    UpQuad u;
    u.ee = (he[0] + he[1] * 6.f + he[2]) * s;
    u.eo = (ho[0] + ho[1] * 6.f + ho[2]) * s;
    u.oe = ((he[1] + he[2]) * 4.f) * s;
    u.oo = ((ho[1] + ho[2]) * 4.f) * s;
    return u;
}

template <bool U8>
__global__ void __launch_bounds__(256) k_collapse(const void* __restrict__ gL, const void* __restrict__ gR, const float* __restrict__ gM,
                                                  const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                                  float* __restrict__ outB, int w, int h, int nw, int nh, CollapsePitch cp) {
    const int sxe = blockIdx.x * blockDim.x + threadIdx.x;
    const int sy = blockIdx.y;
    if (sxe >= nw * 3) return;
    const int sx = sxe / 3, c = sxe - sx * 3;
    const int x0 = 2 * sx, y0 = 2 * sy;
    // interior quads: the low-resolution 3 x 3 neighbourhood inside the coarser level AND all four outputs inside this one (an odd width's last
    // quad has one column)
    if (sx >= 1 && sx <= nw - 2 && sy >= 1 && sy <= nh - 2 && x0 + 1 < w && y0 + 1 < h) {
        const int ns = cp.n * 3;
        const size_t lo = (size_t)sy * ns + sxe;
        const UpQuad uL = pyrup_quad(nL + lo, ns), uR = pyrup_quad(nR + lo, ns), uB = pyrup_quad(nB + lo, ns);
        const size_t e00 = ((size_t)y0 * cp.g + x0) * 3 + c, e10 = e00 + (size_t)cp.g * 3;
        const size_t m00 = (size_t)y0 * cp.m + x0, m10 = m00 + cp.m;
        outB[e00]     = uB.ee + mix_lr(ld<U8>(gL, e00) - uL.ee,     ld<U8>(gR, e00) - uR.ee,     gM[m00]);
        outB[e00 + 3] = uB.eo + mix_lr(ld<U8>(gL, e00 + 3) - uL.eo, ld<U8>(gR, e00 + 3) - uR.eo, gM[m00 + 1]);
        outB[e10]     = uB.oe + mix_lr(ld<U8>(gL, e10) - uL.oe,     ld<U8>(gR, e10) - uR.oe,     gM[m10]);
        outB[e10 + 3] = uB.oo + mix_lr(ld<U8>(gL, e10 + 3) - uL.oo, ld<U8>(gR, e10 + 3) - uR.oo, gM[m10 + 1]);
    } else {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = x0 + dx, y = y0 + dy;
                if (x < w && y < h)
                    outB[((size_t)y * cp.g + x) * 3 + c] = collapse_elem<U8>(gL, gR, gM, nL, nR, nB, w, h, nw, nh, y, x * 3 + c, cp);
            }
    }
}
void launch_collapse(const void* gL, const void* gR, bool g_u8, const float* gM, const float* nL, const float* nR, const float* nB,
                     float* outB, int w, int h, int nw, int nh, hipStream_t s, const double* mask_ab, int gp, int mp, int np) {
    if (launch_collapse_vec(gL, gR, g_u8, gM, nL, nR, nB, outB, w, h, nw, nh, s, mask_ab, gp, mp, np)) return;
    if (mask_ab) abort();
    const CollapsePitch cp = make_collapse_pitch(w, nw, gp, mp, np);
    dim3 grid((nw * 3 + 255) / 256, nh);
    if (g_u8) hipLaunchKernelGGL(k_collapse<true>, grid, dim3(256), 0, s, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp);
    else      hipLaunchKernelGGL(k_collapse<false>, grid, dim3(256), 0, s, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp);
}

__global__ void k_lbmask(const float* __restrict__ m2, const double* __restrict__ ab, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = mask_value(m2[i], ab[0], ab[1]);
}
void launch_lbmask(const float* m2, const double* mask_ab, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_lbmask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m2, mask_ab, dst, n);
}

// Smallest-level mix of a pyramid too shallow for the LDS tail (blend.hpp:72-76: resultHighestLevel = left * mask + right * (1 - mask)
// with the mask replicated to three channels), straight from and to global memory.
__global__ void __launch_bounds__(256) k_mix_top(const float* __restrict__ l, const float* __restrict__ r, const float* __restrict__ m,
                                                 float* __restrict__ out, int n3) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n3) out[e] = mix_lr(l[e], r[e], m[e / 3]);
}
void launch_mix_top(const float* l, const float* r, const float* m, float* out, int n_px, hipStream_t s) {
    hipLaunchKernelGGL(k_mix_top, dim3((n_px * 3 + 255) / 256), dim3(256), 0, s, l, r, m, out, n_px * 3);
}

// ------------------------------------------------------------------------------------------------
// unsharp_mask(radius 1 -> 9 taps) + u8 conversion.            util.cpp:113-148, algo.cpp:263-265
//   row pass   : s = x0*k0; s = xk*kk + s                      filter.simd.hpp:1682-1730,2477-2487
//   column pass: s = ky0*c + 0; s = kyk*(S[+k] + S[-k]) + s    filter.simd.hpp:2753-2759
//   median 3x3 : 19-exchange sorting network, replicated edges median_blur.simd.hpp:692-713
//   norm       : sqrt of a double sum of squares               matx.hpp:929-932
// ------------------------------------------------------------------------------------------------
__constant__ float c_gauss9[9] = {0x1.18a9c4p-13f, 0x1.22724cp-8f, 0x1.ba4b9ap-5f, 0x1.ef8ebap-3f, 0x1.9884a4p-2f,
                                  0x1.ef8ebap-3f, 0x1.ba4b9ap-5f, 0x1.22724cp-8f, 0x1.18a9c4p-13f};

__global__ void __launch_bounds__(256) k_gauss_row(const float* __restrict__ src, float* __restrict__ dst, int W, int H) {
    const int xe = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xe >= W * 3) return;
    const int px = xe / 3, c = xe - px * 3;
    const float* row = src + (size_t)y * W * 3;
    float acc = row[reflect101(px - 4, W) * 3 + c] * c_gauss9[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) acc = row[reflect101(px - 4 + k, W) * 3 + c] * c_gauss9[k] + acc;
    dst[(size_t)y * W * 3 + xe] = acc;
}

__global__ void __launch_bounds__(256) k_gauss_col_diff(const float* __restrict__ src, const float* __restrict__ tmp, float* __restrict__ diff, int W, int H) {
    const int xe = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (xe >= W * 3) return;
    const size_t stride = (size_t)W * 3;
    float acc = c_gauss9[4] * tmp[y * stride + xe] + 0.f;
#pragma unroll
    for (int k = 1; k <= 4; ++k)
        acc = c_gauss9[4 + k] * (tmp[reflect101(y + k, H) * stride + xe] + tmp[reflect101(y - k, H) * stride + xe]) + acc;
    diff[y * stride + xe] = src[y * stride + xe] - acc;
}

__device__ __forceinline__ void mnmx(float& a, float& b) {
    float t = a;
    a = (b < a) ? b : a;
    b = (b < t) ? t : b;
}

__device__ __forceinline__ float median9(const float* __restrict__ d, int W, int H, int x, int y, int c) {
    if (W == 1 || H == 1) {     // 1-D special case of the reference (median_blur.simd.hpp:694-711)
        int len = W + H - 1, i = (H == 1) ? x : y;
        float p0 = d[(size_t)(i > 0 ? i - 1 : i) * 3 + c], p1 = d[(size_t)i * 3 + c], p2 = d[(size_t)(i < len - 1 ? i + 1 : i) * 3 + c];
        mnmx(p0, p1); mnmx(p1, p2); mnmx(p0, p1);
        return p1;
    }
    const int x0 = x > 0 ? x - 1 : x, x2 = x < W - 1 ? x + 1 : x;
    const float* r0 = d + (size_t)(y > 0 ? y - 1 : 0) * W * 3;
    const float* r1 = d + (size_t)y * W * 3;
    const float* r2 = d + (size_t)(y < H - 1 ? y + 1 : H - 1) * W * 3;
    float p0 = r0[x0 * 3 + c], p1 = r0[x * 3 + c], p2 = r0[x2 * 3 + c];
    float p3 = r1[x0 * 3 + c], p4 = r1[x * 3 + c], p5 = r1[x2 * 3 + c];
    float p6 = r2[x0 * 3 + c], p7 = r2[x * 3 + c], p8 = r2[x2 * 3 + c];
    mnmx(p1, p2); mnmx(p4, p5); mnmx(p7, p8); mnmx(p0, p1);
    mnmx(p3, p4); mnmx(p6, p7); mnmx(p1, p2); mnmx(p4, p5);
    mnmx(p7, p8); mnmx(p0, p3); mnmx(p5, p8); mnmx(p4, p7);
    mnmx(p3, p6); mnmx(p1, p4); mnmx(p2, p5); mnmx(p4, p7);
    mnmx(p4, p2); mnmx(p6, p4); mnmx(p4, p2);
    return p4;
}

__global__ void __launch_bounds__(256) k_median_apply(const float* __restrict__ src, const float* __restrict__ diff, uint8_t* __restrict__ out,
                                                      float* __restrict__ outF, int W, int H, float amount, float threshold) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float d0 = median9(diff, W, H, x, y, 0), d1 = median9(diff, W, H, x, y, 1), d2 = median9(diff, W, H, x, y, 2);
    double s = (double)d0 * (double)d0 + (double)d1 * (double)d1 + (double)d2 * (double)d2;
    const size_t p = ((size_t)y * W + x) * 3;
    float v0 = src[p], v1 = src[p + 1], v2 = src[p + 2];
    if (sqrt(s) >= (double)threshold) {
        v0 = v0 + amount * d0; v1 = v1 + amount * d1; v2 = v2 + amount * d2;
    }
    if (outF) { outF[p] = v0; outF[p + 1] = v1; outF[p + 2] = v2; }
    out[p] = sat_u8(cv_round_x86(v0 * 255.f + 0.f));
    out[p + 1] = sat_u8(cv_round_x86(v1 * 255.f + 0.f));
    out[p + 2] = sat_u8(cv_round_x86(v2 * 255.f + 0.f));
}

// Median of nine as three-input min / max / median instructions: sort the three columns, then
// med3(max of the minima, median of the medians, min of the maxima).  The reference's exchange network
// (median_blur.simd.hpp:692-713) returns the same number: the 5th smallest of nine finite values is unique.
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float median9_cols(float p0, float p1, float p2, float p3, float p4, float p5, float p6, float p7, float p8) {
    const float lo = max3f(min3f(p0, p3, p6), min3f(p1, p4, p7), min3f(p2, p5, p8));
    const float mi = med3f(med3f(p0, p3, p6), med3f(p1, p4, p7), med3f(p2, p5, p8));
    const float hi = min3f(max3f(p0, p3, p6), max3f(p1, p4, p7), max3f(p2, p5, p8));
    return med3f(lo, mi, hi);
}

// --- fused, LDS-tiled form -------------------------------------------------------------------------
// One workgroup produces a kUTx x kUTy pixel tile of the final 8-bit frame: the source tile with a halo of 5
// (4 for the 9-tap Gaussian + 1 for the 3x3 median) is staged in LDS once with reflect-101 addressing, the row
// pass, the column pass + difference and the median/threshold/apply run out of LDS.  HBM traffic drops from
// 87 B/px (three kernels through two f32 scratch images) to ~26 B/px read + 3 B/px written.  Every value is the
// same expression tree as in the three-kernel form above (kept for 1-pixel-wide images).
//
// All three LDS arrays are indexed [row][u] with ONE element coordinate u = 1 + (x - (tx0 - 5)) * 3 + channel, so a
// Gaussian tap is "u + 3k" and a column neighbour "row + k".  The leading pad element makes the staged row start on
// a 16-byte boundary of the source image (for W % 4 == 0), so interior tiles are staged with 16-byte loads; every
// later phase is register-blocked over 6 / 8 / 6 neighbouring elements so that one wide LDS read feeds several
// outputs (LDS read instructions per tile: ~16k instead of ~60k).
constexpr int kUTx = 32, kUTy = 16;                       // 32 x 32 tiles measured slower (2 workgroups per CU): 32.3 vs 28.4 us
constexpr int kUSy = kUTy + 10;                         // staged rows
constexpr int kUDy = kUTy + 2;                          // difference rows (halo 1 for the median)
// Row strides (floats).  For the lanes of an LDS lane group that straddle two rows to continue the bank pattern of one row
// (MI355X_MICROARCH, LDS: 8- and 16-byte reads go in groups of 32 / 16 lanes over 64 banks) the row pass wants an S stride of 108
// (mod 64) — 6 floats per lane, 18 lanes per row —, the column pass 2 x (R stride) = 104 (mod 64) — 4 floats per lane, 26 lanes per
// row pair —, the median a D stride of 96 (mod 64) — 6 floats per lane, 16 lanes per row.  R and D get theirs (116, 160); S would need
// 172, which is 30 KB per workgroup = 5 workgroups per CU: with 132 (25.8 KB, 6 per CU) the row pass pays its bank conflicts
// (model: 1099 instead of 889 conflict-adjusted LDS cycles per tile; 1407 with the round-2 strides) and the kernel is still faster:
// 24.9 -> 20.6 us at 1080p, 78.9 -> 75.1 us at 4K.  The 8-byte reads are kept from being merged into ds_read2_b64, which costs
// twice the cycles.
constexpr int kUSs = 132;                               // S row stride: >= 1 pad + 42 px * 3, multiple of 4 (see above)
constexpr int kURs = 116;                               // R row stride: >= 12 + 34 px * 3
constexpr int kUDs = 160;                               // D row stride (D takes R's place after the column pass)
static_assert(kUDy * kUDs <= kUSy * kURs, "the difference rows live in the row-pass buffer");
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) f2 lds_f2;      // an LDS read the compiler keeps as it is written

struct Col3 { float lo, mi, hi; };
__device__ __forceinline__ Col3 sort3(float a, float b, float c) { return Col3{min3f(a, b, c), med3f(a, b, c), max3f(a, b, c)}; }
__device__ __forceinline__ float median_of_cols(const Col3& a, const Col3& b, const Col3& c) {
    return med3f(max3f(a.lo, b.lo, c.lo), med3f(a.mi, b.mi, c.mi), min3f(a.hi, b.hi, c.hi));
}

__global__ void __launch_bounds__(256) k_unsharp_tile(const float* __restrict__ src, uint8_t* __restrict__ out, float* __restrict__ outF,
                                                      int W, int H, float amount_arg, const float* __restrict__ amount_ptr, double norm2_min, int SP, int stagger, int dbg) {
    // dbg (POPPY_UNSHARP_SKIP, a -DPOPPY_EXPERIMENTS build only; 0 otherwise): phases left out for timing — 1 row pass, 2 column pass, 4 the last phase's arithmetic, 8 the source loads, 16 the stores (wrong frames)
    // SP: pixels per row of src (the blended level 0: PyrLevel::pitch); the frame goes out tight
    if (dbg == 63) return;                                          // (timing: the launch alone)
    const float amount = amount_ptr ? *amount_ptr : amount_arg;     // per-frame value kept in HBM when the launch is a graph node
    __shared__ __attribute__((aligned(16))) float S[kUSy * kUSs];
    __shared__ __attribute__((aligned(16))) float R[kUSy * kURs];
    float* const D = R;                 // the difference rows take R's place once every thread has read its R rows (phase 3)
    const int tid = threadIdx.x;
    stagger_priority(blockIdx.x, stagger);
    const int tiles_x = (W + kUTx - 1) / kUTx;
    const int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tile_y = blk / tiles_x, tile_x = blk - tile_y * tiles_x;
    const int tx0 = tile_x * kUTx, ty0 = tile_y * kUTy;
    // 1. stage: S(r, u(col, c)) = src(reflect(ty0 - 5 + r), reflect(tx0 - 5 + col))
    const bool interior = (SP & 3) == 0 && tx0 >= 8 && tx0 + 38 <= W && ty0 >= 5 && ty0 + kUTy + 5 <= H;
    if (dbg & 8) {                                              // (timing: no source loads)
        for (int i = tid; i < kUSy * kUSs; i += 256) S[i] = (float)i * 1e-4f;
    } else if (interior) {
        const float* base = src + ((size_t)(ty0 - 5) * SP + (tx0 - 5)) * 3 - 1;     // 16-byte aligned: SP % 4 == 0, tx0 % 32 == 0
        for (int i = tid; i < kUSy * 32; i += 256) {
            const int r = i >> 5, v = i & 31;
            *(f4*)(S + r * kUSs + 4 * v) = *(const f4*)(base + (size_t)r * SP * 3 + 4 * v);
        }
    } else {
        for (int i = tid; i < kUSy * (kUTx + 10); i += 256) {
            const int r = i / (kUTx + 10), c = i - r * (kUTx + 10);
            const int yy = reflect101(ty0 - 5 + r, H), xx = reflect101(tx0 - 5 + c, W);
            const float* p = src + ((size_t)yy * SP + xx) * 3;
            float* d = S + r * kUSs + 1 + c * 3;
            d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
        }
    }
    __syncthreads();
    // 2. row pass.  R(r, u) = sum_k S(r, u - 12 + 3k) * g[k] for u in [13, 115); a thread takes 6 neighbouring u.
    for (int i = tid; i < kUSy * 18 && !(dbg & 1); i += 256) {
        const int r = i / 18, m = i - r * 18;
        // volatile: kept as 8-byte reads.  Merged into ds_read2_b64 they take twice the LDS cycles (2 x 4 groups of 16 lanes on a
        // 32-bank map instead of 2 groups of 32 on 64 banks each: MI355X_MICROARCH, LDS)
        const lds_f2* sp = (const lds_f2*)(S + r * kUSs + 6 * m);          // S(r, u0 - 12 ..), u0 = 12 + 6m
        f2 v[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) v[k] = sp[k];
        const float* f = (const float*)v;                            // f[o + 3k] = S(r, u0 + o - 12 + 3k)
        f2 acc[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            acc[o] = f2{f[2 * o], f[2 * o + 1]} * c_gauss9[0];
#pragma unroll
            for (int k = 1; k < 9; ++k) acc[o] = f2{f[2 * o + 3 * k], f[2 * o + 1 + 3 * k]} * c_gauss9[k] + acc[o];
        }
        f2* rp = (f2*)(R + r * kURs + 12 + 6 * m);
        rp[0] = acc[0]; rp[1] = acc[1]; rp[2] = acc[2];
    }
    __syncthreads();
    // 3. column pass + difference.  D(q, u) for q in [0, 18) <-> image row ty0 - 1 + q <-> S / R row q + 4.
    //    A thread takes 4 neighbouring u and 2 neighbouring rows: 10 R rows feed 8 outputs.
    static_assert(26 * (kUDy / 2) <= 256, "one item per thread: the results wait in registers for the barrier below");
    {
        const int i = tid;
        const bool has = i < 26 * (kUDy / 2) && !(dbg & 2);
        const int qi = i / 26, t = i - qi * 26;
        const int u0 = 12 + 4 * t, q0 = 2 * qi;
        f4 dd[2];
        if (has) {
            f4 rr[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) rr[k] = *(const f4*)(R + (q0 + k) * kURs + u0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f4 acc = c_gauss9[4] * rr[4 + j] + 0.f;
#pragma unroll
                for (int k = 1; k <= 4; ++k) acc = c_gauss9[4 + k] * (rr[4 + j + k] + rr[4 + j - k]) + acc;
                dd[j] = *(const f4*)(S + (q0 + j + 4) * kUSs + u0) - acc;
            }
        }
        __syncthreads();                                   // R has been read: D may overwrite it
        if (has) {
            *(f4*)(D + q0 * kUDs + u0) = dd[0];
            *(f4*)(D + (q0 + 1) * kUDs + u0) = dd[1];
        }
    }
    __syncthreads();
    // 4. median of the difference (replicated image edges), threshold, apply, convert.  A thread takes 2 neighbouring
    //    pixels of a row: 4 sorted columns per channel serve both medians.  Tiles that do not touch the left / right image edge
    //    (a workgroup-uniform test) skip the per-lane edge selects.
    auto phase4 = [&](auto edge_tag) {
        constexpr bool kEdge = decltype(edge_tag)::value;
        for (int it = 0; it < kUTy / 16; ++it) {
            const int ly = (tid >> 4) + 16 * it, pi = tid & 15;
            const int x0 = tx0 + 2 * pi, y = ty0 + ly;
            if ((kEdge && x0 >= W) || y >= H) continue;
            if (dbg & 16) { if (S[tid] == 12345.f) out[tid] = 1; continue; }                     // (timing: no stores)
            if (dbg & 4) { uint16_t* o16 = (uint16_t*)(out + ((size_t)y * W + x0) * 3); o16[0] = (uint16_t)tid; o16[1] = 0; o16[2] = 0; continue; }
            const bool has1 = !kEdge || x0 + 1 < W;
            const int qm = y > 0 ? ly : ly + 1, qc = ly + 1, qp = y < H - 1 ? ly + 2 : ly + 1;     // D rows
            f2 a[7], b[7], c[7];
            const lds_f2 *ap = (const lds_f2*)(D + qm * kUDs + 12 + 6 * pi), *bp = (const lds_f2*)(D + qc * kUDs + 12 + 6 * pi),
                         *cp = (const lds_f2*)(D + qp * kUDs + 12 + 6 * pi);   // 8-byte reads, see the row pass
#pragma unroll
            for (int k = 0; k < 7; ++k) { a[k] = ap[k]; b[k] = bp[k]; c[k] = cp[k]; }
            const float *fa = (const float*)a, *fb = (const float*)b, *fc = (const float*)c;    // index 1 + 3*col + ch, col 0..3 <-> x0-1 .. x0+2
            const lds_f2* sp = (const lds_f2*)(S + (ly + 5) * kUSs + 16 + 6 * pi);
            const f2 s01 = sp[0], s23 = sp[1], s45 = sp[2];
            const float sv[6] = {s01.x, s01.y, s23.x, s23.y, s45.x, s45.y};
            const bool left_edge = kEdge && x0 == 0, right_edge = kEdge && x0 + 1 >= W - 1;
            float d0[3], d1[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                Col3 cB = sort3(fa[4 + ch], fb[4 + ch], fc[4 + ch]);
                Col3 cC = sort3(fa[7 + ch], fb[7 + ch], fc[7 + ch]);
                Col3 cA = sort3(fa[1 + ch], fb[1 + ch], fc[1 + ch]);
                Col3 cD = sort3(fa[10 + ch], fb[10 + ch], fc[10 + ch]);
                if (kEdge) {
                    if (left_edge) cA = cB;
                    if (right_edge) cD = cC;
                    if (!has1) cC = cB;                              // x0 is the last column: its right neighbour is itself
                }
                d0[ch] = median_of_cols(cA, cB, cC);
                d1[ch] = median_of_cols(cB, cC, cD);
            }
            float v[6] = {sv[0], sv[1], sv[2], sv[3], sv[4], sv[5]};
            // |d|^2 >= norm2_min (the double sum of squares against the smallest double whose square root reaches the threshold).  As in k_unsharp_stream the float sum
            // decides unless it lies within 1e-6 of the bound (its error is below 3e-7 relative); only a wave that holds such a pixel forms the doubles (round 6: the ten
            // double-precision operations per thread were a quarter of this phase's arithmetic)
            const float t2f = (float)norm2_min;
            const float f0 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2], f1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
            bool sh0 = f0 > t2f, sh1 = f1 > t2f;
            if (__builtin_amdgcn_ballot_w64(fabsf(f0 - t2f) <= t2f * 1e-6f || fabsf(f1 - t2f) <= t2f * 1e-6f) != 0) {
                const double n0 = (double)d0[0] * (double)d0[0] + (double)d0[1] * (double)d0[1] + (double)d0[2] * (double)d0[2];
                const double n1 = (double)d1[0] * (double)d1[0] + (double)d1[1] * (double)d1[1] + (double)d1[2] * (double)d1[2];
                sh0 = n0 >= norm2_min; sh1 = n1 >= norm2_min;
            }
            if (sh0) { v[0] = v[0] + amount * d0[0]; v[1] = v[1] + amount * d0[1]; v[2] = v[2] + amount * d0[2]; }
            if (sh1) { v[3] = v[3] + amount * d1[0]; v[4] = v[4] + amount * d1[1]; v[5] = v[5] + amount * d1[2]; }
            const size_t p = ((size_t)y * W + x0) * 3;
            uint32_t o[6];
            // convertTo(CV_8U, 255): v * 255 + 0; the "+ 0" only turns -0 into +0, which rounds to the same byte
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = sat_u8(cv_round_x86(v[k] * 255.f));
            const int nv = has1 ? 6 : 3;
            if (outF)
                for (int k = 0; k < nv; ++k) outF[p + k] = v[k];
            if (!kEdge && !(W & 3) && !outF) {
                // tiles inside the image, rows of a multiple of 4 pixels (round 6): two neighbouring lanes hold 12 bytes = three dwords that begin on a dword — the even lane
                // stores the first two (its six bytes + the neighbour's first two, fetched by a DPP swap inside the lane pair), the odd lane the third.  One store per lane
                // instead of three 16-bit ones: the stores were 3.9 us of the kernel's 23.5 at 1080p (tools/experiments/unsharp_skip.sh)
                const uint32_t lo32 = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), hi16 = o[4] | (o[5] << 8);
                const uint32_t nb_lo16 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(lo32 & 0xffffu), 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]: the pair's other lane
                if (!(pi & 1)) *(uint2*)(out + p) = make_uint2(lo32, hi16 | (nb_lo16 << 16));
                else *(uint32_t*)(out + p + 2) = (lo32 >> 16) | (hi16 << 16);
            } else if (has1 && !(p & 1)) {                          // (even W): three aligned 16-bit stores
                uint16_t* o16 = (uint16_t*)(out + p);
                o16[0] = (uint16_t)(o[0] | (o[1] << 8)); o16[1] = (uint16_t)(o[2] | (o[3] << 8)); o16[2] = (uint16_t)(o[4] | (o[5] << 8));
            } else {
                for (int k = 0; k < nv; ++k) out[p + k] = (uint8_t)o[k];
            }
        }
    };
    if (tx0 > 0 && tx0 + kUTx < W) phase4(std::false_type{});
    else phase4(std::true_type{});
}

int stagger_flag(int bit) {
#ifdef POPPY_EXPERIMENTS
    static const int ab = getenv("POPPY_STAGGER_AB") ? atoi(getenv("POPPY_STAGGER_AB")) : 0;      // kernels that alternate, launch by launch
    static std::atomic<unsigned> calls[16];
    if ((ab >> bit) & 1) return (int)(calls[bit & 15].fetch_add(1) & 1u);
#endif
    return (POPPY_STAGGER >> bit) & 1;
}

void launch_unsharp(const float* src, float* tmpRow, float* diff, uint8_t* out_u8, float* out_f32_or_null,
                    int w, int h, float amount, const float* d_amount, float threshold, hipStream_t s, hipEvent_t done, int src_pitch) {
    if (src_pitch <= 0) src_pitch = w;
    if (w > 1 && h > 1) {
        // norm(d) >= threshold with norm = correctly rounded sqrt of a double: equivalent to |d|^2 >= x*, where x* is the
        // smallest double whose square root rounds to >= threshold (found here with the host's IEEE sqrt).
        const double t = (double)threshold;
        double x = t <= 0 ? 0.0 : t * t;
        if (t > 0) {
            while (x > 0 && std::sqrt(std::nextafter(x, 0.0)) >= t) x = std::nextafter(x, 0.0);
            while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
        }
        static const bool tile_only = getenv("POPPY_UNSHARP_TILE") != nullptr;
        static const int unsharp_dbg = poppy_experiment_env_i("POPPY_UNSHARP_SKIP");
        if (!tile_only && unsharp_stream_eligible(w, h)) { launch_unsharp_stream(src, out_u8, out_f32_or_null, w, h, amount, d_amount, x, s, done, src_pitch); return; }
        dim3 grid(((w + kUTx - 1) / kUTx) * ((h + kUTy - 1) / kUTy));
        hipExtLaunchKernelGGL(k_unsharp_tile, grid, dim3(256), 0, s, nullptr, done, 0, src, out_u8, out_f32_or_null, w, h, amount, d_amount, x, src_pitch, stagger_flag(1), unsharp_dbg);
        return;
    }
    dim3 ge((w * 3 + 255) / 256, h), gp((w + 255) / 256, h);
    hipLaunchKernelGGL(k_gauss_row, ge, dim3(256), 0, s, src, tmpRow, w, h);
    hipLaunchKernelGGL(k_gauss_col_diff, ge, dim3(256), 0, s, src, tmpRow, diff, w, h);
    hipLaunchKernelGGL(k_median_apply, gp, dim3(256), 0, s, src, diff, out_u8, out_f32_or_null, w, h, amount, threshold);
    if (done) (void)hipEventRecord(done, s);
}

// ------------------------------------------------------------------------------------------------
// Plan upload.  The per-frame plan (~200 KB) sits in pinned, device-mapped host memory; this kernel pulls it over
// PCIe into the slot's device copy.  A kernel instead of hipMemcpyAsync because of what the call costs the
// submitting host thread: 60-100 us per frame for the API call against ~5 us for a launch (profiles/r01_e_streams.md).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_upload(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
void launch_upload(const void* host_mapped, void* dst, size_t bytes, hipStream_t s) {
    const int n16 = (int)((bytes + 15) / 16);
    if (n16 > 0) hipLaunchKernelGGL(k_upload, dim3((n16 + 255) / 256), dim3(256), 0, s, (const uint4*)host_mapped, (uint4*)dst, n16);
}

// ------------------------------------------------------------------------------------------------
// u8 addWeighted fallback: dst = sat(round(a*wa + b*wb))          arithm.simd.hpp:131-135,1705-1755
// ------------------------------------------------------------------------------------------------
__global__ void k_dissolve(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* __restrict__ dst, size_t n, float wa, float wb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = (float)a[i] * wa + ((float)b[i] * wb + 0.f);
    dst[i] = sat_u8(cv_round_x86(v));
}
void launch_dissolve(const uint8_t* a, const uint8_t* b, uint8_t* dst, size_t n, float wa, float wb, hipStream_t s) {
    hipLaunchKernelGGL(k_dissolve, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, dst, n, wa, wb);
}

}  // namespace poppy_hip
