// pyramid_device.h — device-side primitives shared by the pyramid kernels (kernels_frame.hip, kernels_pyramid_vec.hip).
// Per-element forms of pyrDown / pyrUp / the Laplacian mix, written so that every output is one fixed expression
// tree equal to the reference's (OCV/imgproc/src/pyramids.cpp, src/blend.hpp); the wide kernels reuse them for
// border handling and as the definition they must match.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>

namespace poppy_hip {

// Workgroups are handed to the 8 XCDs round-robin (hardware block b runs on XCD b % 8) and each XCD has its own L2.
// Kernels whose neighbouring tiles share halo rows / columns renumber their blocks with this bijection of [0, n) so
// that the blocks of one XCD form one contiguous run of tiles: the halo is then an L2 hit instead of a second trip
// to the memory side.
__device__ __forceinline__ int xcd_swizzle(int b, int n) {
    const int per = n >> 3, rem = n & 7;
    const int x = b & 7, s = b >> 3;
    return x * per + (x < rem ? x : rem) + s;
}

// Wave priorities against lock step (round 6).  The workgroups of a launch that fits the chip in one round start together, and the waves that share a
// SIMD then move through the kernel's phases together — all wait for their loads, all compute, all store —, so memory time and arithmetic add up
// instead of overlapping.  Giving every second workgroup OF A COMPUTE UNIT the high priority (hardware block b runs on XCD b % 8 and, there, on CU
// (b / 8) % 32: the co-resident workgroups of a CU differ in b >> 8) lets those run ahead: k_warp_bin 17.6 -> 14.6 us at 1080p, bit-identical
// (profiles/r06_notes.md section 2; at 4K, four rounds, the workgroups are out of step by themselves and it changes nothing; changing the priority in
// mid-kernel, or a start offset by s_sleep, was worth nothing or cost a factor).  POPPY_STAGGER = the kernels that do it, one bit each (timing builds:
// tools/experiments/stagger_build.sh):  1 k_warp_bin (one-round grids), 2 k_unsharp_tile, 4 k_pyrdown_level<true>, 8 k_collapse_level<true>, 16 k_collapse_cone,
// 128 k_unsharp_stream, 256 k_warp_bin (larger grids)
// Shipped: 1 + 4 + 8 + 16 + 256 — in-process A/B, on / off (profiles/r06_notes.md section 3): k_warp_bin 1080p 15.0 / 17.4 us (4K 44.1 / 44.5), k_collapse_level<true> 21.7 / 22.9
// (4K 66.0 / 67.4), k_pyrdown_level<true> 4K 41.5 / 42.3, k_collapse_cone 13.7 / 13.9; k_unsharp_tile is SLOWER with it (24.5 / 23.5), k_unsharp_stream indifferent.
#ifndef POPPY_STAGGER
#define POPPY_STAGGER 285
#endif
// `on`: a kernel argument from stagger_flag(bit) (kernels.h) — the bit of POPPY_STAGGER, or, in a -DPOPPY_EXPERIMENTS build under POPPY_STAGGER_AB=<mask>,
// on and off launch by launch, so that ONE traced process holds both forms of a kernel side by side (tools/experiments/stagger_ab.py splits a kernel's dispatches
// by parity; between processes the same kernel's average moves by +-10 % on these boxes, more than most of the effects looked for)
__device__ __forceinline__ void stagger_priority(unsigned hw_block, int on) {
    if (on && ((hw_block >> 8) & 1u)) __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

// cvRound(float) on x86 (cvtss2si): half-to-even; NaN / |v| >= 2^31 -> 0x80000000
__device__ __forceinline__ int cv_round_x86(float v) {
    return (fabsf(v) < 2147483648.f) ? __float2int_rn(v) : INT_MIN;
}
__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

constexpr float kInv255 = (float)(1.0 / 255.0);

// lbmask = clamp((1-mr) - m2*mr)            arithm.simd.hpp:1160-1216,1808 (double, one rounding)
__device__ __forceinline__ float mask_value(float m2, double alpha, double beta) {
    double t = (double)m2 * beta + 0.0;
    float v = (float)(1.0 * alpha + t);
    if (v < 0.f) v = 0.f;
    if (v > 1.f) v = 1.f;
    return v;
}

// How the level-0 blend kernels see the blend mask: either the materialised lbmask (ab == nullptr: the values are used as
// they are) or the pair's m2 field with the frame's (alpha, beta) — then every loaded value goes through mask_value, and
// the warp kernel does not have to write lbmask (4 B/px written + 4 B/px read less per frame).
struct MaskSource {
    double alpha, beta;
    bool lazy;
    __device__ __forceinline__ float operator()(float raw) const { return lazy ? mask_value(raw, alpha, beta) : raw; }
};
__device__ __forceinline__ MaskSource mask_source(const double* __restrict__ ab) {
    MaskSource m;
    m.lazy = ab != nullptr;
    m.alpha = m.lazy ? ab[0] : 0.0;
    m.beta = m.lazy ? ab[1] : 0.0;
    return m;
}
struct MaskPlain { __device__ __forceinline__ float operator()(float raw) const { return raw; } };

template <bool U8> __device__ __forceinline__ float ld(const void* p, size_t i) {
    if (U8) return (float)((const uint8_t*)p)[i] * kInv255;      // convertTo(CV_32F, 1/255): v*a + 0
    return ((const float*)p)[i];
}

struct DownGeom {            // host-computed constants of one pyrDown (source sw x sh, cn channels)
    int sw, sh, dw, dh, cn;
    int sp, dp;              // pixels per row of the source / destination BUFFER (>= sw / dw: kernels.h PyrLevel::pitch)
    int w0;                  // width0 in pixels: columns reachable without the right border table
    int hBodyEnd;            // element index where the SIMD-body association of the H pass stops
    int vBodyEnd;            // same for the V pass: (dw*cn/4)*4
};

__host__ __device__ inline DownGeom make_down_geom(int sw, int sh, int cn, int sp = 0, int dp = 0) {
    DownGeom g;
    g.sw = sw; g.sh = sh; g.cn = cn; g.dw = (sw + 1) / 2; g.dh = (sh + 1) / 2;
    g.sp = sp > 0 ? sp : sw; g.dp = dp > 0 ? dp : g.dw;
    int w0 = (sw - 3) / 2 + 1;                 // C division truncates toward zero, as in the reference
    g.w0 = w0 < g.dw ? w0 : g.dw;
    int width = g.w0 * cn - cn;                // elements offered to the SIMD body (starts after pixel 0)
    int covered = 0;
    if (width >= 4) covered = (cn == 1) ? ((width - 4) / 4 + 1) * 4 : ((width - 4) / 3 + 1) * 3;
    g.hBodyEnd = cn + covered;
    g.vBodyEnd = (g.dw * cn / 4) * 4;
    return g;
}

template <bool U8>
__device__ __forceinline__ float pyrdown_elem(const void* src, const DownGeom& g, int y, int xe) {
    const int cn = g.cn;
    const int px = xe / cn, c = xe - px * cn;
    const bool hBody = (xe >= cn) && (xe < g.hBodyEnd);
    int col[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) col[k] = reflect101(2 * px + k - 2, g.sw) * cn + c;
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int sy = reflect101(2 * y + k - 2, g.sh);
        size_t base = (size_t)sy * g.sp * cn;
        float t0 = ld<U8>(src, base + col[0]), t1 = ld<U8>(src, base + col[1]), t2 = ld<U8>(src, base + col[2]);
        float t3 = ld<U8>(src, base + col[3]), t4 = ld<U8>(src, base + col[4]);
        r[k] = hBody ? t2 * 6.f + ((t1 + t3) * 4.f + (t0 + t4))
                     : t2 * 6.f + (t1 + t3) * 4.f + t0 + t4;
    }
    const float s = 1.f / 256;
    return (xe < g.vBodyEnd) ? ((r[1] + r[3] + r[2]) * 4.f + (r[0] + r[4] + (r[2] + r[2]))) * s
                             : (r[2] * 6.f + (r[1] + r[3]) * 4.f + r[0] + r[4]) * s;
}

// horizontal pyrUp value of source row `row` at destination element dxe
__device__ __forceinline__ float pyrup_h(const float* __restrict__ row, int sw, int cn, int dxe) {
    const int dpx = dxe / cn, c = dxe - dpx * cn;
    const int spx = dpx >> 1;
    const bool odd = dpx & 1;
    if (sw == 1) return row[c] * 8.f;
    if (spx == 0) {
        float s0 = row[c], s1 = row[cn + c];
        return odd ? (s0 + s1) * 4.f : s0 * 6.f + s1 * 2.f;
    }
    if (spx >= sw - 1) {
        float sm = row[(sw - 2) * cn + c], s0 = row[(sw - 1) * cn + c];
        return odd ? s0 * 8.f : sm + s0 * 7.f;
    }
    float sm = row[(spx - 1) * cn + c], s0 = row[spx * cn + c], sp = row[(spx + 1) * cn + c];
    return odd ? (s0 + sp) * 4.f : sm + s0 * 6.f + sp;
}

// (sp: pixels per row of the source buffer, 0 = sw)
__device__ __forceinline__ float pyrup_elem(const float* __restrict__ src, int sw, int sh, int cn, int dy, int dxe, int sp = 0) {
    const int sy = dy >> 1;
    const size_t stride = (size_t)(sp > 0 ? sp : sw) * cn;
    const float s = 1.f / 64;
    if (dy & 1) {
        int syp = reflect101((sy + 1) * 2, sh * 2) >> 1;
        float r1 = pyrup_h(src + sy * stride, sw, cn, dxe), r2 = pyrup_h(src + syp * stride, sw, cn, dxe);
        return ((r1 + r2) * 4.f) * s;
    }
    int sym = reflect101((sy - 1) * 2, sh * 2) >> 1, syp = reflect101((sy + 1) * 2, sh * 2) >> 1;
    float r0 = pyrup_h(src + sym * stride, sw, cn, dxe), r1 = pyrup_h(src + sy * stride, sw, cn, dxe);
    float r2 = pyrup_h(src + syp * stride, sw, cn, dxe);
    return (r0 + r1 * 6.f + r2) * s;
}

// blend of one Laplacian level (blend.hpp:67-77): A = lapL*m; B = lapR*(1-m); A + B
__device__ __forceinline__ float mix_lr(float l, float r, float m) {
    float a = l * m;
    float anti = 1.f - m;
    float b = r * anti;
    return a + b;
}

// pixels per row of the buffers of one collapse step: the Gaussian level's images (and the output), its mask, the coarser level
struct CollapsePitch { int g, m, n; };
__host__ __device__ inline CollapsePitch make_collapse_pitch(int w, int nw, int gp = 0, int mp = 0, int np = 0) {
    CollapsePitch p;
    p.g = gp > 0 ? gp : w; p.m = mp > 0 ? mp : w; p.n = np > 0 ? np : nw;
    return p;
}

template <bool U8>
__device__ __forceinline__ float collapse_elem(const void* gL, const void* gR, const float* gM, const float* nL, const float* nR,
                                               const float* nB, int w, int h, int nw, int nh, int y, int xe, CollapsePitch cp) {
    const size_t i = (size_t)y * cp.g * 3 + xe;
    const float m = gM[(size_t)y * cp.m + xe / 3];
    float lapL = ld<U8>(gL, i) - pyrup_elem(nL, nw, nh, 3, y, xe, cp.n);
    float lapR = ld<U8>(gR, i) - pyrup_elem(nR, nw, nh, 3, y, xe, cp.n);
    float res = mix_lr(lapL, lapR, m);
    return pyrup_elem(nB, nw, nh, 3, y, xe, cp.n) + res;
}
template <bool U8>
__device__ __forceinline__ float collapse_elem(const void* gL, const void* gR, const float* gM, const float* nL, const float* nR,
                                               const float* nB, int w, int h, int nw, int nh, int y, int xe) {
    return collapse_elem<U8>(gL, gR, gM, nL, nR, nB, w, h, nw, nh, y, xe, make_collapse_pitch(w, nw));
}

// ---- branch-free variants for levels of at least 3 x 3 pixels ------------------------------------------------
// Same expression trees as above; only the ADDRESSING is restated without data-dependent control flow (one
// reflection suffices when the level is at least 3 wide/high), so all loads of an output can be in flight at once.
// They serve the border frames of the wide kernels, where a few thousand outputs would otherwise sit on a long
// chain of dependent branches and loads.
__device__ __forceinline__ int reflect101_once(int p, int len) { return p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p); }

template <bool U8, int CN, typename F = MaskPlain>
__device__ __forceinline__ float pyrdown_elem_wide(const void* src, const DownGeom& g, int y, int xe, F fn = F()) {
    constexpr int cn = CN;                                   // == g.cn; a constant keeps xe / cn off the division path
    const int px = xe / cn, c = xe - px * cn;
    const bool hBody = (xe >= cn) && (xe < g.hBodyEnd);
    int col[5];
    size_t rowo[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        col[k] = reflect101_once(2 * px + k - 2, g.sw) * cn + c;
        rowo[k] = (size_t)reflect101_once(2 * y + k - 2, g.sh) * g.sp * cn;
    }
    float t[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int m = 0; m < 5; ++m) t[k][m] = fn(ld<U8>(src, rowo[k] + col[m]));
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
        r[k] = hBody ? t[k][2] * 6.f + ((t[k][1] + t[k][3]) * 4.f + (t[k][0] + t[k][4]))
                     : t[k][2] * 6.f + (t[k][1] + t[k][3]) * 4.f + t[k][0] + t[k][4];
    const float s = 1.f / 256;
    return (xe < g.vBodyEnd) ? ((r[1] + r[3] + r[2]) * 4.f + (r[0] + r[4] + (r[2] + r[2]))) * s
                             : (r[2] * 6.f + (r[1] + r[3]) * 4.f + r[0] + r[4]) * s;
}

// horizontal pyrUp value from three unconditionally loaded neighbours (sw >= 2)
__device__ __forceinline__ float pyrup_h_wide(const float* __restrict__ row, int sw, int dxe) {
    const int dpx = dxe / 3, c = dxe - dpx * 3;
    const int spx = dpx >> 1;
    const bool odd = dpx & 1;
    const float sm = row[max(spx - 1, 0) * 3 + c], s0 = row[spx * 3 + c], sp = row[min(spx + 1, sw - 1) * 3 + c];
    const bool left = spx == 0, right = spx >= sw - 1;
    const float ev = left ? s0 * 6.f + sp * 2.f : right ? sm + s0 * 7.f : sm + s0 * 6.f + sp;
    const float od = right ? s0 * 8.f : (s0 + sp) * 4.f;
    return odd ? od : ev;
}

__device__ __forceinline__ float pyrup_elem_wide(const float* __restrict__ src, int sw, int sh, int dy, int dxe, int sp = 0) {
    const int sy = dy >> 1;
    const size_t stride = (size_t)(sp > 0 ? sp : sw) * 3;
    const int sym = sy >= 1 ? sy - 1 : 1, syp = sy + 1 <= sh - 1 ? sy + 1 : sh - 1;      // sh >= 2
    const float r0 = pyrup_h_wide(src + sym * stride, sw, dxe), r1 = pyrup_h_wide(src + sy * stride, sw, dxe);
    const float r2 = pyrup_h_wide(src + syp * stride, sw, dxe);
    const float s = 1.f / 64;
    return (dy & 1) ? ((r1 + r2) * 4.f) * s : (r0 + r1 * 6.f + r2) * s;
}

template <bool U8, typename F = MaskPlain>
__device__ __forceinline__ float collapse_elem_wide(const void* gL, const void* gR, const float* gM, const float* nL, const float* nR,
                                                    const float* nB, int w, int h, int nw, int nh, int y, int xe, CollapsePitch cp, F fn = F()) {
    const size_t i = (size_t)y * cp.g * 3 + xe;
    const float m = fn(gM[(size_t)y * cp.m + xe / 3]);
    const float gl = ld<U8>(gL, i), gr = ld<U8>(gR, i);
    const float uL = pyrup_elem_wide(nL, nw, nh, y, xe, cp.n), uR = pyrup_elem_wide(nR, nw, nh, y, xe, cp.n), uB = pyrup_elem_wide(nB, nw, nh, y, xe, cp.n);
    return uB + mix_lr(gl - uL, gr - uR, m);
}

}  // namespace poppy_hip
