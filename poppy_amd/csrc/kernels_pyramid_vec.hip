// kernels_pyramid_vec.hip — wide-access forms of the two pyramid kernels that carry most of a frame's bytes
// (pyrDown and the fused collapse step) for levels whose rows are 16-byte aligned (width % 4 == 0).
//
// Why a second form: the per-element kernels in kernels_frame.hip issue one 4-byte (or 1-byte) access per lane
// and walk 3-channel rows with a stride of 3 elements, so a wave needs ~10x more memory instructions than the
// bytes justify and the address pipeline, not HBM, sets the pace (profiles/r01_c_pmc_4k.md).  Here every thread
// owns a small block of OUTPUT pixels (pyrDown: 2 x 2, collapse: 4 of one row = three 16-byte stores), pulls its inputs
// with 16-byte loads and keeps the stencil in registers.  Each output is the same expression tree as in the
// per-element form (pyrdown_elem / pyrup_elem / mix_lr), so results stay bit-identical; threads whose stencil
// touches an image border, where the reference switches formulas, call the per-element functions.
#include "kernels.h"
#include "pyramid_device.h"
#include <algorithm>

namespace poppy_hip {

// Interior = the threads whose whole stencil lies inside the image: t in [1, t1], row in [1, y1].  Everything else
// (a frame of a few pixels around the level) is produced by the per-element border kernels below, so that no wave
// of the wide kernels ever takes the slow path.
struct VecBounds { int t1, y1; };

// enumerates the output pixels outside the interior rectangle [x0, x1] x [y0, y1] of a w x h image
__device__ __forceinline__ bool border_pixel(int i, int w, int h, int x0, int x1, int y0, int y1, int& x, int& y) {
    const int top = y0 * w, bottom = (h - 1 - y1) * w, mid_h = y1 - y0 + 1, left = x0, right = w - 1 - x1;
    if (i < top) { y = i / w; x = i - y * w; return true; }
    i -= top;
    if (i < bottom) { y = y1 + 1 + i / w; x = i % w; return true; }
    i -= bottom;
    if (i < left * mid_h) { y = y0 + i / left; x = i % left; return true; }
    i -= left * mid_h;
    if (i < right * mid_h) { y = y0 + i / right; x = x1 + 1 + i % right; return true; }
    return false;
}
__host__ __device__ inline int border_count(int w, int h, int x0, int x1, int y0, int y1) {
    return y0 * w + (h - 1 - y1) * w + (x0 + (w - 1 - x1)) * (y1 - y0 + 1);
}

// one thread per border ELEMENT (7 per pixel for pyrDown: 3 + 3 + mask; 3 per pixel for the collapse)
template <bool U8, typename F>
__device__ __forceinline__ void pyrdown_border_body(int block, const void* __restrict__ srcL, const void* __restrict__ srcR, const float* __restrict__ srcM,
                                                    float* __restrict__ dstL, float* __restrict__ dstR, float* __restrict__ dstM,
                                                    const DownGeom& g3, const DownGeom& g1, int x0, int x1, int y0, int y1, F fn) {
    const int i = block * 256 + threadIdx.y * 64 + threadIdx.x;
    const int pix = i / 7, k = i - pix * 7;
    int x, y;
    if (!border_pixel(pix, g3.dw, g3.dh, x0, x1, y0, y1, x, y)) return;
    if (k < 3)      dstL[((size_t)y * g3.dp + x) * 3 + k] = pyrdown_elem_wide<U8, 3>(srcL, g3, y, x * 3 + k);
    else if (k < 6) dstR[((size_t)y * g3.dp + x) * 3 + (k - 3)] = pyrdown_elem_wide<U8, 3>(srcR, g3, y, x * 3 + (k - 3));
    else            dstM[(size_t)y * g1.dp + x] = pyrdown_elem_wide<false, 1>(srcM, g1, y, x, fn);
}

template <bool U8, typename F>
__device__ __forceinline__ void collapse_border_body(int block, const void* __restrict__ gL, const void* __restrict__ gR, const float* __restrict__ gM,
                                                     const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                                     float* __restrict__ outB, int w, int h, int nw, int nh, CollapsePitch cp, int x0, int x1, int y0, int y1, F fn) {
    const int i = block * 256 + threadIdx.y * 64 + threadIdx.x;
    const int pix = i / 3, c = i - pix * 3;
    int x, y;
    if (!border_pixel(pix, w, h, x0, x1, y0, y1, x, y)) return;
    outB[((size_t)y * cp.g + x) * 3 + c] = collapse_elem_wide<U8>(gL, gR, gM, nL, nR, nB, w, h, nw, nh, y, x * 3 + c, cp, fn);
}

// ------------------------------------------------------------------------------------------------
// pyrDown, 3 channels.  Thread (t, yb): output pixels 2t, 2t+1 of output rows 2yb and 2yb+1, from source pixels 4t-2 .. 4t+4 (21 elements,
// read as six dwords / six float4 from element 12t-8) of source rows 4yb-2 .. 4yb+4.  The two output rows share three of their five source
// rows, so a thread runs the row pass over 7 rows instead of 10.  Measured against the earlier form (4 pixels of ONE row per thread, 88 / 112
// VGPRs): 62 / 74 VGPRs, 8 / 6 waves per SIMD, 16 % fewer vector instructions; level 0 15.3 -> 14.3 us at 1080p and 43.5 -> 41.0 us at 4K,
// the float levels 8.3 -> 7.4 us and 16.3 -> 14.8 us.  (Row pairs with 4 pixels per thread — 102-112 VGPRs, half the threads — and 2 pixels of
// one row — 27 % more source elements converted per output — both measured slower: profiles/r02_notes.md section 8.)  Same expression trees
// as pyrdown_elem, same interior rectangle (pixels 4 .. 4 t1 + 3 of rows 1 .. y1) as the border blocks enumerate.
// ------------------------------------------------------------------------------------------------
// row pass of source row `row` for the 6 output elements of thread t
template <bool U8>
__device__ __forceinline__ void pyrdown3_row2(const void* __restrict__ src, size_t srow, int row, int t, const DownGeom& g, float* r /*6*/) {
    float v[24];
    if (U8) {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)src + (size_t)row * srow + (size_t)(12 * t - 8));
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = p[i];
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = (float)((w[i >> 2] >> (8 * (i & 3))) & 255u) * kInv255;
    } else {
        const float4* p = (const float4*)((const float*)src + (size_t)row * srow + (size_t)(12 * t - 8));
#pragma unroll
        for (int i = 0; i < 6; ++i) { float4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int j = e / 3, c = e - 3 * j;
        const int bb = 2 + 6 * j + c;
        const float t0 = v[bb], t1 = v[bb + 3], t2 = v[bb + 6], t3 = v[bb + 9], t4 = v[bb + 12];
        const int xe = 6 * t + e;
        const bool hBody = (xe >= 3) && (xe < g.hBodyEnd);
        r[e] = hBody ? t2 * 6.f + ((t1 + t3) * 4.f + (t0 + t4))
                     : t2 * 6.f + (t1 + t3) * 4.f + t0 + t4;
    }
}
__device__ __forceinline__ void pyrdown3_store2(float* __restrict__ dst, int dwe, int y, int t, const DownGeom& g,
                                                const float* r0, const float* r1, const float* r2, const float* r3, const float* r4) {
    const float s = 1.f / 256;
    float o[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int xe = 6 * t + e;
        o[e] = (xe < g.vBodyEnd) ? ((r1[e] + r3[e] + r2[e]) * 4.f + (r0[e] + r4[e] + (r2[e] + r2[e]))) * s
                                 : (r2[e] * 6.f + (r1[e] + r3[e]) * 4.f + r0[e] + r4[e]) * s;
    }
    float2* d = (float2*)(dst + (size_t)y * dwe + 6 * t);
    d[0] = make_float2(o[0], o[1]); d[1] = make_float2(o[2], o[3]); d[2] = make_float2(o[4], o[5]);
}
// thread (t, yb): output pixels 2t, 2t+1 of output rows 2yb and 2yb+1 — 7 source rows of 7 source pixels
template <bool U8>
__device__ __forceinline__ void pyrdown3_body2(int bx, int by, const void* __restrict__ src, float* __restrict__ dst, const DownGeom& g, VecBounds b) {
    const int t = bx * 64 + threadIdx.x;
    const int y = (by * 4 + threadIdx.y) * 2;
    if (y >= g.dh) return;
    if (!(t >= 2 && t <= 2 * b.t1 + 1)) return;                        // border outputs: k_pyrdown_border
    const size_t srow = (size_t)g.sp * 3;
    const bool in0 = y >= 1 && y <= b.y1, in1 = y + 1 <= b.y1;
    if (in0 && in1) {
        float r[7][6];
#pragma unroll
        for (int k = 0; k < 7; ++k) pyrdown3_row2<U8>(src, srow, 2 * y - 2 + k, t, g, r[k]);
        pyrdown3_store2(dst, g.dp * 3, y, t, g, r[0], r[1], r[2], r[3], r[4]);
        pyrdown3_store2(dst, g.dp * 3, y + 1, t, g, r[2], r[3], r[4], r[5], r[6]);
    } else if (in0 || in1) {
        const int yy = in0 ? y : y + 1;
        float r[5][6];
#pragma unroll
        for (int k = 0; k < 5; ++k) pyrdown3_row2<U8>(src, srow, 2 * yy - 2 + k, t, g, r[k]);
        pyrdown3_store2(dst, g.dp * 3, yy, t, g, r[0], r[1], r[2], r[3], r[4]);
    }
}

// pyrDown, 1 channel (mask).  Thread (t, y): output pixels 4t..4t+3; source pixels 8t-2 .. 8t+8 of rows 2y-2 .. 2y+2,
// fetched as four float4 from the aligned window 8t-4 .. 8t+11.
template <typename F>
__device__ __forceinline__ void pyrdown1_body(int bx, int by, const float* __restrict__ src, float* __restrict__ dst, const DownGeom& g, VecBounds b, F fn) {
    const int t = bx * 64 + threadIdx.x;
    const int y = by * 4 + threadIdx.y;
    const int nt = (g.dw + 3) >> 2;
    if (t >= nt || y >= g.dh) return;
    if (!(t >= 1 && t <= b.t1 && y >= 1 && y <= b.y1)) return;        // border outputs: k_pyrdown_border
    float r[5][4];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float4* p = (const float4*)(src + (size_t)(2 * y - 2 + k) * g.sp + (8 * t - 4));
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { float4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
#pragma unroll
        for (int i = 2; i <= 12; ++i) v[i] = fn(v[i]);    // the eleven values the four outputs use
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int b = 2 + 2 * e;                       // source pixel 8t + 2e - 2 -> window index 2 + 2e
            const float t0 = v[b], t1 = v[b + 1], t2 = v[b + 2], t3 = v[b + 3], t4 = v[b + 4];
            const int xe = 4 * t + e;
            const bool hBody = (xe >= 1) && (xe < g.hBodyEnd);
            r[k][e] = hBody ? t2 * 6.f + ((t1 + t3) * 4.f + (t0 + t4))
                            : t2 * 6.f + (t1 + t3) * 4.f + t0 + t4;
        }
    }
    const float s = 1.f / 256;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int xe = 4 * t + e;
        o[e] = (xe < g.vBodyEnd) ? ((r[1][e] + r[3][e] + r[2][e]) * 4.f + (r[0][e] + r[4][e] + (r[2][e] + r[2][e]))) * s
                                 : (r[2][e] * 6.f + (r[1][e] + r[3][e]) * 4.f + r[0][e] + r[4][e]) * s;
    }
    *(float4*)(dst + (size_t)y * g.dp + 4 * t) = make_float4(o[0], o[1], o[2], o[3]);
}

// One launch per level: blocks [0, 3*nbi) are the interior tiles of L, R and the mask, the rest enumerate the border
// elements.  The role of a block is uniform, so nothing diverges inside a wave.
template <bool U8, typename F>
__device__ __forceinline__ void pyrdown_level_body(const void* __restrict__ srcL, const void* __restrict__ srcR, const float* __restrict__ srcM,
                                                   float* __restrict__ dstL, float* __restrict__ dstR, float* __restrict__ dstM,
                                                   const DownGeom& g3, const DownGeom& g1, VecBounds b, int gx, int gy, int nborder, int x0, int x1, int y0, int y1, F fn) {
    // L and R in the 2 x 2 form (gx2 block columns, gy2 block rows), the mask in the 4-pixel form
    const int gx2 = (g3.dw / 2 + 63) / 64;
    const int gy2 = (gy + 1) / 2;
    const int nbi3 = gx2 * gy2, nbi1 = gx * gy;
    int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    if (blk >= nborder) {                 // border blocks come first: their load chains are the longest
        blk -= nborder;
        if (blk < 2 * nbi3) {
            const int which = blk / nbi3; blk -= which * nbi3;
            const int by = blk / gx2, bx = blk - by * gx2;
            pyrdown3_body2<U8>(bx, by, which ? srcR : srcL, which ? dstR : dstL, g3, b);
        } else {
            blk -= 2 * nbi3;
            if (blk >= nbi1) return;
            const int by = blk / gx, bx = blk - by * gx;
            pyrdown1_body(bx, by, srcM, dstM, g1, b, fn);
        }
    } else {
        pyrdown_border_body<U8>(blk, srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, x0, x1, y0, y1, fn);
    }
}
// mask_ab: nullptr = srcM holds the mask itself; else srcM is the pair's m2 and mask_ab -> (alpha, beta) of this frame (level 0 only)
template <bool U8>
__global__ void __launch_bounds__(256) k_pyrdown_level(const void* __restrict__ srcL, const void* __restrict__ srcR, const float* __restrict__ srcM,
                                                       float* __restrict__ dstL, float* __restrict__ dstR, float* __restrict__ dstM,
                                                       DownGeom g3, DownGeom g1, VecBounds b, int gx, int gy, int nborder, int x0, int x1, int y0, int y1,
                                                       const double* __restrict__ mask_ab, int stagger) {
    stagger_priority(blockIdx.x, stagger);
    if (U8 && mask_ab) pyrdown_level_body<U8>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, b, gx, gy, nborder, x0, x1, y0, y1, mask_source(mask_ab));
    else               pyrdown_level_body<U8>(srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, b, gx, gy, nborder, x0, x1, y0, y1, MaskPlain());
}

// sp / mp / dp: pixels per row of the source images, the source mask and the destination level (0: the level's own width).  The 16-byte accesses
// need rows that begin on 16-byte boundaries: pitches that are multiples of 4 — the widths themselves may be anything.
static bool pyrdown_vec_bounds(int sw, int sh, VecBounds& b, int sp = 0, int mp = 0, int dp = 0) {
    const DownGeom g3 = make_down_geom(sw, sh, 3, sp, dp), g1 = make_down_geom(sw, sh, 1, mp, dp);
    if ((g3.dp & 3) != 0 || (g3.sp & 3) != 0 || (g1.sp & 3) != 0 || g3.dw < 32 || g3.dh < 8) return false;
    // interior threads: 24t + 27 < 3 sw  (this also covers the 1-channel window 8t + 11 < sw) and 2y + 2 <= sh - 1
    b.t1 = std::min(g3.dw / 4 - 1, (3 * sw - 28) / 24);
    b.t1 = std::min(b.t1, (sw - 12) / 8);
    b.y1 = std::min(g3.dh - 1, (sh - 3) / 2);
    return b.t1 >= 1 && b.y1 >= 1;
}
// (gp / mp: pixels per row of this level's images and of its mask; the coarser level's rows are read 12 bytes at a time: any pitch)
static bool collapse_vec_ok(int w, int h, int nw, int nh, int gp = 0, int mp = 0) {
    (void)h;
    const CollapsePitch cp = make_collapse_pitch(w, nw, gp, mp, 0);
    return (cp.g & 3) == 0 && (cp.m & 3) == 0 && (nw * 2 == w || nw * 2 == w + 1) && cp.g >= 2 * nw && nw >= 16 && nh >= 8;
}

// both level-0 kernels of a w x h frame take the wide forms (the ones that can read the mask through m2)
bool pyr_level0_vec_ok(int w, int h) {
    VecBounds b;
    return pyrdown_vec_bounds(w, h, b) && collapse_vec_ok(w, h, (w + 1) / 2, (h + 1) / 2);
}

bool launch_pyrdown_vec(const void* srcL, const void* srcR, const float* srcM, bool src_u8,
                        float* dstL, float* dstR, float* dstM, int sw, int sh, hipStream_t s, const double* mask_ab, int sp, int mp, int dp) {
    const DownGeom g3 = make_down_geom(sw, sh, 3, sp, dp), g1 = make_down_geom(sw, sh, 1, mp, dp);
    VecBounds b;
    if (!pyrdown_vec_bounds(sw, sh, b, sp, mp, dp)) return false;
    const int gx = (g3.dw / 4 + 63) / 64, gy = (g3.dh + 3) / 4;
    const int x0 = 4, x1 = 4 * b.t1 + 3, y0 = 1, y1 = b.y1;
    const int nb = border_count(g3.dw, g3.dh, x0, x1, y0, y1);
    const int nborder = (nb * 7 + 255) / 256, blocks = 2 * ((g3.dw / 2 + 63) / 64) * ((gy + 1) / 2) + gx * gy + nborder;
    if (src_u8) hipLaunchKernelGGL(k_pyrdown_level<true>, dim3(blocks), dim3(64, 4), 0, s, srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, b, gx, gy, nborder, x0, x1, y0, y1, mask_ab, src_u8 ? stagger_flag(2) : 0);
    else        hipLaunchKernelGGL(k_pyrdown_level<false>, dim3(blocks), dim3(64, 4), 0, s, srcL, srcR, srcM, dstL, dstR, dstM, g3, g1, b, gx, gy, nborder, x0, x1, y0, y1, (const double*)nullptr, 0);
    return true;
}

// ------------------------------------------------------------------------------------------------
// Collapse step.  Thread (t, sy): low-resolution pixels 2t, 2t+1 of low-resolution row sy -> output pixels
// 4t..4t+3 of rows 2sy and 2sy+1 (24 floats, six 16-byte stores).  Low-resolution pixels 2t-1 .. 2t+2 of rows
// sy-1 .. sy+1 are fetched as 12-byte loads for each of the three upsampled images in turn.
// ------------------------------------------------------------------------------------------------
struct float3u { float x, y, z; };     // 12-byte load unit (4-byte aligned)

// up[r][j][c]: pyrUp of one image at output row r (0: even row 2sy, 1: odd row 2sy+1), output pixel j (0..3), channel c
__device__ __forceinline__ void up_2x4(const float* __restrict__ n, size_t nrow, int sy, int t, float up[2][4][3]) {
    float he[3][2][3], ho[3][2][3];     // [low-res row][low-res pixel 2t / 2t+1][channel]
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float3u* p = (const float3u*)(n + (size_t)(sy - 1 + r) * nrow + (size_t)(2 * t - 1) * 3);
        const float3u a = p[0], b = p[1], c = p[2], d = p[3];
        const float A[3] = {a.x, a.y, a.z}, B[3] = {b.x, b.y, b.z}, C[3] = {c.x, c.y, c.z}, D[3] = {d.x, d.y, d.z};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            he[r][0][ch] = A[ch] + B[ch] * 6.f + C[ch];
            ho[r][0][ch] = (B[ch] + C[ch]) * 4.f;
            he[r][1][ch] = B[ch] + C[ch] * 6.f + D[ch];
            ho[r][1][ch] = (C[ch] + D[ch]) * 4.f;
        }
    }
    const float s = 1.f / 64;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            up[0][2 * q][ch]     = (he[0][q][ch] + he[1][q][ch] * 6.f + he[2][q][ch]) * s;
            up[0][2 * q + 1][ch] = (ho[0][q][ch] + ho[1][q][ch] * 6.f + ho[2][q][ch]) * s;
            up[1][2 * q][ch]     = ((he[1][q][ch] + he[2][q][ch]) * 4.f) * s;
            up[1][2 * q + 1][ch] = ((ho[1][q][ch] + ho[2][q][ch]) * 4.f) * s;
        }
}

template <bool U8>
__device__ __forceinline__ void load_g12(const void* g, size_t elem_off, float v[12]) {
    if (U8) {
        const uint32_t* p = (const uint32_t*)((const uint8_t*)g + elem_off);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = (float)((w0 >> (8 * i)) & 255u) * kInv255;
            v[4 + i] = (float)((w1 >> (8 * i)) & 255u) * kInv255;
            v[8 + i] = (float)((w2 >> (8 * i)) & 255u) * kInv255;
        }
    } else {
        const float4* p = (const float4*)((const float*)g + elem_off);
        const float4 a = p[0], b = p[1], c = p[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    }
}

template <bool U8, typename F>
__device__ __forceinline__ void collapse_body(int bx, int by, const void* __restrict__ gL, const void* __restrict__ gR, const float* __restrict__ gM,
                                              const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                              float* __restrict__ outB, int w, int h, int nw, int nh, CollapsePitch cp, F fn) {
    const int t = bx * 64 + threadIdx.x;
    const int sy = by * 4 + threadIdx.y;
    if (t >= (nw >> 1) || sy >= nh) return;
    if (!(t >= 1 && 2 * t + 2 <= nw - 1 && sy >= 1 && sy <= nh - 2)) return;     // border outputs: k_collapse_border
    const size_t nrow = (size_t)cp.n * 3, orow = (size_t)cp.g * 3;
    float res[2][12];
    float m[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float4 q = *(const float4*)(gM + (size_t)(2 * sy + r) * cp.m + 4 * t);
        m[r][0] = fn(q.x); m[r][1] = fn(q.y); m[r][2] = fn(q.z); m[r][3] = fn(q.w);
    }
    // The three upsampled images one after the other in a loop that is NOT unrolled: unrolled, the scheduler hoists all 36 three-dword
    // loads of the three images together and the kernel needs 206 VGPRs (2 waves per SIMD).
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const float* n = k == 0 ? nL : k == 1 ? nR : nB;
        float up[2][4][3];
        up_2x4(n, nrow, sy, t, up);
        if (k < 2) {                                         // A = (G_L - up(nL)) * m;  + (G_R - up(nR)) * (1 - m)
            const void* g_img = k == 0 ? gL : gR;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float g[12];
                load_g12<U8>(g_img, (size_t)(2 * sy + r) * orow + 12 * t, g);
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    const float wgt = k == 0 ? m[r][e / 3] : 1.f - m[r][e / 3];
                    const float b = (g[e] - up[r][e / 3][e % 3]) * wgt;
                    res[r][e] = k == 0 ? b : res[r][e] + b;
                }
            }
        } else {                                             // out = up(nB) + res
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                float o[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) o[e] = up[r][e / 3][e % 3] + res[r][e];
                float4* d = (float4*)(outB + (size_t)(2 * sy + r) * orow + 12 * t);
                d[0] = make_float4(o[0], o[1], o[2], o[3]);
                d[1] = make_float4(o[4], o[5], o[6], o[7]);
                d[2] = make_float4(o[8], o[9], o[10], o[11]);
            }
        }
    }
}

template <bool U8, typename F>
__device__ __forceinline__ void collapse_level_body(const void* __restrict__ gL, const void* __restrict__ gR, const float* __restrict__ gM,
                                                    const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                                    float* __restrict__ outB, int w, int h, int nw, int nh, CollapsePitch cp, int gx, int gy, int nborder,
                                                    int x0, int x1, int y0, int y1, F fn) {
    int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    if (blk >= nborder) {
        blk -= nborder;
        const int by = blk / gx, bx = blk - by * gx;
        collapse_body<U8>(bx, by, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, fn);
    } else {
        collapse_border_body<U8>(blk, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, x0, x1, y0, y1, fn);
    }
}
template <bool U8>
__global__ void __launch_bounds__(256) k_collapse_level(const void* __restrict__ gL, const void* __restrict__ gR, const float* __restrict__ gM,
                                                        const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                                        float* __restrict__ outB, int w, int h, int nw, int nh, CollapsePitch cp, int gx, int gy, int nborder,
                                                        int x0, int x1, int y0, int y1, const double* __restrict__ mask_ab, int stagger) {
    stagger_priority(blockIdx.x, stagger);
    if (U8 && mask_ab) collapse_level_body<U8>(gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, gx, gy, nborder, x0, x1, y0, y1, mask_source(mask_ab));
    else               collapse_level_body<U8>(gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, gx, gy, nborder, x0, x1, y0, y1, MaskPlain());
}

bool launch_collapse_vec(const void* gL, const void* gR, bool g_u8, const float* gM, const float* nL, const float* nR, const float* nB,
                         float* outB, int w, int h, int nw, int nh, hipStream_t s, const double* mask_ab, int gp, int mp, int np) {
    if (!collapse_vec_ok(w, h, nw, nh, gp, mp)) return false;
    const CollapsePitch cp = make_collapse_pitch(w, nw, gp, mp, np);
    const int t1 = (nw - 3) / 2;                      // last interior thread: 2t + 2 <= nw - 1
    const int gx = (nw / 2 + 63) / 64, gy = (nh + 3) / 4;
    const int x0 = 4, x1 = 4 * t1 + 3, y0 = 2, y1 = 2 * (nh - 2) + 1;
    const int nb = border_count(w, h, x0, x1, y0, y1);
    const int nborder = (nb * 3 + 255) / 256, blocks = gx * gy + nborder;
    if (g_u8) hipLaunchKernelGGL(k_collapse_level<true>, dim3(blocks), dim3(64, 4), 0, s, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, gx, gy, nborder, x0, x1, y0, y1, mask_ab, stagger_flag(3));
    else      hipLaunchKernelGGL(k_collapse_level<false>, dim3(blocks), dim3(64, 4), 0, s, gL, gR, gM, nL, nR, nB, outB, w, h, nw, nh, cp, gx, gy, nborder, x0, x1, y0, y1, (const double*)nullptr, 0);
    return true;
}

}  // namespace poppy_hip
