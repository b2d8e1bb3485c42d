// kernels_orb.hip — ORB keypoint detection kernels for gfx950 (integer / byte work, wave64).
//
//   pyramid  : 8 levels, INTER_LINEAR_EXACT (8.8 fixed point) + 32-pixel reflect-101 border
//              OCV/features2d/src/orb.cpp:1033-1113, OCV/imgproc/src/resize.cpp:345-398,619-760
//   FAST     : FAST-9/16 corner test + corner score, threshold 20       fast.cpp:57-266, fast_score.cpp:120-211
//   NMS      : strict 3x3 maximum, border filter (edgeThreshold 31), unordered append (the host
//              sorts the few thousand survivors into raster order)       fast.cpp:271-290, keypoint.cpp:106-118
//   Harris   : 7x7 block, Sobel-like integer sums                        orb.cpp:130-177
//   IC angle : intensity centroid over the circular patch + fastAtan2    orb.cpp:181-215, mathfuncs_core.simd.hpp:34-64
//   rBRIEF   : 7x7 sigma-2 fixed-point blur + 256 steered comparisons    orb.cpp:219-285,1188
//   Hamming  : brute-force 1-NN, xor + popcount, wave min-reduction       batch_distance.cpp:103-110,199-262
// Selection of the best keypoints (std::nth_element / std::partition order) stays on the host by design.
#include "kernels_orb.h"
#include <cfloat>
#include <climits>

namespace poppy_hip {

__device__ __forceinline__ int reflect101i(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
    return p;
}

// ------------------------------------------------------------------------------------------------
// pyramid
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_orb_level0(const uint8_t* __restrict__ src, int sw, int sh, size_t sstride, uint8_t* __restrict__ atlas, OrbLevel L) {
    int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= L.w + 2 * kOrbBorder) return;
    int x = reflect101i(px - kOrbBorder, L.w), y = reflect101i(py - kOrbBorder, L.h);
    atlas[L.offset + (size_t)py * L.stride + px] = src[(size_t)y * sstride + x];
}

// 8.8 coefficient pair of destination index v (interpolationLinear::getCoeffs); returns the source offset,
// kind: 0 = interpolate, -1 = left of the image (use src[0]), +1 = right (use src[last])
__device__ __forceinline__ int lin_coeff(int v, double scale, int ssize, int& c0, int& c1, int& kind) {
    double f = scale * ((double)v + 0.5) - 0.5;
    int iv = (int)floor(f);
    if (iv >= 0 && ssize > 1) {
        if (iv < ssize - 1) {
            double frac = f - (double)iv;
            c1 = frac < 0 ? 0 : __double2int_rn(frac * 256.0);
            c0 = 256 > c1 ? 256 - c1 : 0;
            kind = 0;
            return iv;
        }
        kind = 1; c0 = c1 = 0;
        return ssize - 1;
    }
    kind = -1; c0 = c1 = 0;
    return 0;
}

// One padded pixel of level `L` from the interior of the previous level (`P`, or the input image for level 1).
// The reference computes "left/right of the image" as contiguous index ranges [0,min) and [max,dsize); because
// the source coordinate is monotonic in v these coincide with the per-index kinds computed here.
__global__ void __launch_bounds__(256) k_orb_resize(const uint8_t* __restrict__ src, int sw, int sh, size_t sstride,
                                                   uint8_t* __restrict__ atlas, OrbLevel L, double scale_x, double scale_y) {
    int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= L.w + 2 * kOrbBorder) return;
    int x = reflect101i(px - kOrbBorder, L.w), y = reflect101i(py - kOrbBorder, L.h);
    int xc0, xc1, xk, yc0, yc1, yk;
    int xo = lin_coeff(x, scale_x, sw, xc0, xc1, xk);
    int yo = lin_coeff(y, scale_y, sh, yc0, yc1, yk);
    auto hline = [&](int row) -> unsigned {
        const uint8_t* s = src + (size_t)row * sstride;
        if (xk == 0) return (unsigned)(xc0 * s[xo] + xc1 * s[xo + 1]) & 0xFFFFu;
        return (unsigned)s[xk < 0 ? 0 : sw - 1] << 8;
    };
    unsigned out;
    if (yk == 0) {
        unsigned r0 = hline(yo), r1 = hline(yo + 1);
        out = (r0 * (unsigned)yc0 + r1 * (unsigned)yc1 + (1u << 15)) >> 16;
    } else {
        unsigned r = hline(yk < 0 ? 0 : sh - 1);
        out = ((r + 128u) & 0xFFFFu) >> 8;
    }
    atlas[L.offset + (size_t)py * L.stride + px] = (uint8_t)out;
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16
// ------------------------------------------------------------------------------------------------
constexpr int kFastRows = 4;      // rows of one level per block: the grid spans the LARGEST level, so most blocks of the small levels fall outside their
                                  // level and only cost their dispatch; fewer, taller blocks cut that four-fold

__device__ __forceinline__ constexpr int ring_dx(int k) { constexpr int t[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}; return t[k]; }
__device__ __forceinline__ constexpr int ring_dy(int k) { constexpr int t[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3}; return t[k]; }

// FAST-9/16 corner score of the pixel at c (0: no corner), threshold t (fast.cpp:57-292, fast_score.cpp:115-200).  A 9-arc of the 16-ring holds one of every
// two opposite ring pixels, so four opposite pairs are looked at first, as the reference's own loop does (fast.cpp:110-131): a pair with no pixel brighter
// than v + t rules the bright arc out, one with none darker than v - t the dark arc; most pixels end there, after two loads.
__device__ __forceinline__ int fast_score_at(const uint8_t* __restrict__ c, int st, int threshold) {
    const int v = c[0];
    int d[16];
    unsigned poss = 3;                                          // bit 0: a dark arc (ring < v - t) is still possible, bit 1: a bright one
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
        d[k] = v - (int)c[ring_dy(k) * st + ring_dx(k)];
        d[k + 8] = v - (int)c[ring_dy(k + 8) * st + ring_dx(k + 8)];
        poss &= ((d[k] > threshold || d[k + 8] > threshold) ? 1u : 0u) | ((d[k] < -threshold || d[k + 8] < -threshold) ? 2u : 0u);
        if (!poss) return 0;
    }
#pragma unroll
    for (int k = 1; k < 8; k += 2) {
        d[k] = v - (int)c[ring_dy(k) * st + ring_dx(k)];
        d[k + 8] = v - (int)c[ring_dy(k + 8) * st + ring_dx(k + 8)];
    }
    // With d = v - ring: a dark 9-arc exists iff some arc's MINIMUM of d exceeds t, a bright one iff some arc's MAXIMUM lies below -t; the score (the largest
    // threshold at which the pixel is still a corner, fast_score.cpp:115-200) is max(largest arc minimum, -(smallest arc maximum)) - 1.  So the two extremes
    // decide AND score.  The sixteen arcs share their pieces: three-in-a-row extremes (16 x min3 / max3), an arc = three of those.
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo3[k] = min(min(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
        hi3[k] = max(max(d[k], d[(k + 1) & 15]), d[(k + 2) & 15]);
    }
    int best_min = -1000, best_max = 1000;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        best_min = max(best_min, min(min(lo3[k], lo3[(k + 3) & 15]), lo3[(k + 6) & 15]));
        best_max = min(best_max, max(max(hi3[k], hi3[(k + 3) & 15]), hi3[(k + 6) & 15]));
    }
    const int sc = max(best_min, -best_max);
    return sc > threshold ? (int)(uint8_t)(sc - 1) : 0;
}

__global__ void __launch_bounds__(256) k_fast_score(const uint8_t* __restrict__ atlas, OrbLevelSet S, uint8_t* __restrict__ scores, int threshold) {
    const OrbLevel L = S.lv[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= L.w) return;
    for (int rr = 0; rr < kFastRows; ++rr) {
        const int y = blockIdx.y * kFastRows + rr;
        if (y >= L.h) return;
        uint8_t result = 0;
        if (x >= 3 && x < L.w - 3 && y >= 3 && y < L.h - 3)
            result = (uint8_t)fast_score_at(atlas + L.offset + (size_t)(y + kOrbBorder) * L.stride + (x + kOrbBorder), (int)L.stride, threshold);
        scores[L.score_offset + (size_t)y * L.w + x] = result;
    }
}

// strict 3x3 maximum + border filter; survivors appended unordered: (level, y*w + x, score).
// Non-maximum suppression + compaction IN RASTER ORDER (fast.cpp:271-290 emits row by row, left to right, and the order of the list is part of
// the contract: ORB's retainBest permutes it with std::nth_element).  A block takes 256 columns x kNmsRows rows of a level.  Three launches:
//   k_fast_nms_count   survivors per (row, block of columns) -> rowblk[], each thread's survivor bits -> keep[] (one byte);
//   k_fast_nms_scan    one block per level: exclusive prefix of rowblk[] in (row, column block) order -> the slot of every row piece, and the
//                      level's total -> counters[];
//   k_fast_nms_emit    every survivor writes (position, score) at its row piece's slot + its rank among the piece's survivors.
// The host gets the candidates in FAST's own emission order and no longer re-sorts ~10^5 of them per image (0.4 ms of the pair set-up); the
// round-2 form claimed slots with one atomic per block, in whatever order the blocks ran.
constexpr int kNmsRows = 8;
struct NmsLayout { int base[kOrbLevels + 1]; int nbx[kOrbLevels]; };            // rowblk[] offsets per level (entries = rows x column blocks)

__device__ __forceinline__ unsigned nms_keep_bits(const uint8_t* __restrict__ scores, const OrbLevel& L, int edge, int x, int y0, int* sc) {
    const int W = L.w;
    const bool col_ok = x >= edge && x < W - edge;
    unsigned keep = 0;
#pragma unroll
    for (int r = 0; r < kNmsRows; ++r) {
        const int y = y0 + r;
        sc[r] = 0;
        if (col_ok && y >= edge && y < L.h - edge) {
            const uint8_t* p = scores + L.score_offset + (size_t)y * W + x;
            const int v = p[0];
            sc[r] = v;
            if (v != 0 && v > p[-1] && v > p[1] && v > p[-W - 1] && v > p[-W] && v > p[-W + 1] && v > p[W - 1] && v > p[W] && v > p[W + 1]) keep |= 1u << r;
        }
    }
    return keep;
}

__global__ void __launch_bounds__(256) k_fast_nms_count(const uint8_t* __restrict__ scores, OrbLevelSet S, NmsLayout lay, int edge,
                                                        int* __restrict__ rowblk, uint8_t* __restrict__ keepbuf, size_t keep_stride) {
    __shared__ int s_cnt[kNmsRows][4];
    const int lvl = blockIdx.z;
    const OrbLevel L = S.lv[lvl];
    const int W = L.w, nbx = lay.nbx[lvl];
    const int bx = blockIdx.x, y0 = blockIdx.y * kNmsRows;
    if (bx >= nbx || y0 >= L.h) return;                                   // uniform: outside this level's grid
    const int x = bx * 256 + threadIdx.x;
    const bool dead = W <= 2 * edge || L.h <= 2 * edge || y0 >= L.h - edge || y0 + kNmsRows <= edge || bx * 256 >= W - edge;   // uniform: nothing can survive
    unsigned keep = 0;
    if (!dead) { int sc[kNmsRows]; keep = nms_keep_bits(scores, L, edge, x, y0, sc); }
    if (x < W) keepbuf[(size_t)lvl * keep_stride + (size_t)blockIdx.y * W + x] = (uint8_t)keep;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kNmsRows; ++r) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((keep >> r) & 1u);
        if (lane == 0) s_cnt[r][wv] = __builtin_popcountll(m);
    }
    __syncthreads();
    if (threadIdx.x < kNmsRows && y0 + (int)threadIdx.x < L.h)
        rowblk[lay.base[lvl] + (y0 + threadIdx.x) * nbx + bx] = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
}

__global__ void __launch_bounds__(1024) k_fast_nms_scan(NmsLayout lay, int* __restrict__ rowblk, int* __restrict__ counters) {
    __shared__ int s_part[1024];
    const int lvl = blockIdx.x, tid = threadIdx.x;
    const int n = lay.base[lvl + 1] - lay.base[lvl];
    int* const a = rowblk + lay.base[lvl];
    const int per = (n + 1023) / 1024, lo = min(tid * per, n), hi = min(lo + per, n);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                                  // inclusive scan of the threads' sums
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int i = lo; i < hi; ++i) { const int c = a[i]; a[i] = run; run += c; }
    if (tid == 1023) counters[lvl] = s_part[1023];
}

__global__ void __launch_bounds__(256) k_fast_nms_emit(const uint8_t* __restrict__ scores, OrbLevelSet S, NmsLayout lay, const int* __restrict__ rowblk,
                                                       const uint8_t* __restrict__ keepbuf, size_t keep_stride, const int* __restrict__ counters,
                                                       int* __restrict__ cand, int cap) {
    __shared__ int s_cnt[kNmsRows][4];
    const int lvl = blockIdx.z;
    // the levels' lists lie one behind the other (the host fetches counts and candidates in ONE copy of a guessed length): this level starts
    // where the survivors of the finer levels end (the scan kernel's totals)
    int level_base = 0;
    for (int l = 0; l < lvl; ++l) level_base += min(counters[l], cap);
    const int total_cap = kOrbLevels * cap;
    const OrbLevel L = S.lv[lvl];
    const int W = L.w, nbx = lay.nbx[lvl];
    const int bx = blockIdx.x, y0 = blockIdx.y * kNmsRows;
    if (bx >= nbx || y0 >= L.h) return;
    const int x = bx * 256 + threadIdx.x;
    const unsigned keep = x < W ? keepbuf[(size_t)lvl * keep_stride + (size_t)blockIdx.y * W + x] : 0u;
    if (__syncthreads_or(keep != 0) == 0) return;                         // uniform
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int rank[kNmsRows];
#pragma unroll
    for (int r = 0; r < kNmsRows; ++r) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((keep >> r) & 1u);
        rank[r] = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[r][wv] = __builtin_popcountll(m);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kNmsRows; ++r)
        if (keep & (1u << r)) {
            int slot = rowblk[lay.base[lvl] + (y0 + r) * nbx + bx] + rank[r];
            for (int w = 0; w < wv; ++w) slot += s_cnt[r][w];
            if (slot < cap && level_base + slot < total_cap) {
                const int pos = (y0 + r) * W + x;
                cand[((size_t)level_base + slot) * 2] = x | ((y0 + r) << 16);          // both fit 16 bits (levels up to 65535 x 65535)
                cand[((size_t)level_base + slot) * 2 + 1] = scores[L.score_offset + pos];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Harris response and intensity-centroid angle, one thread per keypoint: kp = (level, x, y)
// ------------------------------------------------------------------------------------------------
__global__ void k_harris(const uint8_t* __restrict__ atlas, OrbLevelSet S, const int* __restrict__ kp, int n, float* __restrict__ resp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const OrbLevel L = S.lv[kp[3 * i]];
    const int st = (int)L.stride;
    const uint8_t* c = atlas + L.offset + (size_t)(kp[3 * i + 2] + kOrbBorder) * L.stride + (kp[3 * i + 1] + kOrbBorder);
    int a = 0, b = 0, cc = 0;
    for (int dy = -3; dy <= 3; ++dy)
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t* p = c + dy * st + dx;
            int Ix = ((int)p[1] - p[-1]) * 2 + ((int)p[-st + 1] - p[-st - 1]) + ((int)p[st + 1] - p[st - 1]);
            int Iy = ((int)p[st] - p[-st]) * 2 + ((int)p[st - 1] - p[-st - 1]) + ((int)p[st + 1] - p[-st + 1]);
            a += Ix * Ix; b += Iy * Iy; cc += Ix * Iy;
        }
    const float harris_k = 0.04f;
    float scale = 1.f / ((1 << 2) * 7 * 255.f);
    float s4 = scale * scale * scale * scale;
    resp[i] = ((float)a * (float)b - (float)cc * (float)cc - harris_k * ((float)a + (float)b) * ((float)a + (float)b)) * s4;
}

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.141592653589793238462643383279502884197169399375105820974944592307816406286),
                p3 = -0.3258083974640975f * (float)(180 / 3.141592653589793238462643383279502884197169399375105820974944592307816406286),
                p5 = 0.1555786518463281f * (float)(180 / 3.141592653589793238462643383279502884197169399375105820974944592307816406286),
                p7 = -0.04432655554792128f * (float)(180 / 3.141592653589793238462643383279502884197169399375105820974944592307816406286);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// one wave per keypoint: lane = patch row v = lane - 15 (31 rows), integer moments reduced across the wave (the sums are integers: any
// order gives the reference's m01 / m10, orb.cpp:181-215)
__global__ void __launch_bounds__(64) k_ic_angle(const uint8_t* __restrict__ atlas, OrbLevelSet S, const int* __restrict__ kp, int n, float* __restrict__ angle) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const OrbLevel L = S.lv[kp[3 * i]];
    const int st = (int)L.stride;
    const uint8_t* c = atlas + L.offset + (size_t)(kp[3 * i + 2] + kOrbBorder) * L.stride + (kp[3 * i + 1] + kOrbBorder);
    int m01 = 0, m10 = 0;
    if (lane < 31) {
        const int v = lane - 15, d = c_umax[v < 0 ? -v : v];
        const uint8_t* row = c + v * st;
        int rsum = 0;
        for (int u = -d; u <= d; ++u) { const int p = row[u]; rsum += p; m10 += u * p; }
        m01 = v * rsum;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m01 += __shfl_down(m01, o); m10 += __shfl_down(m10, o); }
    if (lane == 0) angle[i] = fast_atan2_deg((float)m01, (float)m10);
}

// ------------------------------------------------------------------------------------------------
// rBRIEF.  Blur: the level is blurred inside the padded atlas by the generic float separable filter
// (smooth.dispatch.cpp:646 skips the fixed-point path for a non-isolated submatrix): row pass
// acc = k0*s0; acc += kj*sj (filter.simd.hpp:2477-2487), column pass acc = k3*c + 0; acc += kj*(c[+j] + c[-j])
// (:2753-2759), cvRound to u8.  The border ring keeps the unblurred pixels, exactly like the in-place reference.
// ------------------------------------------------------------------------------------------------
__constant__ float c_gauss7[7] = {0x1.1f5f62p-4f, 0x1.0c70fcp-3f, 0x1.869472p-3f, 0x1.ba95c0p-3f, 0x1.869472p-3f, 0x1.0c70fcp-3f, 0x1.1f5f62p-4f};
__constant__ signed char c_pattern[1024] = {
#include "orb_pattern.inc"
};

__global__ void __launch_bounds__(256) k_orb_blur(const uint8_t* __restrict__ atlas, uint8_t* __restrict__ blurred, OrbLevelSet S) {
    const OrbLevel L = S.lv[blockIdx.z];
    int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= L.w + 2 * kOrbBorder || py >= L.h + 2 * kOrbBorder) return;
    const size_t o = L.offset + (size_t)py * L.stride + px;
    int x = px - kOrbBorder, y = py - kOrbBorder;
    if (x < 0 || x >= L.w || y < 0 || y >= L.h) { blurred[o] = atlas[o]; return; }
    const uint8_t* c = atlas + o;
    const int st = (int)L.stride;
    float row[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint8_t* s = c + (r - 3) * st;
        float acc = c_gauss7[0] * (float)s[-3];
#pragma unroll
        for (int k = 1; k < 7; ++k) acc += c_gauss7[k] * (float)s[k - 3];
        row[r] = acc;
    }
    float acc = c_gauss7[3] * row[3] + 0.f;
#pragma unroll
    for (int k = 1; k <= 3; ++k) acc += c_gauss7[3 + k] * (row[3 + k] + row[3 - k]);
    int v = (fabsf(acc) < 2147483648.f) ? __float2int_rn(acc) : INT_MIN;
    blurred[o] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

// one thread per (keypoint, descriptor byte).  cs = (cos, sin) of the keypoint angle, computed on the host
// with the same libm call as the reference (orb.cpp:236).
__global__ void __launch_bounds__(256) k_orb_describe(const uint8_t* __restrict__ blurred, OrbLevelSet S, const int* __restrict__ kp,
                                                      const float2* __restrict__ cs, int n, uint8_t* __restrict__ desc) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int j = t >> 5, byte = t & 31;
    if (j >= n) return;
    const OrbLevel L = S.lv[kp[3 * j]];
    const int st = (int)L.stride;
    const uint8_t* center = blurred + L.offset + (size_t)(kp[3 * j + 2] + kOrbBorder) * L.stride + (kp[3 * j + 1] + kOrbBorder);
    const float a = cs[j].x, b = cs[j].y;
    const signed char* pat = c_pattern + byte * 32;
    int val = 0;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
        int tv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float px = (float)pat[bit * 4 + q * 2], py = (float)pat[bit * 4 + q * 2 + 1];
            float x = px * a - py * b, y = px * b + py * a;
            tv[q] = center[__float2int_rn(y) * st + __float2int_rn(x)];
        }
        val |= (tv[0] < tv[1]) << bit;
    }
    desc[(size_t)j * 32 + byte] = (uint8_t)val;
}

// ------------------------------------------------------------------------------------------------
// Brute-force Hamming 1-NN (BFMatcher(NORM_HAMMING).match).  One wave per query descriptor; the train set is
// streamed through LDS in tiles shared by the 4 waves of the block; xor + popcount on 8 dwords; the running
// best is (distance, index) with the lower index winning ties, reduced across the wave with shuffles.
// ------------------------------------------------------------------------------------------------
constexpr int kHamTile = 512;
// LDS tile of train descriptors, word-major: tile[k * kHamStride + j] = word k of descriptor j.  The lanes of a wave read consecutive j
// (one bank each); descriptor-major (tile[j * 8 + k]) put them 8 dwords apart, an 8-way bank conflict on every read.
constexpr int kHamStride = kHamTile + 1;

__global__ void __launch_bounds__(256) k_hamming(const uint32_t* __restrict__ query, int nq, const uint32_t* __restrict__ train, int nt, int* __restrict__ out2) {
    __shared__ uint32_t tile[kHamStride * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + wave;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = qi < nq ? query[(size_t)qi * 8 + k] : 0u;
    int bestD = INT_MAX, bestJ = INT_MAX;
    for (int base = 0; base < nt; base += kHamTile) {
        const int cnt = min(kHamTile, nt - base);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 8; e += 256) tile[(e & 7) * kHamStride + (e >> 3)] = train[(size_t)base * 8 + e];
        __syncthreads();
        for (int j = lane; j < cnt; j += 64) {
            int d = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += __popc(q[k] ^ tile[k * kHamStride + j]);
            if (d < bestD) { bestD = d; bestJ = base + j; }      // ascending j per lane: strict < keeps the lowest index
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int od = __shfl_xor(bestD, off), oj = __shfl_xor(bestJ, off);
        if (od < bestD || (od == bestD && oj < bestJ)) { bestD = od; bestJ = oj; }
    }
    if (lane == 0 && qi < nq) { out2[2 * qi] = bestJ; out2[2 * qi + 1] = bestD; }
}

// BFMatcher::knnMatch(k = 2): the two nearest train descriptors per query, ordered by (distance, train index) — OpenCV's
// sorted insertion uses strict comparisons, so equal distances keep the lower index first (batch_distance.cpp:225-248).
// Same streaming as k_hamming; each lane keeps its two best, the wave merges them.  out4 = (idx0, d0, idx1, d1), -1 = none.
__device__ __forceinline__ bool knn_less(int d, int j, int od, int oj) { return d < od || (d == od && j < oj); }

__global__ void __launch_bounds__(256) k_hamming_knn2(const uint32_t* __restrict__ query, int nq, const uint32_t* __restrict__ train, int nt, int* __restrict__ out4) {
    __shared__ uint32_t tile[kHamStride * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = blockIdx.x * 4 + wave;
    uint32_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = qi < nq ? query[(size_t)qi * 8 + k] : 0u;
    int d0 = INT_MAX, j0 = INT_MAX, d1 = INT_MAX, j1 = INT_MAX;
    for (int base = 0; base < nt; base += kHamTile) {
        const int cnt = min(kHamTile, nt - base);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 8; e += 256) tile[(e & 7) * kHamStride + (e >> 3)] = train[(size_t)base * 8 + e];
        __syncthreads();
        for (int j = lane; j < cnt; j += 64) {
            int d = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) d += __popc(q[k] ^ tile[k * kHamStride + j]);
            const int jj = base + j;                              // ascending per lane
            if (d < d0) { d1 = d0; j1 = j0; d0 = d; j0 = jj; }
            else if (d < d1) { d1 = d; j1 = jj; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {                      // merge two sorted pairs, keep the best two
        const int a0 = __shfl_xor(d0, off), b0 = __shfl_xor(j0, off), a1 = __shfl_xor(d1, off), b1 = __shfl_xor(j1, off);
        if (knn_less(a0, b0, d0, j0)) {
            if (knn_less(a1, b1, d0, j0)) { d1 = a1; j1 = b1; } else { d1 = d0; j1 = j0; }
            d0 = a0; j0 = b0;
        } else if (knn_less(a0, b0, d1, j1)) { d1 = a0; j1 = b0; }
    }
    if (lane == 0 && qi < nq) {
        out4[4 * qi] = d0 == INT_MAX ? -1 : j0; out4[4 * qi + 1] = d0 == INT_MAX ? -1 : d0;
        out4[4 * qi + 2] = d1 == INT_MAX ? -1 : j1; out4[4 * qi + 3] = d1 == INT_MAX ? -1 : d1;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out4, hipStream_t s) {
    if (nq > 0) hipLaunchKernelGGL(k_hamming_knn2, dim3((nq + 3) / 4), dim3(256), 0, s, (const uint32_t*)query, nq, (const uint32_t*)train, nt, out4);
}
void launch_orb_blur(const uint8_t* atlas, uint8_t* blurred, const OrbLevelSet& S, hipStream_t s) {
    dim3 grid((S.lv[0].w + 2 * kOrbBorder + 255) / 256, S.lv[0].h + 2 * kOrbBorder, S.n);
    hipLaunchKernelGGL(k_orb_blur, grid, dim3(256), 0, s, atlas, blurred, S);
}
void launch_orb_describe(const uint8_t* blurred, const OrbLevelSet& S, const int* kp, const float* cos_sin, int n, uint8_t* desc, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_orb_describe, dim3((n * 32 + 255) / 256), dim3(256), 0, s, blurred, S, kp, (const float2*)cos_sin, n, desc);
}
void launch_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out2, hipStream_t s) {
    if (nq > 0) hipLaunchKernelGGL(k_hamming, dim3((nq + 3) / 4), dim3(256), 0, s, (const uint32_t*)query, nq, (const uint32_t*)train, nt, out2);
}

void launch_orb_pyramid(const uint8_t* d_img, int w, int h, size_t stride, uint8_t* atlas, const OrbLevelSet& S, hipStream_t s) {
    for (int l = 0; l < S.n; ++l) {
        const OrbLevel& L = S.lv[l];
        dim3 grid((L.w + 2 * kOrbBorder + 255) / 256, L.h + 2 * kOrbBorder);
        if (l == 0) {
            hipLaunchKernelGGL(k_orb_level0, grid, dim3(256), 0, s, d_img, w, h, stride, atlas, L);
        } else {
            const uint8_t* src; int sw, sh; size_t sst;
            if (l == 1) { src = d_img; sw = w; sh = h; sst = stride; }
            else { const OrbLevel& P = S.lv[l - 1]; src = atlas + P.offset + (size_t)kOrbBorder * P.stride + kOrbBorder; sw = P.w; sh = P.h; sst = P.stride; }
            double sx = 1.0 / ((double)L.w / sw), sy = 1.0 / ((double)L.h / sh);
            hipLaunchKernelGGL(k_orb_resize, grid, dim3(256), 0, s, src, sw, sh, sst, atlas, L, sx, sy);
        }
    }
}

size_t fast_nms_scratch_bytes(const OrbLevelSet& S) {
    size_t rows = 0, keep = 0;
    for (int l = 0; l < S.n; ++l) {
        rows += (size_t)S.lv[l].h * ((S.lv[l].w + 255) / 256);
        keep = std::max(keep, (size_t)((S.lv[l].h + kNmsRows - 1) / kNmsRows) * S.lv[l].w);
    }
    return rows * sizeof(int) + 256 + (size_t)S.n * ((keep + 255) & ~(size_t)255);
}
void launch_fast(const uint8_t* atlas, const OrbLevelSet& S, uint8_t* scores, int threshold, int edge, int* counters, int* cand, int cap, void* scratch, hipStream_t s) {
    const dim3 grid_score((S.lv[0].w + 255) / 256, (S.lv[0].h + kFastRows - 1) / kFastRows, S.n);
    const dim3 grid_nms((S.lv[0].w + 255) / 256, (S.lv[0].h + kNmsRows - 1) / kNmsRows, S.n);
    NmsLayout lay;
    size_t rows = 0, keep = 0;
    for (int l = 0; l < kOrbLevels; ++l) {
        lay.base[l] = (int)rows;
        lay.nbx[l] = l < S.n ? (S.lv[l].w + 255) / 256 : 0;
        if (l < S.n) { rows += (size_t)S.lv[l].h * lay.nbx[l]; keep = std::max(keep, (size_t)((S.lv[l].h + kNmsRows - 1) / kNmsRows) * S.lv[l].w); }
    }
    lay.base[kOrbLevels] = (int)rows;
    for (int l = S.n; l < kOrbLevels; ++l) lay.base[l] = (int)rows;
    const size_t keep_stride = (keep + 255) & ~(size_t)255;
    int* rowblk = (int*)scratch;
    uint8_t* keepbuf = (uint8_t*)scratch + ((rows * sizeof(int) + 255) & ~(size_t)255);
    hipLaunchKernelGGL(k_fast_score, grid_score, dim3(256), 0, s, atlas, S, scores, threshold);
    hipLaunchKernelGGL(k_fast_nms_count, grid_nms, dim3(256), 0, s, scores, S, lay, edge, rowblk, keepbuf, keep_stride);
    hipLaunchKernelGGL(k_fast_nms_scan, dim3(S.n), dim3(1024), 0, s, lay, rowblk, counters);
    hipLaunchKernelGGL(k_fast_nms_emit, grid_nms, dim3(256), 0, s, scores, S, lay, rowblk, keepbuf, keep_stride, counters, cand, cap);
}

void launch_harris(const uint8_t* atlas, const OrbLevelSet& S, const int* kp, int n, float* resp, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_harris, dim3((n + 63) / 64), dim3(64), 0, s, atlas, S, kp, n, resp);
}
void launch_ic_angle(const uint8_t* atlas, const OrbLevelSet& S, const int* kp, int n, float* angle, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_ic_angle, dim3(n), dim3(64), 0, s, atlas, S, kp, n, angle);
}

}  // namespace poppy_hip
