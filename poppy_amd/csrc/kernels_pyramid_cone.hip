// kernels_pyramid_cone.hip — the small levels of the Laplacian blend's COLLAPSE in one launch (round 6).
//
//   src/blend.hpp:58-77   cur = pyrUp(cur) + (lapL_l * m_l + lapR_l * (1 - m_l)),  lap_l = G_l - pyrUp(G_{l+1})
//   OCV/imgproc/src/pyramids.cpp:903-1005 (pyrUp)
//
// Until round 5 the way up from the tail kernel's level to level 1 was three dependent launches at 1080p (k_collapse2 x 2, k_collapse_level<false>:
// 24.5 us of 148 per chained frame, 7-9 us each for work that is a few hundred KB) and five at 4K.  Going UP, a pixel of level l depends on a
// 3 x 3 neighbourhood of level l+1 only — the cone under a tile does not grow, it converges to ~6 x 6 pixels per level — so a workgroup that owns a
// 64 x 16 tile of the output level can rebuild the blended levels under its tile itself, out of LDS: one trip to memory for the Gaussian levels'
// patches (L, R, mask of every level under the tile, the tail's blended level at the bottom), then one short LDS stage per level.  The
// recomputation is ~1.9x the outputs of levels that hold 4 % of a frame's pixels; what it removes is the launch floor + the dependent memory round trips
// of every level in between.  Every value is the per-element expression tree used everywhere else (pyrup_elem_patch below = pyrup_elem_wide's tree on an
// LDS patch; mix_lr), so the result is bit-identical to the per-level kernels'.
#include "kernels.h"
#include "pyramid_device.h"

namespace poppy_hip {

namespace {

__device__ __forceinline__ int div_small_c(int e, int d, float inv) {     // e / d for 0 <= e < 2^20, inv ~ 1.f / d (the hardware's reciprocal will do: the quotient is corrected)
    int q = (int)((float)e * inv);
    const int r = e - q * d;
    return r < 0 ? q - 1 : (r >= d ? q + 1 : q);
}

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 ld3(const float* __restrict__ p) { return F3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* __restrict__ p, F3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// pyrUp as a thread sees it here: ONE source pixel (spx, sy) of the low-resolution level and the 2 x 2 block of output pixels
// (2 spx + {0, 1}, 2 sy + {0, 1}) it stands under.  The four outputs share the 3 x 3 source neighbourhood and the six horizontal values
// (even / odd column form of the rows sy-1, sy, sy+1), pyramids.cpp:945-977 (rows) and :987-992 (columns); computing them per output,
// as the per-element kernels do, costs 4.5 times the arithmetic (the first build of this kernel did: 32 us where this one takes well under half;
// the kernel is bound by the vector instructions it issues, not by a wave's latency: a thread per block AND channel — three times the waves, a third
// of the work each — took 19 us against 14.8).
// The patch holds low(py0 + r, px0 + p, c) at patch[r * pstride + p * 3 + c]; the low level is sw x sh, at least 2 x 2.
struct UpBlock { int r0, r1, r2, cm, c0, cp; bool left, right, edge; };
__device__ __forceinline__ UpBlock up_block_taps(int pstride, int px0, int py0, int sw, int sh, int spx, int sy) {
    UpBlock u;
    const int sym = sy >= 1 ? sy - 1 : 1, syp = sy + 1 <= sh - 1 ? sy + 1 : sh - 1;      // reflect-101 of the doubled rows (pyramids.cpp:979-985)
    u.r0 = (sym - py0) * pstride; u.r1 = (sy - py0) * pstride; u.r2 = (syp - py0) * pstride;
    u.cm = (max(spx - 1, 0) - px0) * 3; u.c0 = (spx - px0) * 3; u.cp = (min(spx + 1, sw - 1) - px0) * 3;
    u.left = spx == 0; u.right = !u.left && spx >= sw - 1; u.edge = u.left || u.right;      // (sw >= 2: never both)
    return u;
}
// out[row parity][column parity], one plane (three channels)
__device__ __forceinline__ void up_block(const float* __restrict__ patch, const UpBlock& u, F3 out[2][2]) {
    float o[2][2][3];
    // the three edge forms of a row (pyramids.cpp:945-977) with shared operations and operand selects instead of three expression trees
    // behind branches:  left  s0*6 + sp*2   right  sm + s0*7   else  sm + s0*6 + sp   (sp*2 = sp + sp and the first sum commutes: same bits);
    // odd column:  right  s0*8 = (s0 + s0)*4   else  (s0 + sp)*4
    const float k = u.right ? 7.f : 6.f;
    const int rows[3] = {u.r0, u.r1, u.r2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float ev[3], od[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float sm = patch[rows[q] + u.cm + c], s0 = patch[rows[q] + u.c0 + c], sp = patch[rows[q] + u.cp + c];
            const float x = u.left ? sp + sp : sm;
            const float t = x + s0 * k;
            const float t2 = t + sp;
            ev[q] = u.edge ? t : t2;
            od[q] = (s0 + (u.right ? s0 : sp)) * 4.f;
        }
        const float s = 1.f / 64;
        o[0][0][c] = (ev[0] + ev[1] * 6.f + ev[2]) * s; o[0][1][c] = (od[0] + od[1] * 6.f + od[2]) * s;
        o[1][0][c] = ((ev[1] + ev[2]) * 4.f) * s;       o[1][1][c] = ((od[1] + od[2]) * 4.f) * s;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) out[a][b] = F3{o[a][b][0], o[a][b][1], o[a][b][2]};
}
__device__ __forceinline__ F3 blend_px(F3 l, F3 r, float m, F3 uL, F3 uR, F3 uB) {      // blend.hpp:67-77 for one pixel
    return F3{uB.x + mix_lr(l.x - uL.x, r.x - uR.x, m), uB.y + mix_lr(l.y - uL.y, r.y - uR.y, m), uB.z + mix_lr(l.z - uL.z, r.z - uR.z, m)};
}

constexpr int kTx = kConeTileX, kTy = kConeTileY;            // tile of the output level: 64 x 16 = 32 x 8 blocks of 2 x 2 pixels, one per thread
constexpr int kThreads = (kTx / 2) * (kTy / 2);
// The region of level j+1 under a region [a, b] of level j, plus the ring pyrUp reads: [max((a >> 1) - 1, 0), min((b >> 1) + 1, w - 1)]:
// at most ((b - a + 1) >> 1) + 3 pixels (2 when a is even).  From 64 x 16 at an even origin: 34 x 10, then 20 x 8, 13 x 7, 9 x 6, 7 x 6, 6 x 6.
// The cone's stages cost the same for a tile twice as high (the deeper regions hardly grow), which is why the tile is not 64 x 8 (14.8 us against 12).
constexpr int kPw1 = kTx / 2 + 2, kPwN = (kPw1 >> 1) + 3;    // pixels per row of the level-1 slot and of the deeper levels' slots
constexpr int kPh1 = kTy / 2 + 2, kPhN = (kPh1 >> 1) + 3;    // rows
constexpr int kS1 = kPw1 * 3 + 2, kSN = kPwN * 3 + 2;        // row strides in floats
constexpr int kSlot1 = kPh1 * kS1, kSlotN = kPhN * kSN;
static_assert(kPw1 * kPh1 <= 2 * kThreads && kPwN * kPhN <= kThreads, "two pixels of level 1's region, one of every deeper region per thread");

struct Region { int x0, y0, pw, ph; };                       // first pixel, size in pixels

template <int NL>
__global__ void __launch_bounds__(kThreads) k_collapse_cone(const float* __restrict__ pyrL, const float* __restrict__ pyrR, const float* __restrict__ pyrM,
                                                             float* __restrict__ pyrB, ConeArgs a, int tiles_x, int stagger) {
    // planes of level j = 1 .. NL (relative to the output level): L, R, blended; the mask of levels 1 .. NL-1
    __shared__ float s1[3][kSlot1], sN[NL > 1 ? NL - 1 : 1][3][kSlotN], sM1[kPh1 * kPw1], sMN[NL > 1 ? NL - 1 : 1][kPhN * kPwN];
    const int tid = threadIdx.x;
    // tiles in row-major order, an XCD's workgroups a contiguous run of them: the band of the output level an XCD writes is the band the next
    // launch's workgroups on that XCD read (k_collapse_level numbers its blocks the same way), so it is still in that XCD's L2
    // (without: k_collapse_level<true> 18.9 -> 23.7 us at 1080p)
    stagger_priority(blockIdx.x, stagger);
    const int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = blk / tiles_x, tx = blk - ty * tiles_x;
    const ConeLevel& l0 = a.lv[0];
    const int x0 = tx * kTx, y0 = ty * kTy;
    const int x1 = min(x0 + kTx - 1, l0.w - 1), y1 = min(y0 + kTy - 1, l0.h - 1);
    Region rg[NL + 1];
    rg[0] = Region{x0, y0, x1 - x0 + 1, y1 - y0 + 1};
#pragma unroll
    for (int j = 1; j <= NL; ++j) {
        const Region& p = rg[j - 1];
        const int ax0 = max((p.x0 >> 1) - 1, 0), ax1 = min(((p.x0 + p.pw - 1) >> 1) + 1, a.lv[j].w - 1);
        const int ay0 = max((p.y0 >> 1) - 1, 0), ay1 = min(((p.y0 + p.ph - 1) >> 1) + 1, a.lv[j].h - 1);
        rg[j] = Region{ax0, ay0, ax1 - ax0 + 1, ay1 - ay0 + 1};
    }
    auto plane = [&](int j, int k) -> float* { return j == 1 ? s1[k] : sN[j - 2][k]; };       // (j, k are compile-time constants after unrolling)
    auto maskp = [&](int j) -> float* { return j == 1 ? sM1 : sMN[j - 2]; };
    auto stride = [](int j) { return j == 1 ? kS1 : kSN; };
    auto mstride = [](int j) { return j == 1 ? kPw1 : kPwN; };
    // ---- one trip to memory: every level's region (L, R, and the mask or — bottom level — the blended image; a thread holds two pixels of level 1's
    // region, one of every other) and the thread's own 2 x 2 block of the tile ----
    F3 vl[NL + 1], vr[NL + 1], vb;
    float vm[NL + 1];
    int at[NL + 1], atm[NL + 1];
#pragma unroll
    for (int j = 0; j <= NL; ++j) {                           // j = 0: the second pixel of level 1's region
        const int jj = j == 0 ? 1 : j;
        const ConeLevel& lv = a.lv[jj];
        const int n = rg[jj].pw * rg[jj].ph;
        const int t = j == 0 ? tid + kThreads : tid;
        const bool on = t < n;
        const int e = on ? t : 0;                             // (a thread without a pixel loads pixel 0: a valid address)
        const int r = div_small_c(e, rg[jj].pw, __builtin_amdgcn_rcpf((float)rg[jj].pw)), c = e - r * rg[jj].pw;
        const size_t g1 = (size_t)(rg[jj].y0 + r) * lv.pitch + rg[jj].x0 + c;
        const size_t g = lv.off3 + g1 * 3;
        at[j] = on ? r * stride(jj) + c * 3 : -1;
        atm[j] = r * mstride(jj) + c;
        vl[j] = ld3(pyrL + g); vr[j] = ld3(pyrR + g);
        if (jj == NL) vb = ld3(pyrB + g);
        else vm[j] = pyrM[lv.off1 + g1];
    }
    static_assert(NL >= 2, "level 1 is never the bottom level (its second pixel has no blended value to stage)");
    const int bx = tid & (kTx / 2 - 1), by = tid >> 5;        // the thread's block of the tile: pixels (x0 + 2 bx + {0, 1}, y0 + 2 by + {0, 1})
    static_assert(kTx / 2 == 32, "by = tid >> 5");
    F3 gl[2][2], gr[2][2];
    float gm[2][2];
    bool gon[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = x0 + 2 * bx + dx, y = y0 + 2 * by + dy;
            gon[dy][dx] = x < l0.w && y < l0.h;
            const size_t g1 = (size_t)(gon[dy][dx] ? y : y0) * l0.pitch + (gon[dy][dx] ? x : x0);
            gl[dy][dx] = ld3(pyrL + l0.off3 + g1 * 3); gr[dy][dx] = ld3(pyrR + l0.off3 + g1 * 3); gm[dy][dx] = pyrM[l0.off1 + g1];
        }
    // every load above is on its way before the first value is asked for: the tile's own pixels arrive with the patches, not a trip later
    asm volatile("" :: "v"(gl[0][0].x), "v"(gr[0][0].x), "v"(gm[0][0]), "v"(gl[1][1].z), "v"(gr[1][1].z), "v"(gm[1][1]));
#pragma unroll
    for (int j = 0; j <= NL; ++j) {
        const int jj = j == 0 ? 1 : j;
        if (at[j] >= 0) {
            st3(plane(jj, 0) + at[j], vl[j]); st3(plane(jj, 1) + at[j], vr[j]);
            if (jj == NL) st3(plane(jj, 2) + at[j], vb);
            else maskp(jj)[atm[j]] = vm[j];
        }
    }
    __syncthreads();
    // ---- the blended levels under the tile, coarsest first: B_j = pyrUp(B_{j+1}) + mix(L_j - pyrUp(L_{j+1}), R_j - pyrUp(R_{j+1}), M_j).
    // A thread = one pixel of level j+1 whose 2 x 2 block of level j touches the region of level j ----
#pragma unroll
    for (int j = NL - 1; j >= 1; --j) {
        const int bx0 = rg[j].x0 >> 1, nbx = ((rg[j].x0 + rg[j].pw - 1) >> 1) - bx0 + 1;
        const int by0 = rg[j].y0 >> 1, nby = ((rg[j].y0 + rg[j].ph - 1) >> 1) - by0 + 1;
        if (tid < nbx * nby) {
            const int r = div_small_c(tid, nbx, __builtin_amdgcn_rcpf((float)nbx)), c = tid - r * nbx;
            const int spx = bx0 + c, sy = by0 + r;
            const UpBlock u = up_block_taps(stride(j + 1), rg[j + 1].x0, rg[j + 1].y0, a.lv[j + 1].w, a.lv[j + 1].h, spx, sy);
            F3 uL[2][2], uR[2][2], uB[2][2];
            up_block(plane(j + 1, 0), u, uL); up_block(plane(j + 1, 1), u, uR); up_block(plane(j + 1, 2), u, uB);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int lx = 2 * spx + dx - rg[j].x0, ly = 2 * sy + dy - rg[j].y0;
                    const bool in = (unsigned)lx < (unsigned)rg[j].pw && (unsigned)ly < (unsigned)rg[j].ph;      // inside the region?  (otherwise: pixel 0's values, not stored)
                    const int o = in ? ly * stride(j) + lx * 3 : 0, om = in ? ly * mstride(j) + lx : 0;
                    const F3 b = blend_px(ld3(plane(j, 0) + o), ld3(plane(j, 1) + o), maskp(j)[om], uL[dy][dx], uR[dy][dx], uB[dy][dx]);
                    if (in) st3(plane(j, 2) + o, b);
                }
        }
        __syncthreads();
    }
    // ---- the tile ----
    if (gon[0][0]) {                                          // (a block that starts outside the level stores nothing)
        const UpBlock u = up_block_taps(kS1, rg[1].x0, rg[1].y0, a.lv[1].w, a.lv[1].h, (x0 >> 1) + bx, (y0 >> 1) + by);
        F3 uL[2][2], uR[2][2], uB[2][2];
        up_block(s1[0], u, uL); up_block(s1[1], u, uR); up_block(s1[2], u, uB);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                if (gon[dy][dx]) {
                    const F3 b = blend_px(gl[dy][dx], gr[dy][dx], gm[dy][dx], uL[dy][dx], uR[dy][dx], uB[dy][dx]);
                    st3(pyrB + l0.off3 + ((size_t)(y0 + 2 * by + dy) * l0.pitch + x0 + 2 * bx + dx) * 3, b);
                }
    }
}

}  // namespace

// levels[0] = the output level, levels[n] = the level whose blended image is known; every level below the output at least 2 x 2
bool collapse_cone_eligible(const PyrLevel* levels, int n) {
    if (n < 2 || n > kConeMaxLevels) return false;
    for (int j = 1; j <= n; ++j) if (levels[j].w < 2 || levels[j].h < 2) return false;
    if ((size_t)levels[0].w * levels[0].h > kConeMaxPixels) return false;
    return levels[0].off3 + (size_t)levels[0].pitch * levels[0].h * 3 < (1ull << 32);
}

void launch_collapse_cone(const float* pyrL, const float* pyrR, const float* pyrM, float* pyrB, const PyrLevel* levels, int n, hipStream_t s) {
    ConeArgs a;
    a.n = n;
    for (int j = 0; j <= n; ++j) a.lv[j] = ConeLevel{levels[j].w, levels[j].h, levels[j].pitch, (unsigned)levels[j].off3, (unsigned)levels[j].off1};
    const int tiles_x = (levels[0].w + kTx - 1) / kTx, tiles_y = (levels[0].h + kTy - 1) / kTy;
    const dim3 grid(tiles_x * tiles_y), block(kThreads);
    switch (n) {
        case 2: hipLaunchKernelGGL(k_collapse_cone<2>, grid, block, 0, s, pyrL, pyrR, pyrM, pyrB, a, tiles_x, stagger_flag(4)); break;
        case 3: hipLaunchKernelGGL(k_collapse_cone<3>, grid, block, 0, s, pyrL, pyrR, pyrM, pyrB, a, tiles_x, stagger_flag(4)); break;
        case 4: hipLaunchKernelGGL(k_collapse_cone<4>, grid, block, 0, s, pyrL, pyrR, pyrM, pyrB, a, tiles_x, stagger_flag(4)); break;
        case 5: hipLaunchKernelGGL(k_collapse_cone<5>, grid, block, 0, s, pyrL, pyrR, pyrM, pyrB, a, tiles_x, stagger_flag(4)); break;
        default: hipLaunchKernelGGL(k_collapse_cone<6>, grid, block, 0, s, pyrL, pyrR, pyrM, pyrB, a, tiles_x, stagger_flag(4)); break;
    }
}

}  // namespace poppy_hip
