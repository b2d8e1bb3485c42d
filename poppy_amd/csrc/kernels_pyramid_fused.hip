// kernels_pyramid_fused.hip — two pyramid levels per launch for the small levels of the Laplacian blend.
//
// Below ~480x270 a pyramid level is a 5-6 us launch whose time is the launch floor plus two dependent HBM round trips,
// not its work (profiles/r01_e_streams.md).  The chained frame loop has ~10 of those on its critical path.  These
// kernels halve that count without any inter-workgroup synchronisation: a workgroup owns a tile of the SECOND level
// of the pair and computes the part of the intermediate level that tile needs (tile + halo) into LDS itself.  The halo
// is recomputed by the neighbouring workgroups too (1.5-1.7x redundant work on levels that have almost none).
//
// Every value is produced by the same per-element expression trees as everywhere else (pyramid_device.h: the
// association split points of pyrDown depend on ABSOLUTE element positions, which is why the patch variants below keep
// absolute coordinates and only translate the final LDS address).  Reference: OCV/imgproc/src/pyramids.cpp:745-1005,
// src/blend.hpp:25-77.
#include "kernels.h"
#include "pyramid_device.h"

namespace poppy_hip {

namespace {

__device__ __forceinline__ int div_small_i(int e, int d, float inv) {     // e / d for 0 <= e < 2^20, inv = 1.f / d
    int q = (int)((float)e * inv);
    const int r = e - q * d;
    return r < 0 ? q - 1 : (r >= d ? q + 1 : q);
}

// ---- pyrDown out of an LDS patch of the source level ------------------------------------------------------------------
// patch(row r, element e) = source(py0 + r, pxe0 + e); the source level itself is g.sw x g.sh (>= 3 x 3).
template <int CN>
__device__ __forceinline__ float pyrdown_elem_patch(const float* __restrict__ patch, int pstride, int pxe0, int py0,
                                                    const DownGeom& g, int y, int xe) {
    const int px = xe / CN, c = xe - px * CN;
    const bool hBody = (xe >= CN) && (xe < g.hBodyEnd);
    int col[5], rowo[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        col[k] = reflect101_once(2 * px + k - 2, g.sw) * CN + c - pxe0;
        rowo[k] = (reflect101_once(2 * y + k - 2, g.sh) - py0) * pstride;
    }
    float t[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int m = 0; m < 5; ++m) t[k][m] = patch[rowo[k] + col[m]];
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
        r[k] = hBody ? t[k][2] * 6.f + ((t[k][1] + t[k][3]) * 4.f + (t[k][0] + t[k][4]))
                     : t[k][2] * 6.f + (t[k][1] + t[k][3]) * 4.f + t[k][0] + t[k][4];
    const float s = 1.f / 256;
    return (xe < g.vBodyEnd) ? ((r[1] + r[3] + r[2]) * 4.f + (r[0] + r[4] + (r[2] + r[2]))) * s
                             : (r[2] * 6.f + (r[1] + r[3]) * 4.f + r[0] + r[4]) * s;
}

constexpr int kD2Tx = 8, kD2Ty = 4;                     // tile of the second destination level, pixels
constexpr int kD2Ps = (2 * kD2Tx + 3) * 3 + 3;          // B patch row stride in floats (35 px x 3 ch, padded)
constexpr int kD2Pr = 2 * kD2Ty + 3;                    // B patch rows
constexpr int kD2As = (4 * kD2Tx + 9) * 3 + 2;          // A patch row stride (73 px x 3 ch, padded)
constexpr int kD2Ar = 4 * kD2Ty + 9;                    // A patch rows
constexpr int kStageBatch = 24;                         // loads a thread keeps in flight while staging

// rows [y0, y0 + rows) x elements [xe0, xe0 + width) of a level with `stride` floats per row -> LDS, every thread's loads issued
// before its first store (ONE trip to memory for the whole patch instead of one per output and tap)
__device__ __forceinline__ void stage_rect(const float* __restrict__ src, size_t stride, int y0, int xe0, int width, int rows,
                                           float* __restrict__ dst, int dstride) {
    const int total = width * rows;
    const float inv_w = 1.f / (float)width;
    for (int base = 0; base < total; base += 256 * kStageBatch) {
        float v[kStageBatch];
        int at[kStageBatch];
#pragma unroll
        for (int i = 0; i < kStageBatch; ++i) {
            const int e = base + i * 256 + (int)threadIdx.x;
            const int r = div_small_i(e < total ? e : 0, width, inv_w), c = (e < total ? e : 0) - r * width;
            at[i] = e < total ? r * dstride + c : -1;
            v[i] = e < total ? src[(size_t)(y0 + r) * stride + xe0 + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kStageBatch; ++i) if (at[i] >= 0) dst[at[i]] = v[i];
    }
}

// A (gA.sw x gA.sh) -> B (gA.dw x gA.dh) -> C (gB.dw x gB.dh); gB is the geometry with B as its source.
template <int CN>
__device__ __forceinline__ void pyrdown2_body(const float* __restrict__ A, float* __restrict__ B, float* __restrict__ C,
                                              const DownGeom& gA, const DownGeom& gB, int tx, int ty, float* __restrict__ patch,
                                              float* __restrict__ apatch, int pitchA) {
    const int tid = threadIdx.x;
    const int wa = gA.sw, ha = gA.sh, wb = gA.dw, hb = gA.dh, wc = gB.dw, hc = gB.dh;
    const int cx0 = tx * kD2Tx, cy0 = ty * kD2Ty;
    const int cx1 = min(cx0 + kD2Tx - 1, wc - 1), cy1 = min(cy0 + kD2Ty - 1, hc - 1);
    // the B pixels the tile's 5x5 windows touch, after reflection, form one clipped rectangle
    const int bx0 = max(2 * cx0 - 2, 0), bx1 = min(2 * cx1 + 2, wb - 1);
    const int by0 = max(2 * cy0 - 2, 0), by1 = min(2 * cy1 + 2, hb - 1);
    // ... and so do the A pixels under those
    const int ax0 = max(2 * bx0 - 2, 0), ax1 = min(2 * bx1 + 2, wa - 1);
    const int ay0 = max(2 * by0 - 2, 0), ay1 = min(2 * by1 + 2, ha - 1);
    stage_rect(A, (size_t)pitchA * CN, ay0, ax0 * CN, (ax1 - ax0 + 1) * CN, ay1 - ay0 + 1, apatch, kD2As);      // (pitchA: pixels per row of A's buffer; B and C are tight)
    __syncthreads();
    const int pw = (bx1 - bx0 + 1) * CN, ph = by1 - by0 + 1;
    const float inv_pw = 1.f / (float)pw;
    for (int e = tid; e < pw * ph; e += 256) {
        const int py = div_small_i(e, pw, inv_pw), pxe = e - py * pw;
        const int by = by0 + py, bxe = bx0 * CN + pxe;
        const float v = pyrdown_elem_patch<CN>(apatch, kD2As, ax0 * CN, ay0, gA, by, bxe);
        patch[py * kD2Ps + pxe] = v;
        const int bx = bxe / CN;
        if (bx >= 2 * cx0 && bx <= 2 * cx1 + 1 && by >= 2 * cy0 && by <= 2 * cy1 + 1)      // the part of B this tile owns
            B[(size_t)by * wb * CN + bxe] = v;
    }
    __syncthreads();
    const int cw = (cx1 - cx0 + 1) * CN, chh = cy1 - cy0 + 1;
    const float inv_cw = 1.f / (float)cw;
    for (int e = tid; e < cw * chh; e += 256) {
        const int ly = div_small_i(e, cw, inv_cw), lxe = e - ly * cw;
        const int cy = cy0 + ly, cxe = cx0 * CN + lxe;
        C[(size_t)cy * wc * CN + cxe] = pyrdown_elem_patch<CN>(patch, kD2Ps, bx0 * CN, by0, gB, cy, cxe);
    }
}

__global__ void __launch_bounds__(256) k_pyrdown2(const float* __restrict__ aL, const float* __restrict__ aR, const float* __restrict__ aM,
                                                  float* __restrict__ bL, float* __restrict__ bR, float* __restrict__ bM,
                                                  float* __restrict__ cL, float* __restrict__ cR, float* __restrict__ cM,
                                                  DownGeom gA3, DownGeom gA1, DownGeom gB3, DownGeom gB1, int tiles_x, int pitchA) {
    __shared__ float patch[kD2Pr * kD2Ps];
    __shared__ float apatch[kD2Ar * kD2As];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int which = blockIdx.y;
    if (which == 0)      pyrdown2_body<3>(aL, bL, cL, gA3, gB3, tx, ty, patch, apatch, pitchA);
    else if (which == 1) pyrdown2_body<3>(aR, bR, cR, gA3, gB3, tx, ty, patch, apatch, pitchA);
    else                 pyrdown2_body<1>(aM, bM, cM, gA1, gB1, tx, ty, patch, apatch, pitchA);
}

// ---- pyrUp out of an LDS patch of the low-resolution level ------------------------------------------------------------
// patch(row r, pixel p, channel c) = low(py0 + r, px0 + p, c); the low level is sw x sh (>= 2 x 2).
__device__ __forceinline__ float pyrup_elem_patch(const float* __restrict__ patch, int pstride, int px0, int py0,
                                                  int sw, int sh, int dy, int dxe) {
    const int dpx = dxe / 3, c = dxe - dpx * 3;
    const int spx = dpx >> 1;
    const bool odd = dpx & 1;
    const int sy = dy >> 1;
    const int sym = sy >= 1 ? sy - 1 : 1, syp = sy + 1 <= sh - 1 ? sy + 1 : sh - 1;
    const int cm = (max(spx - 1, 0) - px0) * 3 + c, c0 = (spx - px0) * 3 + c, cp = (min(spx + 1, sw - 1) - px0) * 3 + c;
    const bool left = spx == 0, right = spx >= sw - 1;
    float r[3];
    const int rows[3] = {sym, sy, syp};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* row = patch + (rows[k] - py0) * pstride;
        const float sm = row[cm], s0 = row[c0], sp = row[cp];
        const float ev = left ? s0 * 6.f + sp * 2.f : right ? sm + s0 * 7.f : sm + s0 * 6.f + sp;
        const float od = right ? s0 * 8.f : (s0 + sp) * 4.f;
        r[k] = odd ? od : ev;
    }
    const float s = 1.f / 64;
    return (dy & 1) ? ((r[1] + r[2]) * 4.f) * s : (r[0] + r[1] * 6.f + r[2]) * s;
}

constexpr int kC2Tx = 16, kC2Ty = 4;                    // tile of the output level, pixels
constexpr int kC2Pw = kC2Tx / 2 + 2, kC2Pr = kC2Ty / 2 + 2;      // level k+1 under the tile plus the ring pyrUp reads: 18 x 6 pixels
constexpr int kC2Ps = kC2Pw * 3 + 2;                    // its row stride in floats (padded)
constexpr int kC2Nw = kC2Pw / 2 + 3, kC2Nr = kC2Pr / 2 + 3;      // level k+2 under that, plus its ring: 12 x 6 pixels (the region starts at an odd pixel)
constexpr int kC2Ns = kC2Nw * 3 + 1;

// level k (w x h): g*;  level k+1 (w1 x h1): m*;  level k+2 (w2 x h2): n* (nB = blended level k+2).  Writes blended level k.
// Everything the tile needs of levels k+1 and k+2 is staged in LDS and the tile's own level-k values are loaded into registers in
// the same trip to memory; the blended level k+1 under the tile is then built from LDS, and the tile from that.
__global__ void __launch_bounds__(256) k_collapse2(const float* __restrict__ gL, const float* __restrict__ gR, const float* __restrict__ gM,
                                                   const float* __restrict__ mL, const float* __restrict__ mR, const float* __restrict__ mM,
                                                   const float* __restrict__ nL, const float* __restrict__ nR, const float* __restrict__ nB,
                                                   float* __restrict__ outB, int w, int h, int w1, int h1, int w2, int h2, int tiles_x) {
    __shared__ float patch[kC2Pr * kC2Ps];               // blended level k+1
    __shared__ float sML[kC2Pr * kC2Ps], sMR[kC2Pr * kC2Ps], sMM[kC2Pr * kC2Pw];
    __shared__ float sNL[kC2Nr * kC2Ns], sNR[kC2Nr * kC2Ns], sNB[kC2Nr * kC2Ns];
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x0 = tx * kC2Tx, y0 = ty * kC2Ty;
    const int x1 = min(x0 + kC2Tx - 1, w - 1), y1 = min(y0 + kC2Ty - 1, h - 1);
    // blended level k+1 under the tile, plus the one-pixel ring pyrUp reads
    const int px0 = max((x0 >> 1) - 1, 0), px1 = min((x1 >> 1) + 1, w1 - 1);
    const int py0 = max((y0 >> 1) - 1, 0), py1 = min((y1 >> 1) + 1, h1 - 1);
    const int pw = (px1 - px0 + 1) * 3, ph = py1 - py0 + 1;
    // level k+2 under that
    const int nx0 = max((px0 >> 1) - 1, 0), nx1 = min((px1 >> 1) + 1, w2 - 1);
    const int ny0 = max((py0 >> 1) - 1, 0), ny1 = min((py1 >> 1) + 1, h2 - 1);
    const int nw = (nx1 - nx0 + 1) * 3, nh = ny1 - ny0 + 1;
    // one trip to memory: the level k+1 / k+2 regions (a thread has at most two elements of each) and the tile's level-k values
    {
        const float inv_pw = 1.f / (float)pw, inv_nw = 1.f / (float)nw, inv_pp = 1.f / (float)(pw / 3);
        float vl[2], vr[2], vm, vnl, vnr, vnb;
        int al[2], am = -1, an = -1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            const bool on = e < pw * ph;
            const int r = div_small_i(on ? e : 0, pw, inv_pw), c = (on ? e : 0) - r * pw;
            const size_t g = (size_t)(py0 + r) * w1 * 3 + px0 * 3 + c;
            al[i] = on ? r * kC2Ps + c : -1;
            vl[i] = on ? mL[g] : 0.f; vr[i] = on ? mR[g] : 0.f;
        }
        {
            const int ppw = pw / 3;
            const bool on = tid < ppw * ph;
            const int r = div_small_i(on ? tid : 0, ppw, inv_pp), c = (on ? tid : 0) - r * ppw;
            am = on ? r * kC2Pw + c : -1;
            vm = on ? mM[(size_t)(py0 + r) * w1 + px0 + c] : 0.f;
        }
        {
            const bool on = tid < nw * nh;
            const int r = div_small_i(on ? tid : 0, nw, inv_nw), c = (on ? tid : 0) - r * nw;
            const size_t g = (size_t)(ny0 + r) * w2 * 3 + nx0 * 3 + c;
            an = on ? r * kC2Ns + c : -1;
            vnl = on ? nL[g] : 0.f; vnr = on ? nR[g] : 0.f; vnb = on ? nB[g] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) if (al[i] >= 0) { sML[al[i]] = vl[i]; sMR[al[i]] = vr[i]; }
        if (am >= 0) sMM[am] = vm;
        if (an >= 0) { sNL[an] = vnl; sNR[an] = vnr; sNB[an] = vnb; }
    }
    const int cw = (x1 - x0 + 1) * 3, chh = y1 - y0 + 1;
    const float inv_cw = 1.f / (float)cw;
    float gl[3], gr[3], gm[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {                         // 32 x 8 x 3 outputs: three per thread
        const int e = tid + i * 256;
        const bool on = e < cw * chh;
        const int ly = div_small_i(on ? e : 0, cw, inv_cw), lxe = (on ? e : 0) - ly * cw;
        const int y = y0 + ly, xe = x0 * 3 + lxe;
        const size_t g = (size_t)y * w * 3 + xe;
        gl[i] = on ? gL[g] : 0.f; gr[i] = on ? gR[g] : 0.f; gm[i] = on ? gM[(size_t)y * w + xe / 3] : 0.f;
    }
    __syncthreads();
    {
        const float inv_pw = 1.f / (float)pw;
        for (int e = tid; e < pw * ph; e += 256) {
            const int py = div_small_i(e, pw, inv_pw), pxe = e - py * pw;
            const int Y = py0 + py, XE = px0 * 3 + pxe;
            const float m = sMM[py * kC2Pw + pxe / 3];
            const float ml = sML[py * kC2Ps + pxe], mr = sMR[py * kC2Ps + pxe];
            const float uL = pyrup_elem_patch(sNL, kC2Ns, nx0, ny0, w2, h2, Y, XE), uR = pyrup_elem_patch(sNR, kC2Ns, nx0, ny0, w2, h2, Y, XE);
            const float uB = pyrup_elem_patch(sNB, kC2Ns, nx0, ny0, w2, h2, Y, XE);
            patch[py * kC2Ps + pxe] = uB + mix_lr(ml - uL, mr - uR, m);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = tid + i * 256;
        if (e < cw * chh) {
            const int ly = div_small_i(e, cw, inv_cw), lxe = e - ly * cw;
            const int y = y0 + ly, xe = x0 * 3 + lxe;
            const float uL = pyrup_elem_patch(sML, kC2Ps, px0, py0, w1, h1, y, xe), uR = pyrup_elem_patch(sMR, kC2Ps, px0, py0, w1, h1, y, xe);
            const float uB = pyrup_elem_patch(patch, kC2Ps, px0, py0, w1, h1, y, xe);
            outB[(size_t)y * w * 3 + xe] = uB + mix_lr(gl[i] - uL, gr[i] - uR, gm[i]);
        }
    }
}

}  // namespace

bool pyrdown2_eligible(int sw, int sh) {
    // A -> B -> C with B at least 3 x 3 (single reflection in the patch) and small enough to be launch-latency bound
    const int wb = (sw + 1) / 2, hb = (sh + 1) / 2;
    return sw >= 3 && sh >= 3 && wb >= 3 && hb >= 3 && (size_t)sw * sh <= kFuseMaxPixels;
}

void launch_pyrdown2(const float* aL, const float* aR, const float* aM, float* bL, float* bR, float* bM,
                     float* cL, float* cR, float* cM, int sw, int sh, hipStream_t s, int src_pitch) {
    const DownGeom gA3 = make_down_geom(sw, sh, 3), gA1 = make_down_geom(sw, sh, 1);
    const DownGeom gB3 = make_down_geom(gA3.dw, gA3.dh, 3), gB1 = make_down_geom(gA3.dw, gA3.dh, 1);
    const int tiles_x = (gB3.dw + kD2Tx - 1) / kD2Tx, tiles_y = (gB3.dh + kD2Ty - 1) / kD2Ty;
    hipLaunchKernelGGL(k_pyrdown2, dim3(tiles_x * tiles_y, 3), dim3(256), 0, s, aL, aR, aM, bL, bR, bM, cL, cR, cM, gA3, gA1, gB3, gB1, tiles_x, src_pitch > 0 ? src_pitch : sw);
}

bool collapse2_eligible(int w, int h, int w1, int h1, int w2, int h2) {
    return w1 >= 2 && h1 >= 2 && w2 >= 2 && h2 >= 2 && (size_t)w * h <= kFuseMaxPixels;
}

void launch_collapse2(const float* gL, const float* gR, const float* gM, const float* mL, const float* mR, const float* mM,
                      const float* nL, const float* nR, const float* nB, float* outB,
                      int w, int h, int w1, int h1, int w2, int h2, hipStream_t s) {
    const int tiles_x = (w + kC2Tx - 1) / kC2Tx, tiles_y = (h + kC2Ty - 1) / kC2Ty;
    hipLaunchKernelGGL(k_collapse2, dim3(tiles_x * tiles_y), dim3(256), 0, s, gL, gR, gM, mL, mR, mM, nL, nR, nB, outB,
                       w, h, w1, h1, w2, h2, tiles_x);
}

}  // namespace poppy_hip
