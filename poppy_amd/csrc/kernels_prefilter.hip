// kernels_prefilter.hip — pre-ORB filter chain, part 1: Extractor::foreground (src/extractor.cpp:136-229) on gfx950.
//
// Integer / per-pixel float work, HBM- or LDS-bound; no MFMA.  Arithmetic follows the reference's SSE3-baseline build
// (this file is compiled with -ffp-contract=off; OCV = third/opencv-4.6.0/modules):
//   k_bgr2gray        cvtColor(BGR2GRAY), 15-bit fixed point          OCV/imgproc/src/color_rgb.simd.hpp:646-730
//   k_mog2_all        BackgroundSubtractorMOG2(500,16,true)::apply x 13     OCV/video/src/bgfg_gaussmix2.cpp:479-523,539-755,847-884
//                     + acc += flow * (1/6) (u8 convertTo, saturating add; matrix_expressions.cpp:270-275,1330-1336)
//   k_median_u8       medianBlur, ksize 9..89, replicated border       OCV/imgproc/src/median_blur.simd.hpp (exact median)
//   k_gauss23_h/_v    GaussianBlur 23x23 sigma 1, 8.8 fixed point      OCV/imgproc/src/smooth.simd.hpp:1136-1199,1780-1866
//   k_fg_mask         u8 -> f32, *19 + 1, cv::log, / log 20, * grey, -> u8, histogram
//                                                                      OCV/core/src/mathfuncs_core.simd.hpp:683-752
//   k_equalize_lut / k_apply_lut   equalizeHist                        OCV/imgproc/src/histogram.cpp:3436-3493
#include "kernels_prefilter.h"
#include <cstdlib>
#include "pyramid_device.h"

namespace poppy_hip {

__global__ void __launch_bounds__(256) k_bgr2gray(const uint8_t* __restrict__ bgr, size_t stride, uint8_t* __restrict__ gray, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* p = bgr + (size_t)y * stride + (size_t)x * 3;
    gray[(size_t)y * W + x] = (uint8_t)((p[0] * 3735 + p[1] * 19235 + p[2] * 9798 + (1 << 14)) >> 15);
}
void launch_bgr2gray(const uint8_t* bgr, size_t stride, uint8_t* gray, int w, int h, hipStream_t s) {
    hipLaunchKernelGGL(k_bgr2gray, dim3((w + 255) / 256, h), dim3(256), 0, s, bgr, stride, gray, w, h);
}

// ---- MOG2 ---------------------------------------------------------------------------------------------------------
// One thread per pixel; the model is structure-of-arrays [mode][pixel] so that every access is coalesced.  The five
// modes live in registers for the duration of the update (sorting swaps become register moves).
constexpr int kMog2Modes = 5;

// One BackgroundSubtractorMOG2::apply step on one pixel; returns the mask value (0, 127, 255).
// Kept out of line on purpose: inlined into k_mog2_all's step loop, hipcc 7.2 drops the `mu[k] = data; v[k] = varInit` stores of a
// newly opened mode (found by comparing the fused kernel with a step-by-step one on the same inputs; an optimisation barrier
// or a memory round trip of the model between steps did not help).  The call costs nothing that matters here.
// The mixture travels BY VALUE (in and out in registers: 17 dwords); passed by reference the arrays of an out-of-line call live in scratch
// memory, 30 scratch accesses per step and pixel.
typedef float mog2_vec __attribute__((ext_vector_type(16)));      // w[5], v[5], mu[5], and nmodes | out << 8 as the bits of the last lane
__device__ __forceinline__ int mog2_step_body(float (&w)[kMog2Modes], float (&v)[kMog2Modes], float (&mu)[kMog2Modes], int& nmodes,
                                              float data, float alphaT, float prune);
__device__ __noinline__ mog2_vec mog2_step(mog2_vec m, float data, float alphaT, float prune) {
    float w[kMog2Modes], v[kMog2Modes], mu[kMog2Modes];
#pragma unroll
    for (int k = 0; k < kMog2Modes; ++k) { w[k] = m[k]; v[k] = m[kMog2Modes + k]; mu[k] = m[2 * kMog2Modes + k]; }
    int nmodes = __float_as_int(m[15]) & 255;
    const int out = mog2_step_body(w, v, mu, nmodes, data, alphaT, prune);
#pragma unroll
    for (int k = 0; k < kMog2Modes; ++k) { m[k] = w[k]; m[kMog2Modes + k] = v[k]; m[2 * kMog2Modes + k] = mu[k]; }
    m[15] = __int_as_float(nmodes | (out << 8));
    return m;
}
__device__ __forceinline__ int mog2_step_body(float (&w)[kMog2Modes], float (&v)[kMog2Modes], float (&mu)[kMog2Modes], int& nmodes,
                                              float data, float alphaT, float prune) {
    const float kBgSigma2 = 16.f, kBgShare = 0.9f, kMatchSigma2 = 9.f, kVar0 = 15.f, kVarLo = 4.f, kVarHi = 75.f, kShadowLo = 0.5f;
    const float keep = 1.f - alphaT;
    bool background = false, matched = false;
    float wsum = 0.f;
#pragma unroll
    for (int mode = 0; mode < kMog2Modes; ++mode) {
        if (mode < nmodes) {                               // nmodes shrinks inside the loop when a mode is pruned
            float wgt = keep * w[mode] + prune;
            int dst = mode;
            if (!matched) {
                const float var = v[mode];
                const float delta = mu[mode] - data;
                float d2 = 0.f;
                d2 += delta * delta;
                if (wsum < kBgShare && d2 < kBgSigma2 * var) background = true;
                if (d2 < kMatchSigma2 * var) {
                    matched = true;
                    wgt += alphaT;
                    const float k = __fdiv_rn(alphaT, wgt);
                    mu[mode] -= k * delta;
                    float var_upd = var + k * (d2 - var);
                    var_upd = fmaxf(var_upd, kVarLo);
                    var_upd = fminf(var_upd, kVarHi);
                    v[mode] = var_upd;
#pragma unroll
                    for (int i = kMog2Modes - 1; i > 0; --i) {   // bubble the matched mode up while its new weight is not smaller
                        if (i <= dst && i == dst && !(wgt < w[i - 1])) {
                            float t;
                            t = w[i]; w[i] = w[i - 1]; w[i - 1] = t;
                            t = v[i]; v[i] = v[i - 1]; v[i - 1] = t;
                            t = mu[i]; mu[i] = mu[i - 1]; mu[i - 1] = t;
                            --dst;
                        }
                    }
                }
            }
            if (wgt < -prune) { wgt = 0.f; --nmodes; }
#pragma unroll
            for (int k = 0; k < kMog2Modes; ++k) if (k == dst) w[k] = wgt;          // w[dst] = wgt without indexing the registers
            wsum += wgt;
        }
    }
    float wnorm = 0.f;
    if (fabsf(wsum) > 1.1920928955078125e-7f) wnorm = __fdiv_rn(1.f, wsum);
#pragma unroll
    for (int mode = 0; mode < kMog2Modes; ++mode)
        if (mode < nmodes) w[mode] *= wnorm;
    if (!matched && alphaT > 0.f) {
        const int mode = nmodes == kMog2Modes ? kMog2Modes - 1 : nmodes++;
        if (nmodes == 1) {
#pragma unroll
            for (int k = 0; k < kMog2Modes; ++k) if (k == mode) w[k] = 1.f;
        } else {
#pragma unroll
            for (int k = 0; k < kMog2Modes; ++k) {
                if (k == mode) w[k] = alphaT;
                else if (k < nmodes - 1) w[k] *= keep;
            }
        }
#pragma unroll
        for (int k = 0; k < kMog2Modes; ++k) if (k == mode) { mu[k] = data; v[k] = kVar0; }
        int pos = nmodes - 1;
#pragma unroll
        for (int i = kMog2Modes - 1; i > 0; --i) {
            if (i == pos && !(alphaT < w[i - 1])) {
                float t;
                t = w[i]; w[i] = w[i - 1]; w[i - 1] = t;
                t = v[i]; v[i] = v[i - 1]; v[i - 1] = t;
                t = mu[i]; mu[i] = mu[i - 1]; mu[i - 1] = t;
                --pos;
            }
        }
    }
    int out = 0;
    if (!background) {
        out = 255;
        float wacc = 0.f;                               // detectShadowGMM
        bool done = false;
#pragma unroll
        for (int mode = 0; mode < kMog2Modes; ++mode) {
            if (mode < nmodes && !done) {
                float dot_xm = 0.f, dot_mm = 0.f;
                dot_xm += data * mu[mode];
                dot_mm += mu[mode] * mu[mode];
                if (dot_mm == 0.f) done = true;
                else {
                    if (dot_xm <= dot_mm && dot_xm >= kShadowLo * dot_mm) {
                        const float a = __fdiv_rn(dot_xm, dot_mm);
                        float resid2 = 0.f;
                        const float resid = a * mu[mode] - data;
                        resid2 += resid * resid;
                        if (resid2 < kBgSigma2 * v[mode] * a * a) { out = 127; done = true; }
                    }
                    if (!done) {
                        wacc += w[mode];
                        if (wacc > kBgShare) done = true;
                    }
                }
            }
        }
    }
    return out;
}

// All applies of Extractor::foreground in ONE launch.  A pixel's mixture depends on that pixel only, and the chain of input
// images (grey, then the progressive medians) does not depend on the masks, so once the inputs exist the mixture never has to
// leave the registers: 13 bytes in, 13 bytes out per pixel instead of 13 x 122 (the per-step kernel kept 5 modes x 3 floats in HBM).
struct Mog2Inputs { const uint8_t* img[kMog2Steps]; float alphaT[kMog2Steps], prune[kMog2Steps]; };

__global__ void __launch_bounds__(256) k_mog2_all(Mog2Inputs in, uint8_t* __restrict__ flows, int n, int steps) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    mog2_vec m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // no modes yet
    for (int i = 0; i < steps; ++i) {
        const float data = (float)in.img[i][p];
        m = mog2_step(m, data, in.alphaT[i], in.prune[i]);
        flows[(size_t)i * n + p] = (uint8_t)(__float_as_int(m[15]) >> 8);
    }
}

// fgMask += flow * scale: convertTo(u8, alpha) rounds to nearest even and saturates, then a saturating add
// (matrix_expressions.cpp:270-275,1330-1336)
// (first: the accumulator is all zeros — Mat::zeros, src/extractor.cpp — and is written, not read: saves the launch that would clear it)
__global__ void __launch_bounds__(256) k_acc_flow(uint8_t* __restrict__ acc, const uint8_t* __restrict__ flow, int n, float acc_scale, int first) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int t = cv_round_x86((float)flow[p] * acc_scale + 0.f);
    t = t < 0 ? 0 : t > 255 ? 255 : t;
    const int sum = (first ? 0 : (int)acc[p]) + t;
    acc[p] = (uint8_t)(sum > 255 ? 255 : sum);
}

void launch_mog2_all(const uint8_t* const* imgs, const float* alphaT, const float* prune, int steps, uint8_t* flows, int n_px, hipStream_t s) {
    Mog2Inputs in;
    for (int i = 0; i < kMog2Steps; ++i) { in.img[i] = imgs[i < steps ? i : 0]; in.alphaT[i] = alphaT[i < steps ? i : 0]; in.prune[i] = prune[i < steps ? i : 0]; }
    hipLaunchKernelGGL(k_mog2_all, dim3((n_px + 255) / 256), dim3(256), 0, s, in, flows, n_px, steps);
}
void launch_acc_flow(uint8_t* acc, const uint8_t* flow, int n_px, float acc_scale, hipStream_t s, bool first) {
    hipLaunchKernelGGL(k_acc_flow, dim3((n_px + 255) / 256), dim3(256), 0, s, acc, flow, n_px, acc_scale, first ? 1 : 0);
}

// ---- large-kernel median ---------------------------------------------------------------------------------------------
// One lane = one image column, sliding DOWN a segment of rows.  Each lane keeps its own 256-bin histogram of the current
// ksize x ksize window in LDS, plus a 16-bin coarse histogram for the search.  Counts are at most 89^2 < 2^16, so a counter
// is 16 bits (34 KB per wave instead of 68: four waves per CU) and a bin is 32 dwords: lanes l and l + 32 share dword l, in
// the low and the high half.  The two lanes sit in different halves of the wave's LDS access (32 lanes each), so the 32
// lanes of one half always hit 32 different banks whatever their bins are; a count never goes negative, so adding or
// subtracting 1 << 16 never borrows across the halves.  With this layout the increment is a per-lane constant and the
// address is `value * 128 + lane constant`: ~4 VALU instructions per value next to its two `ds_add`/`ds_sub` (the kernel
// runs one wave per SIMD and is bound by instruction issue, so that count IS the run time).
// A step down = add one row of ksize values, remove one: the two rows are read as unaligned dwords (adjacent lanes read
// overlapping, consecutive addresses).  The median is the first value whose cumulative count exceeds ksize^2 / 2:
// 16 coarse + 16 fine reads.
constexpr int kMedLanes = 64;
constexpr int kMedBinBytes = kMedLanes * 2;                // one bin = 64 u16 counters in 32 dwords: lanes l and l + 32 share dword l
constexpr int kMedCoarseOff = 256 * kMedBinBytes;          // 256 fine bins, then 16 coarse ones
constexpr int kMedWords = (256 + 16) * kMedBinBytes / 4 / kMedLanes;   // dwords per lane (136)

__device__ __forceinline__ void med_update(char* __restrict__ lane_base, uint32_t unit, uint32_t v, bool add) {
    uint32_t* f = reinterpret_cast<uint32_t*>(lane_base + v * kMedBinBytes);
    uint32_t* c = reinterpret_cast<uint32_t*>(lane_base + kMedCoarseOff + (v >> 4) * kMedBinBytes);
    if (add) { atomicAdd(f, unit); atomicAdd(c, unit); }
    else     { atomicSub(f, unit); atomicSub(c, unit); }
}

// The source is first copied into a buffer with kMedPad replicated columns on either side (k_pad_cols), so that every
// lane reads its window row as plain unaligned dwords without clamping; rows are clamped through the row pointer.
constexpr int kMedPad = 48;                                 // >= 44 (ksize 89) + 3 spare bytes for the last dword
constexpr int kMedMaxDw = 23;                               // dwords covering 89 bytes: the largest instantiation
constexpr int kMedWavesLong = 4;                            // waves per histogram set for rows of 4 dwords and more (1 / 2 / 4 measured per ksize: profiles/r02_notes.md 6.9)

__global__ void __launch_bounds__(256) k_pad_cols(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H) {
    const int xp = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int Wp = W + 2 * kMedPad;
    if (xp >= Wp) return;
    dst[(size_t)y * Wp + xp] = src[(size_t)y * W + min(max(xp - kMedPad, 0), W - 1)];
}

// NDW = dwords per window row, a template parameter (ksize = 8 i + 1 -> NDW = 2 i + 1, one byte used of the last dword): with
// a run-time count the unrolled update loops are chopped into basic blocks by uniform branches.
// NW = waves that SHARE one set of 64 histograms (a workgroup of NW waves): wave w applies the dwords i = w (mod NW) of every
// window row, all waves search (wave 0 writes).  The kernel is bound by LDS capacity to 4 histogram sets per CU; with one wave
// per set a SIMD holds a single wave, which pays the full latency of every instruction it issues (its time follows its
// instruction count: without the coarse-histogram atomic 30-36 % less, with three more vector instructions per value 25 % more,
// profiles/r02_notes.md 6.7).  Two waves per set put two waves on every SIMD, each with half the updates; the price is two
// workgroup barriers per row step (updates | search | updates) and the search done twice.  Counts cannot go negative whatever the
// order of the two waves' atomics: a value removed in a step is in the window at its start.
template <int NDW, int NW, int WV>
__device__ __forceinline__ void median_body(uint32_t* __restrict__ hist, const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst,
                                            uint8_t* __restrict__ padded_out, int W, int H, int ksize, int rows_per_block) {
    const int lane = threadIdx.x & (kMedLanes - 1);
    const int x = blockIdx.x * kMedLanes + lane;
    const int y_begin = blockIdx.y * rows_per_block, y_end = min(y_begin + rows_per_block, H);
    const int r = ksize >> 1, half = (ksize * ksize) >> 1;
    const int Wp = W + 2 * kMedPad;
    for (int b = WV; b < kMedWords; b += NW) hist[b * kMedLanes + lane] = 0;
    if (NW > 1) __syncthreads();
    char* const hist_b = reinterpret_cast<char*>(hist);
    char* const lane_base = hist_b + (lane & 31) * 4;         // the dword lanes l and l + 32 share, in bin 0
    const uint32_t unit = 1u << ((lane >> 5) << 4);           // +1 in this lane's half of it
    auto count_of = [&](int bin) {
        return (int)*reinterpret_cast<const uint16_t*>(hist_b + bin * kMedBinBytes + (lane & 31) * 4 + (lane >> 5) * 2);
    };
    const bool live = x < W;                                  // lanes past the right edge idle along
    const int xc = live ? x : W - 1;
    constexpr int ndw = NDW;                                  // dwords per window row (the last one is partly used)
    const int tail = ksize - 4 * (ndw - 1);                   // bytes used of the last dword (1 for the ksizes of the chain)
    auto row_ptr = [&](int yy) { return srcp + (size_t)min(max(yy, 0), H - 1) * Wp + (kMedPad + xc - r); };
    auto load_row = [&](const uint8_t* p, uint32_t* regs) {   // this wave's dwords of the row
#pragma unroll
        for (int i = WV; i < NDW; i += NW)
            __builtin_memcpy(&regs[i], p + 4 * i, 4);
    };
    constexpr bool kSumCoarse = NDW >= 7;                     // ksize 25 and up (625+ coarse atomics per lane against 256 reads shared by the waves; at 17 the reads lose: 65 -> 75 us)
    auto apply_row = [&](const uint32_t* regs, bool add) {
#pragma unroll
        for (int i = WV; i < NDW; i += NW) {
            {
                const int nb = (i == ndw - 1) ? tail : 4;
                // On the plateaus the chain's later stages consist of, the four bytes of a dword are one value in every lane: one update of 4 instead of four
                // (wave-uniform test; the warm-up is ksize * ksize additions per lane and none of them cancels)
                const uint32_t w4 = regs[i];
                if (nb == 4 && add && __all(w4 == (w4 & 255u) * 0x01010101u)) {
                    const uint32_t v = w4 & 255u;
                    atomicAdd(reinterpret_cast<uint32_t*>(lane_base + v * kMedBinBytes), 4u * unit);
                    if (!kSumCoarse) atomicAdd(reinterpret_cast<uint32_t*>(lane_base + kMedCoarseOff + (v >> 4) * kMedBinBytes), 4u * unit);
                    continue;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nb) {
                        const uint32_t v = (regs[i] >> (8 * k)) & 255u;
                        if (kSumCoarse) {                       // fine bins only: the warm-up's coarse counts are summed up once, below
                            uint32_t* f = reinterpret_cast<uint32_t*>(lane_base + v * kMedBinBytes);
                            if (add) atomicAdd(f, unit); else atomicSub(f, unit);
                        } else med_update(lane_base, unit, v, add);
                    }
            }
        }
    };
    // A step down adds row y + r + 1 and removes row y - r.  Where the value entering a column of the window equals the one leaving it the
    // two updates cancel: nothing is sent to LDS (the kernel is bound by its LDS atomics, four per such pair).  The chain feeds every median
    // with the previous median's output — plateaus — so most pairs cancel from the second stage on; a wave whose 64 lanes all cancel
    // skips the instructions altogether.
    auto apply_step = [&](const uint32_t* add_regs, const uint32_t* sub_regs) {
#pragma unroll
        for (int i = WV; i < NDW; i += NW) {
            const int nb = (i == ndw - 1) ? tail : 4;
            if (add_regs[i] != sub_regs[i] || nb < 4) {                  // (a dword of four equal pairs is the common case)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nb) {
                        const uint32_t va = (add_regs[i] >> (8 * k)) & 255u, vs = (sub_regs[i] >> (8 * k)) & 255u;
                        if (va != vs) { med_update(lane_base, unit, va, true); med_update(lane_base, unit, vs, false); }
                    }
            }
        }
    };
    uint32_t ra[NDW], rs[NDW];
    // warm-up: rows y_begin - r .. y_begin + r, loaded one ahead of the row being applied
    load_row(row_ptr(y_begin - r), ra);
    for (int yy = y_begin - r; yy <= y_begin + r; ++yy) {
        uint32_t cur[NDW];
#pragma unroll
        for (int i = WV; i < NDW; i += NW) cur[i] = ra[i];
        if (yy < y_begin + r) load_row(row_ptr(yy + 1), ra);
        apply_row(cur, true);
    }
    uint32_t ra2[NDW], rs2[NDW];                              // the rows of the step after the next one: two steps of prefetch
    if (y_begin + 1 < y_end) { load_row(row_ptr(y_begin + r + 1), ra); load_row(row_ptr(y_begin - r), rs); }
    if (y_begin + 2 < y_end) { load_row(row_ptr(y_begin + r + 2), ra2); load_row(row_ptr(y_begin + 1 - r), rs2); }
    // The warm-up is ksize * ksize additions per lane — 40 % of a segment's atomics at ksize 89 — and none of them cancels; its coarse level is
    // therefore not kept by atomics but summed from the fine bins once, every wave a share of the 16 groups (256 reads per lane instead of
    // ksize * ksize atomics).
    if (kSumCoarse && NW > 1) __syncthreads();                // every wave's fine additions are in
#pragma unroll 1
    for (int g = WV; kSumCoarse && g < 16; g += NW) {
        int c[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = count_of(g * 16 + j);
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += c[j];
        *reinterpret_cast<uint16_t*>(hist_b + kMedCoarseOff + g * kMedBinBytes + (lane & 31) * 4 + (lane >> 5) * 2) = (uint16_t)sum;
    }
    for (int y = y_begin; y < y_end; ++y) {
        if (NW > 1) __syncthreads();                          // every wave's updates of this window are in
        if (WV == 0) {                                        // one wave searches; the others wait at the barrier below
            int s = 0, cb = 0, below = 0;
#pragma unroll
            for (int c = 0; c < 16; ++c) {                    // first coarse bin whose cumulative count exceeds `half`
                s += count_of(256 + c);
                const bool hit = s > half;
                cb += hit ? 0 : 1;
                below = hit ? below : s;
            }
            int fb = 0;
            s = below;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                s += count_of(cb * 16 + k);
                fb += (s > half) ? 0 : 1;
            }
            if (live) {
                const uint8_t m = (uint8_t)(cb * 16 + fb);
                dst[(size_t)y * W + x] = m;
                if (padded_out) {                             // the next median of the chain reads its source with replicated side columns: written here
                    uint8_t* prow = padded_out + (size_t)y * Wp;
                    prow[kMedPad + x] = m;
                    if (x == 0) for (int j = 0; j < kMedPad; ++j) prow[j] = m;
                    if (x == W - 1) for (int j = 0; j < kMedPad; ++j) prow[kMedPad + W + j] = m;
                }
            }
        }
        if (NW > 1) __syncthreads();                          // every wave has read the window's counts
        if (y + 1 < y_end) {
            uint32_t ca[NDW], cs[NDW];
#pragma unroll
            for (int i = WV; i < NDW; i += NW) { ca[i] = ra[i]; cs[i] = rs[i]; ra[i] = ra2[i]; rs[i] = rs2[i]; }
            if (y + 3 < y_end) { load_row(row_ptr(y + r + 3), ra2); load_row(row_ptr(y + 2 - r), rs2); }   // rows of the step after the next
            apply_step(ca, cs);
        }
    }
}

template <int NDW, int NW>
__global__ void __launch_bounds__(kMedLanes * NW) k_median_u8(const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst, uint8_t* __restrict__ padded_out,
                                                              int W, int H, int ksize, int rows_per_block) {
    __shared__ uint32_t hist[kMedWords * kMedLanes];
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8, "waves per histogram set");
    static_assert(NW <= NDW, "every wave needs a dword of the row");
    const int wv = threadIdx.x >> 6;
    if (NW == 1 || wv == 0) median_body<NDW, NW, 0>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (NW == 2 || wv == 1) median_body<NDW, NW, 1 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (wv == 2) median_body<NDW, NW, 2 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (NW == 4 || wv == 3) median_body<NDW, NW, 3 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (wv == 4) median_body<NDW, NW, 4 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (wv == 5) median_body<NDW, NW, 5 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else if (wv == 6) median_body<NDW, NW, 6 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
    else median_body<NDW, NW, 7 % NW>(hist, srcp, dst, padded_out, W, H, ksize, rows_per_block);
}
void launch_pad_cols(const uint8_t* src, uint8_t* padded, int w, int h, hipStream_t s) {
    const int wp = w + 2 * kMedPad;
    hipLaunchKernelGGL(k_pad_cols, dim3((wp + 255) / 256, h), dim3(256), 0, s, src, padded, w, h);
}
void launch_median_u8(const uint8_t* src, uint8_t* padded_tmp, uint8_t* dst, int w, int h, int ksize, hipStream_t s) {
    launch_pad_cols(src, padded_tmp, w, h, s);
    launch_median_padded(padded_tmp, dst, nullptr, w, h, ksize, s);
}
void launch_median_padded(const uint8_t* padded_src, uint8_t* dst, uint8_t* padded_next, int w, int h, int ksize, hipStream_t s) {
    const uint8_t* padded_tmp = padded_src;
    // Segments of rows.  A segment pays a ksize-row warm-up, so longer is cheaper in total work; but a wave is a long
    // serial instruction stream and 35 KB of LDS limits a CU to 4 of them, so the time is that of ONE segment as long as
    // there are no more than ~1000: aim for that many, never shorter than ksize / 2 rows.
    const int col_blocks = (w + kMedLanes - 1) / kMedLanes;
    // Histogram sets per launch (4 fit a CU, 1024 the GPU).  The two images of a pair run their chains side by side on two streams, so
    // a launch that asks for all 1024 only queues behind the other image's; and a set pays a ksize-row warm-up, so longer segments are
    // less work.  Measured per launch, both images interleaved (us at 1024 / 768 / 512 / 384 sets): ksize 17: 75 / 78 / 81 / 104; 41: 142 / 148 /
    // 140 / 174; 65: 259 / 245 / 224 / 273; 89: 373 / 345 / 251 / 356.  POPPY_MED_SETS forces a number.
    static const int forced_sets = getenv("POPPY_MED_SETS") ? std::max(64, atoi(getenv("POPPY_MED_SETS"))) : 0;
    // (Round 3: with the warm-up's coarse level summed instead of kept by atomics and cancelling updates skipped, 1024 sets are the fastest for
    // every ksize — us per launch at 1024 / 512 sets, both images: ksize 41: 116 / 124, 65: 178 / 192, 89: 238 / 259, the set-up 3.69 / 3.91 ms; up to round
    // 2 the long windows did better with 512.  tools/experiments/med_sets_sweep.sh.)
    const int seg_target = forced_sets ? forced_sets : 1024;
    int segs = std::max(1, seg_target / col_blocks);
    int rows = std::max((h + segs - 1) / segs, std::min(h, (ksize + 1) / 2));
    segs = (h + rows - 1) / rows;
    const int ndw = (ksize + 3) >> 2;
    // waves per histogram set: POPPY_MED_WAVES forces 1 / 2 / 4 / 8 (experiments); default by row length
    static const int forced = getenv("POPPY_MED_WAVES") ? atoi(getenv("POPPY_MED_WAVES")) : 0;
    const int nw = forced == 1 || ndw < 2 ? 1 : forced == 8 && ndw >= 12 ? 8 : forced == 4 && ndw >= 4 ? 4 : forced == 2 ? 2 : (ndw >= 13 && ndw != 17 ? 8 : ndw >= 4 ? kMedWavesLong : 1);      // eight waves from ksize 49 up: 8-10 % off the long windows' launches (round 3); 17 dwords share out badly among eight (3 rounds): 191 against 184 us with four
#define MEDW(N, NWV) hipLaunchKernelGGL((k_median_u8<N, NWV>), dim3(col_blocks, segs), dim3(kMedLanes * NWV), 0, s, padded_tmp, dst, padded_next, w, h, ksize, rows)
#define MED(N) case N: if (nw == 8) MEDW(N, (N >= 12 ? 8 : N >= 4 ? 4 : 1)); else if (nw == 4) MEDW(N, (N >= 4 ? 4 : 1)); else if (nw == 2) MEDW(N, (N >= 2 ? 2 : 1)); else MEDW(N, 1); break;
    switch (ndw) {
        MED(1) MED(2) MED(3) MED(4) MED(5) MED(6) MED(7) MED(8) MED(9) MED(10) MED(11) MED(12) MED(13) MED(14) MED(15) MED(16)
        MED(17) MED(18) MED(19) MED(20) MED(21) MED(22) MED(23)
        default: static_assert(kMedMaxDw == 23, "instantiations cover 1..23"); break;   // ksize > 89 is rejected by the caller
    }
#undef MEDW
#undef MED
}
size_t median_padded_bytes(int w, int h) { return (size_t)(w + 2 * kMedPad) * h + 16; }
int median_cols_min_ksize() {
    // every window of the chain since the column kernel needs 34 KB of LDS (four workgroups per compute unit): alone on the GPU k_median_u8 is as fast up to ksize 17
    // (44 / 47 against 40 / 41 us), beside other chains the column kernel shares the compute units better — 4K set-up 7.95 -> 7.72 ms, a pool step 57.5 -> 56.6 ms
    // (tools/experiments/median_min_ab.sh); 25 until then
    static const int v = getenv("POPPY_MED_COLS_MIN") ? atoi(getenv("POPPY_MED_COLS_MIN")) : 9;
    return v;
}
int median_cols_min_ksize_hard() {
    // never by default: on photographs a launch takes as long as its slowest tile (one whose medians need a second or third window of ranks), 200 - 250 us at
    // 1080p beside 115 - 190 for the lane-per-column kernel; set-up with 57: photographs 4.43 against 4.27 ms, noise 4.9 - 5.0 against 5.3 (profiles/r05_notes.md)
    static const int v = getenv("POPPY_MED_COLS_MIN_HARD") ? atoi(getenv("POPPY_MED_COLS_MIN_HARD")) : 999;
    return v;
}
bool prepare_median_u8() { return true; }

// ---- GaussianBlur 23x23, sigma 1, 8 bit: taps 1 14 62 102 62 14 1 (the other 16 taps are 0 in 8.8 fixed point) ------------
__constant__ int c_g7[7] = {1, 14, 62, 102, 62, 14, 1};

__global__ void __launch_bounds__(256) k_gauss23_h(const uint8_t* __restrict__ src, uint16_t* __restrict__ dst, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t* row = src + (size_t)y * W;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += (uint32_t)c_g7[k] * row[reflect101(x + k - 3, W)];
    dst[(size_t)y * W + x] = (uint16_t)s;                     // <= 255 * 256: fits
}
__global__ void __launch_bounds__(256) k_gauss23_v(const uint16_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) s += (uint32_t)c_g7[k] * src[(size_t)reflect101(y + k - 3, H) * W + x];
    const uint32_t v = (s + (1u << 15)) >> 16;
    dst[(size_t)y * W + x] = (uint8_t)(v > 255 ? 255 : v);
}
// fgMask += flow * scale; GaussianBlur(fgMask, 23 x 23, sigma 1) in one launch (src/extractor.cpp:150-154): a workgroup owns 64 x 16 outputs,
// stages the accumulated values of the tile and its halo of 3 in LDS (the accumulation is per pixel, so halo values are simply
// recomputed), runs the row pass into LDS and the column pass to the output.  Same integer arithmetic as k_acc_flow + k_gauss23_h / _v
// (3 launches, 47 us per step at 1080p, 12 dependent steps per image).
constexpr int kAgTx = 64, kAgTy = 16, kAgW = kAgTx + 6, kAgH = kAgTy + 6;
__global__ void __launch_bounds__(256) k_acc_gauss23(const uint8_t* __restrict__ acc, const uint8_t* __restrict__ flow, uint8_t* __restrict__ dst,
                                                     int W, int H, float acc_scale) {
    __shared__ uint8_t tile[kAgH * kAgW];
    __shared__ uint16_t hs[kAgH * kAgTx];
    const int tx0 = blockIdx.x * kAgTx, ty0 = blockIdx.y * kAgTy, tid = threadIdx.x;
    for (int i = tid; i < kAgH * kAgW; i += 256) {
        const int r = i / kAgW, c = i - r * kAgW;
        const size_t p = (size_t)reflect101(ty0 - 3 + r, H) * W + reflect101(tx0 - 3 + c, W);
        int t = cv_round_x86((float)flow[p] * acc_scale + 0.f);
        t = t < 0 ? 0 : t > 255 ? 255 : t;
        const int sum = acc[p] + t;
        tile[i] = (uint8_t)(sum > 255 ? 255 : sum);
    }
    __syncthreads();
    for (int i = tid; i < kAgH * kAgTx; i += 256) {
        const int r = i / kAgTx, c = i - r * kAgTx;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) sum += (uint32_t)c_g7[k] * tile[r * kAgW + c + k];
        hs[i] = (uint16_t)sum;
    }
    __syncthreads();
    for (int i = tid; i < kAgTy * kAgTx; i += 256) {
        const int r = i / kAgTx, c = i - r * kAgTx;
        const int x = tx0 + c, y = ty0 + r;
        if (x >= W || y >= H) continue;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) sum += (uint32_t)c_g7[k] * hs[(r + k) * kAgTx + c];
        const uint32_t v = (sum + (1u << 15)) >> 16;
        dst[(size_t)y * W + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}
// The same for widths divisible by 4, four pixels per dword throughout, and S dependent steps per launch (the byte-per-thread form above spends
// its time on byte loads, byte LDS reads and index divisions — 21 us per step at 1080p, 84 us at 4K, for 6 / 25 MB of traffic — and twelve
// dependent launches per image are mostly launch latency).  A workgroup owns 128 x 16 outputs of the LAST step and carries a halo of 3 rows and
// 4 columns (dwords stay whole) per step.  Values outside the image need no special case in any step: source and masks are read through
// reflect-101, and the blur of a reflected extension with a symmetric kernel is the reflected extension of the blur (the saturating add is
// pointwise), so the intermediate images come out right on the halo positions outside the image too.  Per step: (1) the row pass as two
// `v_dot4_u32_u8` per output (taps 1 14 62 102 | 62 14 1 0 against the window's two byte quadruples, cut out with `v_alignbyte`) into 16-bit sums
// in LDS, (2) the column pass on those; its result, plus the next step's mask, goes back to LDS as packed bytes, the last step's to memory.
// Same integers as k_acc_gauss23 in every step.
constexpr int kAvTx = 128, kAvTy = 16;
__device__ __forceinline__ uint32_t acc_add4(uint32_t a, uint32_t f, float acc_scale) {      // sat_u8(a + sat_u8(cvRound(f * scale))) per byte
    uint32_t out = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        int t = cv_round_x86((float)((f >> (8 * b)) & 255u) * acc_scale + 0.f);
        t = t < 0 ? 0 : t > 255 ? 255 : t;
        const uint32_t sum = ((a >> (8 * b)) & 255u) + (uint32_t)t;
        out |= (sum > 255u ? 255u : sum) << (8 * b);
    }
    return out;
}
// four bytes of plane `p` at row offset `row`, columns x .. x + 3 (x % 4 == 0, W % 4 == 0: inside the image or outside, never across)
__device__ __forceinline__ uint32_t load4_reflect(const uint8_t* __restrict__ p, size_t row, int x, int W) {
    if (x >= 0 && x < W) return *reinterpret_cast<const uint32_t*>(p + row + x);
    uint32_t v = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) v |= (uint32_t)p[row + min(max(reflect101_once(x + b, W), 0), W - 1)] << (8 * b);   // (columns beyond the halo in use: kept in range)
    return v;
}
template <int S>
__global__ void __launch_bounds__(256) k_acc_gauss23_v4(const uint8_t* __restrict__ acc, const uint8_t* __restrict__ flows, size_t plane, uint8_t* __restrict__ dst,
                                                        int W, int H, float acc_scale) {
    constexpr int kRows = kAvTy + 6 * S, kDw = kAvTx / 4 + 2 * S, kHsDw = 2 * kDw + 2;    // region rows, dwords per row, 16-bit-pair dwords per row (+2: bank spread)
    __shared__ uint32_t tile[kRows * kDw];
    __shared__ uint32_t hs[kRows * kHsDw];
    const int tx0 = blockIdx.x * kAvTx - 4 * S, ty0 = blockIdx.y * kAvTy - 3 * S, tid = threadIdx.x;      // image position of region row 0 / dword 0
    for (int i = tid; i < kRows * kDw; i += 256) {
        const int r = i / kDw, j = i - r * kDw;
        const size_t row = (size_t)reflect101(ty0 + r, H) * W;
        tile[i] = acc_add4(load4_reflect(acc, row, tx0 + 4 * j, W), load4_reflect(flows, row, tx0 + 4 * j, W), acc_scale);
    }
    __syncthreads();
    constexpr uint32_t kTapsLo = 1u | 14u << 8 | 62u << 16 | 102u << 24, kTapsHi = 62u | 14u << 8 | 1u << 16;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        {   // row pass: rows 3 s .. kRows - 3 s, dwords s + 1 .. kDw - s - 1
            const int r_lo = 3 * s, nr = kRows - 6 * s, q_lo = s + 1, nq = kDw - 2 * s - 2;
            for (int i = tid; i < nr * nq; i += 256) {
                const int r = r_lo + i / nq, q = q_lo + i % nq;
                const uint32_t d0 = tile[r * kDw + q - 1], d1 = tile[r * kDw + q], d2 = tile[r * kDw + q + 1];    // bytes x - 4 .. x + 7 of the dword's first pixel x
                // pixel k's window is bytes x + k - 3 .. x + k + 3 = offsets 1 + k .. 7 + k of the twelve
                const uint32_t l0 = __builtin_amdgcn_alignbyte(d1, d0, 1), l1 = __builtin_amdgcn_alignbyte(d1, d0, 2), l2 = __builtin_amdgcn_alignbyte(d1, d0, 3), l3 = d1;
                const uint32_t h0 = __builtin_amdgcn_alignbyte(d2, d1, 1), h1 = __builtin_amdgcn_alignbyte(d2, d1, 2), h2 = __builtin_amdgcn_alignbyte(d2, d1, 3), h3 = d2;
                const uint32_t s0 = __builtin_amdgcn_udot4(h0, kTapsHi, __builtin_amdgcn_udot4(l0, kTapsLo, 0u, false), false);
                const uint32_t s1 = __builtin_amdgcn_udot4(h1, kTapsHi, __builtin_amdgcn_udot4(l1, kTapsLo, 0u, false), false);
                const uint32_t s2 = __builtin_amdgcn_udot4(h2, kTapsHi, __builtin_amdgcn_udot4(l2, kTapsLo, 0u, false), false);
                const uint32_t s3 = __builtin_amdgcn_udot4(h3, kTapsHi, __builtin_amdgcn_udot4(l3, kTapsLo, 0u, false), false);
                hs[r * kHsDw + 2 * q] = s0 | s1 << 16;              // <= 255 * 256: 16 bits each
                hs[r * kHsDw + 2 * q + 1] = s2 | s3 << 16;
            }
        }
        __syncthreads();
        {   // column pass: rows 3 s + 3 .. kRows - 3 s - 3, same dwords
            const int r_lo = 3 * s + 3, nr = kRows - 6 * s - 6, q_lo = s + 1, nq = kDw - 2 * s - 2;
            constexpr uint32_t g7[7] = {1, 14, 62, 102, 62, 14, 1};
            for (int i = tid; i < nr * nq; i += 256) {
                const int r = r_lo + i / nq, q = q_lo + i % nq;
                uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const uint32_t lo = hs[(r + k - 3) * kHsDw + 2 * q], hi = hs[(r + k - 3) * kHsDw + 2 * q + 1];
                    s0 += g7[k] * (lo & 0xffffu); s1 += g7[k] * (lo >> 16);
                    s2 += g7[k] * (hi & 0xffffu); s3 += g7[k] * (hi >> 16);
                }
                auto fin = [](uint32_t sum) { const uint32_t v = (sum + (1u << 15)) >> 16; return v > 255u ? 255u : v; };
                const uint32_t out = fin(s0) | fin(s1) << 8 | fin(s2) << 16 | fin(s3) << 24;
                const int x = tx0 + 4 * q, y = ty0 + r;
                if (s + 1 < S) {                                   // the next step's input: + its mask (read where the reflected position lies)
                    const size_t row = (size_t)reflect101(y, H) * W;
                    tile[r * kDw + q] = acc_add4(out, load4_reflect(flows + (size_t)(s + 1) * plane, row, x, W), acc_scale);
                } else if (x < W && y < H) *reinterpret_cast<uint32_t*>(dst + (size_t)y * W + x) = out;
            }
        }
        if (s + 1 < S) __syncthreads();
    }
}
bool acc_gauss23_fused_takes(int w, int h) { return w % 4 == 0 && w >= 2 * kAvTx && h >= 4 * kAvTy; }
void launch_acc_gauss23_fused(const uint8_t* acc, const uint8_t* flows, size_t plane, int steps, uint8_t* dst, int w, int h, float acc_scale, hipStream_t s) {
    const dim3 grid((w + kAvTx - 1) / kAvTx, (h + kAvTy - 1) / kAvTy);
#define ACCV4(N) case N: hipLaunchKernelGGL(k_acc_gauss23_v4<N>, grid, dim3(256), 0, s, acc, flows, plane, dst, w, h, acc_scale); break;
    switch (steps) { ACCV4(1) ACCV4(2) ACCV4(3) ACCV4(4) ACCV4(6) default: break; }
#undef ACCV4
}
void launch_acc_gauss23(const uint8_t* acc, const uint8_t* flow, uint8_t* dst, int w, int h, float acc_scale, hipStream_t s) {
    hipLaunchKernelGGL(k_acc_gauss23, dim3((w + kAgTx - 1) / kAgTx, (h + kAgTy - 1) / kAgTy), dim3(256), 0, s, acc, flow, dst, w, h, acc_scale);
}

void launch_gauss23_u8(const uint8_t* src, uint16_t* tmp, uint8_t* dst, int w, int h, hipStream_t s) {
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_gauss23_h, grid, dim3(256), 0, s, src, tmp, w, h);
    hipLaunchKernelGGL(k_gauss23_v, grid, dim3(256), 0, s, tmp, dst, w, h);
}

// ---- blur_margin's GaussianBlur(127x127, sigma 6) of one strip of a 3-channel 8-bit canvas (src/util.cpp:574-602) -------------
// The strip (rx, ry, rw, rh) is filtered as an image of its own: reflect-101 inside the strip; a strip dimension of 1 is not
// filtered (the taps collapse to {256}).  taps: n fixed-point (8.8) values.  Pass 1 reads the unblurred canvas, pass 2 writes
// the output canvas.
__global__ void __launch_bounds__(256) k_strip_h(const uint8_t* __restrict__ canvas, int cw, uint32_t* __restrict__ tmp, const int* __restrict__ taps, int n,
                                                 int rx, int ry, int rw, int rh) {
    const int e = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;      // e = x * 3 + channel inside the strip
    if (e >= rw * 3) return;
    const int x = e / 3, c = e - x * 3;
    uint32_t s = 0;
    if (rw == 1) s = 256u * canvas[((size_t)(ry + y) * cw + rx) * 3 + c];
    else
        for (int k = 0; k < n; ++k)
            if (taps[k]) s += (uint32_t)taps[k] * canvas[((size_t)(ry + y) * cw + rx + reflect101(x + k - n / 2, rw)) * 3 + c];
    tmp[(size_t)y * rw * 3 + e] = s;
}
__global__ void __launch_bounds__(256) k_strip_v(const uint32_t* __restrict__ tmp, uint8_t* __restrict__ out, int cw, const int* __restrict__ taps, int n,
                                                 int rx, int ry, int rw, int rh) {
    const int e = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (e >= rw * 3) return;
    uint32_t s = 0;
    if (rh == 1) s = 256u * tmp[e];
    else
        for (int k = 0; k < n; ++k)
            if (taps[k]) s += (uint32_t)taps[k] * tmp[(size_t)reflect101(y + k - n / 2, rh) * rw * 3 + e];
    const uint32_t v = (s + (1u << 15)) >> 16;
    out[((size_t)(ry + y) * cw + rx) * 3 + e] = (uint8_t)(v > 255 ? 255 : v);
}
void launch_strip_blur(const uint8_t* canvas, uint8_t* out, int cw, uint32_t* tmp, const int* d_taps, int n, int rx, int ry, int rw, int rh, hipStream_t s) {
    if (rw <= 0 || rh <= 0) return;
    dim3 grid((rw * 3 + 255) / 256, rh);
    hipLaunchKernelGGL(k_strip_h, grid, dim3(256), 0, s, canvas, cw, tmp, d_taps, n, rx, ry, rw, rh);
    hipLaunchKernelGGL(k_strip_v, grid, dim3(256), 0, s, tmp, out, cw, d_taps, n, rx, ry, rw, rh);
}

// ---- log mask, multiply, 8 bit, histogram --------------------------------------------------------------------------------
// cv::log for floats: 256-entry table of (ln(1 + i/256), 1/(1 + i/256)) — the last entry is (ln 2, 1/2) with the argument
// shifted by -1/512 — and a cubic; the table is computed on the host in long double and handed over as floats.
__device__ __forceinline__ float cv_log32f(float x, const float* __restrict__ tab) {
    const float A0 = 0.3333333333333333333333333f, A1 = -0.5f, A2 = 1.f;
    const float ln2 = (float)0.69314718055994530941723212145818;
    const int i0 = __float_as_int(x);
    const float bf = __int_as_float((i0 & ((1 << 15) - 1)) | (127 << 23));
    const int idx = (i0 >> 14) & 510;
    const float y0 = (float)(((i0 >> 23) & 0xff) - 127) * ln2 + tab[idx];
    const float x0 = (bf - 1.f) * tab[idx + 1] + (idx == 510 ? -1.f / 512 : 0.f);
    return ((A0 * x0 + A1) * x0 + A2) * x0 + y0;
}

__global__ void __launch_bounds__(256) k_fg_mask(const uint8_t* __restrict__ grey, const uint8_t* __restrict__ fg, const float* __restrict__ tab,
                                                 const float* __restrict__ radial, uint8_t* __restrict__ masked, unsigned* __restrict__ hist,
                                                 float* __restrict__ dbg, int n) {
    __shared__ unsigned lh[256];
    __shared__ float ltab[512];
    lh[threadIdx.x] = 0;
    ltab[threadIdx.x] = tab[threadIdx.x]; ltab[threadIdx.x + 256] = tab[threadIdx.x + 256];
    __syncthreads();
    const float ln20 = cv_log32f(20.f, ltab);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        const float g = (float)grey[p] * kInv255 + 0.f;        // convertTo(CV_32F, 1/255)
        float m = (float)fg[p] * kInv255 + 0.f;
        if (radial) m = m * radial[p];                         // Settings::enable_radial_mask: multiply(fgMaskFloat, radialMaskFloat), src/extractor.cpp:195-197
        const float lin = m * 19.f + 1.f;                      // convertTo(CV_32F, 19, 1)
        const float lg = cv_log32f(lin, ltab);
        const float fin = __fdiv_rn(lg, ln20);
        const float mk = g * fin;
        const uint8_t o = sat_u8(cv_round_x86(mk * 255.f + 0.f));
        masked[p] = o;
        atomicAdd(&lh[o], 1u);
        if (dbg) { dbg[p] = lin; dbg[(size_t)n + p] = lg; dbg[2 * (size_t)n + p] = fin; }
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// lut[i] = saturate(round(sum_{first < j <= i} hist[j] * 255 / (total - hist[first]))); a constant image maps to itself
__global__ void __launch_bounds__(256) k_equalize_lut(const unsigned* __restrict__ hist, int total, uint8_t* __restrict__ lut) {
    __shared__ int s_h[256];
    __shared__ int s_first;
    const int t = threadIdx.x;
    const int h = (int)hist[t];
    s_h[t] = h;
    if (t == 0) s_first = 255;
    __syncthreads();
    if (h) atomicMin(&s_first, t);                         // first non-empty bin
    __syncthreads();
    const int first = s_first, hf = s_h[first];
    if (hf == total) { lut[t] = (uint8_t)first; return; }  // a constant image maps to itself
    // inclusive prefix sums of the 256 counts (Hillis-Steele in LDS; integer sums: the reference's running sum)
    for (int d = 1; d < 256; d <<= 1) {
        const int add = t >= d ? s_h[t - d] : 0;
        __syncthreads();
        s_h[t] += add;
        __syncthreads();
    }
    const float scale = __fdiv_rn(256 - 1.f, (float)(total - hf));
    const int sum = s_h[t] - s_h[first];                   // sum over first < j <= t
    lut[t] = t <= first ? (uint8_t)0 : sat_u8(cv_round_x86((float)sum * scale));
}
__global__ void __launch_bounds__(256) k_apply_lut(const uint8_t* __restrict__ src, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst, int n) {
    __shared__ uint8_t l[256];
    l[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) dst[p] = l[src[p]];
}
void launch_equalize_from_hist(const uint8_t* src, const unsigned* hist, uint8_t* lut, uint8_t* out, int n_px, hipStream_t s) {
    const int blocks = std::min((n_px + 255) / 256, 2048);
    hipLaunchKernelGGL(k_equalize_lut, dim3(1), dim3(256), 0, s, hist, n_px, lut);
    hipLaunchKernelGGL(k_apply_lut, dim3(blocks), dim3(256), 0, s, src, lut, out, n_px);
}
void launch_fg_tail(const uint8_t* grey, const uint8_t* fg, const float* d_logtab, const float* radial_or_null, uint8_t* masked, unsigned* hist, uint8_t* lut,
                    uint8_t* out, float* dbg_or_null, int n_px, hipStream_t s) {
    (void)hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), s);
    const int blocks = std::min((n_px + 255) / 256, 2048);
    hipLaunchKernelGGL(k_fg_mask, dim3(blocks), dim3(256), 0, s, grey, fg, d_logtab, radial_or_null, masked, hist, dbg_or_null, n_px);
    hipLaunchKernelGGL(k_equalize_lut, dim3(1), dim3(256), 0, s, hist, n_px, lut);
    hipLaunchKernelGGL(k_apply_lut, dim3(blocks), dim3(256), 0, s, masked, lut, out, n_px);
}

}  // namespace poppy_hip
