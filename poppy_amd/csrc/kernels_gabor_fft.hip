// kernels_gabor_fft.hip — the 16-orientation Gabor banks (Extractor::keypoints' 31 x 31 bank, gabor_filter's 13 x 13 x 3 bank) by
// tiled double-precision FFTs instead of 15 376 (2 704) double multiply-adds per pixel.
//
//   src/util.cpp:31-61 (gabor_filter: filter2D per angle, clamp to [0, 1], mean of 16), src/extractor.cpp:63-71
//   OCV/imgproc/src/filter.dispatch.cpp:1291-1292 -> OCV/imgproc/src/templmatch.cpp:566-760 (crossCorr, maxDepth CV_64F)
//
// For float images and a kernel this large the reference's filter2D IS an FFT correlation in double precision, rounded to float once
// at the end; its planes equal the exact correlation sums up to the DFT's noise (~1e-13 relative).  The direct kernel
// (kernels_prefilter2.hip: k_gabor_bank) forms those sums in double and is bound by the FP64 pipe (1.6 ms per 1080p image, half of the
// pair set-up's exclusive GPU time together with the second image's).  Here the same sums come from 64 x 64 FFTs, overlap-save:
//   a workgroup owns a block of B x B output pixels, B = 64 - ks + 1 (34 for the 31 x 31 bank), and the 64 x 64 input patch around it;
//   forward 2-D FFT of the patch (reflect-101 borders as filter2D's BORDER_DEFAULT);
//   per PAIR of orientations (a, b): spectrum x (conj K_a^ + i conj K_b^) / 4096, inverse 2-D FFT: the real part is orientation a's
//   plane, the imaginary part orientation b's (both are real) — 8 inverse transforms serve the 16 orientations;
//   clamp, accumulate in float in the reference's order (a then b), write mean.
// Round 4: the forward transform runs columns then rows and the inverse ones rows then columns, so that (1) the patch spectrum stays in
// the registers of the forward transform's last pass — the inverse transforms' first pass is the same job on the same elements — and is
// multiplied by a pair's kernel spectrum there, and (2) the planes leave the last inverse pass in registers with a wave's lanes along x:
// clamp, accumulate and the final store need no LDS.  56 LDS phases per tile instead of 89, 9 barriers fewer: alone on the GPU at 1080p
// 257 -> 226 us (31 x 31), 306 -> 252 us (13 x 13 x 3) (`tools/experiments/gabor_ab.sh`).
// A 64-point transform is two passes of 8-point butterflies held in registers (64 = 8 x 8) with the twiddle W64^(n2 k1) between
// them; a pass reads 8 values of a line from LDS and writes them back in place.  The forward transform leaves frequency
// k1 + 8 k2 at position 8 k1 + k2; the kernel spectra are stored in that order and the inverse transform consumes it.
// ~2.3 kFLOP per pixel instead of 30.7 k.  Error: ~1e-15 relative to the magnitude of the sums, two orders below the reference's
// own DFT noise; the planes are rounded to float afterwards, so a plane value could differ from the direct sum's only where the exact
// sum lies within that distance of a float rounding boundary (one float in ~1e8, seen in round 4's fuzz and on the photograph
// fixtures) or of zero (the borders of all-black regions).  Those values are detected (rounding_in_doubt) and formed as the direct
// kernel forms them (k_gabor_redo): the two forms give the same bits; tests/test_gpu_prefilter2.py compares them.
#include "kernels_prefilter.h"
#include "pyramid_device.h"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace poppy_hip {

namespace {

typedef double cd __attribute__((ext_vector_type(2)));       // (re, im)
constexpr int kFN = 64;                                       // transform size
constexpr int kFS = 65;                                       // LDS row stride in elements: lanes that walk 16 consecutive lines hit 16 different 16-byte slots
constexpr double kSqrtHalf = 0.70710678118654752440;
constexpr int kFT = 512;                                      // threads per tile: one 8-point job per thread and pass, 8 spectrum elements in registers;
                                                              // two tiles per CU (LDS) = 4 waves per SIMD.  (256 threads with two jobs each = 2 waves per SIMD: 0.37 / 0.40 ms; 512: 0.27 / 0.27 ms)

__constant__ double c_w64[2 * kFN];                           // W64^m = exp(-2 pi i m / 64) as (cos, -sin)

__device__ __forceinline__ cd cmul(cd a, cd w) { return cd{fma(a.x, w.x, -(a.y * w.y)), fma(a.x, w.y, a.y * w.x)}; }
// multiplication by SIGN * i
template <int SIGN> __device__ __forceinline__ cd rot90(cd a) { return SIGN < 0 ? cd{a.y, -a.x} : cd{-a.y, a.x}; }

template <int SIGN>
__device__ __forceinline__ void fft4(cd y0, cd y1, cd y2, cd y3, cd& Y0, cd& Y1, cd& Y2, cd& Y3) {
    const cd t0 = y0 + y2, t1 = y0 - y2, t2 = y1 + y3, t3 = rot90<SIGN>(y1 - y3);
    Y0 = t0 + t2; Y2 = t0 - t2; Y1 = t1 + t3; Y3 = t1 - t3;
}
// X[k] = sum_n v[n] exp(SIGN 2 pi i n k / 8), natural order in and out
template <int SIGN>
__device__ __forceinline__ void fft8(cd (&v)[8]) {
    cd E[4], O[4];
    fft4<SIGN>(v[0], v[2], v[4], v[6], E[0], E[1], E[2], E[3]);
    fft4<SIGN>(v[1], v[3], v[5], v[7], O[0], O[1], O[2], O[3]);
    const cd w1 = cd{kSqrtHalf, SIGN * kSqrtHalf}, w3 = cd{-kSqrtHalf, SIGN * kSqrtHalf};
    O[1] = cmul(O[1], w1); O[2] = rot90<SIGN>(O[2]); O[3] = cmul(O[3], w3);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = E[k] + O[k]; v[k + 4] = E[k] - O[k]; }
}

// One pass over all 64 lines of the 64 x 64 array: 512 jobs (line, g), two per thread; a job transforms the 8 elements
// {8 i + g} (STRIDED) or {8 g + i} of its line in place and, with TWIDDLE, multiplies output i by W64^(SIGN g i).
// COLS: the lines are the columns.  A wave's 64 lanes are 64 consecutive lines with one g.
// LOAD = false: the job's inputs are in `reg` already; STORE = false: its outputs stay in `reg` (one job per thread: kFT == 512).  The
// forward transform's last pass and the inverse transforms' first pass are the same jobs on the same elements ({8 g + i} of column
// `line`), so the patch spectrum never goes back to LDS: it is multiplied by a pair's kernel spectrum in the registers it was born in.
template <int SIGN, bool STRIDED, bool TWIDDLE, bool COLS, bool LOAD = true, bool STORE = true>
__device__ __forceinline__ void fft_pass(cd* __restrict__ L, int tid, cd (&reg)[8]) {
    static_assert(kFT == 512, "one job per thread");
    const int line = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
    int at[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = STRIDED ? 8 * i + g : 8 * g + i;
        at[i] = COLS ? idx * kFS + line : line * kFS + idx;
        if (LOAD) reg[i] = L[at[i]];
    }
    fft8<SIGN>(reg);
    if (TWIDDLE) {
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            const int m = (g * i) & 63;
            reg[i] = cmul(reg[i], cd{c_w64[2 * m], SIGN < 0 ? c_w64[2 * m + 1] : -c_w64[2 * m + 1]});
        }
    }
    if (STORE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) L[at[i]] = reg[i];
    }
}
template <int SIGN, bool STRIDED, bool TWIDDLE, bool COLS>
__device__ __forceinline__ void fft_pass(cd* __restrict__ L, int tid) { cd v[8]; fft_pass<SIGN, STRIDED, TWIDDLE, COLS>(L, tid, v); }

// ---- the values whose float could differ from the direct sum's -----------------------------------------------------------------------
// The transform's planes are ~1e-15 x (the size of the patch) from the direct kernel's sums (k_gabor_bank: an fma chain in double over
// the taps, row by row; measured on the GPU over ~1e5 such values of photographs, textures and flat shapes at 1080p: at most 1.1e-14 x
// max(1, the patch's largest magnitude), rms ~1.5e-15).  Rounded to float and clamped to [0, 1], a plane value v contributes the same
// as the direct sum unless it lies within that distance of the midpoint between two floats, or of zero (the clamp's corner: the noise
// around an exact zero at the border of a black region would survive it).  `band` is 1e-13 x max(1, the patch's largest magnitude): ten
// times the measured worst case — an empirical margin (the transform's error follows the patch's and the kernel's norms), so "the two forms
// agree in every bit" means: on everything tested, with that margin; POPPY_GABOR_BAND=1e30 re-forms every pixel for a check without it.
// Not in doubt: v below -band or above 1 + band (clamped either way) and the planes of a window that holds only zeros (exact zeros in
// the direct sums; the kernel knows such windows from ballots taken while it loads the patch and adds nothing for them).  A pixel
// with a value in doubt — 1 value in ~1e4, most of them small: a float's neighbours are 2^-24 of its size apart — goes on a list, and
// k_gabor_redo, launched behind the transform kernel, forms the listed pixels again as the direct kernel forms them: 16 lanes per
// pixel, one orientation's chain each.  (Re-forming a value inside the transform kernel holds its whole tile at a barrier for the
// length of a 961-step chain: 0.27 -> 1.1 ms at 1080p; all listed pixels of an image side by side take ~20 us.)
__device__ unsigned long long g_doubt[3];                     // since the last read: values near zero, near a midpoint; pixels re-formed (poppy_hip_gabor_doubt)
__device__ __forceinline__ int rounding_in_doubt(double v, double band) {      // 0: no, 1: near zero, 2: near a midpoint
    if (v < -band || v > 1.0 + band) return 0;
    if (v < band) return 1;
    // v = f + d with f the nearest float; the midpoints lie half a spacing away: 2^(e - 24) above f and below it, except below a power of two
    // (2^(e - 25)).  In doubt when d comes within `band` of one of them.
    const float f = (float)v;
    const double d = v - (double)f;
    const int fb = __float_as_int(f);
    float h = __int_as_float((fb & 0x7f800000) - (24 << 23));
    if ((fb & 0x007fffff) == 0 && d < 0.0) h *= 0.5f;
    return fabs(d) > (double)h - band ? 2 : 0;
}

// list[0]: entries, list[2 + e]: (y * W + x) * CN + ch.  A group of 16 lanes takes an entry; lane o of the group runs orientation o's chain.
// A chain is KS * KS dependent steps, so everything it reads is staged where a step costs an LDS read: the 16 windows of a workgroup's
// entries once (all loads in flight together), the bank's taps row by row, two rows in LDS, the next one fetched while a row is summed.
// (With the operands read from global memory as the chain went, a launch took one memory latency per window row: 76 us at 1080p however
// few the entries; 120 - 145 us with ~7 000.)
template <int KS>
__global__ void __launch_bounds__(256) k_gabor_redo(const float* __restrict__ src, const double* __restrict__ bank, const unsigned* __restrict__ list,
                                                    float* __restrict__ dst, int W, int H, int CN) {
    constexpr int R = KS / 2, N = KS * KS, RT = KS * 16;      // RT: taps of one window row, [dx][orientation]
    extern __shared__ __attribute__((aligned(16))) double redo_lds[];
    double* const taps = redo_lds;                            // [2][RT]
    float* const win = (float*)(redo_lds + 2 * RT);           // [16][N]
    const unsigned n = list[0];
    const int tid = threadIdx.x, o = tid & 15, g = tid >> 4;
    if (n && blockIdx.x == 0 && tid == 0) atomicAdd(&g_doubt[2], (unsigned long long)n);
    for (unsigned e0 = blockIdx.x * 16; e0 < n; e0 += gridDim.x * 16) {                 // uniform over the workgroup
        const bool have = e0 + g < n;
        const unsigned idx = have ? list[2 + e0 + g] : 0;
        const int ch = idx % CN, p = idx / CN, y = p / W, x = p - y * W;
        __syncthreads();                                      // the previous batch's windows and taps have been read
        if (have)
            for (int k0 = o; k0 < N; k0 += 16 * 16) {        // 16 loads in flight per lane (a load waited for before the next one is issued costs a latency each: 60 of them)
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = min(k0 + 16 * u, N - 1), dy = k / KS, dx = k - dy * KS;
                    v[u] = src[((size_t)reflect101(y - R + dy, H) * W + reflect101(x - R + dx, W)) * CN + ch];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) if (k0 + 16 * u < N) win[g * N + k0 + 16 * u] = v[u];
            }
        for (int k = tid; k < RT; k += 256) taps[k] = bank[k];
        double acc = 0.0;
#pragma unroll 1                                              // (unrolled, the 961 steps are 30 KB of straight-line code fetched once per batch)
        for (int dy = 0; dy < KS; ++dy) {
            __syncthreads();                                  // row dy's taps (and, the first time, the windows) are in LDS
            double nx0 = 0.0, nx1 = 0.0;                      // row dy + 1's taps: RT <= 512 doubles, at most two per thread
            if (dy + 1 < KS) {
                if (tid < RT) nx0 = bank[(size_t)(dy + 1) * RT + tid];
                if (tid + 256 < RT) nx1 = bank[(size_t)(dy + 1) * RT + tid + 256];
            }
            const double* tr = taps + (dy & 1) * RT + o;
            const float* wr = win + g * N + dy * KS;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) acc = fma(tr[dx * 16], (double)wr[dx], acc);
            __builtin_amdgcn_sched_barrier(0);                // the wait for the next row's taps stays behind this row's chain
            if (dy + 1 < KS) {
                double* tn = taps + ((dy + 1) & 1) * RT;      // last read two barriers ago
                if (tid < RT) tn[tid] = nx0;
                if (tid + 256 < RT) tn[tid + 256] = nx1;
            }
        }
        const float f = fminf(fmaxf((float)acc, 0.f), 1.f);
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += __shfl(f, (tid & 48) + k);         // dst += plane, orientation by orientation
        if (have && o == 0) dst[idx] = sum * 0.0625f;
    }
}

template <int KS>
__global__ void __launch_bounds__(kFT, 4) k_gabor_fft(const float* __restrict__ src, const cd* __restrict__ G, unsigned* __restrict__ list,
                                                   float* __restrict__ dst, int W, int H, int CN, int tiles_x, double band_unit) {
    constexpr int R = KS / 2, B = kFN - KS + 1, kEl = kFN * kFN / kFT;
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];      // 64 x 65 complex doubles: just over the static 64 KB limit
    cd* const L = (cd*)lds_raw;
    unsigned long long* const rowmask = (unsigned long long*)(lds_raw + 2 * kFN * kFS);   // [64] which pixels of a patch row are not zero; [64..64+B) the same over KS rows
    float* const wavemax = (float*)(rowmask + 2 * kFN);                                   // [8] the largest magnitude each wave has loaded
    const int tid = threadIdx.x, lane = tid & 63;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, ch = blockIdx.y;
    const int bx = tx * B, by = ty * B;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < kEl; ++i) {
        const int e = tid + kFT * i, y = e >> 6, x = e & 63;
        const float v = src[((size_t)reflect101(by - R + y, H) * W + reflect101(bx - R + x, W)) * CN + ch];
        L[y * kFS + x] = cd{(double)v, 0.0};
        const unsigned long long nz = __ballot(v != 0.f);                  // a wave holds one patch row (x = lane)
        if (x == 0) rowmask[y] = nz;
        if (KS != 13) amax = fmaxf(amax, fabsf(v));                        // the 13 x 13 bank runs on bytes / 255: at most 1
    }
    if (KS != 13) {
#pragma unroll
        for (int d = 32; d; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
        if (lane == 0) wavemax[tid >> 6] = amax;
    }
    __syncthreads();
    if (tid < B) {                                                         // the rows of output row tid's windows together
        unsigned long long m = 0;
        for (int dy = 0; dy < KS; ++dy) m |= rowmask[tid + dy];
        rowmask[kFN + tid] = m;
    }
    // forward: columns, then rows; the last pass leaves the spectrum in registers: row `lane`, columns 8 g + i (g = tid / 64)
    fft_pass<-1, true, true, true>(L, tid);    __syncthreads();
    fft_pass<-1, false, false, true>(L, tid);  __syncthreads();
    fft_pass<-1, true, true, false>(L, tid);   __syncthreads();
    cd P[8];
    fft_pass<-1, false, false, false, true, false>(L, tid, P);
    amax = 1.f;
    if (KS != 13) {
#pragma unroll
        for (int w = 0; w < kFT / 64; ++w) amax = fmaxf(amax, wavemax[w]);
    }
    const double band = KS != 13 ? band_unit * (double)amax : band_unit;
    // inverse: rows, then columns; the last pass leaves the planes in registers: column `lane`, rows g + 8 i — output pixel (lane, g + 8 i) of
    // the tile's B x B block, a wave's lanes along x
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[8];
    unsigned state = 0;                                                    // bit 2 i: output i is inside the block and the image and not an all-zero window (whose planes are exact zeros: nothing to add); bit 2 i + 1: on the list
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] = 0.f;
        const int ny = g + 8 * i;
        if (lane < B && ny < B && bx + lane < W && by + ny < H && ((rowmask[kFN + min(ny, B - 1)] >> lane) & ((1ull << KS) - 1)) != 0) state |= 1u << (2 * i);
    }
    for (int j = 0; j < 8; ++j) {
        const cd* Gj = G + (size_t)j * (kFN * kFN) + (size_t)g * 8 * kFN + lane;       // the kernel spectra lie transposed: [column][row]
        cd v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = cmul(P[i], Gj[i * kFN]);
        __syncthreads();                                    // the previous pair's last pass (or the forward transform's) has read its inputs
        fft_pass<1, false, true, false, false, true>(L, tid, v);  __syncthreads();
        fft_pass<1, true, false, false>(L, tid);   __syncthreads();
        fft_pass<1, false, true, true>(L, tid);    __syncthreads();
        fft_pass<1, true, false, true, true, false>(L, tid, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (state & (1u << (2 * i))) {
                const cd c = v[i];
                const int dx_ = rounding_in_doubt(c.x, band), dy_ = rounding_in_doubt(c.y, band);
                if ((dx_ | dy_) && list) {                                  // rare
                    if (dx_) atomicAdd(&g_doubt[dx_ - 1], 1ull);
                    if (dy_) atomicAdd(&g_doubt[dy_ - 1], 1ull);
                    if (!(state & (2u << (2 * i)))) { state |= 2u << (2 * i); list[2 + atomicAdd(list, 1u)] = (unsigned)(((by + g + 8 * i) * W + bx + lane) * CN + ch); }
                }
                acc[i] += fminf(fmaxf((float)c.x, 0.f), 1.f);            // plane.setTo(1, plane > 1); setTo(0, plane < 0); dst += plane
                acc[i] += fminf(fmaxf((float)c.y, 0.f), 1.f);            // ... orientation 2 j, then 2 j + 1
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ny = g + 8 * i;
        if (lane < B && ny < B && bx + lane < W && by + ny < H) dst[((size_t)(by + ny) * W + bx + lane) * CN + ch] = acc[i] * 0.0625f;   // dst /= 16
    }
}

int perm64(int p) { return (p >> 3) + 8 * (p & 7); }         // the frequency held at position p after the forward transform

}  // namespace

// Spectra of the 16 kernels of a bank (`bank`: [orientation][ks * ks] floats, tap (dy, dx) multiplies the pixel at (x + dx - ks/2,
// y + dy - ks/2)), paired, conjugated (correlation), scaled by 1/4096 and stored in the transform's position order:
// 8 x 4096 complex doubles.  Computed once per context on the host, in long double.
std::vector<double> gabor_fft_tables(const std::vector<float>& bank, int ks) {
    std::vector<double> out((size_t)8 * kFN * kFN * 2);
    std::vector<long double> cs(kFN), sn(kFN);
    for (int m = 0; m < kFN; ++m) { cs[m] = cosl(2.0L * M_PIl * m / kFN); sn[m] = sinl(2.0L * M_PIl * m / kFN); }
    std::vector<long double> hr((size_t)16 * kFN * kFN), hi((size_t)16 * kFN * kFN);
    std::vector<long double> rr((size_t)ks * kFN), ri((size_t)ks * kFN);
    for (int o = 0; o < 16; ++o) {
        const float* K = &bank[(size_t)o * ks * ks];
        for (int dy = 0; dy < ks; ++dy)                       // along x first
            for (int kc = 0; kc < kFN; ++kc) {
                long double a = 0, b = 0;
                for (int dx = 0; dx < ks; ++dx) { const int m = (kc * dx) & (kFN - 1); a += (long double)K[dy * ks + dx] * cs[m]; b -= (long double)K[dy * ks + dx] * sn[m]; }
                rr[(size_t)dy * kFN + kc] = a; ri[(size_t)dy * kFN + kc] = b;
            }
        for (int kr = 0; kr < kFN; ++kr)
            for (int kc = 0; kc < kFN; ++kc) {
                long double a = 0, b = 0;
                for (int dy = 0; dy < ks; ++dy) {
                    const int m = (kr * dy) & (kFN - 1);
                    const long double c = cs[m], s = -sn[m], xr = rr[(size_t)dy * kFN + kc], xi = ri[(size_t)dy * kFN + kc];
                    a += xr * c - xi * s; b += xr * s + xi * c;
                }
                hr[((size_t)o * kFN + kr) * kFN + kc] = a; hi[((size_t)o * kFN + kr) * kFN + kc] = b;
            }
    }
    const long double scale = 1.0L / (kFN * kFN);
    for (int j = 0; j < 8; ++j)
        for (int r = 0; r < kFN; ++r)
            for (int c = 0; c < kFN; ++c) {
                const size_t f = (size_t)perm64(r) * kFN + perm64(c);
                const long double ar = hr[(size_t)(2 * j) * kFN * kFN + f], ai = -hi[(size_t)(2 * j) * kFN * kFN + f];          // conj
                const long double br = hr[(size_t)(2 * j + 1) * kFN * kFN + f], bi = -hi[(size_t)(2 * j + 1) * kFN * kFN + f];
                // conj(Ka) + i conj(Kb)
                out[(((size_t)j * kFN + r) * kFN + c) * 2] = (double)((ar - bi) * scale);
                out[(((size_t)j * kFN + r) * kFN + c) * 2 + 1] = (double)((ai + br) * scale);
            }
    return out;
}

constexpr size_t kFftLds = (size_t)kFN * kFS * sizeof(cd) + (size_t)2 * kFN * sizeof(unsigned long long) + (kFT / 64) * sizeof(float);

// Since the last call, on the current device: plane values of the FFT form [0] within the band of zero, [1] within the band of a float
// midpoint; [2] pixels formed again as direct sums because of them (diagnostic: tools/experiments/setup_content.py prints them).
bool gabor_fft_doubt(unsigned long long out[3]) {
    const unsigned long long zero[3] = {0, 0, 0};
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_doubt), sizeof(zero)) == hipSuccess && hipMemcpyToSymbol(HIP_SYMBOL(g_doubt), zero, sizeof(zero)) == hipSuccess;
}

constexpr size_t redo_lds_bytes(int ks) { return (size_t)2 * ks * 16 * sizeof(double) + (size_t)16 * ks * ks * sizeof(float); }

bool gabor_fft_prepare() {                                     // the twiddle table and the kernels' LDS limit on this device (idempotent)
    double w[2 * kFN];
    for (int m = 0; m < kFN; ++m) { w[2 * m] = (double)cosl(2.0L * M_PIl * m / kFN); w[2 * m + 1] = (double)-sinl(2.0L * M_PIl * m / kFN); }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_w64), w, sizeof(w)) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)k_gabor_fft<31>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFftLds) != hipSuccess) return false;
    if (hipFuncSetAttribute((const void*)k_gabor_fft<13>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFftLds) != hipSuccess) return false;
    return hipFuncSetAttribute((const void*)k_gabor_redo<31>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)redo_lds_bytes(31)) == hipSuccess;
}

// POPPY_GABOR_NO_REDO (measurement aid): the transform alone, as until round 4 — a plane value in ~1e8 then lands on the other neighbouring float
static bool gabor_redo_on() { static const bool on = poppy_experiment_env("POPPY_GABOR_NO_REDO") == nullptr; return on; }
// The width of the doubt band in units of max(1, the patch's largest magnitude): 1e-13 = ten times the largest distance between the two forms
// measured on ~1e5 plane values (1.1e-14).  The equality of the forms rests on that measurement, not on a proof: POPPY_GABOR_BAND widens the
// band for checks (1e30: every pixel is re-formed as direct sums — a fuzz run under it and one without must agree in every bit).
static double gabor_band_unit() { static const double v = poppy_experiment_env("POPPY_GABOR_BAND") ? atof(poppy_experiment_env("POPPY_GABOR_BAND")) : 1e-13; return v; }

// list: 2 + w * h * channels words, device (this launch resets and fills it; k_gabor_redo reads it)
void launch_gabor_fft31(const float* src, const double* d_tables, const double* d_bank, unsigned* d_list, float* dst, int w, int h, hipStream_t s) {
    constexpr int B = kFN - 31 + 1;
    const int tiles_x = (w + B - 1) / B, tiles_y = (h + B - 1) / B;
    if (!gabor_redo_on()) d_list = nullptr;
    if (d_list) (void)hipMemsetAsync(d_list, 0, 4, s);
    hipLaunchKernelGGL(k_gabor_fft<31>, dim3(tiles_x * tiles_y, 1), dim3(kFT), kFftLds, s, src, (const cd*)d_tables, d_list, dst, w, h, 1, tiles_x, gabor_band_unit());
    if (d_list) hipLaunchKernelGGL(k_gabor_redo<31>, dim3(512), dim3(256), redo_lds_bytes(31), s, src, d_bank, d_list, dst, w, h, 1);
}
void launch_gabor_fft13_c3(const float* src, const double* d_tables, const double* d_bank, unsigned* d_list, float* dst, int w, int h, hipStream_t s) {
    constexpr int B = kFN - 13 + 1;
    const int tiles_x = (w + B - 1) / B, tiles_y = (h + B - 1) / B;
    if (!gabor_redo_on()) d_list = nullptr;
    if (d_list) (void)hipMemsetAsync(d_list, 0, 4, s);
    hipLaunchKernelGGL(k_gabor_fft<13>, dim3(tiles_x * tiles_y, 3), dim3(kFT), kFftLds, s, src, (const cd*)d_tables, d_list, dst, w, h, 3, tiles_x, gabor_band_unit());
    if (d_list) hipLaunchKernelGGL(k_gabor_redo<13>, dim3(512), dim3(256), redo_lds_bytes(13), s, src, d_bank, d_list, dst, w, h, 3);
}

}  // namespace poppy_hip
