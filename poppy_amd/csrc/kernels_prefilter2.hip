// kernels_prefilter2.hip — pre-ORB filter chain, part 2 (src/extractor.cpp:33-83, src/poppy.hpp:119-122) on gfx950:
// the float stages between `goodFeatures` and the ORB input, the `gabor2` mask field, and dft_detail2.
//
// Exact stages (same expression trees as the reference's SSE3-baseline build, -ffp-contract=off):
//   u8 -> f32, unsharp_mask(sigma 2: 17-tap separable Gaussian, 3x3 median of the difference, threshold, amount),
//   BGR2GRAY on float, multiplies, f32 -> u8, equalizeHist, magnitude / log / min-max normalisation of the spectrum.
// The two Gabor banks: OpenCV evaluates filter2D with a 31x31 (13x13) float kernel through its DFT-based cross-correlation
// in DOUBLE precision and rounds to float once (OCV/imgproc/src/filter.dispatch.cpp:1291, templmatch.cpp:592); here the
// same once-rounded exact sum is accumulated directly in double (see k_gabor_bank).  dft_detail2's complex float DFT is
// cv::dft restated operation for operation (dft_c2c_forward), because its result is read back as raw bytes.
#include "kernels_prefilter.h"
#include <algorithm>
#include "pyramid_device.h"

namespace poppy_hip {

// ---- unsharp_mask(src, radius 2, amount 6, threshold 0.1) on ONE channel ----------------------------------------------
// The reference runs it on the 3-channel replication of a grey image (triple_channel): all three channels carry the
// same numbers through the same operations, so one channel is computed; the norm of the difference is the double
// sqrt(d*d + d*d + d*d) as for the Vec3f.  src/util.cpp:113-148, filter.simd.hpp:1625-1700,1884-1990.
__global__ void __launch_bounds__(256) k_u8_to_f32(const uint8_t* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i] * kInv255 + 0.f;
}
__global__ void __launch_bounds__(256) k_sep_row(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ taps, int ksize, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float* row = src + (size_t)y * W;
    const int r = ksize >> 1;
    float acc = row[reflect101(x - r, W)] * taps[0];
    for (int k = 1; k < ksize; ++k) acc = row[reflect101(x - r + k, W)] * taps[k] + acc;
    dst[(size_t)y * W + x] = acc;
}
__global__ void __launch_bounds__(256) k_sep_col_diff(const float* __restrict__ src, const float* __restrict__ tmp, float* __restrict__ diff,
                                                      const float* __restrict__ taps, int ksize, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int r = ksize >> 1;
    float acc = taps[r] * tmp[(size_t)y * W + x] + 0.f;
    for (int k = 1; k <= r; ++k)
        acc = taps[r + k] * (tmp[(size_t)reflect101(y + k, H) * W + x] + tmp[(size_t)reflect101(y - k, H) * W + x]) + acc;
    diff[(size_t)y * W + x] = src[(size_t)y * W + x] - acc;
}
__device__ __forceinline__ float med9(const float* __restrict__ d, int W, int H, int x, int y) {
    const int x0 = x > 0 ? x - 1 : x, x2 = x < W - 1 ? x + 1 : x;
    const float* r0 = d + (size_t)(y > 0 ? y - 1 : 0) * W;
    const float* r1 = d + (size_t)y * W;
    const float* r2 = d + (size_t)(y < H - 1 ? y + 1 : H - 1) * W;
    const float a0 = r0[x0], a1 = r0[x], a2 = r0[x2], b0 = r1[x0], b1 = r1[x], b2 = r1[x2], c0 = r2[x0], c1 = r2[x], c2 = r2[x2];
    const float lo = fmaxf(fmaxf(fminf(fminf(a0, b0), c0), fminf(fminf(a1, b1), c1)), fminf(fminf(a2, b2), c2));
    const float mi = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a0, b0, c0), __builtin_amdgcn_fmed3f(a1, b1, c1), __builtin_amdgcn_fmed3f(a2, b2, c2));
    const float hi = fminf(fminf(fmaxf(fmaxf(a0, b0), c0), fmaxf(fmaxf(a1, b1), c1)), fmaxf(fmaxf(a2, b2), c2));
    return __builtin_amdgcn_fmed3f(lo, mi, hi);
}
// out = grey of the unsharp-masked triple: v' = v + amount * d where |(d,d,d)| >= threshold, then v'*0.114 + v'*0.587 + v'*0.299
__global__ void __launch_bounds__(256) k_unsharp1_gray(const float* __restrict__ src, const float* __restrict__ diff, float* __restrict__ out,
                                                       int W, int H, float amount, float threshold) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    float d;
    if (W == 1 || H == 1) {            // 1-D special case of medianBlur (median_blur.simd.hpp:694-711)
        const int len = W + H - 1, i = (H == 1) ? x : y;
        float p0 = diff[i > 0 ? i - 1 : i], p1 = diff[i], p2 = diff[i < len - 1 ? i + 1 : i];
        d = __builtin_amdgcn_fmed3f(p0, p1, p2);
    } else d = med9(diff, W, H, x, y);
    const size_t p = (size_t)y * W + x;
    float v = src[p];
    const double n2 = (double)d * (double)d + (double)d * (double)d + (double)d * (double)d;
    if (sqrt(n2) >= (double)threshold) v = v + amount * d;
    out[p] = v * 0.299f + (v * 0.587f + v * 0.114f);           // v_fma(r, cr, v_fma(g, cg, b*cb)), color_rgb.simd.hpp:630,638
}
void launch_unsharp1_gray(const uint8_t* gf, float* f32, float* tmp, float* diff, float* out, const float* d_taps17, int w, int h, hipStream_t s) {
    const int n = w * h;
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_u8_to_f32, dim3((n + 255) / 256), dim3(256), 0, s, gf, f32, n);
    hipLaunchKernelGGL(k_sep_row, grid, dim3(256), 0, s, f32, tmp, d_taps17, 17, w, h);
    hipLaunchKernelGGL(k_sep_col_diff, grid, dim3(256), 0, s, f32, tmp, diff, d_taps17, 17, w, h);
    hipLaunchKernelGGL(k_unsharp1_gray, grid, dim3(256), 0, s, f32, diff, out, w, h, 6.f, 0.1f);
}

// ---- Gabor bank: mean over 16 orientations of clamp(filter2D(src, kernel_i), 0, 1) ------------------------------------
// For float images OpenCV's filter2D with a kernel this large runs crossCorr (OCV/imgproc/src/templmatch.cpp:566-760) with
// maxDepth = CV_64F: the correlation is evaluated through DOUBLE-precision DFTs and converted to float at the end, i.e. each
// plane is the exact sum rounded once to float (the double DFT's error, ~1e-13 relative, only matters if the exact sum sits
// within that distance of a float rounding boundary).  The same value is produced here directly: the products of two floats
// are exact in double, they are accumulated in double and rounded once.  Measured: bit-identical to the reference on every
// pixel of the fixtures.  Tile in LDS (reflect-101 borders); every source value is read once per tap position and feeds the
// 16 orientation accumulators; the taps (doubles, uniform across the wave) come in through scalar loads.
template <int K, int CN>
__global__ void __launch_bounds__(256) k_gabor_bank(const float* __restrict__ src, const double* __restrict__ bank, float* __restrict__ dst, int W, int H) {
    constexpr int R = K / 2, TX = 32, TY = 8, SX = TX + 2 * R, SY = TY + 2 * R;
    __shared__ float tile[SY * SX];
    const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY, ch = blockIdx.z;
    for (int i = threadIdx.x; i < SY * SX; i += 256) {
        const int r = i / SX, c = i - r * SX;
        tile[i] = src[((size_t)reflect101(ty0 - R + r, H) * W + reflect101(tx0 - R + c, W)) * CN + ch];
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    double acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.0;
    for (int dy = 0; dy < K; ++dy)
        for (int dx = 0; dx < K; ++dx) {
            const double v = (double)tile[(ly + dy) * SX + lx + dx];
            const double* wv = bank + (dy * K + dx) * 16;                             // [tap][orientation]: one 128-byte scalar fetch per tap
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fma(wv[o], v, acc[o]);              // the product is exact in double
        }
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    float sum = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) sum += fminf(fmaxf((float)acc[o], 0.f), 1.f);   // plane.setTo(1, plane > 1); setTo(0, plane < 0); dst += plane
    dst[((size_t)y * W + x) * CN + ch] = sum * 0.0625f;                            // dst /= 16
}
void launch_gabor_bank31(const float* src, const double* d_bank, float* dst, int w, int h, hipStream_t s) {
    hipLaunchKernelGGL((k_gabor_bank<31, 1>), dim3((w + 31) / 32, (h + 7) / 8, 1), dim3(256), 0, s, src, d_bank, dst, w, h);
}
void launch_gabor_bank13_c3(const float* src, const double* d_bank, float* dst, int w, int h, hipStream_t s) {
    hipLaunchKernelGGL((k_gabor_bank<13, 3>), dim3((w + 31) / 32, (h + 7) / 8, 3), dim3(256), 0, s, src, d_bank, dst, w, h);
}

// u8 BGR -> f32 BGR * (1/255)  (corrected2.convertTo(CV_32F, 1.0/255))
void launch_u8_to_f32(const uint8_t* src, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_u8_to_f32, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
}

// ---- g = gabor * us * radial -> u8 -> equalizeHist ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_orb_input(const float* __restrict__ gb, const float* __restrict__ us, const float* __restrict__ radial,
                                                   uint8_t* __restrict__ out, unsigned* __restrict__ hist, int n) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    for (int p = blockIdx.x * 256 + threadIdx.x; p < n; p += gridDim.x * 256) {
        float g = gb[p] * us[p];
        g = g * radial[p];
        const uint8_t o = sat_u8(cv_round_x86(g * 255.f + 0.f));
        out[p] = o;
        atomicAdd(&lh[o], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}
void launch_orb_input(const float* gb, const float* us, const float* radial, uint8_t* tmp_u8, unsigned* hist, uint8_t* lut, uint8_t* out,
                      int n_px, hipStream_t s) {
    (void)hipMemsetAsync(hist, 0, 256 * sizeof(unsigned), s);
    const int blocks = std::min((n_px + 255) / 256, 2048);
    hipLaunchKernelGGL(k_orb_input, dim3(blocks), dim3(256), 0, s, gb, us, radial, tmp_u8, hist, n_px);
    launch_equalize_from_hist(tmp_u8, hist, lut, out, n_px, s);
}

// ---- dft_detail2: spectrum post-processing (src/experiments.hpp:267-318) ----------------------------------------------------
// spec = interleaved complex DFT of the zero-padded image, M x N.  mag = log(sqrt(re^2 + im^2) + 1) and its min / max.
__device__ __forceinline__ float cv_log32f_dev(float x, const float* __restrict__ tab) {
    const float A0 = 0.3333333333333333333333333f, A1 = -0.5f, A2 = 1.f;
    const float ln2 = (float)0.69314718055994530941723212145818;
    const int i0 = __float_as_int(x);
    const float bf = __int_as_float((i0 & ((1 << 15) - 1)) | (127 << 23));
    const int idx = (i0 >> 14) & 510;
    const float y0 = (float)(((i0 >> 23) & 0xff) - 127) * ln2 + tab[idx];
    const float x0 = (bf - 1.f) * tab[idx + 1] + (idx == 510 ? -1.f / 512 : 0.f);
    return ((A0 * x0 + A1) * x0 + A2) * x0 + y0;
}
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// (also resets the two reductions of the spectrum kernels that follow it on the stream: min / max as ordered words, the sum of squares)
__global__ void __launch_bounds__(256) k_pad_complex(const uint8_t* __restrict__ src, float2* __restrict__ dst, int W, int H, int N, int M,
                                                     unsigned* __restrict__ minmax, unsigned long long* __restrict__ powsum) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x == 0 && y == 0) { minmax[0] = 0xffffffffu; minmax[1] = 0u; powsum[0] = 0ull; }
    if (x >= N) return;
    dst[(size_t)y * N + x] = make_float2((x < W && y < H) ? (float)src[(size_t)y * W + x] : 0.f, 0.f);
}
// Both spectrum kernels walk the image with a fixed, small grid (one row per step of a block) and reduce inside the block, so
// that the global atomics are a few hundred per launch: one per block.  (With a block per 256 pixels the 8100 blocks' atomics on
// one address WERE the kernels: 199 and 171 us at 1080p.)
constexpr int kSpectrumBlocks = 1024;

__global__ void __launch_bounds__(256) k_spectrum_log(const float2* __restrict__ spec, float* __restrict__ mag, const float* __restrict__ tab,
                                                      unsigned* __restrict__ minmax, int N, int M, int Nc, int Mc) {
    __shared__ float ltab[512];
    __shared__ unsigned smin, smax;
    ltab[threadIdx.x] = tab[threadIdx.x]; ltab[threadIdx.x + 256] = tab[threadIdx.x + 256];
    if (threadIdx.x == 0) { smin = 0xffffffffu; smax = 0; }
    __syncthreads();
    unsigned lo = 0xffffffffu, hi = 0;
    const int xb = (Nc + 255) / 256, jobs = xb * Mc;                       // a job = 256 consecutive pixels of one row
    for (int job = blockIdx.x; job < jobs; job += gridDim.x) {
        const int y = job / xb, x = (job - y * xb) * 256 + threadIdx.x;
        if (x < Nc) {                                             // the crop to even sizes happens before min / max
            const float2 c = spec[(size_t)y * N + x];
            // hal::magnitude32f = correctly rounded sqrt of the float sum; through double (53 >= 2*24 + 2 bits, so the second rounding
            // cannot change the result) because the float intrinsic is not correctly rounded on this target
            float m = (float)sqrt((double)(c.x * c.x + c.y * c.y));
            m = m + 1.f;
            m = cv_log32f_dev(m, ltab);
            mag[(size_t)y * Nc + x] = m;
            const unsigned o = f2ord(m);
            lo = min(lo, o); hi = max(hi, o);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (unsigned)__shfl_down(lo, o)); hi = max(hi, (unsigned)__shfl_down(hi, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMin(&smin, lo); atomicMax(&smax, hi); }
    __syncthreads();
    if (threadIdx.x == 0) { atomicMin(&minmax[0], smin); atomicMax(&minmax[1], smax); }
}
// sum over rows r and bytes b < Nc of (byte b of row r of the quadrant-swapped, min-max normalised image)^2
// cv::normalize(.., 0, 1, NORM_MINMAX) into CV_32F (OCV/core/src/norm.cpp:1384-1397): the scale is rounded to float first and the shift is built from that
// rounded scale, then convertTo(CV_32F, scale, shift).  Every thread forms the two from the min / max words of k_spectrum_log (the reference's double
// arithmetic: one correctly rounded division, one product, three roundings to float) — on the host until round 4, at the price of a round trip in mid-chain.
__global__ void __launch_bounds__(256) k_spectrum_bytes(const float* __restrict__ mag, const unsigned* __restrict__ minmax, int Nc, int Mc,
                                                        unsigned long long* __restrict__ powsum) {
    __shared__ unsigned long long ssum;
    const unsigned o0 = minmax[0], o1 = minmax[1];
    const double smin = (double)__uint_as_float((o0 & 0x80000000u) ? (o0 & 0x7fffffffu) : ~o0), smax = (double)__uint_as_float((o1 & 0x80000000u) ? (o1 & 0x7fffffffu) : ~o1);
    const double scale_d = (double)(float)((1.0 - 0.0) * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.));
    const float scale = (float)scale_d, shift = (float)((double)((float)0.0 - (float)(smin * scale_d)));
    if (threadIdx.x == 0) ssum = 0;
    __syncthreads();
    const int cx = Nc / 2, cy = Mc / 2;
    const int cols = (Nc + 3) / 4, xb = (cols + 255) / 256, jobs = xb * Mc;  // float column c < Nc / 4 (+ a partial one)
    unsigned long long s = 0;
    for (int job = blockIdx.x; job < jobs; job += gridDim.x) {
        const int r = job / xb, c = (job - r * xb) * 256 + threadIdx.x;
        if (4 * c < Nc) {
            const float v = mag[(size_t)((r + cy) % Mc) * Nc + (c + cx) % Nc] * scale + shift;
            const unsigned u = __float_as_uint(v);
            const int nb = min(4, Nc - 4 * c);
            for (int k = 0; k < nb; ++k) { const unsigned b = (u >> (8 * k)) & 255; s += b * b; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&ssum, s);
    __syncthreads();
    if (threadIdx.x == 0 && ssum) atomicAdd(powsum, ssum);
}
// ---- cv::dft's complex float transform, one thread per 1-D transform ------------------------------------------------------
// OCV/core/src/dxt.cpp:835-1190 with the SSE3 specialisations that run for float (radix 4: :645-727, radix 2 / 3 identical
// in value to the scalar forms).  The operation ORDER is the reference's: the permuted copy, radix-4 passes over the power
// of two, one radix-2 pass if the power is odd, then the odd factors (5s before 3s: DFTFactorize reverses them).  Each
// product and each sum rounds on its own (-ffp-contract=off).  A thread walks its transform sequentially in global
// memory: rows of the image in the first launch (elements contiguous), columns in the second (adjacent threads touch
// adjacent addresses).  This is the exactness path of dft_detail2, where the result is read back as raw float bytes.


#define DFT_AT(p, i) (p)[(size_t)(i) * stride]
__device__ void dft_c2c_forward(const float2* __restrict__ src, float2* __restrict__ dst, int stride, const DftPlanDev& c) {
    const float2* __restrict__ wave = c.wave;
    const int N = c.n;
    for (int i = 0; i < N; ++i) DFT_AT(dst, i) = DFT_AT(src, c.itab[i]);        // 0. shuffle data
    int n = 1, dw0 = N;
    if ((c.factors[0] & 1) == 0) {
        const int f0 = c.factors[0];
        for (; n * 4 <= f0;) {                                                   // radix-4 passes (DFT_VecR4<float>)
            const int nx = n;
            n *= 4; dw0 /= 4;
            for (int i = 0; i < N; i += n) {
                for (int j = 0, dw = 0; j < nx; ++j, dw += dw0) {
                    float2 x0 = DFT_AT(dst, i + j), x1 = DFT_AT(dst, i + j + nx), x2 = DFT_AT(dst, i + j + 2 * nx), x3 = DFT_AT(dst, i + j + 3 * nx);
                    if (j > 0) {
                        const float2 w1 = wave[dw], w2 = wave[dw * 2], w3 = wave[dw * 3];
                        const float2 a = x1, b = x2, d = x3;
                        x1 = make_float2(a.x * w2.x - a.y * w2.y, a.x * w2.y + a.y * w2.x);
                        x3 = make_float2(d.x * w3.x - d.y * w3.y, d.x * w3.y + d.y * w3.x);
                        x2 = make_float2(b.x * w1.x - b.y * w1.y, b.x * w1.y + b.y * w1.x);
                    }
                    const float2 s01 = make_float2(x0.x + x1.x, x0.y + x1.y), s23 = make_float2(x2.x + x3.x, x2.y + x3.y);
                    const float2 d01 = make_float2(x0.x - x1.x, x0.y - x1.y), d23 = make_float2(x2.x - x3.x, x2.y - x3.y);
                    DFT_AT(dst, i + j) = make_float2(s01.x + s23.x, s01.y + s23.y);
                    DFT_AT(dst, i + j + nx) = make_float2(d01.x + d23.y, d01.y - d23.x);
                    DFT_AT(dst, i + j + 2 * nx) = make_float2(s01.x - s23.x, s01.y - s23.y);
                    DFT_AT(dst, i + j + 3 * nx) = make_float2(d01.x - d23.y, d01.y + d23.x);
                }
            }
        }
        for (; n < f0;) {                                                        // the remaining radix-2 pass
            n *= 2; dw0 /= 2;
            const int nx = n / 2;
            for (int i = 0; i < N; i += n)
                for (int j = 0, dw = 0; j < nx; ++j, dw += dw0) {
                    const float2 v0 = DFT_AT(dst, i + j), vn = DFT_AT(dst, i + j + nx);
                    float2 x1 = vn;
                    if (j > 0) { const float2 w = wave[dw]; x1 = make_float2(vn.x * w.x - vn.y * w.y, vn.y * w.x + vn.x * w.y); }
                    DFT_AT(dst, i + j) = make_float2(v0.x + x1.x, v0.y + x1.y);
                    DFT_AT(dst, i + j + nx) = make_float2(v0.x - x1.x, v0.y - x1.y);
                }
        }
    }
    for (int f_idx = (c.factors[0] & 1) ? 0 : 1; f_idx < c.nf; ++f_idx) {        // 2. odd factors
        const int factor = c.factors[f_idx];
        const int nx = n;
        n *= factor; dw0 /= factor;
        if (factor == 3) {
            const float sin_120 = (float)0.86602540378443864676372317075294;
            for (int i = 0; i < N; i += n)
                for (int j = 0, dw = 0; j < nx; ++j, dw += dw0) {
                    const float2 v0 = DFT_AT(dst, i + j), a = DFT_AT(dst, i + j + nx), b = DFT_AT(dst, i + j + 2 * nx);
                    float r0, i0, r1, i1, r2, i2;
                    if (j == 0) {
                        r1 = a.x + b.x; i1 = a.y + b.y;
                        r2 = sin_120 * (a.y - b.y); i2 = sin_120 * (b.x - a.x);
                    } else {
                        const float2 w1 = wave[dw], w2 = wave[dw * 2];
                        const float ar = a.x * w1.x - a.y * w1.y, ai = a.x * w1.y + a.y * w1.x;
                        const float br = b.x * w2.x - b.y * w2.y, bi = b.x * w2.y + b.y * w2.x;
                        r1 = ar + br; i1 = ai + bi;
                        r2 = sin_120 * (ai - bi); i2 = sin_120 * (br - ar);
                    }
                    r0 = v0.x; i0 = v0.y;
                    DFT_AT(dst, i + j) = make_float2(r0 + r1, i0 + i1);
                    r0 -= 0.5f * r1; i0 -= 0.5f * i1;
                    DFT_AT(dst, i + j + nx) = make_float2(r0 + r2, i0 + i2);
                    DFT_AT(dst, i + j + 2 * nx) = make_float2(r0 - r2, i0 - i2);
                }
        } else if (factor == 5) {
            const float fft5_2 = (float)0.559016994374947424102293417182819, fft5_3 = (float)-0.951056516295153572116439333379382;
            const float fft5_4 = (float)-1.538841768587626701285145288018455, fft5_5 = (float)0.363271264002680442947733378740309;
            for (int i = 0; i < N; i += n)
                for (int j = 0, dw = 0; j < nx; ++j, dw += dw0) {
                    const float2 a0 = DFT_AT(dst, i + j), a1 = DFT_AT(dst, i + j + nx), a2 = DFT_AT(dst, i + j + 2 * nx),
                                 a3 = DFT_AT(dst, i + j + 3 * nx), a4 = DFT_AT(dst, i + j + 4 * nx);
                    const float2 w1 = wave[dw], w2 = wave[dw * 2], w3 = wave[dw * 3], w4 = wave[dw * 4];
                    float r0, i0, r1, i1, r2, i2, r3, i3, r4, i4, r5, i5;
                    r3 = a1.x * w1.x - a1.y * w1.y; i3 = a1.x * w1.y + a1.y * w1.x;
                    r2 = a4.x * w4.x - a4.y * w4.y; i2 = a4.x * w4.y + a4.y * w4.x;
                    r1 = r3 + r2; i1 = i3 + i2;
                    r3 -= r2; i3 -= i2;
                    r4 = a3.x * w3.x - a3.y * w3.y; i4 = a3.x * w3.y + a3.y * w3.x;
                    r0 = a2.x * w2.x - a2.y * w2.y; i0 = a2.x * w2.y + a2.y * w2.x;
                    r2 = r4 + r0; i2 = i4 + i0;
                    r4 -= r0; i4 -= i0;
                    r0 = a0.x; i0 = a0.y;
                    r5 = r1 + r2; i5 = i1 + i2;
                    DFT_AT(dst, i + j) = make_float2(r0 + r5, i0 + i5);
                    r0 -= 0.25f * r5; i0 -= 0.25f * i5;
                    r1 = fft5_2 * (r1 - r2); i1 = fft5_2 * (i1 - i2);
                    r2 = -fft5_3 * (i3 + i4); i2 = fft5_3 * (r3 + r4);
                    i3 *= -fft5_5; r3 *= fft5_5;
                    i4 *= -fft5_4; r4 *= fft5_4;
                    r5 = r2 + i3; i5 = i2 + r3;
                    r2 -= i4; i2 -= r4;
                    r3 = r0 + r1; i3 = i0 + i1;
                    r0 -= r1; i0 -= i1;
                    DFT_AT(dst, i + j + nx) = make_float2(r3 + r2, i3 + i2);
                    DFT_AT(dst, i + j + 4 * nx) = make_float2(r3 - r2, i3 - i2);
                    DFT_AT(dst, i + j + 2 * nx) = make_float2(r0 + r5, i0 + i5);
                    DFT_AT(dst, i + j + 3 * nx) = make_float2(r0 - r5, i0 - i5);
                }
        }
    }
}
#undef DFT_AT

__global__ void __launch_bounds__(64) k_dft_lines(const float2* __restrict__ src, float2* __restrict__ dst, int count, int line_pitch,
                                                  int stride, DftPlanDev plan) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= count) return;
    dft_c2c_forward(src + (size_t)t * line_pitch, dst + (size_t)t * line_pitch, stride, plan);
}
// The same transform with one WORKGROUP per line: the line sits in LDS and the butterflies of a pass — which are independent
// of each other — are spread over the threads, with a barrier between passes.  Every butterfly is the same expression as
// in dft_c2c_forward, so the result is bit-identical; only the order in which independent butterflies run differs.
constexpr int kDftMaxN = 4096;

__device__ __forceinline__ float2 cmul_f(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

__global__ void __launch_bounds__(256) k_dft_line_wg(const float2* __restrict__ src, float2* __restrict__ dst, int line_pitch, int stride, DftPlanDev c) {
    extern __shared__ float2 v[];
    const float2* __restrict__ wave = c.wave;
    const int N = c.n, tid = threadIdx.x;
    const float2* s = src + (size_t)blockIdx.x * line_pitch;
    float2* d = dst + (size_t)blockIdx.x * line_pitch;
    for (int i = tid; i < N; i += 256) v[i] = s[(size_t)c.itab[i] * stride];
    __syncthreads();
    int n = 1, dw0 = N;
    if ((c.factors[0] & 1) == 0) {
        const int f0 = c.factors[0];
        for (; n * 4 <= f0;) {
            const int nx = n;
            n *= 4; dw0 /= 4;
            for (int b = tid; b < N / 4; b += 256) {
                const int blk = b / nx, j = b - blk * nx, p = blk * n + j, dw = j * dw0;
                float2 x0 = v[p], x1 = v[p + nx], x2 = v[p + 2 * nx], x3 = v[p + 3 * nx];
                if (j > 0) { x1 = cmul_f(x1, wave[dw * 2]); x3 = cmul_f(x3, wave[dw * 3]); x2 = cmul_f(x2, wave[dw]); }
                const float2 s01 = make_float2(x0.x + x1.x, x0.y + x1.y), s23 = make_float2(x2.x + x3.x, x2.y + x3.y);
                const float2 d01 = make_float2(x0.x - x1.x, x0.y - x1.y), d23 = make_float2(x2.x - x3.x, x2.y - x3.y);
                v[p] = make_float2(s01.x + s23.x, s01.y + s23.y);
                v[p + nx] = make_float2(d01.x + d23.y, d01.y - d23.x);
                v[p + 2 * nx] = make_float2(s01.x - s23.x, s01.y - s23.y);
                v[p + 3 * nx] = make_float2(d01.x - d23.y, d01.y + d23.x);
            }
            __syncthreads();
        }
        for (; n < f0;) {
            n *= 2; dw0 /= 2;
            const int nx = n / 2;
            for (int b = tid; b < N / 2; b += 256) {
                const int blk = b / nx, j = b - blk * nx, p = blk * n + j;
                const float2 v0 = v[p], vn = v[p + nx];
                float2 x1 = vn;
                if (j > 0) { const float2 w = wave[j * dw0]; x1 = make_float2(vn.x * w.x - vn.y * w.y, vn.y * w.x + vn.x * w.y); }
                v[p] = make_float2(v0.x + x1.x, v0.y + x1.y);
                v[p + nx] = make_float2(v0.x - x1.x, v0.y - x1.y);
            }
            __syncthreads();
        }
    }
    for (int f_idx = (c.factors[0] & 1) ? 0 : 1; f_idx < c.nf; ++f_idx) {
        const int factor = c.factors[f_idx];
        const int nx = n;
        n *= factor; dw0 /= factor;
        if (factor == 3) {
            const float sin_120 = (float)0.86602540378443864676372317075294;
            for (int b = tid; b < N / 3; b += 256) {
                const int blk = b / nx, j = b - blk * nx, p = blk * n + j, dw = j * dw0;
                const float2 v0 = v[p], a = v[p + nx], bb = v[p + 2 * nx];
                float r0, i0, r1, i1, r2, i2;
                if (j == 0) {
                    r1 = a.x + bb.x; i1 = a.y + bb.y;
                    r2 = sin_120 * (a.y - bb.y); i2 = sin_120 * (bb.x - a.x);
                } else {
                    const float2 aw = cmul_f(a, wave[dw]), bw = cmul_f(bb, wave[dw * 2]);
                    r1 = aw.x + bw.x; i1 = aw.y + bw.y;
                    r2 = sin_120 * (aw.y - bw.y); i2 = sin_120 * (bw.x - aw.x);
                }
                r0 = v0.x; i0 = v0.y;
                v[p] = make_float2(r0 + r1, i0 + i1);
                r0 -= 0.5f * r1; i0 -= 0.5f * i1;
                v[p + nx] = make_float2(r0 + r2, i0 + i2);
                v[p + 2 * nx] = make_float2(r0 - r2, i0 - i2);
            }
        } else if (factor == 5) {
            const float fft5_2 = (float)0.559016994374947424102293417182819, fft5_3 = (float)-0.951056516295153572116439333379382;
            const float fft5_4 = (float)-1.538841768587626701285145288018455, fft5_5 = (float)0.363271264002680442947733378740309;
            for (int b = tid; b < N / 5; b += 256) {
                const int blk = b / nx, j = b - blk * nx, p = blk * n + j, dw = j * dw0;
                const float2 a0 = v[p], a1 = v[p + nx], a2 = v[p + 2 * nx], a3 = v[p + 3 * nx], a4 = v[p + 4 * nx];
                const float2 w1 = wave[dw], w2 = wave[dw * 2], w3 = wave[dw * 3], w4 = wave[dw * 4];
                float r0, i0, r1, i1, r2, i2, r3, i3, r4, i4, r5, i5;
                r3 = a1.x * w1.x - a1.y * w1.y; i3 = a1.x * w1.y + a1.y * w1.x;
                r2 = a4.x * w4.x - a4.y * w4.y; i2 = a4.x * w4.y + a4.y * w4.x;
                r1 = r3 + r2; i1 = i3 + i2;
                r3 -= r2; i3 -= i2;
                r4 = a3.x * w3.x - a3.y * w3.y; i4 = a3.x * w3.y + a3.y * w3.x;
                r0 = a2.x * w2.x - a2.y * w2.y; i0 = a2.x * w2.y + a2.y * w2.x;
                r2 = r4 + r0; i2 = i4 + i0;
                r4 -= r0; i4 -= i0;
                r0 = a0.x; i0 = a0.y;
                r5 = r1 + r2; i5 = i1 + i2;
                v[p] = make_float2(r0 + r5, i0 + i5);
                r0 -= 0.25f * r5; i0 -= 0.25f * i5;
                r1 = fft5_2 * (r1 - r2); i1 = fft5_2 * (i1 - i2);
                r2 = -fft5_3 * (i3 + i4); i2 = fft5_3 * (r3 + r4);
                i3 *= -fft5_5; r3 *= fft5_5;
                i4 *= -fft5_4; r4 *= fft5_4;
                r5 = r2 + i3; i5 = i2 + r3;
                r2 -= i4; i2 -= r4;
                r3 = r0 + r1; i3 = i0 + i1;
                r0 -= r1; i0 -= i1;
                v[p + nx] = make_float2(r3 + r2, i3 + i2);
                v[p + 4 * nx] = make_float2(r3 - r2, i3 - i2);
                v[p + 2 * nx] = make_float2(r0 + r5, i0 + i5);
                v[p + 3 * nx] = make_float2(r0 - r5, i0 - i5);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < N; i += 256) d[(size_t)i * stride] = v[i];
}

// 2-D forward transform of an m x n complex image: rows (src -> tmp), then columns (tmp -> dst)
void launch_dft2d_exact(const float2* src, float2* tmp, float2* dst, int n, int m, const DftPlanDev& rows, const DftPlanDev& cols, hipStream_t s) {
    if (n <= kDftMaxN && m <= kDftMaxN) {
        hipLaunchKernelGGL(k_dft_line_wg, dim3(m), dim3(256), (size_t)n * 8, s, src, tmp, n, 1, rows);
        hipLaunchKernelGGL(k_dft_line_wg, dim3(n), dim3(256), (size_t)m * 8, s, tmp, dst, 1, n, cols);
        return;
    }
    hipLaunchKernelGGL(k_dft_lines, dim3((m + 63) / 64), dim3(64), 0, s, src, tmp, m, n, 1, rows);
    hipLaunchKernelGGL(k_dft_lines, dim3((n + 63) / 64), dim3(64), 0, s, tmp, dst, n, 1, n, cols);
}

void launch_pad_complex(const uint8_t* src, float2* dst, int w, int h, int n, int m, unsigned* minmax, unsigned long long* powsum, hipStream_t s) {
    hipLaunchKernelGGL(k_pad_complex, dim3((n + 255) / 256, m), dim3(256), 0, s, src, dst, w, h, n, m, minmax, powsum);
}
void launch_spectrum_log(const float2* spec, float* mag, const float* d_logtab, unsigned* minmax, int n, int m, int nc, int mc, hipStream_t s) {
    hipLaunchKernelGGL(k_spectrum_log, dim3(std::min(kSpectrumBlocks, ((nc + 255) / 256) * mc)), dim3(256), 0, s, spec, mag, d_logtab, minmax, n, m, nc, mc);
}
void launch_spectrum_bytes(const float* mag, const unsigned* minmax, int nc, int mc, unsigned long long* powsum, hipStream_t s) {
    hipLaunchKernelGGL(k_spectrum_bytes, dim3(std::min(kSpectrumBlocks, (((nc + 3) / 4 + 255) / 256) * mc)), dim3(256), 0, s, mag, minmax, nc, mc, powsum);
}

}  // namespace poppy_hip
