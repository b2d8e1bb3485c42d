// kernels_align.hip — cv::warpAffine(u8x3, INTER_LINEAR, BORDER_CONSTANT 0) for the auto-align steps (auto_align.h).
// OCV/imgproc/src/imgwarp.cpp:2155-2290 (WarpAffineInvoker: 10-bit fixed-point coordinates, 5-bit fractions) feeding
// remapBilinear (:721-731,808-852).  The per-column / per-row coordinate terms are the reference's saturate_cast<int> of
// double products; the host computes them with the same libm-free double arithmetic and uploads 2 * (w + h) ints, the
// kernel is integer only: memory-bound gather, one thread per pixel.
#include "auto_align.h"
#include "warp_device.h"
#include <cmath>
#include <vector>

namespace poppy_hip {

__global__ void __launch_bounds__(256) k_warp_affine(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H,
                                                     const int* __restrict__ adelta, const int* __restrict__ bdelta,
                                                     const int* __restrict__ x0row, const int* __restrict__ y0row) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int X = (int)((unsigned)x0row[y] + (unsigned)adelta[x]) >> 5;      // the SIMD adds wrap
    const int Y = (int)((unsigned)y0row[y] + (unsigned)bdelta[x]) >> 5;
    const int ix = max(-32768, min(32767, X >> 5)), iy = max(-32768, min(32767, Y >> 5));
    uint8_t o[3];
    sample3_fixed(src, W, H, ix, iy, X & 31, Y & 31, o);
    uint8_t* d = dst + ((size_t)y * W + x) * 3;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

static int round_x86(double v) {                  // cvRound(double) = cvtsd2si: nearest even, "indefinite" when out of range
    const double r = std::nearbyint(v);
    return (r >= -2147483648.0 && r <= 2147483647.0) ? (int)r : INT_MIN;
}

bool warp_affine_device(const uint8_t* d_src, uint8_t* d_dst, int w, int h, const double Mfwd[6], int* d_tables, hipStream_t s) {
    double M[6] = {Mfwd[0], Mfwd[1], Mfwd[2], Mfwd[3], Mfwd[4], Mfwd[5]};
    {   // dst -> src map (imgwarp.cpp:2622-2631)
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5];
        const double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    std::vector<int> t((size_t)2 * (w + h));
    int *ad = t.data(), *bd = ad + w, *x0 = bd + w, *y0 = x0 + h;
    for (int x = 0; x < w; ++x) { ad[x] = round_x86(M[0] * x * 1024); bd[x] = round_x86(M[3] * x * 1024); }
    for (int y = 0; y < h; ++y) {
        x0[y] = (int)((unsigned)round_x86((M[1] * y + M[2]) * 1024) + 16u);
        y0[y] = (int)((unsigned)round_x86((M[4] * y + M[5]) * 1024) + 16u);
    }
    // the table is tiny and the caller waits for each step's result anyway: a plain synchronous copy keeps the staging buffer simple
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    if (hipMemcpy(d_tables, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return false;
    hipLaunchKernelGGL(k_warp_affine, dim3((w + 255) / 256, h), dim3(256), 0, s, d_src, d_dst, w, h,
                       d_tables, d_tables + w, d_tables + 2 * w, d_tables + 2 * w + h);
    return hipGetLastError() == hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------
// Candidate scoring for the auto-align searches (Transformer::rerotate tries 1080 angles, retranslate a handful of shifts;
// src/transformer.cpp:99-217): the O(N^2) parts of morph_distance (src/util.cpp:351-431) for many candidate versions of
// the second point set at once.  The host prepares the candidate sets (its libm defines the rotations).
//   candidate_pairs   : one wave per candidate.  make_distance_map's greedy claims (every first-set point, in list order,
//                       takes the nearest unclaimed candidate point; strict <, so the lowest index wins ties), then the pairs'
//                       distances summed in float in ascending (distance, insertion) order, as the multimap is walked.
//                       hypotf(x, y) = (float)sqrt((double)x*x + (double)y*y): glibc's formula (checked bit for bit on 4e8
//                       inputs, tools/micro/hypotf_check.c), IEEE double multiply / add / sqrt on the device.
//   candidate_inner   : one wave per candidate: the float sum over all (i, j) of (set[i].x - p1[j].x) + (set[i].y - p1[j].y),
//                       one chain of additions in that order; the lanes prepare 64 terms at a time.
// The hull areas (Sklansky scan, Douglas-Peucker) and the long-double combination stay on the host (auto_align.cpp).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hypotf_glibc(float x, float y) { return (float)sqrt((double)x * (double)x + (double)y * (double)y); }

__device__ __forceinline__ void candidate_pairs(unsigned char* lds, int a, const float2* __restrict__ p1, const float2* __restrict__ sets, int n,
                                                float* __restrict__ total, int* __restrict__ n_pairs) {
    float2* pool = (float2*)lds;                    // the candidate set; a claimed entry becomes (-1,-1), the reference's tombstone
    float* dist = (float*)(pool + n);               // distance of pair i, < 0 where point i found nothing
    float* sorted = dist + n;
    const int lane = threadIdx.x;
    for (int j = lane; j < n; j += 64) pool[j] = sets[(size_t)a * n + j];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int i = 0; i < n; ++i) {
        const float2 q = p1[i];
        unsigned long long best = ~0ull;            // (distance bits, index): distances are >= 0, so the bit pattern orders them
        for (int j = lane; j < n; j += 64) {
            const float2 c = pool[j];
            if (c.x == -1.f && c.y == -1.f) continue;
            const float d = hypotf_glibc(c.x - q.x, c.y - q.y);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, off), hi = __shfl_xor((unsigned)(best >> 32), off);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o < best ? o : best;
        }
        if (lane == 0) {
            if (best != ~0ull) { dist[i] = __uint_as_float((unsigned)(best >> 32)); pool[(unsigned)best] = make_float2(-1.f, -1.f); }
            else dist[i] = -1.f;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    // ascending distance, insertion order among equals: rank by counting
    int count = 0;
    for (int i = lane; i < n; i += 64) {
        const float d = dist[i];
        if (d < 0.f) continue;
        int rank = 0;
        for (int k = 0; k < n; ++k) {
            const float e = dist[k];
            rank += (e >= 0.f && (e < d || (e == d && k < i))) ? 1 : 0;
        }
        sorted[rank] = d;
        ++count;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (lane == 0) {
        float t = 0;
        for (int k = 0; k < count; ++k) t += sorted[k];
        total[a] = t;
        n_pairs[a] = count;
    }
}

// One wave per candidate.  The 64 lanes form the terms of 64 consecutive j in parallel; the accumulation itself has to stay one
// chain of float additions in (i, j) order, so the terms are handed to it one lane after the other (v_readlane + v_add).
__device__ __forceinline__ void candidate_inner(unsigned char* lds, int a, const float2* __restrict__ p1, const float2* __restrict__ sets, int n,
                                                float* __restrict__ inner) {
    float2* first = (float2*)lds;
    const int lane = threadIdx.x;
    const float2* mine = sets + (size_t)a * n;
    for (int j = lane; j < n; j += 64) first[j] = p1[j];
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float acc = 0;
    const int full = n & ~63;
    for (int i = 0; i < n; ++i) {
        const float2 r = mine[i];
        int jb = 0;
        for (; jb < full; jb += 64) {
            const float2 q = first[jb + lane];
            const float term = (r.x - q.x) + (r.y - q.y);
#pragma unroll
            for (int k = 0; k < 64; ++k) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), k));
        }
        if (jb < n) {
            const float2 q = first[min(jb + lane, n - 1)];
            const float term = (r.x - q.x) + (r.y - q.y);
            for (int k = 0; k < n - jb; ++k) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, term), k));
        }
    }
    if (lane == 0) inner[a] = acc;
}

// both kinds of wave in one launch (even blocks pair, odd blocks sum), so that they run side by side
__global__ void __launch_bounds__(64) k_candidate_scores(const float2* __restrict__ p1, const float2* __restrict__ sets, int n,
                                                         float* __restrict__ total, int* __restrict__ n_pairs, float* __restrict__ inner) {
    extern __shared__ unsigned char lds[];
    const int a = blockIdx.x >> 1;
    if (blockIdx.x & 1) candidate_inner(lds, a, p1, sets, n, inner);
    else candidate_pairs(lds, a, p1, sets, n, total, n_pairs);
}

void launch_candidate_scores(const float* d_p1, const float* d_sets, int n, int n_cand, float* d_total, int* d_npairs, float* d_inner, hipStream_t s) {
    hipLaunchKernelGGL(k_candidate_scores, dim3(2 * n_cand), dim3(64), (size_t)n * 16, s, (const float2*)d_p1, (const float2*)d_sets, n,
                       d_total, d_npairs, d_inner);
}

}  // namespace poppy_hip
