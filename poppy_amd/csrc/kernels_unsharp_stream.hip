// kernels_unsharp_stream.hip — unsharp_mask(lapBlend, 1, amount, 0.3) + convertTo(CV_8U, 255) as a streaming kernel.
//
//   src/util.cpp:113-148 (unsharp_mask), src/algo.cpp:263-265 (the call and the 8-bit conversion)
//   row pass   : s = x0*k0; s = xk*kk + s                      OCV/imgproc/src/filter.simd.hpp:1682-1730,2477-2487
//   column pass: s = ky0*c + 0; s = kyk*(S[+k] + S[-k]) + s    OCV/imgproc/src/filter.simd.hpp:2753-2759
//   median 3x3 : the 5th of nine, replicated edges             OCV/imgproc/src/median_blur.simd.hpp:692-713
//   norm       : sqrt of a double sum of squares               OCV/core/include/opencv2/core/matx.hpp:929-932
//
// One WAVE owns a column strip of 60 pixels (lanes 1..60; lanes 0 and 61 carry the median's left / right neighbour column) and
// walks down a segment of rows.  Per row it stages the source row (72 pixels) in a wave-private LDS ring — the only LDS traffic:
// 3 floats written, 27 read per pixel —, runs the 9-tap row pass for its pixel, keeps the last nine row-pass rows in REGISTERS
// (the loop is unrolled nine times so that the ring's indices are compile-time), forms the column pass, the difference and its
// last three rows in registers, exchanges the sorted difference columns with the neighbouring lanes by DPP wave shifts, and
// stores bytes.  No barrier anywhere (a wave's LDS operations are ordered), no halo in the column direction beyond the ten
// rows a segment needs to warm up, a 72 / 60 halo in the row direction.  The tile kernel this replaces (k_unsharp_tile: 32 x 16
// tiles, halo 1.73 in the row pass, four barrier-separated phases through ~26 KB of LDS) was bound by its LDS traffic and
// barriers, not by arithmetic (profiles/r03_notes.md).
//
// Every value is the same expression tree as in k_unsharp_tile / the three-kernel form (kernels_frame.hip).
#include "kernels.h"
#include "pyramid_device.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>

// Source rows requested POPPY_US_AHEAD steps ahead (1, 3 or 9) and the waves per SIMD the registers must leave room for.  Round 3 ran 3 / 4 (104 VGPRs);
// round 4: one step ahead needs 88 VGPRs = FIVE waves per SIMD — the 5120 waves of a 4K frame (64 strips x 80 segments of 27 rows) are then resident at once —
// and the fifth wave hides what the shorter prefetch exposes: 4K 64.4-67.6 -> 60.6-61.0 us on one box (tools/experiments/ab_lib.sh).
#ifndef POPPY_US_AHEAD
#define POPPY_US_AHEAD 1
#endif
#ifndef POPPY_US_WAVES
#define POPPY_US_WAVES 5
#endif

namespace poppy_hip {

namespace {

constexpr int kStripPx = 60;                       // pixels a wave produces per row
constexpr int kRowPx = 72;                         // pixels staged per row: image columns X0 - 5 .. X0 + 66
constexpr int kRowFloats = kRowPx * 3;
constexpr int kRingRows = 8;                       // source rows kept in LDS (the apply step reads the row staged 5 steps ago)
constexpr int kWaveLds = kRingRows * kRowFloats;   // floats per wave

// getGaussianKernel(9, 1) as float (smooth.dispatch.cpp:200-221), as literals: a VOP2 multiply by a literal issues at full rate
constexpr float kG0 = 0x1.18a9c4p-13f, kG1 = 0x1.22724cp-8f, kG2 = 0x1.ba4b9ap-5f, kG3 = 0x1.ef8ebap-3f, kG4 = 0x1.9884a4p-2f;

typedef float f3 __attribute__((ext_vector_type(3)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));

__device__ __forceinline__ int reflect_clamp(int p, int len) {        // reflect-101 once, then into range (far halo columns / rows only)
    p = p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p);
    return p < 0 ? 0 : (p >= len ? len - 1 : p);
}
// lane l takes lane l - 1's / lane l + 1's value; the wave's first / last lane keeps its own
__device__ __forceinline__ float from_prev_lane(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x138, 0xf, 0xf, false));     // wave_shr:1
}
__device__ __forceinline__ float from_next_lane(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x130, 0xf, 0xf, false));     // wave_shl:1
}
// max / min over (lane - 1, lane, lane + 1) of one register: two VOP2 instructions with a DPP source instead of two DPP moves + a
// three-input instruction (every one of them issues at half rate: profiles/r03_notes.md section 1).  The wave's end lanes read 0 for
// the missing neighbour (bound_ctrl): they are halo lanes whose results are never stored.
__device__ __forceinline__ float max_with_neighbours(float v) {
    float t, r;
    // s_nop: a DPP read of a VGPR needs two wait states after the VALU write; the compiler does not see into the asm text
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(t) : "v"(v));
    asm("v_max_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r) : "v"(v), "v"(t));
    return r;
}
__device__ __forceinline__ float min_with_neighbours(float v) {
    float t, r;
    asm("s_nop 1\n\tv_min_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(t) : "v"(v));
    asm("v_min_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r) : "v"(v), "v"(t));
    return r;
}
__device__ __forceinline__ float min3s(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3s(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float med3s(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

}  // namespace

__global__ void __launch_bounds__(256, POPPY_US_WAVES) k_unsharp_stream(const float* __restrict__ src, uint8_t* __restrict__ out, float* __restrict__ outF,
                                                        int W, int H, int SP, int seg_rows, int n_strips, int blocks_x,
                                                        float amount_arg, const float* __restrict__ amount_ptr, double norm2_min, int stagger) {
    // SP: pixels per row of src (the blended level 0: PyrLevel::pitch, a multiple of 4 or W itself); the frame and outF go out tight, W pixels per row.
    // Any width (round 6): a width that is no multiple of 4 leaves a last group of fewer than four pixels per row — those go out byte by byte — and
    // rows that begin on any byte — the other groups' dword stores are then unaligned, which global stores take (3838 x 2160: 358 -> ~330 us per frame,
    // it ran on the tile kernel before)
    __shared__ float lds[4 * kWaveLds];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stagger_priority(blockIdx.x, stagger);
    const int blk = xcd_swizzle(blockIdx.x, gridDim.x);
    const int by = blk / blocks_x, bx = blk - by * blocks_x;
    const int strip = bx * 4 + wv;
    if (strip >= n_strips) return;                                  // no barriers below: a wave may leave on its own
    const float amount = amount_ptr ? *amount_ptr : amount_arg;     // per-frame value kept in HBM when the launch is a graph node
    float* const ring = lds + wv * kWaveLds;
    const int X0 = strip * kStripPx, Y0 = by * seg_rows;
    const int Yend = min(Y0 + seg_rows, H);
    const int n_steps = (Yend - Y0) + 10;
    // the two staged columns of this lane (reflect-101 at the image edge happens HERE, so the row pass has no edge cases)
    // (byte offsets into a source row; lanes 8..63 have no second column: their offset is out of the buffer's range, which a buffer
    // load answers with zeros without touching memory — no branch around the load, so the compiler can count the loads in flight)
    const uint32_t ca = (uint32_t)reflect_clamp(X0 - 5 + lane, W) * 12u;
    const uint32_t cb = lane < 8 ? (uint32_t)reflect_clamp(X0 - 5 + 64 + lane, W) * 12u : 0xfffffff0u;
    // this lane's pixel, clamped into the image: a lane beyond the edge repeats the edge pixel, which is the median's replicated border
    const int xl = min(max(X0 - 1 + lane, 0), W - 1);
    const int wl = (xl - X0 + 1) * 3;                               // window start in a staged row: column xl - 4
    const bool stores = lane >= 1 && lane <= kStripPx;
    const int j4 = (lane - 1) & 3;                                   // position in the group of four pixels that becomes three dwords
    const int xg = X0 + ((lane - 1) & ~3);                           // first pixel of that group
    const bool st_dw = stores && j4 < 3 && xg + 3 < W;              // a whole group of four pixels: three dwords
    const bool st_px = stores && xg + 3 >= W && X0 - 1 + lane < W;  // the row's last, shorter group: every lane its own three bytes

    const float t2f = (float)norm2_min;
    float R[9][3], D[3][3];
#pragma unroll
    for (int a = 0; a < 9; ++a) { R[a][0] = R[a][1] = R[a][2] = 0.f; }
#pragma unroll
    for (int a = 0; a < 3; ++a) { D[a][0] = D[a][1] = D[a][2] = 0.f; }
    const size_t pitch = (size_t)W * 3, spitch = (size_t)SP * 3;
    // source rows are requested kAhead steps before they are staged: a step is ~400 issue cycles per wave, memory is 1-2 us away
    constexpr int kAhead = POPPY_US_AHEAD;
    static_assert(9 % kAhead == 0, "the request ring is indexed by the step number modulo 9");
    u3 pa[kAhead], pb[kAhead];
    const uint32_t row_bytes = (uint32_t)W * 12u;
    auto request = [&](int logical_row, u3& a, u3& b) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (size_t)reflect_clamp(logical_row, H) * spitch),
                                                                            0, (int)row_bytes, 0x00020000);
        a = __builtin_amdgcn_raw_buffer_load_b96(rs, ca, 0, 0);
        b = __builtin_amdgcn_raw_buffer_load_b96(rs, cb, 0, 0);
    };
#pragma unroll
    for (int a = 0; a < kAhead; ++a) request(Y0 - 5 + a, pa[a], pb[a]);
    // one row step; p = i % 9 is a compile-time constant so that the register rings R and D are indexed statically
    auto step = [&](auto p_tag, const int i) {
            constexpr int p = decltype(p_tag)::value;
            // 1. the prefetched source row (logical row Y0 - 5 + i) goes into the ring; the next one is requested
            float* const slot = ring + (i & (kRingRows - 1)) * kRowFloats;
            constexpr int q = p % kAhead;
            uint32_t* const slot_u = (uint32_t*)slot;
            slot_u[lane * 3] = pa[q].x; slot_u[lane * 3 + 1] = pa[q].y; slot_u[lane * 3 + 2] = pa[q].z;
            if (lane < 8) { slot_u[(64 + lane) * 3] = pb[q].x; slot_u[(64 + lane) * 3 + 1] = pb[q].y; slot_u[(64 + lane) * 3 + 2] = pb[q].z; }
            __builtin_amdgcn_wave_barrier();                       // compiler-only: the other lanes' values are read below
            request(Y0 - 5 + kAhead + i, pa[q], pb[q]);            // (past the segment's last row: a harmless row of the image)
            // 2. row pass of this lane's pixel
            {
                const float* wp = slot + wl;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = wp[c] * kG0;
                    acc = wp[3 + c] * kG1 + acc;
                    acc = wp[6 + c] * kG2 + acc;
                    acc = wp[9 + c] * kG3 + acc;
                    acc = wp[12 + c] * kG4 + acc;
                    acc = wp[15 + c] * kG3 + acc;
                    acc = wp[18 + c] * kG2 + acc;
                    acc = wp[21 + c] * kG1 + acc;
                    acc = wp[24 + c] * kG0 + acc;
                    R[p][c] = acc;
                }
            }
            // 3. column pass + difference for image row yb (R[p] is logical row yb + 4)
            const int yb = Y0 - 9 + i;
            if (i >= 8) {
                if (yb >= 0 && yb < H) {
                    const float* sc = ring + ((i - 4) & (kRingRows - 1)) * kRowFloats + wl + 12;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float acc = kG4 * R[(p + 5) % 9][c] + 0.f;
                        acc = kG3 * (R[(p + 6) % 9][c] + R[(p + 4) % 9][c]) + acc;
                        acc = kG2 * (R[(p + 7) % 9][c] + R[(p + 3) % 9][c]) + acc;
                        acc = kG1 * (R[(p + 8) % 9][c] + R[(p + 2) % 9][c]) + acc;
                        acc = kG0 * (R[p][c] + R[(p + 1) % 9][c]) + acc;
                        D[p % 3][c] = sc[c] - acc;
                    }
                    if (yb == 0) { D[(p + 2) % 3][0] = D[p % 3][0]; D[(p + 2) % 3][1] = D[p % 3][1]; D[(p + 2) % 3][2] = D[p % 3][2]; }
                } else if (yb >= H) {                                // below the image: the median's replicated last row
                    D[p % 3][0] = D[(p + 2) % 3][0]; D[p % 3][1] = D[(p + 2) % 3][1]; D[p % 3][2] = D[(p + 2) % 3][2];
                }
            }
            // 4. image row yo: median of the difference, threshold, apply, convert, store
            if (i >= 10) {
                const int yo = Y0 - 10 + i;
                float d[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = D[(p + 1) % 3][c], b = D[(p + 2) % 3][c], e = D[p % 3][c];     // rows yo - 1, yo, yo + 1
                    const float lo = min3s(a, b, e), mi = med3s(a, b, e), hi = max3s(a, b, e);
                    const float lo3 = max_with_neighbours(lo);
                    const float mi3 = med3s(from_prev_lane(mi), mi, from_next_lane(mi));
                    const float hi3 = min_with_neighbours(hi);
                    d[c] = med3s(lo3, mi3, hi3);
                }
                const float* so = ring + ((i - 5) & (kRingRows - 1)) * kRowFloats + wl + 12;
                float v[3] = {so[0], so[1], so[2]};
                // |d|^2 >= norm2_min (the double sum of squares against the smallest double whose square root reaches the threshold).
                // The float sum decides unless it lies within 1e-6 of the bound (its error is below 3e-7 relative); only then the doubles.
                const float nf = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                bool sharpen = nf > t2f;
                if (__builtin_amdgcn_ballot_w64(fabsf(nf - t2f) <= t2f * 1e-6f) != 0) {
                    const double n2 = (double)d[0] * (double)d[0] + (double)d[1] * (double)d[1] + (double)d[2] * (double)d[2];
                    sharpen = n2 >= norm2_min;
                }
                if (sharpen) { v[0] = v[0] + amount * d[0]; v[1] = v[1] + amount * d[1]; v[2] = v[2] + amount * d[2]; }
                // convertTo(CV_8U, 255): v * 255 + 0; the "+ 0" only turns -0 into +0, which rounds to the same byte.  v_cvt_pk_u8_f32
                // equals saturate_cast<uchar>(cvRound(x)) for every float except x >= 2^31, where x86's cvRound returns INT_MIN, i.e. byte 0
                // (tools/micro/cvt_u8_probe.hip, all 2^32 bit patterns)
                const float t0 = v[0] * 255.f, t1 = v[1] * 255.f, t2 = v[2] * 255.f;
                uint32_t P = __builtin_amdgcn_cvt_pk_u8_f32(t2, 2, __builtin_amdgcn_cvt_pk_u8_f32(t1, 1, __builtin_amdgcn_cvt_pk_u8_f32(t0, 0, 0u)));
                if (__builtin_amdgcn_ballot_w64(max3s(t0, t1, t2) >= 2147483648.f) != 0)
                    P = (uint32_t)sat_u8(cv_round_x86(t0)) | ((uint32_t)sat_u8(cv_round_x86(t1)) << 8) | ((uint32_t)sat_u8(cv_round_x86(t2)) << 16);
                const uint32_t Pn = (uint32_t)__builtin_amdgcn_update_dpp((int)P, (int)P, 0x130, 0xf, 0xf, false);
                // four pixels = three dwords: lane j of a group writes dword j = bytes of its own pixel from byte j on + the first bytes of the next pixel
                if (st_dw) {
                    const uint32_t dw = (P >> (8 * j4)) | (Pn << (24 - 8 * j4));
                    *(uint32_t*)(out + (size_t)yo * pitch + (size_t)xg * 3 + 4 * j4) = dw;
                }
                if (st_px) {
                    uint8_t* o = out + (size_t)yo * pitch + (size_t)(X0 - 1 + lane) * 3;
                    o[0] = (uint8_t)P; o[1] = (uint8_t)(P >> 8); o[2] = (uint8_t)(P >> 16);
                }
                if (outF && stores && X0 - 1 + lane < W) {
                    float* o = outF + (size_t)yo * pitch + (size_t)(X0 - 1 + lane) * 3;
                    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
                }
            }
            __builtin_amdgcn_wave_barrier();                       // ... and the ring slot written next step is not read after this point
    };
    for (int i0 = 0; i0 < n_steps; i0 += 9) {
#define POPPY_STEP(P) if (i0 + P >= n_steps) break; step(std::integral_constant<int, P>{}, i0 + P);
        POPPY_STEP(0) POPPY_STEP(1) POPPY_STEP(2) POPPY_STEP(3) POPPY_STEP(4) POPPY_STEP(5) POPPY_STEP(6) POPPY_STEP(7) POPPY_STEP(8)
#undef POPPY_STEP
    }
}

// Geometry the kernel can take, and where it pays: at 4K it runs in 57-63 us against 75-80 for the tile kernel; a 1080p frame is ONE
// round of 3-4 waves per SIMD whose 18-22 dependent row steps cannot hide each other's latency (26-28 us against 21-25: the tile
// kernel stays below 4 Mpx).  POPPY_UNSHARP_STREAM forces it for every geometry it can take (the GPU tests run both).
bool unsharp_stream_takes(int w, int h) { return w >= 64 && h >= 16; }
bool unsharp_stream_eligible(int w, int h) {
    static const bool forced = getenv("POPPY_UNSHARP_STREAM") != nullptr;
    return unsharp_stream_takes(w, h) && (forced || (long long)w * h >= 4000000);
}

void launch_unsharp_stream(const float* src, uint8_t* out_u8, float* out_f32_or_null, int w, int h, float amount, const float* d_amount,
                           double norm2_min, hipStream_t s, hipEvent_t done, int src_pitch) {
    const int n_strips = (w + kStripPx - 1) / kStripPx, blocks_x = (n_strips + 3) / 4;
    // rows per segment: about as many waves as the chip holds at once (1024 SIMDs x 4-5), not so few rows that the ten warm-up rows
    // dominate (4K: 27 rows, 56.7 us; 18-36 rows 58-63 us)
    static const int forced = getenv("POPPY_UNSHARP_ROWS") ? atoi(getenv("POPPY_UNSHARP_ROWS")) : 0;
    int seg = forced > 0 ? forced : (int)(((long long)h * n_strips + 2560) / 5120);
    seg = seg < 8 ? 8 : (seg > 64 ? 64 : seg);
    const int blocks_y = (h + seg - 1) / seg;
    hipExtLaunchKernelGGL(k_unsharp_stream, dim3(blocks_x * blocks_y), dim3(256), 0, s, nullptr, done, 0, src, out_u8, out_f32_or_null,
                          w, h, src_pitch > 0 ? src_pitch : w, seg, n_strips, blocks_x, amount, d_amount, norm2_min, stagger_flag(7));
}

}  // namespace poppy_hip
