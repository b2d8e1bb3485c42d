// warp_fast_device.h — the packed-arithmetic pieces shared by the two tiled warp kernels (kernels_warp_fast.hip: ids from the
// id map; kernels_warp_bin.hip: one id byte per pixel from k_tile_expand).  The header comment of kernels_warp_fast.hip explains why each of them
// returns the reference's bits.
#pragma once
#include <type_traits>
#include "kernels.h"
#include "warp_device.h"
#include <hip/hip_ext.h>

// POPPY_WARP_ABL (timing builds only — tools/experiments/abl_build.sh; the product is built without it and the results of such a build are wrong by
// construction): bit 0 = the IEEE division chains become one multiply, bit 1 = constant bilinear weights, bit 2 = no blend arithmetic (the first
// footprint word passes through), bit 3 = no footprint loads (registers filled from the tap offsets).  What each deletion does to the kernel's time says
// whether it is bound by instruction issue (profiles/r05_notes.md).
#ifndef POPPY_WARP_ABL
#define POPPY_WARP_ABL 0
#endif

namespace poppy_hip {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 div_core(f2 n, f2 d, f2 r1) {
    f2 q0 = n * r1;
    f2 s0 = __builtin_elementwise_fma(-d, q0, n);
    f2 q1 = __builtin_elementwise_fma(s0, r1, q0);
    f2 s1 = __builtin_elementwise_fma(-d, q1, n);
    return __builtin_elementwise_fma(s1, r1, q1);
}

// (sx, sy) = cvRound(v) for both coordinates, v = 32 x the map coordinate (the records' numerator rows are scaled by 32 on the host:
// pack_warp_records); out-of-range values come back far outside any image
// (Dropping the two clamps behind a host-side bound on the quotients was measured: same kernel time, and the bound sent 8 % of the
// frames of a real pair — sliver triangles — to the general kernel.)
__device__ __forceinline__ void to_fixed(f2 v, int& sx, int& sy) {
    v.x = __builtin_amdgcn_fmed3f(v.x, -2097152.f, 2097152.f);
    v.y = __builtin_amdgcn_fmed3f(v.y, -2097152.f, 2097152.f);
    const f2 magic = {12582912.f, 12582912.f};
    const f2 t = v + magic;
    sx = __float_as_int(t.x) - 0x4B400000;
    sy = __float_as_int(t.y) - 0x4B400000;
}

struct FastTap { uint32_t wt, wb, off; bool inside; };

// The four bilinear weights of a tap, DOUBLED and held two to a register: wt = 2 w00 | 2 w01 << 16, wb = 2 w10 | 2 w11 << 16 with
// w = BilinearTab_i (imgwarp.cpp:213-287) = 32 (32 - fx or fx)(32 - fy or fy).  Doubling moves the result byte of
// (sum w s + 16384) >> 15 = (sum 2w s + 32768) >> 16 onto a byte boundary, so three channels are packed with two v_perm instead of
// three shift / shift-or pairs.  2 w00 = 65536 at fx = fy = 0 does not fit 16 bits: v_pk_mad_u16 with clamp saturates it to 65535,
// and (65535 s + 32768) >> 16 = s for every byte s — the same result (the other three weights are 0 there).
__device__ __forceinline__ FastTap make_fast_tap(int sx, int sy, int W, int H) {
    FastTap t;
    const int fx = sx & 31, fy = sy & 31, ix = sx >> 5, iy = sy >> 5;
    t.inside = (unsigned)ix < (unsigned)(W - 1) && (unsigned)iy < (unsigned)(H - 1);
    const uint32_t P = __umul24(fx, 65535u) + 32u;                 // (32 - fx) | fx << 16
    const uint32_t M2 = 0x08000800u - __umul24(fy, 0x00400040u);   // 64 (32 - fy) in both halves
#if POPPY_WARP_ABL & 2
    t.wt = 0x40004000u; t.wb = 0x40004000u; (void)P; (void)M2;
#else
    asm("v_pk_mad_u16 %0, %1, %2, 0 clamp" : "=v"(t.wt) : "v"(P), "v"(M2));
    t.wb = __umul24(P, (uint32_t)(fy << 6));                       // 2 w10 | 2 w11 << 16 (no half exceeds 63488)
#endif
    t.off = t.inside ? (uint32_t)(__umul24(iy, W) + ix) * 3u : 0u;
    return t;
}

__device__ __forceinline__ uint32_t blend_fast(const FastTap& t, u2v a, u2v b) {
#if POPPY_WARP_ABL & 4
    return (a.x ^ b.y) + t.wt;
#endif
    uint32_t acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t sel = 0x0c000c00u | (uint32_t)k | ((uint32_t)(k + 3) << 16);
        const uint32_t pt = __builtin_amdgcn_perm(a.y, a.x, sel);                // s00 | s01 << 16
        const uint32_t pb = __builtin_amdgcn_perm(b.y, b.x, sel);
        acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pt), __builtin_bit_cast(us2, t.wt), 32768u, false);
        acc[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, pb), __builtin_bit_cast(us2, t.wb), acc[k], false);
    }
    // byte 2 of each sum (below 2^24) is the channel
    return __builtin_amdgcn_perm(acc[2], __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u), 0x0c060100u);
}

// raw buffer descriptor over [p, p + bytes): 32-bit offsets straight into the load instruction, no 64-bit address arithmetic
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// footprint on or over the image border: the byte-wise definition, from the record's matrix
__device__ __forceinline__ uint32_t slow_pixel(const float* __restrict__ rec, int src, const uint8_t* __restrict__ img, int W, int H, int x, int y) {
    float h[9];
    const float u = 0.03125f;                                     // the numerator rows are stored times 32: exact to undo
    if (src == 0) { h[0] = rec[0] * u; h[3] = rec[1] * u; h[1] = rec[2] * u; h[4] = rec[3] * u; h[2] = rec[4] * u; h[5] = rec[5] * u; }
    else          { h[0] = rec[6] * u; h[3] = rec[7] * u; h[1] = rec[8] * u; h[4] = rec[9] * u; h[2] = rec[10] * u; h[5] = rec[11] * u; }
    h[6] = rec[12 + src]; h[7] = rec[14 + src]; h[8] = rec[16 + src];
    float mx, my;
    map_point(h, x, y, mx, my);
    uint8_t o[3];
    sample3(img, W, H, mx, my, o);
    return o[0] | (o[1] << 8) | (o[2] << 16);
}

typedef unsigned u3v __attribute__((ext_vector_type(3)));

// Footprint fetch + bilinear blend + store of one thread's four pixels (both sources), given the eight taps; then the rare
// byte-wise redo of pixels whose 2x2 footprint touches the image border.  rec_of(k) returns the record pointer of pixel k.
template <bool kAlignedRows, typename RecOf>
__device__ __forceinline__ void warp_fetch_blend_store(FastTap (&t)[2][4], __amdgpu_buffer_rsrc_t rs1, __amdgpu_buffer_rsrc_t rs2,
                                                       __amdgpu_buffer_rsrc_t ro1, __amdgpu_buffer_rsrc_t ro2, uint32_t pitch, uint32_t g,
                                                       const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                       uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H, int x0, int y, RecOf rec_of,
                                                       int out_pitch = 0) {
    // The footprint's second row lies `pitch` bytes behind the first.  Source rows of widths that are no multiple of 4 are no multiple of 4 bytes long, so the second
    // row's dwords begin elsewhere: it gets its own aligned address and byte shift (kAlignedRows = false: an instantiation of its own, the registers it needs would
    // spill in the common one) — loaded at o4 + pitch the dwordx3 loads were UNALIGNED and the kernel took 71 us at 3838 x 2160 where 3840 takes 45 (round 6).
    {
        u3v ra[2][4], rb[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const uint32_t o4 = t[im][k].off & ~3u;
#if POPPY_WARP_ABL & 8
                ra[im][k] = u3v{o4, o4 + 1u, o4 + 2u}; rb[im][k] = u3v{o4 ^ pitch, o4 + 5u, o4 + 7u};
#else
                ra[im][k] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, 0, 0);
                if (kAlignedRows) rb[im][k] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, (int)pitch, 0);
                else rb[im][k] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, (t[im][k].off + pitch) & ~3u, 0, 0);
#endif
            }
        }
#ifdef POPPY_WARP_AFTER_GATHERS
        POPPY_WARP_AFTER_GATHERS;
#endif
        uint32_t p[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const uint32_t bs = t[im][k].off & 3u;
                const uint32_t bsb = kAlignedRows ? bs : (t[im][k].off + pitch) & 3u;
                const u3v a3 = ra[im][k], b3 = rb[im][k];
                const u2v a = {__builtin_amdgcn_alignbyte(a3.y, a3.x, bs), __builtin_amdgcn_alignbyte(a3.z, a3.y, bs)};
                const u2v b = {__builtin_amdgcn_alignbyte(b3.y, b3.x, bsb), __builtin_amdgcn_alignbyte(b3.z, b3.y, bsb)};
                p[im][k] = blend_fast(t[im][k], a, b);
            }
        }
        const u3v o1 = {p[0][0] | (p[0][1] << 24), (p[0][1] >> 8) | (p[0][2] << 16), (p[0][2] >> 16) | (p[0][3] << 8)};
        const u3v o2 = {p[1][0] | (p[1][1] << 24), (p[1][1] >> 8) | (p[1][2] << 16), (p[1][2] >> 16) | (p[1][3] << 8)};
        __builtin_amdgcn_raw_buffer_store_b96(o1, ro1, g * 12u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b96(o2, ro2, g * 12u, 0, 0);
    }
#ifndef POPPY_WARP_COUNT_MAIN          // tools/warp_facts.py counts the main path's instructions with the rare paths compiled out
    uint32_t edges = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) edges |= (t[0][k].inside ? 0u : 1u << k) | (t[1][k].inside ? 0u : 16u << k);
    // pixels past the row's end (widths that are no multiple of 4: the padding of the outputs' last group, never read) are not redone — their maps point
    // outside the image, so EVERY row's last wave took the byte-wise path for them: k_warp_bin 72 us at 3838 x 2160 where 3840 takes 45 (round 6)
    if (x0 + 3 >= W) edges &= x0 + 2 >= W ? (x0 + 1 >= W ? 0x11u : 0x33u) : 0x77u;
    if (__builtin_amdgcn_ballot_w64(edges != 0) != 0 && edges != 0) {
        for (int e = 0; e < 8; ++e) {
            if (!((edges >> e) & 1u)) continue;
            const int k = e & 3, im = e >> 2;
            const uint32_t v = slow_pixel(rec_of(k), im, im ? c2 : c1, W, H, x0 + k, y);
            uint8_t* dst = (uint8_t*)(im ? tr2 : tr1) + ((size_t)y * (out_pitch > 0 ? out_pitch : W) + x0 + k) * 3;
            dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
        }
    }
#endif
}

// the eight taps of one pixel pair from its record (A..E as laid out by pack_warp_records)
__device__ __forceinline__ void warp_taps(float4 A, float4 B, float4 C, float4 D, f2 E, float fx, float fy, int W, int H, FastTap& t0, FastTap& t1) {
    const f2 fx2 = {fx, fx}, fy2 = {fy, fy}, one2 = {1.f, 1.f};
    const f2 n1 = (f2{A.x, A.y} * fx2 + f2{A.z, A.w} * fy2) + f2{B.x, B.y};
    const f2 n2 = (f2{B.z, B.w} * fx2 + f2{C.x, C.y} * fy2) + f2{C.z, C.w};
    const f2 z = (f2{D.x, D.y} * fx2 + f2{D.z, D.w} * fy2) + f2{E.x, E.y};
    const f2 r0 = {__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
    const f2 e0 = __builtin_elementwise_fma(-z, r0, one2);
    const f2 r1 = __builtin_elementwise_fma(e0, r0, r0);
    // the two sources' division chains step by step side by side: a dependent packed-fp32 instruction directly behind its producer costs a hazard
    // s_nop (0.8 issue cycles each beside half-rate instructions: tools/micro/issue_mix.hip), 10 of them per pixel when one chain follows the other
    const f2 d1 = {z.x, z.x}, d2 = {z.y, z.y}, ra = {r1.x, r1.x}, rb = {r1.y, r1.y};
    // (the empty asm statements pin the lock-step order: left alone, the scheduler runs one chain after the other)
#define POPPY_PAIR_HERE(a, b) asm volatile("" : "+v"(a), "+v"(b))
    f2 qa = n1 * ra, qb = n2 * rb;
#if POPPY_WARP_ABL & 1
    const f2 q1 = qa, q2 = qb;
    (void)d1; (void)d2;
#else
    POPPY_PAIR_HERE(qa, qb);
    f2 sa = __builtin_elementwise_fma(-d1, qa, n1), sb = __builtin_elementwise_fma(-d2, qb, n2);
    POPPY_PAIR_HERE(sa, sb);
    qa = __builtin_elementwise_fma(sa, ra, qa); qb = __builtin_elementwise_fma(sb, rb, qb);
    POPPY_PAIR_HERE(qa, qb);
    sa = __builtin_elementwise_fma(-d1, qa, n1); sb = __builtin_elementwise_fma(-d2, qb, n2);
    POPPY_PAIR_HERE(sa, sb);
    const f2 q1 = __builtin_elementwise_fma(sa, ra, qa), q2 = __builtin_elementwise_fma(sb, rb, qb);
#endif
#undef POPPY_PAIR_HERE
    int sx, sy;
    to_fixed(q1, sx, sy);
    t0 = make_fast_tap(sx, sy, W, H);
    to_fixed(q2, sx, sy);
    t1 = make_fast_tap(sx, sy, W, H);
}

}  // namespace
}  // namespace poppy_hip
