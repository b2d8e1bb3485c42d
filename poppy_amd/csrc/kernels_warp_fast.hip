// kernels_warp_fast.hip — the warp kernel (create_map + remap of both sources) with the per-pixel instruction count
// cut to what gfx950 can issue at twice the rate: packed fp32 (v_pk_mul/add/fma_f32), one reciprocal per
// denominator shared by both coordinates, a magic-number float->int rounding, 16-bit packed tap weights and
// v_dot2_u32_u16 for the bilinear sum.  Every result bit equals k_warp4's (kernels_frame.hip), which is the
// reference's arithmetic (src/algo.cpp:146-176; OCV/imgproc/src/imgwarp.cpp:1197-1234,721-731,808-852):
//
//   * affine rows: (h0*x + h1*y) + h2, each operation rounded on its own (this file is built with -ffp-contract=off;
//     the packed instructions round each half exactly like the scalar ones);
//   * division: the instruction sequence hipcc emits for an IEEE float division on this target is
//        r0 = rcp(d); e = fma(-d, r0, 1); r1 = fma(e, r0, r0); q0 = n*r1; s0 = fma(-d, q0, n); q1 = fma(s0, r1, q0);
//        s1 = fma(-d, q1, n); q = fma(s1, r1, q1)
//     wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act when an operand or the quotient is near the
//     ends of the exponent range.  The host admits a frame to this kernel only if every triangle's denominator stays
//     within [2^-20, 2^20] over the whole image (frame_plan.cpp: pack_warp_records); then the wrappers are the identity
//     for every numerator whose quotient can reach the image, the bare sequence above returns the same bits, and r1
//     depends on the denominator only, so mapx and mapy share it.  Quotients too large or too small to matter
//     (|q| >= 2^16, or < 2^-80) are clamped / round to zero on both paths;
//   * cvRound(v*32): v*32 is exact; adding 1.5*2^23 rounds to nearest-even integer for |v*32| <= 2^21, larger
//     values are clamped first and land outside the image, where the exact byte-wise path redoes the pixel;
//   * tap weights: BilinearTab_i's first entry {32767,0,0,1} yields the same byte as the unsaturated {32768,0,0,0}
//     ((32767 a + d + 16384) >> 15 = a for bytes a, d), so the table is replaced by exact 5-bit products, two
//     weights per register;  sum(w*s) + 16384 stays below 2^24, so v_dot2_u32_u16 is exact.
//
// Per-triangle data comes as one 80-byte record holding both inverse matrices in the order the packed
// instructions consume them; record 0 is the identity (id-map value 0 = no triangle = mapx,mapy = x,y).
#include "warp_fast_device.h"
#include <climits>
#include <cstdlib>

namespace poppy_hip {

// ---------------------------------------------------------------------------------------------------------------
// One workgroup = one 64 x 16 pixel tile (16 x 16 threads, 4 pixels per thread), tiles numbered XCD by XCD.
// The tile's few triangles: their records go through a 64-slot direct-mapped cache in LDS (slot = id & 63), filled by
// one cooperative pass and read back with ds_read_b128, which takes 72 bytes per pixel off the vector-memory path; an
// id whose slot was taken by another id of the same tile reads memory instead.  Footprints are fetched as three
// ALIGNED dwords per row and shifted into place (v_alignbyte): the unaligned 8-byte form costs the texture path more.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSlots = 64;

template <int kTileW>
__global__ void __launch_bounds__(256) k_warp_tile(int4* __restrict__ triMap4, const float4* __restrict__ rec,
                                                   const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                   uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H, int n_rec,
                                                   int tiles_x, WarpExtras ex) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    __shared__ int s_tag[kSlots];
    __shared__ float4 s_rec[kSlots * 5];
    __shared__ u4v s_ids[256];                  // each thread's four ids, parked for the rare border path

    const int tid = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * kTileW + (tid % kTileTx) * 4, y = ty * kTileH + tid / kTileTx;
    const bool active = x0 < W && y < H;
    const uint32_t g = (uint32_t)y * (uint32_t)(W >> 2) + (uint32_t)(x0 >> 2);          // 4-pixel group index
    const uint32_t pitch = (uint32_t)W * 3u, npx = (uint32_t)W * (uint32_t)H;
    const __amdgpu_buffer_rsrc_t rrec = make_rsrc(rec, (uint32_t)n_rec * 80u);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(c1, pitch * (uint32_t)H + 16u), rs2 = make_rsrc(c2, pitch * (uint32_t)H + 16u);
    const __amdgpu_buffer_rsrc_t rmap = make_rsrc(triMap4, npx * 4u);
    const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(tr1, npx * 3u), ro2 = make_rsrc(tr2, npx * 3u);

    // -- ids, lbmask; claim cache slots -------------------------------------------------------------
    int id[4] = {0, 0, 0, 0};
    float4 m2v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const u4v raw = __builtin_amdgcn_raw_buffer_load_b128(rmap, g * 16u, 0, 0);
        if (ex.m2) m2v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ex.m2, npx * 4u), g * 16u, 0, 0));   // both in flight together
        id[0] = decode_id(raw.x, ex.id_base); id[1] = decode_id(raw.y, ex.id_base);
        id[2] = decode_id(raw.z, ex.id_base); id[3] = decode_id(raw.w, ex.id_base);
        s_ids[tid] = u4v{(unsigned)id[0], (unsigned)id[1], (unsigned)id[2], (unsigned)id[3]};
        s_tag[id[0] & (kSlots - 1)] = id[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (id[k] != id[k - 1]) s_tag[id[k] & (kSlots - 1)] = id[k];
    }
    __syncthreads();
    // -- fill the claimed slots (a slot nobody claimed holds a stale tag: its load is harmless, nobody asks for it) --
    const int fslot = tid >> 2, fpart = tid & 3;                    // 64 slots x 4 parts, then the 64 fifth parts
    const u4v fill0 = __builtin_amdgcn_raw_buffer_load_b128(rrec, (uint32_t)s_tag[fslot] * 80u + (uint32_t)fpart * 16u, 0, 0);
    u4v fill1 = {0u, 0u, 0u, 0u};
    if (tid < kSlots) fill1 = __builtin_amdgcn_raw_buffer_load_b128(rrec, (uint32_t)s_tag[tid] * 80u + 64u, 0, 0);
    if (active && ex.m2) {                                       // the lbmask rider, while the records are on their way
        const float4 o = make_float4(mask_value(m2v.x, ex.alpha, ex.beta), mask_value(m2v.y, ex.alpha, ex.beta),
                                     mask_value(m2v.z, ex.alpha, ex.beta), mask_value(m2v.w, ex.alpha, ex.beta));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o), make_rsrc(ex.mask, npx * 4u), g * 16u, 0, 0);
    }
    s_rec[fslot * 5 + fpart] = __builtin_bit_cast(float4, fill0);
    if (tid < kSlots) s_rec[tid * 5 + 4] = __builtin_bit_cast(float4, fill1);
    __syncthreads();
    if (!active) return;
    // -- coordinates ----------------------------------------------------------------------------------------------
    const float fy = (float)y;
    const f2 fy2 = {fy, fy}, one2 = {1.f, 1.f};
    FastTap t[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int slot = id[k] & (kSlots - 1);
        float4 A, B, C, D; f2 E;
        if (s_tag[slot] == id[k]) {
            A = s_rec[slot * 5]; B = s_rec[slot * 5 + 1]; C = s_rec[slot * 5 + 2]; D = s_rec[slot * 5 + 3];
            const float4 e4 = s_rec[slot * 5 + 4];
            E = f2{e4.x, e4.y};
        } else {
            const uint32_t ro = __umul24(id[k], 80);
            A = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrec, ro, 0, 0));
            B = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrec, ro + 16, 0, 0));
            C = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrec, ro + 32, 0, 0));
            D = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrec, ro + 48, 0, 0));
            E = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rrec, ro + 64, 0, 0));
        }
        const float fx = (float)(x0 + k);
        const f2 fx2 = {fx, fx};
        const f2 n1 = (f2{A.x, A.y} * fx2 + f2{A.z, A.w} * fy2) + f2{B.x, B.y};
        const f2 n2 = (f2{B.z, B.w} * fx2 + f2{C.x, C.y} * fy2) + f2{C.z, C.w};
        const f2 z = (f2{D.x, D.y} * fx2 + f2{D.z, D.w} * fy2) + f2{E.x, E.y};
        const f2 r0 = {__builtin_amdgcn_rcpf(z.x), __builtin_amdgcn_rcpf(z.y)};
        const f2 e0 = __builtin_elementwise_fma(-z, r0, one2);
        const f2 r1 = __builtin_elementwise_fma(e0, r0, r0);
        const f2 q1 = div_core(n1, f2{z.x, z.x}, f2{r1.x, r1.x});
        const f2 q2 = div_core(n2, f2{z.y, z.y}, f2{r1.y, r1.y});
        int sx, sy;
        to_fixed(q1, sx, sy);
        t[0][k] = make_fast_tap(sx, sy, W, H);
        to_fixed(q2, sx, sy);
        t[1][k] = make_fast_tap(sx, sy, W, H);
    }
    // -- footprints: all 16 row fetches in flight together --------------------------------------------------------
    u3v ra[2][4], rb[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const uint32_t o4 = t[im][k].off & ~3u;
            ra[im][k] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, 0, 0);
            rb[im][k] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, (int)pitch, 0);
        }
    }
    uint32_t p[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const uint32_t bs = t[im][k].off & 3u;
            const u3v a3 = ra[im][k], b3 = rb[im][k];
            const u2v a = {__builtin_amdgcn_alignbyte(a3.y, a3.x, bs), __builtin_amdgcn_alignbyte(a3.z, a3.y, bs)};
            const u2v b = {__builtin_amdgcn_alignbyte(b3.y, b3.x, bs), __builtin_amdgcn_alignbyte(b3.z, b3.y, bs)};
            p[im][k] = blend_fast(t[im][k], a, b);
        }
    }
    const u3v o1 = {p[0][0] | (p[0][1] << 24), (p[0][1] >> 8) | (p[0][2] << 16), (p[0][2] >> 16) | (p[0][3] << 8)};
    const u3v o2 = {p[1][0] | (p[1][1] << 24), (p[1][1] >> 8) | (p[1][2] << 16), (p[1][2] >> 16) | (p[1][3] << 8)};
    __builtin_amdgcn_raw_buffer_store_b96(o1, ro1, g * 12u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b96(o2, ro2, g * 12u, 0, 0);
    // Footprints on or over the image border (wave-uniform test: most waves never see this code): the byte-wise
    // definition, written over the pixel afterwards so that its registers do not count against the main path.
    uint32_t edges = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) edges |= (t[0][k].inside ? 0u : 1u << k) | (t[1][k].inside ? 0u : 16u << k);
    if (__builtin_amdgcn_ballot_w64(edges != 0) != 0 && edges != 0) {
        const u4v ids = s_ids[tid];                                                     // not kept in registers for this rare path
        for (int e = 0; e < 8; ++e) {
            if (!((edges >> e) & 1u)) continue;
            const int k = e & 3, im = e >> 2;
            const uint32_t idk = k == 0 ? ids.x : k == 1 ? ids.y : k == 2 ? ids.z : ids.w;
            const uint32_t v = slow_pixel((const float*)(rec + idk * 5u), im, im ? c2 : c1, W, H, x0 + k, y);
            uint8_t* dst = (uint8_t*)(im ? tr2 : tr1) + ((size_t)y * W + x0 + k) * 3;
            dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
        }
    }
}

bool warp_fast_geometry(int w, int h) {
    return (w & 3) == 0 && w >= 8 && h >= 2 && w <= 16384 && h <= 16384 && (long long)w * h * 3 + 16 < (1ll << 31);
}

void launch_warp_fast(int32_t* triMap, const float* records, int n_records, const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2,
                      int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    // tile = 1024 pixels, 64 x 16 up to 1080p-class frames, 128 x 8 above (measured: 21.0 / 21.4 us at 1080p, 65.5 / 61.2 us at 4K)
    static const int forced = getenv("POPPY_TILE_W") ? atoi(getenv("POPPY_TILE_W")) : 0;
    const int tw = forced ? forced : ((long long)w * h >= 4000000 ? 128 : 64);
#define LW(TW) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW); \
    hipExtLaunchKernelGGL(k_warp_tile<TW>, dim3(tiles_x * tiles_y), dim3(256), 0, s, t0, t1, 0, (int4*)triMap, \
                          (const float4*)records, c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, w, h, n_records, tiles_x, ex); }
    switch (tw) { case 32: LW(32); break; case 128: LW(128); break; case 256: LW(256); break; default: LW(64); }
#undef LW
}

}  // namespace poppy_hip
