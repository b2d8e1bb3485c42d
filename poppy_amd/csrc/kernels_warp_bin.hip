// kernels_warp_bin.hip — paint_triangles + create_map + remap of both sources without a triangle-id map in HBM.
//
//   src/algo.cpp:95-106 (paint_triangles), OCV/imgproc/src/drawing.cpp:80-297,1093-1255 (Line, FillConvexPoly)
//   src/algo.cpp:146-176 (create_map), OCV/imgproc/src/imgwarp.cpp:1197-1234,721-731,808-852 (remap)
//
// The id a pixel ends up with is the LAST triangle whose outline-or-fill raster covers it (fillConvexPoly paints them one
// after another), i.e. the largest covering index.  The host bins the triangles by 1024-pixel tile (64 x 16 or 128 x 8;
// frame_plan.cpp: build_tile_bins, ascending order) and hands every triangle's fill-edge table (RasterTri) and its three outline
// segments, already clipped and ordered as Line() walks them (OutlineSeg).  Two kernels:
//   k_tile_expand   a workgroup per tile.  One thread per (triangle of the tile's list, row of the tile) evaluates that row of the
//                   raster in closed form — the fill span from the two edge chains (one 64-bit multiply-add each), the pixels of
//                   the three Bresenham outlines that fall on the row (an interval per segment: a shallow segment puts a run on a
//                   row, a steep one a single pixel) — as a coverage BIT MASK of the tile row in LDS; then every thread walks the
//                   masks of ITS four pixels in painter's order (one v_perm per entry) and writes ONE BYTE per pixel: the number
//                   of the last covering entry of the tile's list.  The entries' 80-byte warp records go into the tile's 32 fixed
//                   record slots.  This kernel depends on the plan only, not on any image: it runs on the plan-upload stream, off
//                   the chained frames' critical path.
//   k_warp_bin      a workgroup owns a tile: a thread's four ids are one coalesced dword, the tile's record slots go through LDS,
//                   both at addresses that follow from the tile number alone (no list offsets to wait for), then k_warp_tile's
//                   second half (warp_fast_device.h): records from LDS, packed map arithmetic, aligned footprint loads, v_dot2
//                   blends.
// Compared with k_raster + k_warp_tile this removes 3 B/px written and 3 B/px read of id map, the atomics and the frame tags, and
// the warp kernel's dependent memory round trips drop from three (ids -> records -> footprints) to two.  (Round 2 handed the row
// masks themselves to the warp kernel — 208 bytes per entry behind a tile_off lookup — and resolved painter's order there: timing
// builds showed that kernel's front end, not its arithmetic, to be what its duration followed: profiles/r03_notes.md.)
// Round 4 built the persistent forms of k_warp_bin the round-3 review asked for — gathers software-pipelined inside the wave (1, 2 or 4
// pixels per stage, 3-6 waves per SIMD), and an 8-wave form with the next tile's ids and record slots prefetched by LDS-DMA —, all bit-exact,
// all SLOWER (4K: 50-57 us against 45-46), and measured why: the kernel's arithmetic ALONE, on register data with 8 waves per SIMD and no
// memory instruction, takes 0.99 us per tile-wave per SIMD = 31 us at 4K (tools/micro/warp_alu.hip).  The kernels are in the history
// (commits 17fafdb, a82110b), the measurements in profiles/r04_notes.md.
// POPPY_WARP_PRIO (timing builds, tools/experiments/prio_build.sh; results unchanged): wave priorities / start offsets that take the co-resident waves of a
// SIMD out of lock step — a round of workgroups starts together, so its waves wait for their ids together, divide together, gather together and store
// together: the arithmetic (30 us at 4K alone) and the memory time (16.6 us for the skeleton alone, profiles/r06_warp_skeleton.txt) ADD UP instead of overlapping.
//   1  a workgroup's priority = (blockIdx.x >> 8) & 3 (the 8 workgroups of a CU in a round get 4 levels)
//   2  priority 3 from the moment a wave's gathers are issued (blend + store first), 0 before
//   3  both: (blockIdx.x >> 8) & 1 before the gathers, 2 + that after
//   4  start offsets: s_sleep ((blockIdx.x >> 8) & 7) * 8
//   5  priority 3 before the gathers (get the loads out first), 0 after
#ifndef POPPY_WARP_PRIO
#define POPPY_WARP_PRIO 0
#endif
#if POPPY_WARP_PRIO == 2
#define POPPY_WARP_AFTER_GATHERS __builtin_amdgcn_s_setprio(3)
#elif POPPY_WARP_PRIO == 3
#define POPPY_WARP_AFTER_GATHERS do { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); } while (0)
#elif POPPY_WARP_PRIO == 5
#define POPPY_WARP_AFTER_GATHERS __builtin_amdgcn_s_setprio(0)
#endif
#include "warp_fast_device.h"
#ifndef POPPY_WARP_WAVES
#define POPPY_WARP_WAVES 8      // waves per SIMD the register allocation must leave room for (64 VGPRs: the 2025 workgroups of a 1080p frame are then ONE round)
#endif
#include <climits>
#include <cstdlib>

namespace poppy_hip {

namespace {

struct RasterTriDev {                       // frame_plan.h: RasterTri
    int ymin, ystop, n0, n1;
    int ybeg[4];
    long long ex[4], edx[4];
};
static_assert(sizeof(RasterTriDev) == 96, "layout shared with the host plan");

struct Interval { int lo, hi; };            // inclusive; lo > hi = empty

// pixels of triangle row y painted by the fill (drawing.cpp:1228-1245)
__device__ __forceinline__ Interval fill_row(const RasterTriDev& r, int y, int W) {
    Interval iv{1, 0};
    if (y < r.ymin || y >= r.ystop || y < 0) return iv;
    const bool s0 = r.n0 > 1 && y >= r.ybeg[1], s1 = r.n1 > 1 && y >= r.ybeg[3];
    const long long a = (s0 ? r.ex[1] : r.ex[0]) + (long long)(y - (s0 ? r.ybeg[1] : r.ybeg[0])) * (s0 ? r.edx[1] : r.edx[0]);
    const long long b = (s1 ? r.ex[3] : r.ex[2]) + (long long)(y - (s1 ? r.ybeg[3] : r.ybeg[2])) * (s1 ? r.edx[3] : r.edx[2]);
    const long long xl = a > b ? b : a, xr = a > b ? a : b;
    int xx1 = (int)((xl + 32768) >> 16), xx2 = (int)((xr + 32768) >> 16);
    if (xx2 >= 0 && xx1 < W) {
        iv.lo = xx1 < 0 ? 0 : xx1;
        iv.hi = xx2 >= W ? W - 1 : xx2;
    }
    return iv;
}

// floor(a / b) for a >= 0, b > 0 whose quotient is below 2^15 (the callers' coordinates are clipped to the image: < 2^14): a float
// reciprocal with a correction step each way — a relative error of a few 2^-24 moves a quotient below 2^15 by less than one —
// instead of the ~30 instructions of an integer division
__device__ __forceinline__ int div_small_quotient(unsigned a, unsigned b) {
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int r = (int)(a - (unsigned)q * b);                    // (wraps to the true remainder when q is off by one)
    q += (r >= (int)b) - (r < 0);
    return q;
}

// pixels of one outline segment on row y: the steps k of the 8-connected Bresenham walk (drawing.cpp:159-245 with
// leftToRight) whose minor offset m_k = max(0, ceil((2 minor k - major) / (2 major))) puts them on this row
__device__ __forceinline__ Interval outline_row(int4 seg, int y) {
    Interval iv{1, 0};
    const int x0 = seg.x, y0 = seg.y, major = seg.z, minor = seg.w & 0xffffff, flags = seg.w >> 24;
    if (major < 0) return iv;
    const int d = (flags & 2) ? y0 - y : y - y0;                 // rows walked from the start, in the segment's y direction
    if (flags & 1) {                                             // steep: one pixel per row, k = d
        if (d < 0 || d > major) return iv;
        int m = 0;
        const int num = 2 * minor * d - major;
        if (num > 0) m = div_small_quotient((unsigned)(num + 2 * major - 1), (unsigned)(2 * major));
        iv.lo = iv.hi = x0 + m;
    } else {                                                     // shallow: row d holds the run of steps with m_k == d
        if (d < 0 || d > minor) return iv;
        const int k_lo = d == 0 ? 0 : div_small_quotient((unsigned)(major * (2 * d - 1)), (unsigned)(2 * minor)) + 1;
        const int k_hi = d == minor ? major : div_small_quotient((unsigned)(major * (2 * d + 1)), (unsigned)(2 * minor));
        iv.lo = x0 + k_lo; iv.hi = x0 + k_hi;
    }
    return iv;
}

// bits [a, b) of a 128-bit tile-row mask for the part of [lo, hi] inside the tile
template <int kTileW>
__device__ __forceinline__ void add_interval(Interval iv, int tx0, uint64_t& m0, uint64_t& m1) {
    int a = iv.lo - tx0, b = iv.hi + 1 - tx0;
    a = a < 0 ? 0 : a; b = b > kTileW ? kTileW : b;
    if (b <= a) return;
    // bits [a, b) = ones(b) & ~ones(a), ones(n) = low n bits set (n in 0..128)
    auto ones = [](int n, int word) -> uint64_t {
        const int k = n - 64 * word;
        return k <= 0 ? 0ull : (k >= 64 ? ~0ull : ((1ull << k) - 1ull));
    };
    m0 |= ones(b, 0) & ~ones(a, 0);
    if (kTileW > 64) m1 |= ones(b, 1) & ~ones(a, 1);
}

}  // namespace

// What k_tile_expand leaves for k_warp_bin, in one allocation (warp_bin_data_bytes):
//   ids      1 byte per pixel, tile after tile (1024 bytes each, row-major inside the tile): 0 = no triangle, e + 1 = entry e of
//            the tile's list — painter's order is resolved HERE, off the chained frames' critical path;
//   slots    32 x 80 bytes per tile: slot 0 the identity record, slot e + 1 the warp record of entry e (e < 31).  The warp kernel
//            loads a tile's ids and slots at fixed addresses: nothing in its front end waits for a tile_off lookup;
//   overflow the records of ALL entries of tiles with more than 31 of them, tile_off-indexed, 80 bytes each (rare: the warp
//            kernel reads tile_off only on that path).
constexpr int kEntryBytes = 80;
constexpr int kSlots = 32;
constexpr int kTileSlotBytes = kSlots * kEntryBytes;
constexpr int kTileIdBytes = 1024;
constexpr int kMaxTileEntries = 255;                      // an id is a byte
constexpr unsigned kOneRoundBlocks = 2304;                // 256 CUs x 8 workgroups of 256 threads (+ the first finishers' successors): a 1080p frame is 2025 / 2040

template <int kTileW>
__global__ void __launch_bounds__(256) k_tile_expand(const float4* __restrict__ rec, const RasterTriDev* __restrict__ tris,
                                                     const int4* __restrict__ outline, const int* __restrict__ tile_off,
                                                     const uint16_t* __restrict__ tile_tris, uint8_t* __restrict__ tile_data,
                                                     int W, int tiles_x, int n_tiles) {
    constexpr int kTileH = 1024 / kTileW, kPass = 256 / kTileH, kWords = kTileW / 64, kTileTx = kTileW / 4;
    __shared__ __attribute__((aligned(16))) uint64_t s_mask[4 * kPass * kTileH * kWords];     // [interval: fill, three outline segments][entry of the pass][row][word]
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int tx0 = tx * kTileW, ty0 = ty * kTileH;
    const int off0 = tile_off[tile], nb = tile_off[tile + 1] - off0;
    uint8_t* const slots = tile_data + (size_t)n_tiles * kTileIdBytes + (size_t)tile * kTileSlotBytes;
    uint8_t* const overflow = tile_data + (size_t)n_tiles * (kTileIdBytes + kTileSlotBytes);
    // the records: slots for the warp kernel's LDS stage, the whole list once more when it does not fit them
    if (tid < kSlots * 5) {
        const int slot = tid / 5, part = tid - slot * 5;
        if (slot <= nb) {
            const int t1 = slot ? tile_tris[off0 + slot - 1] + 1 : 0;
            *(float4*)(slots + tid * 16) = rec[(size_t)t1 * 5 + part];
        }
    }
    if (nb >= kSlots)
        for (int i = tid; i < nb * 5; i += 256) {
            const int e = i / 5, part = i - e * 5;
            *(float4*)(overflow + (size_t)(off0 + e) * kEntryBytes + part * 16) = rec[(size_t)(tile_tris[off0 + e] + 1) * 5 + part];
        }
    // raster jobs: (entry ji of the pass, tile row jr) x the four intervals of a triangle's row — the fill span and the three outline segments.  A wave
    // takes ONE of the four intervals for 64 jobs at a time (wave 0 the fills, waves 1-3 a segment each), so that the four short instruction streams
    // run side by side on the four SIMDs instead of one after the other in the one wave the usual 3-8 entries of a tile fill (the kernel's time is the
    // latency of a tile's chain of dependent steps, not its instruction count: 79 % of a wave's life was spent waiting); every wave writes the masks of
    // its own interval, the resolve step ORs the four.
    const int wv = tid >> 6, ln = tid & 63;
    const int row = tid / kTileTx, xg = tid % kTileTx;           // resolve job: pixels (4 xg .. 4 xg + 3, row) of the tile
    const int mdword = row * kWords * 2 + ((xg * 4) >> 5), shift = (xg * 4) & 31;
    const uint32_t* s_mask32 = (const uint32_t*)s_mask;
    uint32_t ids = 0;                                            // the four pixels' entry numbers, one per byte
    for (int base = 0; base < nb; base += kPass) {
        const int n_here = min(kPass, nb - base);
        if (base) __syncthreads();                               // the previous pass's masks have been read
        for (int j = ln; j < n_here * kTileH; j += 64) {
            const int ji = j / kTileH, jr = j - ji * kTileH;
            const int t = tile_tris[off0 + base + ji];
            const int yy = ty0 + jr;
            uint64_t m0 = 0, m1 = 0;
            if (wv == 0) add_interval<kTileW>(fill_row(tris[t], yy, W), tx0, m0, m1);
            else add_interval<kTileW>(outline_row(outline[t * 3 + wv - 1], yy), tx0, m0, m1);
            uint64_t* dst = s_mask + ((wv * kPass + ji) * kTileH + jr) * kWords;
            dst[0] = m0;
            if (kWords > 1) dst[kWords - 1] = m1;
        }
        __syncthreads();
        // painter's order, a later entry overwrites: an entry's four coverage bits become the byte selector of one v_perm between
        // the old bytes and the entry's number
        for (int i = 0; i < n_here; ++i) {
            constexpr int kPlane = kPass * kTileH * kWords * 2;                          // dwords of one interval's masks
            const int at = i * kTileH * kWords * 2 + mdword;
            const uint32_t bits = ((s_mask32[at] | s_mask32[at + kPlane] | s_mask32[at + 2 * kPlane] | s_mask32[at + 3 * kPlane]) >> shift) & 15u;
            const uint32_t spread = __umul24(bits, 0x204081u) & 0x01010101u;          // bit k of `bits` -> bit 0 of byte k
            ids = __builtin_amdgcn_perm((uint32_t)(base + i + 1) * 0x01010101u, ids, (spread << 2) | 0x03020100u);
        }
    }
    ((uint32_t*)tile_data)[(size_t)tile * 256 + tid] = ids;
}

// kAlignedRows: the sources' rows are a multiple of 4 bytes long (W % 4 == 0); the other instantiation (warp_fast_device.h) may take fewer waves per SIMD
template <int kTileW, bool kAlignedRows>
__global__ void __launch_bounds__(256, kAlignedRows ? POPPY_WARP_WAVES : POPPY_WARP_WAVES - 1) k_warp_bin(const float4* __restrict__ rec, const uint8_t* __restrict__ tile_data,
                                                  const int* __restrict__ tile_off,
                                                  const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                  uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H,
                                                  int tiles_x, uint32_t data_bytes, WarpExtras ex, int stagger) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    __shared__ float4 s_rec[kSlots * 5];
    __shared__ uint32_t s_ids[256];

    const int tid = threadIdx.x;
#if POPPY_WARP_PRIO == 0
    stagger_priority(blockIdx.x, stagger);
#elif POPPY_WARP_PRIO == 1
    switch ((blockIdx.x >> 8) & 3) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); }
#elif POPPY_WARP_PRIO >= 6
    {
        const unsigned b = blockIdx.x, n = gridDim.x;
        const unsigned pr = POPPY_WARP_PRIO == 6 ? (((b >> 8) & 1u) | (b * 2u < n ? 2u : 0u))
                          : POPPY_WARP_PRIO == 7 ? 3u - min(3u, b >> 11)
                          : POPPY_WARP_PRIO == 8 ? ((b >> 9) & 3u)
                          : POPPY_WARP_PRIO == 9 ? ((b >> 7) & 3u)
                          : POPPY_WARP_PRIO == 10 ? (((b >> 8) & 1u) * 3u)
                          : POPPY_WARP_PRIO == 11 ? ((b >> 10) & 3u)
                          : POPPY_WARP_PRIO == 12 ? ((b >> 3) & 3u)
                          : ((b >> 5) & 3u);
        switch (pr) { case 0: __builtin_amdgcn_s_setprio(0); break; case 1: __builtin_amdgcn_s_setprio(1); break; case 2: __builtin_amdgcn_s_setprio(2); break; default: __builtin_amdgcn_s_setprio(3); }
    }
#elif POPPY_WARP_PRIO == 3
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
#elif POPPY_WARP_PRIO == 4
    switch ((blockIdx.x >> 8) & 7) { case 1: __builtin_amdgcn_s_sleep(8); break; case 2: __builtin_amdgcn_s_sleep(16); break; case 3: __builtin_amdgcn_s_sleep(24); break; case 4: __builtin_amdgcn_s_sleep(32); break;
                                     case 5: __builtin_amdgcn_s_sleep(40); break; case 6: __builtin_amdgcn_s_sleep(48); break; case 7: __builtin_amdgcn_s_sleep(56); break; default: break; }
#elif POPPY_WARP_PRIO == 5
    __builtin_amdgcn_s_setprio(3);
#endif
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x), n_tiles = (int)gridDim.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int row = tid / kTileTx, xg = tid % kTileTx;           // this thread's pixels: (tx0 + 4 xg .. + 3, ty0 + row)
    const int x0 = tx * kTileW + xg * 4, y = ty * kTileH + row;
    const bool active = x0 < W && y < H;
    // the outputs' rows are op pixels long (a multiple of 4: W itself, or W rounded up — then the row's last group also holds up to three
    // pixels of padding, computed like any other and never read); the sources and m2 are tight
    const uint32_t op = ex.out_pitch > 0 ? (uint32_t)ex.out_pitch : (uint32_t)W;
    const uint32_t g = (uint32_t)y * (op >> 2) + (uint32_t)(x0 >> 2);
    const uint32_t pitch = (uint32_t)W * 3u, npx = op * (uint32_t)H;
    const __amdgpu_buffer_rsrc_t rdata = make_rsrc(tile_data, data_bytes);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(c1, pitch * (uint32_t)H + 16u), rs2 = make_rsrc(c2, pitch * (uint32_t)H + 16u);
    const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(tr1, npx * 3u), ro2 = make_rsrc(tr2, npx * 3u);

    // ONE round trip, at addresses that depend on the tile number only: this thread's four ids, the tile's record slots (through
    // LDS) and, when the mask rides along, four m2 values
    const uint32_t ids = __builtin_amdgcn_raw_buffer_load_b32(rdata, (uint32_t)tile * kTileIdBytes + (uint32_t)tid * 4u, 0, 0);
    const bool stages = tid < kSlots * 5;
    float4 staged = make_float4(0.f, 0.f, 0.f, 0.f);
    if (stages)
        staged = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                     rdata, (uint32_t)n_tiles * kTileIdBytes + (uint32_t)tile * kTileSlotBytes + (uint32_t)tid * 16u, 0, 0));
    float4 m2v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active && ex.m2)
        m2v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ex.m2, (uint32_t)W * (uint32_t)H * 4u), ((uint32_t)y * (uint32_t)W + (uint32_t)x0) * 4u, 0, 0));
    if (stages) s_rec[tid] = staged;
    __syncthreads();
    if (active && ex.m2) {                                       // the lbmask rider
        const float4 o = make_float4(mask_value(m2v.x, ex.alpha, ex.beta), mask_value(m2v.y, ex.alpha, ex.beta),
                                     mask_value(m2v.z, ex.alpha, ex.beta), mask_value(m2v.w, ex.alpha, ex.beta));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o), make_rsrc(ex.mask, npx * 4u), g * 16u, 0, 0);
    }
    if (!active) return;
    // the ids are only needed again by the rare border path: parked in LDS so that they do not count against the footprint phase
    s_ids[tid] = ids;

    const float fy = (float)y;
    FastTap t[2][4];
    const uint32_t over_base = (uint32_t)n_tiles * (kTileIdBytes + kTileSlotBytes);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int li = (int)((ids >> (8 * k)) & 255u);
        float4 A, B, C, D; f2 E;
#ifdef POPPY_WARP_COUNT_MAIN
        {
#else
        if (li < kSlots) {
#endif
            A = s_rec[li * 5]; B = s_rec[li * 5 + 1]; C = s_rec[li * 5 + 2]; D = s_rec[li * 5 + 3];
            const float4 e4 = s_rec[li * 5 + 4];
            E = f2{e4.x, e4.y};
        }
#ifndef POPPY_WARP_COUNT_MAIN
        else {                                                   // a tile with more triangles than the slots hold
            const uint32_t ro = over_base + (uint32_t)(tile_off[tile] + li - 1) * kEntryBytes;
            A = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, ro, 0, 0));
            B = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, ro + 16, 0, 0));
            C = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, ro + 32, 0, 0));
            D = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, ro + 48, 0, 0));
            E = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rdata, ro + 64, 0, 0));
        }
#endif
        warp_taps(A, B, C, D, E, (float)(x0 + k), fy, W, H, t[0][k], t[1][k]);
    }
    warp_fetch_blend_store<kAlignedRows>(t, rs1, rs2, ro1, ro2, pitch, g, c1, c2, tr1, tr2, W, H, x0, y, [&](int k) -> const float* {
        const unsigned e = (s_ids[tid] >> (8 * k)) & 255u;
        if (e == 0) return (const float*)rec;
        if (e < (unsigned)kSlots) return (const float*)(tile_data + (size_t)n_tiles * kTileIdBytes + (size_t)tile * kTileSlotBytes + (size_t)e * kEntryBytes);
        return (const float*)(tile_data + (size_t)over_base + (size_t)(tile_off[tile] + (int)e - 1) * kEntryBytes);
    }, (int)op);
}


bool warp_bin_geometry(int w, int h) {
    return w >= 8 && h >= 2 && w <= 16384 && h <= 16384 && (long long)((w + 127) & ~127) * h * 3 + 16 < (1ll << 31);      // (the outputs' rows: level_pitch)
}

int warp_bin_tile_width(int w, int h) {
    static const int forced = getenv("POPPY_TILE_W") ? atoi(getenv("POPPY_TILE_W")) : 0;
    if (forced == 64 || forced == 128) return forced;
    return (long long)w * h >= 4000000 ? 128 : 64;
}
size_t warp_bin_entry_bytes() { return kEntryBytes; }
int warp_bin_max_tile_entries() { return kMaxTileEntries; }
size_t warp_bin_data_bytes(size_t n_tiles, size_t max_entries) { return n_tiles * (kTileIdBytes + kTileSlotBytes) + max_entries * kEntryBytes + 256; }

void launch_tile_expand(const float* records, const void* raster_tris, const void* outline, const int* tile_off, const uint16_t* tile_tris,
                        void* tile_data, int tile_w, int w, int h, hipStream_t s) {
#define LE(TW) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW); \
    hipLaunchKernelGGL(k_tile_expand<TW>, dim3(tiles_x * tiles_y), dim3(256), 0, s, (const float4*)records, (const RasterTriDev*)raster_tris, \
                       (const int4*)outline, tile_off, tile_tris, (uint8_t*)tile_data, w, tiles_x, tiles_x * tiles_y); }
    if (tile_w == 128) LE(128) else LE(64)
#undef LE
}

void launch_warp_bin(const float* records, const void* tile_data, size_t tile_data_bytes, const int* tile_off, int tile_w,
                     const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2,
                     int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
#define LB(TW) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW); \
    hipExtLaunchKernelGGL(((w & 3) ? k_warp_bin<TW, false> : k_warp_bin<TW, true>), dim3(tiles_x * tiles_y), dim3(256), 0, s, t0, t1, 0, (const float4*)records, \
                          (const uint8_t*)tile_data, tile_off, c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, \
                          w, h, tiles_x, (uint32_t)std::min<size_t>(tile_data_bytes, 0xfffffff0u), ex, \
                          (unsigned)(tiles_x * tiles_y) <= kOneRoundBlocks ? stagger_flag(0) : stagger_flag(8)); }
    if (tile_w == 128) LB(128) else LB(64)
#undef LB
}

}  // namespace poppy_hip
