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
        if (num > 0) m = (num + 2 * major - 1) / (2 * major);
        iv.lo = iv.hi = x0 + m;
    } else {                                                     // shallow: row d holds the run of steps with m_k == d
        if (d < 0 || d > minor) return iv;
        const int k_lo = d == 0 ? 0 : (int)((unsigned)(major * (2 * d - 1)) / (unsigned)(2 * minor)) + 1;
        const int k_hi = d == minor ? major : (int)((unsigned)(major * (2 * d + 1)) / (unsigned)(2 * minor));
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

template <int kTileW>
__global__ void __launch_bounds__(256) k_tile_expand(const float4* __restrict__ rec, const RasterTriDev* __restrict__ tris,
                                                     const int4* __restrict__ outline, const int* __restrict__ tile_off,
                                                     const uint16_t* __restrict__ tile_tris, uint8_t* __restrict__ tile_data,
                                                     int W, int tiles_x, int n_tiles) {
    constexpr int kTileH = 1024 / kTileW, kPass = 256 / kTileH, kWords = kTileW / 64, kTileTx = kTileW / 4;
    __shared__ __attribute__((aligned(16))) uint64_t s_mask[kPass * kTileH * kWords];     // [entry of the pass][row][word]
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
    const int ji = tid / kTileH, jr = tid % kTileH;              // raster job: (entry ji of the pass, tile row jr)
    const int row = tid / kTileTx, xg = tid % kTileTx;           // resolve job: pixels (4 xg .. 4 xg + 3, row) of the tile
    const int mdword = row * kWords * 2 + ((xg * 4) >> 5), shift = (xg * 4) & 31;
    const uint32_t* s_mask32 = (const uint32_t*)s_mask;
    uint32_t ids = 0;                                            // the four pixels' entry numbers, one per byte
    for (int base = 0; base < nb; base += kPass) {
        const int n_here = min(kPass, nb - base);
        if (base) __syncthreads();                               // the previous pass's masks have been read
        if (ji < n_here) {
            const int t = tile_tris[off0 + base + ji];
            const RasterTriDev r = tris[t];
            const int yy = ty0 + jr;
            uint64_t m0 = 0, m1 = 0;
            add_interval<kTileW>(fill_row(r, yy, W), tx0, m0, m1);
#pragma unroll
            for (int e = 0; e < 3; ++e) add_interval<kTileW>(outline_row(outline[t * 3 + e], yy), tx0, m0, m1);
            uint64_t* dst = s_mask + (ji * kTileH + jr) * kWords;
            dst[0] = m0;
            if (kWords > 1) dst[kWords - 1] = m1;
        }
        __syncthreads();
        // painter's order, a later entry overwrites: an entry's four coverage bits become the byte selector of one v_perm between
        // the old bytes and the entry's number
        for (int i = 0; i < n_here; ++i) {
            const uint32_t bits = (s_mask32[i * kTileH * kWords * 2 + mdword] >> shift) & 15u;
            const uint32_t spread = __umul24(bits, 0x204081u) & 0x01010101u;          // bit k of `bits` -> bit 0 of byte k
            ids = __builtin_amdgcn_perm((uint32_t)(base + i + 1) * 0x01010101u, ids, (spread << 2) | 0x03020100u);
        }
    }
    ((uint32_t*)tile_data)[(size_t)tile * 256 + tid] = ids;
}

template <int kTileW>
__global__ void __launch_bounds__(256, POPPY_WARP_WAVES) k_warp_bin(const float4* __restrict__ rec, const uint8_t* __restrict__ tile_data,
                                                  const int* __restrict__ tile_off,
                                                  const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                  uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H,
                                                  int tiles_x, uint32_t data_bytes, WarpExtras ex) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    __shared__ float4 s_rec[kSlots * 5];
    __shared__ uint32_t s_ids[256];

    const int tid = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x), n_tiles = (int)gridDim.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int row = tid / kTileTx, xg = tid % kTileTx;           // this thread's pixels: (tx0 + 4 xg .. + 3, ty0 + row)
    const int x0 = tx * kTileW + xg * 4, y = ty * kTileH + row;
    const bool active = x0 < W && y < H;
    const uint32_t g = (uint32_t)y * (uint32_t)(W >> 2) + (uint32_t)(x0 >> 2);
    const uint32_t pitch = (uint32_t)W * 3u, npx = (uint32_t)W * (uint32_t)H;
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
    if (active && ex.m2) m2v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(ex.m2, npx * 4u), g * 16u, 0, 0));
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
    warp_fetch_blend_store(t, rs1, rs2, ro1, ro2, pitch, g, c1, c2, tr1, tr2, W, H, x0, y, [&](int k) -> const float* {
        const unsigned e = (s_ids[tid] >> (8 * k)) & 255u;
        if (e == 0) return (const float*)rec;
        if (e < (unsigned)kSlots) return (const float*)(tile_data + (size_t)n_tiles * kTileIdBytes + (size_t)tile * kTileSlotBytes + (size_t)e * kEntryBytes);
        return (const float*)(tile_data + (size_t)over_base + (size_t)(tile_off[tile] + (int)e - 1) * kEntryBytes);
    });
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k_warp_pipe (round 4): the same arithmetic as k_warp_bin, PERSISTENT and software-pipelined.  k_warp_bin's duration is a
// workgroup's lifetime paid once per round of resident workgroups (profiles/r03_notes.md section 3): launch, a round trip for
// ids + records, a barrier, ~1100 issue cycles of map arithmetic, the gathers' round trip, ~800 cycles of blends, stores — with every
// round trip covered only by OTHER waves.  Here a grid of 256 CUs x kPipeWaves workgroups stays resident and every WAVE walks an
// XCD-local run of tiles on its own (wave w of a workgroup takes rows w of every tile of the workgroup's run; no barrier anywhere:
// each wave keeps a private, double-buffered copy of the tile's record slots in LDS):
//     ids + record slots of tile i + 2      in flight (only the slots the tile uses: its entry count comes from tile_off by scalar
//                                           loads, so the 2.5 B/px of slot reads drop to ~0.4)
//     map arithmetic of unit u + 1          computed WHILE the footprint gathers of unit u are in flight (a unit = kUnit of a thread's
//                                           four pixels; the unit after a tile's last is the next tile's first)
//     blends + stores of unit u             behind them
// so a wave has ~1000 issue cycles of its own arithmetic behind every memory round trip, whatever the other waves do.
// Results are those of k_warp_bin bit for bit (same device functions); the border redo and the overflow records are unchanged.
constexpr int kWarpVariantDefault = 0;

struct PipeTap { uint32_t wt, wb, off; };

template <int kTileW, int kUnit, int kWaves>
__global__ void __launch_bounds__(256, kWaves) k_warp_pipe(const float4* __restrict__ rec, const uint8_t* __restrict__ tile_data,
                                                           const int* __restrict__ tile_off,
                                                           const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                           uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H,
                                                           int tiles_x, int n_tiles, uint32_t data_bytes) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4, kNU = 4 / kUnit;
    constexpr int kStage = 2;                                    // float4 per lane staged ahead: the first 25 slots; more (rare) are fetched at use
    __shared__ float4 s_rec[4][2][kSlots * 5];                   // [wave][buffer]: 20 KB per workgroup

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = tid / kTileTx, xg = tid % kTileTx;
    // this workgroup's run: XCD x = blockIdx % 8 owns a contiguous range of tiles, its workgroups stride through it side by side
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nwx = (int)gridDim.x >> 3;
    const int per = n_tiles >> 3, rem = n_tiles & 7;
    const int base = x * per + (x < rem ? x : rem), cnt = per + (x < rem ? 1 : 0);
    if (j >= cnt) return;
    const int n_mine = (cnt - j + nwx - 1) / nwx;
    const int step_q = nwx / tiles_x, step_r = nwx - step_q * tiles_x;

    const uint32_t pitch = (uint32_t)W * 3u, npx = (uint32_t)W * (uint32_t)H;
    const __amdgpu_buffer_rsrc_t rdata = make_rsrc(tile_data, data_bytes);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(c1, pitch * (uint32_t)H + 16u), rs2 = make_rsrc(c2, pitch * (uint32_t)H + 16u);
    const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(tr1, npx * 3u), ro2 = make_rsrc(tr2, npx * 3u);
    const uint32_t slot_base = (uint32_t)n_tiles * kTileIdBytes, over_base = (uint32_t)n_tiles * (kTileIdBytes + kTileSlotBytes);

    // float4s of tile t's slots that are in use (slot 0 = identity, slot e + 1 = entry e), at most all 32 slots; 0 = no such tile
    auto used4 = [&](int t, bool exists) -> int {
        if (!exists) return 0;
        const int nb = tile_off[t + 1] - tile_off[t];
        return (nb + 1 < kSlots ? nb + 1 : kSlots) * 5;
    };
    constexpr uint32_t kNowhere = 0xfffffff0u;                  // beyond every descriptor's range: such a load moves nothing and returns 0, such a store is dropped
    auto load_ids = [&](int t, bool exists) -> uint32_t {
        return __builtin_amdgcn_raw_buffer_load_b32(rdata, exists ? (uint32_t)t * kTileIdBytes + (uint32_t)tid * 4u : kNowhere, 0, 0);
    };
    // no branch around a load: lanes beyond the tile's used slots ask for an address outside the descriptor
    auto load_stage = [&](int t, int n4, float4 (&st)[kStage]) {
#pragma unroll
        for (int s = 0; s < kStage; ++s)
            st[s] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        rdata, lane + 64 * s < n4 ? slot_base + (uint32_t)t * kTileSlotBytes + (uint32_t)(lane + 64 * s) * 16u : kNowhere, 0, 0));
    };
    auto put_stage = [&](int t, int n4, const float4 (&st)[kStage], float4* dst) {
#pragma unroll
        for (int s = 0; s < kStage; ++s) dst[lane + 64 * s] = st[s];
        if (n4 > 64 * kStage && lane < kSlots * 5 - 64 * kStage)  // slots 25..31 of a crowded tile: fetched now (wave-uniform branch, rare)
            dst[lane + 64 * kStage] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                                          rdata, slot_base + (uint32_t)t * kTileSlotBytes + (uint32_t)(lane + 64 * kStage) * 16u, 0, 0));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // map arithmetic of unit u of a tile: pixels 4 xg + u kUnit .. of row `y`.  A pixel of an entry beyond the 32 slots (a tile crossed by more
    // than 31 triangles) is mapped by the identity here and marked for the byte-wise redo, which reads its record from the overflow area.
    auto unit_taps = [&](uint32_t ids, const float4* sr, int x0, int y, int u, PipeTap (&tp)[2][kUnit], uint32_t& edges) {
        const float fy = (float)y;
#pragma unroll
        for (int q = 0; q < kUnit; ++q) {
            const int k = u * kUnit + q;
            int li = (int)((ids >> (8 * k)) & 255u);
#ifndef POPPY_WARP_COUNT_MAIN
            if (li >= kSlots) { li = 0; edges |= 17u << k; }
#endif
            const float4 A = sr[li * 5], B = sr[li * 5 + 1], C = sr[li * 5 + 2], D = sr[li * 5 + 3], e4 = sr[li * 5 + 4];
            FastTap a, b;
            warp_taps(A, B, C, D, f2{e4.x, e4.y}, (float)(x0 + k), fy, W, H, a, b);
            tp[0][q] = PipeTap{a.wt, a.wb, a.off}; tp[1][q] = PipeTap{b.wt, b.wb, b.off};
            edges |= (a.inside ? 0u : 1u << k) | (b.inside ? 0u : 16u << k);
        }
    };
    auto unit_gather = [&](const PipeTap (&tp)[2][kUnit], u3v (&ra)[2][kUnit], u3v (&rb)[2][kUnit]) {
#pragma unroll
        for (int q = 0; q < kUnit; ++q) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const uint32_t o4 = tp[im][q].off & ~3u;
                ra[im][q] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, 0, 0);
                rb[im][q] = __builtin_amdgcn_raw_buffer_load_b96(im ? rs2 : rs1, o4, (int)pitch, 0);
            }
        }
    };
    auto unit_blend = [&](const PipeTap (&tp)[2][kUnit], const u3v (&ra)[2][kUnit], const u3v (&rb)[2][kUnit], int u, uint32_t (&p)[2][4]) {
#pragma unroll
        for (int q = 0; q < kUnit; ++q) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const uint32_t bs = tp[im][q].off & 3u;
                const u3v a3 = ra[im][q], b3 = rb[im][q];
                const u2v a = {__builtin_amdgcn_alignbyte(a3.y, a3.x, bs), __builtin_amdgcn_alignbyte(a3.z, a3.y, bs)};
                const u2v b = {__builtin_amdgcn_alignbyte(b3.y, b3.x, bs), __builtin_amdgcn_alignbyte(b3.z, b3.y, bs)};
                FastTap ft; ft.wt = tp[im][q].wt; ft.wb = tp[im][q].wb; ft.off = 0; ft.inside = true;
                p[im][u * kUnit + q] = blend_fast(ft, a, b);
            }
        }
    };

    float4* const buf0 = s_rec[wv][0];
    float4* const buf1 = s_rec[wv][1];

    // ---- prologue: tile 0 staged and mapped, tile 1 on its way ------------------------------------------------------------
    int t_cur = base + j, ty_c = t_cur / tiles_x, tx_c = t_cur - ty_c * tiles_x;
    int t_nxt = t_cur + nwx, ty_n = ty_c + step_q, tx_n = tx_c + step_r;
    if (tx_n >= tiles_x) { tx_n -= tiles_x; ++ty_n; }
    int t_nn = t_nxt + nwx;
    uint32_t ids_c = load_ids(t_cur, true);
    float4 st[kStage];
    int n4 = used4(t_cur, true);
    load_stage(t_cur, n4, st);
    put_stage(t_cur, n4, st, buf0);
    uint32_t ids_n = load_ids(t_nxt, n_mine > 1);
    n4 = used4(t_nxt, n_mine > 1);
    load_stage(t_nxt, n4, st);
    int x0_c = tx_c * kTileW + xg * 4, y_c = ty_c * kTileH + row;
    PipeTap tp[2][kUnit];
    uint32_t edges_c = 0;
    unit_taps(ids_c, buf0, x0_c, y_c, 0, tp, edges_c);

    for (int i = 0; i < n_mine; ++i) {
        const bool has_next = i + 1 < n_mine, has_nn = i + 2 < n_mine;
        float4* const sr_c = (i & 1) ? buf1 : buf0;
        float4* const sr_n = (i & 1) ? buf0 : buf1;
        const int x0_n = tx_n * kTileW + xg * 4, y_n = ty_n * kTileH + row;
        uint32_t edges_n = 0, p[2][4];
        uint32_t ids_nn = 0;
#pragma unroll
        for (int u = 0; u < kNU; ++u) {
            u3v ra[2][kUnit], rb[2][kUnit];
            unit_gather(tp, ra, rb);
            __builtin_amdgcn_sched_barrier(0);                   // the scheduler would gather every unit's loads at the top and sink every blend to the bottom (spilling the footprints)
            PipeTap tq[2][kUnit];
            if (u == 0) {                                        // tile i + 1's records into the other buffer; tile i + 2 requested
                put_stage(t_nxt, n4, st, sr_n);
                ids_nn = load_ids(t_nn, has_nn);
                n4 = used4(t_nn, has_nn);
                load_stage(t_nn, n4, st);
            }
            if (u + 1 < kNU) unit_taps(ids_c, sr_c, x0_c, y_c, u + 1, tq, edges_c);
            else if (has_next) unit_taps(ids_n, sr_n, x0_n, y_n, 0, tq, edges_n);
            __builtin_amdgcn_sched_barrier(0);
            unit_blend(tp, ra, rb, u, p);
            // ... and the optimiser sinks the blends (pure arithmetic) to their use at the tile's end: an empty asm pins each result here
#pragma unroll
            for (int q = 0; q < kUnit; ++q) { asm volatile("" : "+v"(p[0][u * kUnit + q])); asm volatile("" : "+v"(p[1][u * kUnit + q])); }
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < kNU || has_next) {
#pragma unroll
                for (int q = 0; q < kUnit; ++q) { tp[0][q] = tq[0][q]; tp[1][q] = tq[1][q]; }
            }
        }
        // ---- tile i leaves --------------------------------------------------------------------------------------------
        const bool active = x0_c < W && y_c < H;
        const uint32_t g = (uint32_t)y_c * (uint32_t)(W >> 2) + (uint32_t)(x0_c >> 2);
        const u3v o1 = {p[0][0] | (p[0][1] << 24), (p[0][1] >> 8) | (p[0][2] << 16), (p[0][2] >> 16) | (p[0][3] << 8)};
        const u3v o2 = {p[1][0] | (p[1][1] << 24), (p[1][1] >> 8) | (p[1][2] << 16), (p[1][2] >> 16) | (p[1][3] << 8)};
        __builtin_amdgcn_raw_buffer_store_b96(o1, ro1, active ? g * 12u : kNowhere, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b96(o2, ro2, active ? g * 12u : kNowhere, 0, 0);
#ifndef POPPY_WARP_COUNT_MAIN
        if (!active) edges_c = 0;
        if (__builtin_amdgcn_ballot_w64(edges_c != 0) != 0 && edges_c != 0) {
            for (int e = 0; e < 8; ++e) {
                if (!((edges_c >> e) & 1u)) continue;
                const int k = e & 3, im = e >> 2;
                const unsigned en = (ids_c >> (8 * k)) & 255u;
                const float* rp = en == 0 ? (const float*)rec
                                : en < (unsigned)kSlots ? (const float*)(tile_data + (size_t)slot_base + (size_t)t_cur * kTileSlotBytes + (size_t)en * kEntryBytes)
                                : (const float*)(tile_data + (size_t)over_base + (size_t)(tile_off[t_cur] + (int)en - 1) * kEntryBytes);
                const uint32_t v = slow_pixel(rp, im, im ? c2 : c1, W, H, x0_c + k, y_c);
                uint8_t* dst = (uint8_t*)(im ? tr2 : tr1) + ((size_t)y_c * W + x0_c + k) * 3;
                dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16);
            }
        }
#endif
        // ---- rotate ---------------------------------------------------------------------------------------------------
        t_cur = t_nxt; tx_c = tx_n; ty_c = ty_n; x0_c = x0_n; y_c = y_n; ids_c = ids_n; edges_c = edges_n;
        t_nxt = t_nn; ids_n = ids_nn;
        tx_n += step_r; ty_n += step_q;
        if (tx_n >= tiles_x) { tx_n -= tiles_x; ++ty_n; }
        t_nn = t_nxt + nwx;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k_warp_run (round 4): what the measurements of k_warp_pipe left standing.  Pipelining the gathers inside a wave costs registers, hence
// waves per SIMD, and on this part a SIMD needs its 8 waves (a wave alone issues one instruction per ~5 cycles whatever its dependencies:
// profiles/r03_notes.md section 1) — every k_warp_pipe variant was slower than k_warp_bin.  This kernel keeps k_warp_bin's instruction stream
// and its 8 waves per SIMD and removes only the per-tile fixed part: a resident grid, every WAVE walking its own run of tiles, the next
// tile's ids (one VGPR) and its record slots requested while the current tile is being warped — the slots by LDS-DMA (buffer_load ... lds:
// no staging registers), only the slots the tile uses, into the wave's other LDS buffer.  No barrier, no workgroup launch per tile, no
// exposed first round trip.

// The byte-wise redo of one pixel (slow_pixel) through buffer descriptors only: the persistent kernels keep no raw pointer alive across
// their loop (the scalar registers are what limits them to 8 waves per SIMD).  rec_off = byte offset of the pixel's record in rdata.
__device__ __forceinline__ uint32_t slow_pixel_rsrc(__amdgpu_buffer_rsrc_t rdata, uint32_t rec_off, int src, __amdgpu_buffer_rsrc_t img, int W, int H, int x, int y) {
    auto rf = [&](int i) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdata, rec_off + (uint32_t)i * 4u, 0, 0)); };
    float h[9];
    if (src == 0) { h[0] = rf(0); h[3] = rf(1); h[1] = rf(2); h[4] = rf(3); h[2] = rf(4); h[5] = rf(5); }
    else          { h[0] = rf(6); h[3] = rf(7); h[1] = rf(8); h[4] = rf(9); h[2] = rf(10); h[5] = rf(11); }
    h[6] = rf(12 + src); h[7] = rf(14 + src); h[8] = rf(16 + src);
    float mx, my;
    map_point(h, x, y, mx, my);
    const int sx = cv_round_x86(mx * 32.f), sy = cv_round_x86(my * 32.f);
    int ix = sx >> 5, iy = sy >> 5;
    ix = max(-32768, min(32767, ix)); iy = max(-32768, min(32767, iy));
    int w00, w01, w10, w11;
    bilinear_weights(sx & 31, sy & 31, w00, w01, w10, w11);
    const bool x0 = (unsigned)ix < (unsigned)W, x1 = (unsigned)(ix + 1) < (unsigned)W;
    const bool y0 = (unsigned)iy < (unsigned)H, y1 = (unsigned)(iy + 1) < (unsigned)H;
    const uint32_t o00 = (uint32_t)(iy * W + ix) * 3u, o10 = o00 + (uint32_t)W * 3u;
    auto px = [&](bool ok, uint32_t o) -> int { return ok ? (int)__builtin_amdgcn_raw_buffer_load_b8(img, o, 0, 0) : 0; };
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int acc = __mul24(px(x0 && y0, o00 + k), w00) + __mul24(px(x1 && y0, o00 + 3 + k), w01) +
                        __mul24(px(x0 && y1, o10 + k), w10) + __mul24(px(x1 && y1, o10 + 3 + k), w11);
        out |= (uint32_t)sat_u8((acc + (1 << 14)) >> 15) << (8 * k);
    }
    return out;
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// lanes [0, n) of this wave copy 16 bytes each from rsrc + voff to LDS at lds_dst + 16 * lane (wave-uniform lds_dst); invisible to the
// compiler's vmcnt bookkeeping: issue it BEFORE a load the compiler waits for (buffer loads return in order)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rsrc) : "memory");
}

template <int kTileW, int kWaves>
__global__ void __launch_bounds__(256, kWaves) k_warp_run(const float4* __restrict__ rec, const uint8_t* __restrict__ tile_data,
                                                     const int* __restrict__ tile_off,
                                                     const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                     uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H,
                                                     int tiles_x, int n_tiles, uint32_t data_bytes) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    __shared__ __attribute__((aligned(16))) float4 s_rec[4][2][kSlots * 5];   // [wave][buffer]: 20 KB per workgroup, 8 workgroups per CU

    const int tid0 = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nwx = (int)gridDim.x >> 3;
    const int per = n_tiles >> 3, rem = n_tiles & 7;
    const int base = x * per + (x < rem ? x : rem), cnt = per + (x < rem ? 1 : 0);
    if (j >= cnt) return;
    const int n_mine = (cnt - j + nwx - 1) / nwx;
    const int step_q = nwx / tiles_x, step_r = nwx - step_q * tiles_x;

    const uint32_t pitch = (uint32_t)W * 3u, npx = (uint32_t)W * (uint32_t)H;
    const __amdgpu_buffer_rsrc_t rdata = make_rsrc(tile_data, data_bytes);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(c1, pitch * (uint32_t)H + 16u), rs2 = make_rsrc(c2, pitch * (uint32_t)H + 16u);
    const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(tr1, npx * 3u), ro2 = make_rsrc(tr2, npx * 3u);
    const uint32_t slot_base = (uint32_t)n_tiles * kTileIdBytes, over_base = (uint32_t)n_tiles * (kTileIdBytes + kTileSlotBytes);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_void_ptr)&s_rec[wv][0][0]);
    constexpr uint32_t kBufBytes = kSlots * 5 * 16;

    // the slots tile t uses (slot 0 = identity, slot e + 1 = entry e), by LDS-DMA into buffer b of this wave
    // (the lane-derived offsets are recomputed from the thread number wherever they are used — `tid` is made opaque once per tile —: hoisted out of
    // the loop they were spilled to scratch and reloaded every tile, this kernel has no register to spare at 8 waves per SIMD)
    auto request_slots = [&](int t, int b, int tid) {
        const int lane = tid & 63;
        const int nb = tile_off[t + 1] - tile_off[t];
        const int n4 = (nb + 1 < kSlots ? nb + 1 : kSlots) * 5;
        const uint32_t src = slot_base + (uint32_t)t * kTileSlotBytes + (uint32_t)lane * 16u, dst = lds0 + (uint32_t)b * kBufBytes;
        if (lane < n4) lds_dma16(rdata, src, dst);
        if (n4 > 64) {
            if (lane + 64 < n4) lds_dma16(rdata, src + 1024u, dst + 1024u);
            if (n4 > 128 && lane + 128 < n4) lds_dma16(rdata, src + 2048u, dst + 2048u);
        }
    };
    auto load_ids = [&](int t, int tid) -> uint32_t {
        return __builtin_amdgcn_raw_buffer_load_b32(rdata, (uint32_t)t * kTileIdBytes + (uint32_t)tid * 4u, 0, 0);
    };

    int t_cur = base + j, ty = t_cur / tiles_x, tx = t_cur - ty * tiles_x;
    request_slots(t_cur, 0, tid0);
    uint32_t ids = load_ids(t_cur, tid0);
    for (int i = 0; i < n_mine; ++i) {
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const int tid = tid0 + zero, row = tid / kTileTx, xg = tid % kTileTx;
        const float4* const sr = s_rec[wv][i & 1];
        const int t_nxt = t_cur + nwx;
        uint32_t ids_n = 0;
        if (i + 1 < n_mine) {                                    // wave-uniform: the next tile's slots and ids are on their way while this one is warped
            request_slots(t_nxt, (i + 1) & 1, tid);
            ids_n = load_ids(t_nxt, tid);
        }
        const int x0 = tx * kTileW + xg * 4, y = ty * kTileH + row;
        const bool active = x0 < W && y < H;
        const uint32_t g = (uint32_t)y * (uint32_t)(W >> 2) + (uint32_t)(x0 >> 2);
        const float fy = (float)y;
        FastTap t[2][4];
        uint32_t far = 0;                                        // pixels of entries beyond the 32 slots: identity here, exact in the redo below
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int li = (int)((ids >> (8 * k)) & 255u);
#ifndef POPPY_WARP_COUNT_MAIN
            if (li >= kSlots) { li = 0; far |= 17u << k; }
#endif
            const float4 A = sr[li * 5], B = sr[li * 5 + 1], C = sr[li * 5 + 2], D = sr[li * 5 + 3], e4 = sr[li * 5 + 4];
            warp_taps(A, B, C, D, f2{e4.x, e4.y}, (float)(x0 + k), fy, W, H, t[0][k], t[1][k]);
        }
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
        constexpr uint32_t kNowhere = 0xfffffff0u;               // beyond the descriptor: dropped
        __builtin_amdgcn_raw_buffer_store_b96(o1, ro1, active ? g * 12u : kNowhere, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b96(o2, ro2, active ? g * 12u : kNowhere, 0, 0);
#ifndef POPPY_WARP_COUNT_MAIN
        uint32_t edges = far;
#pragma unroll
        for (int k = 0; k < 4; ++k) edges |= (t[0][k].inside ? 0u : 1u << k) | (t[1][k].inside ? 0u : 16u << k);
        if (!active) edges = 0;
        if (__builtin_amdgcn_ballot_w64(edges != 0) != 0 && edges != 0) {
            for (int e = 0; e < 8; ++e) {
                if (!((edges >> e) & 1u)) continue;
                const int k = e & 3, im = e >> 2;
                const unsigned en = (ids >> (8 * k)) & 255u;          // slot 0 holds the identity record
                const uint32_t ro = en < (unsigned)kSlots ? slot_base + (uint32_t)t_cur * kTileSlotBytes + en * kEntryBytes
                                                          : over_base + (uint32_t)(tile_off[t_cur] + (int)en - 1) * kEntryBytes;
                const uint32_t v = slow_pixel_rsrc(rdata, ro, im, im ? rs2 : rs1, W, H, x0 + k, y);
                const uint32_t po = ((uint32_t)y * (uint32_t)W + (uint32_t)(x0 + k)) * 3u;
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)v, im ? ro2 : ro1, po, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(v >> 8), im ? ro2 : ro1, po + 1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(v >> 16), im ? ro2 : ro1, po + 2, 0, 0);
            }
        }
#endif
        t_cur = t_nxt; ids = ids_n;
        tx += step_r; ty += step_q;
        if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// k_warp_probe: k_warp_bin with its phases fenced and stamped (s_memtime), for tools/experiments/warp_probe.py.  Per wave 8 values:
// entry, after the barrier (ids + records there), after the map arithmetic, after the gathers are issued, after they have all
// returned, after the blends + stores are issued, HW_ID, XCC_ID.  A measuring aid: same results, slightly different schedule
// (the blends wait for ALL gathers here).
__device__ __forceinline__ uint64_t stamp() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int kTileW>
__global__ void __launch_bounds__(256, POPPY_WARP_WAVES) k_warp_probe(const float4* __restrict__ rec, const uint8_t* __restrict__ tile_data,
                                                  const int* __restrict__ tile_off,
                                                  const uint8_t* __restrict__ c1, const uint8_t* __restrict__ c2,
                                                  uint32_t* __restrict__ tr1, uint32_t* __restrict__ tr2, int W, int H,
                                                  int tiles_x, uint32_t data_bytes, uint64_t* __restrict__ probe) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    __shared__ float4 s_rec[kSlots * 5];
    const uint64_t ts0 = stamp();
    const int tid = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x), n_tiles = (int)gridDim.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int row = tid / kTileTx, xg = tid % kTileTx;
    const int x0 = tx * kTileW + xg * 4, y = ty * kTileH + row;
    const bool active = x0 < W && y < H;
    const uint32_t g = (uint32_t)y * (uint32_t)(W >> 2) + (uint32_t)(x0 >> 2);
    const uint32_t pitch = (uint32_t)W * 3u, npx = (uint32_t)W * (uint32_t)H;
    const __amdgpu_buffer_rsrc_t rdata = make_rsrc(tile_data, data_bytes);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(c1, pitch * (uint32_t)H + 16u), rs2 = make_rsrc(c2, pitch * (uint32_t)H + 16u);
    const __amdgpu_buffer_rsrc_t ro1 = make_rsrc(tr1, npx * 3u), ro2 = make_rsrc(tr2, npx * 3u);
    const uint32_t ids = __builtin_amdgcn_raw_buffer_load_b32(rdata, (uint32_t)tile * kTileIdBytes + (uint32_t)tid * 4u, 0, 0);
    if (tid < kSlots * 5)
        s_rec[tid] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                         rdata, (uint32_t)n_tiles * kTileIdBytes + (uint32_t)tile * kTileSlotBytes + (uint32_t)tid * 16u, 0, 0));
    __syncthreads();
    uint32_t idsv = ids;
    asm volatile("" : "+v"(idsv));
    const uint64_t ts1 = stamp();
    const float fy = (float)y;
    FastTap t[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int li = (int)((idsv >> (8 * k)) & 255u);
        if (li >= kSlots) li = 0;                                // (timing aid: crowded tiles are not what it is run on)
        const float4 A = s_rec[li * 5], B = s_rec[li * 5 + 1], C = s_rec[li * 5 + 2], D = s_rec[li * 5 + 3], e4 = s_rec[li * 5 + 4];
        warp_taps(A, B, C, D, f2{e4.x, e4.y}, (float)(x0 + k), fy, W, H, t[0][k], t[1][k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(t[0][k].off), "+v"(t[0][k].wt), "+v"(t[0][k].wb)); asm volatile("" : "+v"(t[1][k].off), "+v"(t[1][k].wt), "+v"(t[1][k].wb)); }
    const uint64_t ts2 = stamp();
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
    const uint64_t ts3 = stamp();
    __builtin_amdgcn_s_waitcnt(0x0070);                          // vmcnt(0) (expcnt 7, lgkmcnt untouched: the stamp has drained it)
#pragma unroll
    for (int k = 0; k < 4; ++k) { asm volatile("" : "+v"(ra[0][k]), "+v"(rb[0][k])); asm volatile("" : "+v"(ra[1][k]), "+v"(rb[1][k])); }
    const uint64_t ts4 = stamp();
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
    constexpr uint32_t kNowhere = 0xfffffff0u;
    __builtin_amdgcn_raw_buffer_store_b96(o1, ro1, active ? g * 12u : kNowhere, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b96(o2, ro2, active ? g * 12u : kNowhere, 0, 0);
    const uint64_t ts5 = stamp();
    if ((tid & 63) == 0) {
        uint64_t* o = probe + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 8;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = ts4; o[5] = ts5;
        o[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));          // HW_REG_HW_ID
        o[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));         // HW_REG_XCC_ID
    }
}

// relaunches a frame's warp as k_warp_probe; `probe` takes 4 * 8 values per tile (device memory)
size_t warp_probe_values(int tile_w, int w, int h) {
    const int tiles_x = (w + tile_w - 1) / tile_w, tiles_y = (h + 1024 / tile_w - 1) / (1024 / tile_w);
    return (size_t)tiles_x * tiles_y * 32;
}
void launch_warp_probe(const float* records, const void* tile_data, size_t tile_data_bytes, const int* tile_off, int tile_w,
                       const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2, int w, int h, uint64_t* probe, hipStream_t s) {
    const uint32_t bytes = (uint32_t)std::min<size_t>(tile_data_bytes, 0xfffffff0u);
#define LQ(TW) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW); \
    hipLaunchKernelGGL(k_warp_probe<TW>, dim3(tiles_x * tiles_y), dim3(256), 0, s, (const float4*)records, (const uint8_t*)tile_data, tile_off, \
                       c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, w, h, tiles_x, bytes, probe); }
    if (tile_w == 128) LQ(128) else LQ(64)
#undef LQ
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

// Which kernel launch_warp_bin runs: 0 = k_warp_bin (a workgroup per tile), v > 0 = k_warp_pipe, persistent: bits 0-3 pixels per pipeline unit
// (1, 2, 4), bits 4-7 resident workgroups per CU.  POPPY_WARP_VARIANT at first use, warp_bin_set_variant() afterwards (A/B in one process).
static int g_warp_variant = -1;
static int g_cus = 0;
void warp_bin_set_variant(int v) { g_warp_variant = v; }
int warp_bin_variant() {
    if (g_warp_variant < 0) g_warp_variant = getenv("POPPY_WARP_VARIANT") ? (int)strtol(getenv("POPPY_WARP_VARIANT"), nullptr, 0) : kWarpVariantDefault;
    return g_warp_variant;
}

void launch_warp_bin(const float* records, const void* tile_data, size_t tile_data_bytes, const int* tile_off, int tile_w,
                     const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2,
                     int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0, hipEvent_t t1) {
    const int variant = ex.m2 ? 0 : warp_bin_variant();
    const uint32_t bytes = (uint32_t)std::min<size_t>(tile_data_bytes, 0xfffffff0u);
    if (variant > 0) {
        if (!g_cus) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) g_cus = pr.multiProcessorCount; else g_cus = 256; }
        if ((variant & 15) == 0) {                               // 0x80, 0x70, 0x60: k_warp_run with that many workgroups per CU
            const int wv = variant >> 4;
#define LR(TW, WV) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW), n_tiles = tiles_x * tiles_y; \
    const int grid = std::min((g_cus * WV) & ~7, (n_tiles + 7) & ~7); \
    hipExtLaunchKernelGGL((k_warp_run<TW, WV>), dim3(grid), dim3(256), 0, s, t0, t1, 0, (const float4*)records, (const uint8_t*)tile_data, tile_off, \
                          c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, w, h, tiles_x, n_tiles, bytes); return; }
            if (wv == 8) { if (tile_w == 128) LR(128, 8) else LR(64, 8) }
            if (wv == 7) { if (tile_w == 128) LR(128, 7) else LR(64, 7) }
            if (wv == 6) { if (tile_w == 128) LR(128, 6) else LR(64, 6) }
#undef LR
        }
        const int unit = variant & 15, waves = (variant >> 4) & 15;
#define LP(TW, U, WV) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW), n_tiles = tiles_x * tiles_y; \
    const int grid = std::min((g_cus * WV) & ~7, (n_tiles + 7) & ~7); \
    hipExtLaunchKernelGGL((k_warp_pipe<TW, U, WV>), dim3(grid), dim3(256), 0, s, t0, t1, 0, (const float4*)records, (const uint8_t*)tile_data, tile_off, \
                          c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, w, h, tiles_x, n_tiles, bytes); return; }
#define LPW(U, WV) { if (tile_w == 128) LP(128, U, WV) else LP(64, U, WV) }
        if (unit == 4 && waves == 4) LPW(4, 4)
        if (unit == 4 && waves == 3) LPW(4, 3)
        if (unit == 2 && waves == 4) LPW(2, 4)
        if (unit == 2 && waves == 5) LPW(2, 5)
        if (unit == 1 && waves == 5) LPW(1, 5)
        if (unit == 1 && waves == 6) LPW(1, 6)
#undef LPW
#undef LP
    }
#define LB(TW) { const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + 1024 / TW - 1) / (1024 / TW); \
    hipExtLaunchKernelGGL(k_warp_bin<TW>, dim3(tiles_x * tiles_y), dim3(256), 0, s, t0, t1, 0, (const float4*)records, \
                          (const uint8_t*)tile_data, tile_off, c1, c2, (uint32_t*)tr1, (uint32_t*)tr2, \
                          w, h, tiles_x, bytes, ex); }
    if (tile_w == 128) LB(128) else LB(64)
#undef LB
}

}  // namespace poppy_hip
