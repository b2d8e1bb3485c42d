// kernels_median_cols.hip — medianBlur on 8-bit single-channel images by COLUMN HISTOGRAMS, for the windows of Extractor::foreground's
// chain (src/extractor.cpp:136-161: ksize 9, 17, ..., 89).  The scheme is the one the reference itself uses for such windows
// (medianBlur_8u_O1, OCV/imgproc/src/median_blur.simd.hpp:84-346, after Perreault & Hebert): every image column keeps the histogram of
// its 2r+1 window rows; the window's histogram slides along the row by adding the entering column's histogram and subtracting the
// leaving one's, so a pixel costs the same whatever ksize is.  The median of a set is unique, so any exact scheme returns the
// reference's bytes; borders are replicated as medianBlur does (clamped rows and columns).
//
// How it sits on a CDNA4 compute unit
//   * Everything is CUMULATIVE and in RANKS.  A 256-bit presence map per tile (written by the median that made the source, or by
//     k_median_presence) says which values the tile's footprint can hold; their ranks 0 .. D-1 stand for them (ranking is monotone: the
//     median's rank is the rank of the median).  A column's table is the cumulative histogram of its window rows over the ranks, one
//     byte per rank (a count is at most 89); the window's table is the sum of its columns' tables.  With cumulative counts C the median's
//     rank is simply the NUMBER of ranks whose count does not exceed half the window: popcount(ballot(C <= half)) — no prefix scan, no
//     search inside a bin, no plain histogram at all.
//   * A workgroup (8 waves) owns a tile of 128 columns by `rows` rows.  The tables of the tile's columns + 2r live in LDS; a thread per
//     column steps them one row down: one value in, one out = the bytes between the two ranks change by one (a few dword atomics).
//   * A WAVE is one sliding window ("chain"): a lane holds the window's count of one rank (D <= 64) or of two (D <= 128, packed in one
//     register).  A step adds the entering and subtracts the leaving column's bytes, a ballot (two) gives the pixel's rank, v_writelane
//     puts it into lane i of a result register; after a run of 16 pixels the ranks go to LDS, are translated to values and stored.
//   * A chain restarts every row at its first pixel from the counts it had there one row up: the change of that window — 2r+1 values in,
//     2r+1 out — is collected by the chain's own lanes as plain counts in a small LDS table (two atomics per column that changed, runs of
//     equal columns combined) and its running sum over the lanes is the cumulative change.
//   * More than 129 different values in a tile's footprint (noise, photographs): the 128 counts cover a WINDOW of ranks, base .. base + 127 —
//     values ranked below the window count at its first rank, values above it have no count — and the pass's answer is the median's rank
//     clamped into the window: exact strictly inside it.  The window is laid around the median of a sample of the tile's own pixels (a
//     median filter's output varies little over 128 x 24 pixels); pixels that come out at the window's edge are noted in two bitmaps and
//     filtered again with the window below / above — a second or third pass of the same code that most tiles never need.  So the tables
//     stay at 132 bytes per column, a workgroup at 35 KB of LDS and 64 registers, and four workgroups share a compute unit (8 waves per
//     SIMD: the launches of a pair's two images overlap).
#include "kernels_prefilter.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace poppy_hip {

namespace {

constexpr int kMedPad = 48;                                  // replicated side columns of the padded source (kernels_prefilter.hip)
constexpr int kMcWaves = 8;                                  // chains per workgroup
constexpr int kMcRun = 16;                                   // pixels a chain takes per row (a tile is 128 columns wide; the run's loop is unrolled)
constexpr int kMcMaxR = 44;                                  // ksize 89
constexpr int kMcCols = kMcWaves * kMcRun + 2 * kMcMaxR;     // column tables per workgroup (216)
// dword stride of a column's table: odd, so that a thread per column (the row step) meets 32 different banks.  128 counts + one spare dword: a value
// of rank 128 (the clamp of the first of two passes) has no count — the last rank's cumulative count is the window's size and never asked for — and
// lands there.  One rank per lane uses the first 64 bytes.
constexpr int kTabStride = 33;
constexpr int kOffTab = 0;                                           // [col][128] bytes: the window rows' values of rank <= the byte's index
constexpr int kOffDelta = kOffTab + kMcCols * kTabStride;            // per chain: the change of the counts of its first window (bytes biased by 128; same stride)
constexpr int kOffRowIn = kOffDelta + kMcWaves * kTabStride;         // the row step's entering / leaving rank of every column
constexpr int kOffRowOut = kOffRowIn + (kMcCols + 3) / 4;
constexpr int kOffRank = kOffRowOut + (kMcCols + 3) / 4;             // value -> rank (256 bytes), rank -> value (128 bytes) of the current pass
constexpr int kOffInv = kOffRank + 64;                               // (129 entries: the answer 128 = the last rank, when the window reaches it)
constexpr int kOffPres = kOffInv + 33;                               // the footprint's presence bits, the output tile's
constexpr int kOffPresOut = kOffPres + 8;
constexpr int kOffOutRow = kOffPresOut + 8;                          // the tile's current output row (ranks), translated and stored by the waves that step no columns
constexpr int kOffMisc = kOffOutRow + kMcWaves * kMcRun / 4;         // [0] bit 0 / 1: some pixel's median lies below / above the first window; [1] the first window's base
constexpr int kMcMaxRows = 128;
constexpr int kOffBelow = kOffMisc + 4;                              // per row 128 bits: the pixels whose median lies below the first window; above it
constexpr int kOffAbove = kOffBelow + kMcMaxRows * 4;
constexpr int kMcLdsWords = kOffAbove + kMcMaxRows * 4;
static_assert(kMcCols <= 256, "the column threads are waves 0..3; waves 4..7 store the rows");
static_assert(kMcLdsWords * 4 <= 39 * 1024, "four workgroups per compute unit");

enum { kRank64 = 1, kRank128 = 2 };                                  // counts per lane x 64
enum { kPassOnly = 0, kPassFirst = 1, kPassBelow = 2, kPassAbove = 3 };   // the one pass of a tile of <= 128 values; a tile with more: the first window, the ones beside it

constexpr uint32_t kBias4 = 0x80808080u;

__device__ __forceinline__ uint32_t spread2(uint32_t x) { return (x & 0xffu) | ((x & 0xff00u) << 8); }      // bytes 0, 1 -> the two 16-bit halves

// a loop whose index is a compile-time constant in the body (LDS offsets and lane selects become immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// v_writelane_b32 (this compiler has no builtin for the instruction): lane kSel of the word becomes the wave-uniform value.  The lane select is an
// immediate (with two scalar operands the instruction would need it in M0); the value comes from the scalar unit as a DATA operand — the instruction's
// one listed hazard is a vector-written lane SELECT.
template <int kSel>
__device__ __forceinline__ void write_lane_imm(int& a, int va) { asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(a) : "s"(va), "i"(kSel)); }

// inclusive prefix sum over the 64 lanes of a wave (row shifts, then the two row broadcasts of the GCN data-parallel primitives)
__device__ __forceinline__ int wave_prefix_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// One value in, one value out of a column's cumulative table: an entry counts the values of rank <= its index, so entering rank ia adds one from ia on
// and leaving rank ib removes one from ib on — the bytes [min, max) change by +1 (ia < ib) or -1.  A thread owns its column's table; counts stay within
// 0..255 (a leaving value was counted).  max <= 128: the last byte touched is 127.
__device__ __forceinline__ void table_step(uint32_t* __restrict__ tab, int ia, int ib) {
    if (ia == ib) return;
    const int lo = min(ia, ib), hi = max(ia, ib);
    for (int d = lo >> 2; d <= (hi - 1) >> 2; ++d) {
        const int j0 = max(lo - 4 * d, 0), j1 = min(hi - 4 * d, 4);
        const uint32_t m = (0x01010101u >> (8 * (4 - (j1 - j0)))) << (8 * j0);
        if (ia < ib) atomicAdd(&tab[d], m); else atomicSub(&tab[d], m);
    }
}

// One pass over the tile.  kForm: counts per lane; pass: which pixels it writes (kPassFirst notes the others in the bitmaps); lo_exact / hi_exact:
// the answers 0 / 128 are the median's rank too (the window begins at rank 0 / reaches the last rank).
template <int kForm>
__device__ __forceinline__ void median_cols_pass(uint32_t* __restrict__ lds, const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst,
                                                 uint8_t* __restrict__ padded_out, int W, int H, int ksize, int rows, int pass, bool lo_exact, bool hi_exact, int& prev_m, int dbg) {
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const uint16_t* const lds16 = reinterpret_cast<const uint16_t*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = ksize >> 1, half = (ksize * ksize) >> 1;
    constexpr int Cw = kMcWaves * kMcRun;
    const int ncols = Cw + 2 * r;
    const int X0 = blockIdx.x * Cw, Y0 = blockIdx.y * rows, Yend = min(Y0 + rows, H);
    const int Wp = W + 2 * kMedPad;
    const uint8_t* const rank8 = lds8 + kOffRank * 4;
    const uint8_t* const inv8 = lds8 + kOffInv * 4;
    const bool col_thread = tid < ncols;
    const int gx = min(max(X0 - r + tid, 0), W - 1);
    const uint8_t* const colp = srcp + kMedPad + gx;                  // this thread's image column (threads past ncols: unused)
    auto row_of = [&](int yy) { return (size_t)min(max(yy, 0), H - 1) * Wp; };
    uint32_t* const my_tab = lds + kOffTab + tid * kTabStride;
    constexpr int kTabDwords = kForm == kRank128 ? 32 : 16;
    // ---- the column tables of the first window row: the counts of rows Y0 - r .. Y0 + r of every column by rank, then their running sums, in place ----
    if (col_thread && !(dbg & 1)) {
        for (int d = 0; d < kTabStride; ++d) my_tab[d] = 0;
        for (int rr0 = -r; rr0 <= r; rr0 += 16) {                     // sixteen rows' loads in flight, then their sixteen additions
            int idx[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) idx[j] = colp[row_of(Y0 + min(rr0 + j, r))];
#pragma unroll
            for (int j = 0; j < 16; ++j) idx[j] = rank8[idx[j]];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (rr0 + j <= r) atomicAdd(&my_tab[idx[j] >> 2], 1u << (8 * (idx[j] & 3)));
        }
        uint32_t below = 0;
        for (int d = 0; d < kTabDwords; ++d) {
            const uint32_t x = my_tab[d];
            const uint32_t b0 = below + (x & 255u), b1 = b0 + ((x >> 8) & 255u), b2 = b1 + ((x >> 16) & 255u), b3 = b2 + (x >> 24);
            my_tab[d] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            below = b3;
        }
    }
    uint32_t* const delta = lds + kOffDelta + wv * kTabStride;
    if (lane < kTabStride) delta[lane] = kBias4;
    __syncthreads();
    // ---- this wave's chain: the window's counts at its first pixel of row Y0 (two counts per lane: packed in one register) ----
    const int c_first = wv * kMcRun;                                  // local column of the window's left edge at the chain's first pixel
    uint32_t C0 = 0;
    for (int j = 0; j <= 2 * r && !(dbg & 16); ++j) {
        const int col = c_first + j;
        if (kForm == kRank128) C0 += spread2(lds16[(kOffTab + col * kTabStride) * 2 + lane]);
        else C0 += lds8[(kOffTab + col * kTabStride) * 4 + lane];
    }
    for (int y = Y0; y < Yend; ++y) {
        const bool more = y + 1 < Yend;
        int nv_in = 0, nv_out = 0;
        if (more && col_thread) { nv_in = colp[row_of(y + r + 1)]; nv_out = colp[row_of(y - r)]; }      // the row step's values, asked for early
        // ---- the run: 16 pixels of row y.  A step to pixel i: column c_first + i + 2r enters, column c_first + i - 1 leaves.  Packed counts: odd
        // steps add 128 + (in - out) per count, even steps subtract 128 - (in - out) — a byte-wise difference without borrows, and the 128s cancel
        // every second step (an odd pixel's ballots compare against half + 128).  The pixel index is a compile-time constant: the columns' LDS
        // offsets and the result's lane are immediates; the columns of the next step are asked for before the current pixel's ballots.
        uint32_t C = C0;
        int res = 0;
        struct ColPair { uint32_t a, s; };
        const int in0 = c_first + 2 * r, out0 = c_first - 1;          // columns of step 0 (step i: + i)
        const uint8_t* const one_in = lds8 + (kOffTab + in0 * kTabStride) * 4 + lane, * const one_out = lds8 + (kOffTab + out0 * kTabStride) * 4 + lane;
        const uint16_t* const two_in = lds16 + (kOffTab + in0 * kTabStride) * 2 + lane, * const two_out = lds16 + (kOffTab + out0 * kTabStride) * 2 + lane;
        auto fetch = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ColPair k;
            if (kForm == kRank128) { k.a = two_in[i * kTabStride * 2]; k.s = two_out[i * kTabStride * 2]; }
            else { k.a = one_in[i * kTabStride * 4]; k.s = one_out[i * kTabStride * 4]; }
            return k;
        };
        auto apply = [&](const ColPair& k, bool odd) {
            if (kForm == kRank128) {
                if (odd) C += spread2((k.a | 0x8080u) - k.s);
                else     C -= spread2((k.s | 0x8080u) - k.a);
            } else C += k.a - k.s;
        };
        auto search = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (kForm == kRank64) write_lane_imm<i>(res, __popcll(__ballot(C <= (uint32_t)half)));
            else {
                const uint32_t thr = half + ((i & 1) ? 128 : 0);
                write_lane_imm<i>(res, __popcll(__ballot((C & 0xffffu) <= thr)) + __popcll(__ballot(C <= ((thr << 16) | 0xffffu))));
            }
        };
        if (!(dbg & 2)) {
            ColPair nx = fetch(std::integral_constant<int, 1>{});     // (past the run's end a fetch reads the next column of the table: unused)
            search(std::integral_constant<int, 0>{});
            static_for<1, kMcRun>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const ColPair cur = nx;
                nx = fetch(std::integral_constant<int, i + 1>{});
                apply(cur, (i & 1) != 0);
                search(ic);
            });
        }
        // ---- lanes 0 .. 15 hold their pixel's rank; the row goes to LDS, waves 4..7 translate and store it (the column threads then never wait for a store) ----
        if (lane < kMcRun) lds8[kOffOutRow * 4 + wv * kMcRun + lane] = (uint8_t)res;
        __syncthreads();                                              // every chain has read the columns of row y; the output row is complete
        if (tid >= 256 && tid - 256 < Cw) {
            const int k = tid - 256, x = X0 + k, rk = lds8[kOffOutRow * 4 + k];
            const int word = (y - Y0) * 4 + (k >> 5);
            const uint32_t bit = 1u << (k & 31);
            bool mine = x < W;
            if (pass == kPassFirst) {                                 // an answer at the window's edge is only a bound: the pixel waits for the window beside it
                if (rk == 0 && !lo_exact) { if (mine) { atomicOr(&lds[kOffBelow + word], bit); atomicOr(&lds[kOffMisc], 1u); } mine = false; }
                else if (rk >= 128 && !hi_exact) { if (mine) { atomicOr(&lds[kOffAbove + word], bit); atomicOr(&lds[kOffMisc], 2u); } mine = false; }
            } else if (pass == kPassBelow) mine = mine && (lds[kOffBelow + word] & bit);
            else if (pass == kPassAbove) mine = mine && (lds[kOffAbove + word] & bit);
            if (mine) {
                const int m = inv8[rk];
                dst[(size_t)y * W + x] = (uint8_t)m;
                if (m != prev_m) { atomicOr(&lds[kOffPresOut + (m >> 5)], 1u << (m & 31)); prev_m = m; }
                if (padded_out) {                                     // the next median of the chain reads its source with replicated side columns
                    uint8_t* prow = padded_out + (size_t)y * Wp;
                    prow[kMedPad + x] = (uint8_t)m;
                    if (x == 0) for (int j = 0; j < kMedPad; ++j) prow[j] = (uint8_t)m;
                    if (x == W - 1) for (int j = 0; j < kMedPad; ++j) prow[kMedPad + W + j] = (uint8_t)m;
                }
            }
        }
        if (!more) break;
        if (col_thread && !(dbg & 4)) {                               // ---- the row step of the column tables ----
            const int ia = rank8[nv_in], ib = rank8[nv_out];
            table_step(my_tab, ia, ib);
            lds8[kOffRowIn * 4 + tid] = (uint8_t)ia;
            lds8[kOffRowOut * 4 + tid] = (uint8_t)ib;
        }
        __syncthreads();
        // ---- the chain's first window one row down: the changes of its 2r+1 columns, collected by the wave as plain counts by rank (two atomics per
        // column that changed); their running sum over the lanes is the cumulative change.  Neighbouring columns with the same pair of ranks — the edges
        // of plateaus — are one atomic pair of their number: atomics of many lanes on ONE address cost a pass each.
        for (int base = 0; base <= 2 * r && !(dbg & 8); base += 64) {
            const int j = base + lane;
            const bool valid = j <= 2 * r;
            const int ia = valid ? lds8[kOffRowIn * 4 + c_first + j] : 0, ib = valid ? lds8[kOffRowOut * 4 + c_first + j] : 0;
            const bool changed = ia != ib;
            if (!__ballot(changed)) continue;
            const int key = ia | (ib << 8);
            const int left = __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false);           // wave_shr:1 (lane 0 keeps -1)
            const unsigned long long stop = __ballot(!changed || key != left);     // lanes that begin a run, or are part of none
            if (changed && key != left) {                             // the head of a run: its length = the distance to the next such lane
                const unsigned long long after = (stop >> lane) >> 1;
                const int len = after ? __ffsll((long long)after) : 64 - lane;
                atomicAdd(&delta[ia >> 2], (uint32_t)len << (8 * (ia & 3)));
                atomicSub(&delta[ib >> 2], (uint32_t)len << (8 * (ib & 3)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // (one wave's LDS operations execute in order; this orders the compiler)
        if (kForm == kRank64) {
            C0 += (uint32_t)wave_prefix_sum((int)reinterpret_cast<uint8_t*>(delta)[lane] - 128);
            if (lane < 16) delta[lane] = kBias4;
        } else {
            const int d2 = reinterpret_cast<uint16_t*>(delta)[lane];
            const int de = (d2 & 0xff) - 128, dsum = de + (d2 >> 8) - 128;
            const int incl = wave_prefix_sum(dsum);
            C0 = (((C0 & 0xffffu) + (uint32_t)(incl - dsum + de)) & 0xffffu) | (((C0 >> 16) + (uint32_t)incl) << 16);
            if (lane < kTabStride) delta[lane] = kBias4;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// force: bits 0-1: 0 = by the footprint's values, 1 = every tile by windows (as if it held more than 129 values), 2 = no tile with one count per lane;
// tests of the windows beside the first: bit 2 = the first window at the top of the ranks, bit 3 = at their bottom (instead of around the sample's median)
__global__ void __launch_bounds__(kMcWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_median_cols(const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst, uint8_t* __restrict__ padded_out, int W, int H, int ksize, int rows,
              const uint32_t* __restrict__ pres_in, uint32_t* __restrict__ pres_out, int force, int dbg) {
    __shared__ uint32_t lds[kMcLdsWords];
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const int tid = threadIdx.x;
    const int r = ksize >> 1;
    constexpr int Cw = kMcWaves * kMcRun;
    const int X0 = blockIdx.x * Cw, Y0 = blockIdx.y * rows;
    // the values present in the tile's footprint: the union of the presence maps of the source tiles it touches
    if (tid < 8) {
        uint32_t p = 0xffffffffu;
        if (pres_in) {
            p = 0;
            const int tx0 = min(max(X0 - r, 0), W - 1) / Cw, tx1 = min(max(X0 + Cw - 1 + r, 0), W - 1) / Cw;
            const int ty0 = min(max(Y0 - r, 0), H - 1) / rows, ty1 = min(max(Y0 + rows - 1 + r, 0), H - 1) / rows;
            for (int ty = ty0; ty <= ty1; ++ty)
                for (int tx = tx0; tx <= tx1; ++tx) p |= pres_in[((size_t)ty * gridDim.x + tx) * 8 + tid];
        }
        lds[kOffPres + tid] = p;
        lds[kOffPresOut + tid] = 0;
    }
    __syncthreads();
    int distinct = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) distinct += __popc(lds[kOffPres + k]);
    const bool windows = (force & 3) == 1 || distinct > 129;
    int rk = 0;                                                       // this thread's value: its rank among the values present
    bool present = false;
    if (tid < 256) {
        for (int k = 0; k < (tid >> 5); ++k) rk += __popc(lds[kOffPres + k]);
        const uint32_t mine = lds[kOffPres + (tid >> 5)];
        rk += __popc(mine & ((1u << (tid & 31)) - 1u));
        present = (mine >> (tid & 31)) & 1u;
    }
    // value -> rank within the window that begins at rank `base` (0 .. 128), and back (a rank has one value)
    auto window_tables = [&](int base) {
        if (tid < 256) {
            lds8[kOffRank * 4 + tid] = (uint8_t)min(max(rk - base, 0), 128);
            if (present && rk >= base && rk <= base + 128) lds8[kOffInv * 4 + rk - base] = (uint8_t)tid;
        }
    };
    int prev_m = -1;                                                  // (storing threads) the value stored last: its presence bit is set
    if (!windows) {
        window_tables(0);
        __syncthreads();
        if (distinct <= 64 && (force & 3) != 2) median_cols_pass<kRank64>(lds, srcp, dst, padded_out, W, H, ksize, rows, kPassOnly, true, true, prev_m, dbg);
        else median_cols_pass<kRank128>(lds, srcp, dst, padded_out, W, H, ksize, rows, kPassOnly, true, true, prev_m, dbg);
    } else {
        // With cumulative counts C[0 .. D-1] the median's rank is m = #{j : C[j] <= half}.  A window at `base` holds the counts C[base .. base + 127]
        // (lower ranks count at `base`, higher ranks nowhere), so its answer is clamp(m - base, 0, 128): m itself for answers 1 .. 127, for 0 if
        // base = 0, for 128 if base + 128 is the last rank.  First window: around the median rank of 512 of the tile's own pixels.
        const int top = max(distinct - 129, 0);                       // the highest base: its window reaches the last rank
        uint16_t* const sample = reinterpret_cast<uint16_t*>(lds + kOffBelow);       // 256 counts by rank (the bitmaps' place, cleared below)
        if (tid < 256) lds8[kOffRank * 4 + tid] = (uint8_t)rk;
        if (tid < 128) lds[kOffBelow + tid] = 0;
        __syncthreads();
        {
            const int sx = min(X0 + (tid & 127), W - 1), sy = min(Y0 + ((tid >> 7) * rows >> 2) + (rows >> 3), H - 1);
            const int sr = lds8[kOffRank * 4 + srcp[(size_t)sy * (W + 2 * kMedPad) + kMedPad + sx]];
            atomicAdd(&lds[kOffBelow + (sr >> 1)], 1u << (16 * (sr & 1)));
        }
        __syncthreads();
        if (tid < 64) {
            const int four = sample[4 * tid] + sample[4 * tid + 1] + sample[4 * tid + 2] + sample[4 * tid + 3];
            const int upto = wave_prefix_sum(four);
            const int ms = 4 * __popcll(__ballot(upto <= kMcWaves * 32)) + 2;        // about the sample's median rank
            const int guess = (force & 4) ? 255 : (force & 8) ? 0 : ms - 64;
            if (tid == 0) { lds[kOffMisc] = 0; lds[kOffMisc + 1] = min(max(guess, 0), top); }
        }
        __syncthreads();
        const int base = lds[kOffMisc + 1];
        for (int k = tid; k < 2 * kMcMaxRows * 4; k += kMcWaves * 64) lds[kOffBelow + k] = 0;
        window_tables(base);
        __syncthreads();
        median_cols_pass<kRank128>(lds, srcp, dst, padded_out, W, H, ksize, rows, kPassFirst, base == 0, base == top, prev_m, dbg);
        __syncthreads();
        const uint32_t again = lds[kOffMisc];
        if (again & 1u) {                                             // medians of rank <= base: the window at 0 holds them all (base <= 127)
            __syncthreads();
            window_tables(0);
            __syncthreads();
            median_cols_pass<kRank128>(lds, srcp, dst, padded_out, W, H, ksize, rows, kPassBelow, true, true, prev_m, dbg);
        }
        if (again & 2u) {                                             // medians of rank >= base + 128 > top: the top window holds them all (top <= 127)
            __syncthreads();
            window_tables(top);
            __syncthreads();
            median_cols_pass<kRank128>(lds, srcp, dst, padded_out, W, H, ksize, rows, kPassAbove, true, true, prev_m, dbg);
        }
    }
    if (pres_out) {
        __syncthreads();
        if (tid < 8) pres_out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + tid] = lds[kOffPresOut + tid];
    }
}

// the presence map of a tight image, tile by tile (the first median of a chain, or one whose source another kernel made)
// easy (may be null): counts the tiles that hold at most kEasyValues different values (what decides, per image, whether its medians take this file's kernel)
constexpr int kEasyValues = 24;
__global__ void __launch_bounds__(256) k_median_presence(const uint8_t* __restrict__ src, uint32_t* __restrict__ pres, int W, int H, int Cw, int rows,
                                                         uint32_t* __restrict__ easy) {
    __shared__ uint32_t p[8];
    const int tid = threadIdx.x;
    if (tid < 8) p[tid] = 0;
    __syncthreads();
    const int X0 = blockIdx.x * Cw, Y0 = blockIdx.y * rows;
    const int tw = min(Cw, W - X0), th = min(rows, H - Y0);
    int last = -1;
    for (int k = tid; k < tw * th; k += 256) {
        const int v = src[(size_t)(Y0 + k / tw) * W + X0 + k % tw];
        if (v != last) { atomicOr(&p[v >> 5], 1u << (v & 31)); last = v; }
    }
    __syncthreads();
    if (tid < 8) pres[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + tid] = p[tid];
    if (easy && tid == 0) {
        int n = 0;
        for (int k = 0; k < 8; ++k) n += __popc(p[k]);
        if (n <= kEasyValues) atomicAdd(easy, 1u);
    }
}

}  // namespace

MedianColsGeom median_cols_geom(int w, int h) {
    MedianColsGeom g{};
    g.run = kMcRun;                                                  // (compiled in: the run's loop is unrolled)
    g.tiles_x = (w + kMcWaves * kMcRun - 1) / (kMcWaves * kMcRun);
    // rows per tile: about three tiles per compute unit for the whole image (a tile pays 2r rows of column warm-up; four workgroups fit a compute unit
    // and a pair's two images run side by side), at least 16.  Measured at 1080p / 4K: tools/experiments/median_rows_ab.sh.
    static const int forced_rows = getenv("POPPY_MED_COLS_ROWS") ? atoi(getenv("POPPY_MED_COLS_ROWS")) : 0;
    const int target_y = std::max(1, 768 / g.tiles_x);
    g.rows = forced_rows > 0 ? forced_rows : std::max((h + target_y - 1) / target_y, 16);
    g.rows = std::min({g.rows, h, kMcMaxRows});
    g.tiles_y = (h + g.rows - 1) / g.rows;
    return g;
}
size_t median_presence_words(int w, int h) { const MedianColsGeom g = median_cols_geom(w, h); return (size_t)g.tiles_x * g.tiles_y * 8; }

void launch_median_presence(const uint8_t* src_tight, uint32_t* pres, int w, int h, uint32_t* easy_or_null, hipStream_t s) {
    const MedianColsGeom g = median_cols_geom(w, h);
    hipLaunchKernelGGL(k_median_presence, dim3(g.tiles_x, g.tiles_y), dim3(256), 0, s, src_tight, pres, w, h, kMcWaves * g.run, g.rows, easy_or_null);
}

// The same question answered on the host from a sparse sample of the image (its green channel stands in for the grey value): 32 blocks of 128 x 32
// pixels, every fourth pixel of every second row.  A hint for speed only — every form of the median returns the same bytes.
int median_cols_hint_from_host(const uint8_t* bgr, size_t stride, int w, int h) {
    if (w < 256 || h < 64) return 0;
    int easy = 0;
    for (int b = 0; b < 32; ++b) {
        const int x0 = (int)((long long)(w - 128) * (b % 8) / 7), y0 = (int)((long long)(h - 32) * (b / 8) / 3);
        uint32_t bits[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = 0; y < 32; y += 2) {
            const uint8_t* row = bgr + (size_t)(y0 + y) * stride + (size_t)x0 * 3 + 1;
            for (int x = 0; x < 128; x += 4) { const int v = row[x * 3]; bits[v >> 5] |= 1u << (v & 31); }
        }
        int n = 0;
        for (int k = 0; k < 8; ++k) n += __builtin_popcount(bits[k]);
        easy += n <= kEasyValues;
    }
    return easy >= 29;
}

void launch_median_cols(const uint8_t* padded_src, uint8_t* dst, uint8_t* padded_next, int w, int h, int ksize, const uint32_t* pres_in, uint32_t* pres_out,
                        int force, hipStream_t s) {
    const MedianColsGeom g = median_cols_geom(w, h);
    static const int dbg = poppy_experiment_env("POPPY_MED_COLS_SKIP") ? atoi(poppy_experiment_env("POPPY_MED_COLS_SKIP")) : 0;    // timing experiments: parts left out (wrong results)
    hipLaunchKernelGGL(k_median_cols, dim3(g.tiles_x, g.tiles_y), dim3(kMcWaves * 64), 0, s, padded_src, dst, padded_next, w, h, ksize, g.rows, pres_in,
                       pres_out, force, dbg);
}

}  // namespace poppy_hip
