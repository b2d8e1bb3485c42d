// kernels_median_cols.hip — medianBlur on 8-bit single-channel images by COLUMN HISTOGRAMS, for the windows of Extractor::foreground's
// chain (src/extractor.cpp:136-161: ksize 9, 17, ..., 89).  The scheme is the one the reference itself uses for such windows
// (medianBlur_8u_O1, OCV/imgproc/src/median_blur.simd.hpp:84-346, after Perreault & Hebert): every image column keeps the histogram of
// its 2r+1 window rows; the window's histogram slides along the row by adding the entering column's histogram and subtracting the
// leaving one's, so a pixel costs the same whatever ksize is.  The median of a set is unique, so any exact scheme returns the
// reference's bytes; borders are replicated as medianBlur does (clamped rows and columns).
//
// How it sits on a CDNA4 compute unit
//   * A workgroup (8 waves) owns a tile of 8 * run columns by `rows` rows.  The column histograms of the tile's columns + 2r live in
//     LDS, a thread per column steps them one row down (one value in, one value out: two LDS atomics, cancelled when the values agree).
//   * A WAVE is one sliding window histogram ("chain"): lane l holds the counts of the values 4l .. 4l+3 (two packed pairs) and C[l],
//     the number of window values below 4(l+1).  Each column also stores that cumulative form (64 bytes), so a step adds and
//     subtracts two dwords and two bytes per lane and NO prefix scan is ever needed: the median's lane is popcount(ballot(C <= half)).
//     That lane's three words go to lane i of three result registers (v_readlane -> v_writelane); after a run of `run` pixels the
//     lanes finish the search inside their four values in parallel.
//   * A chain restarts every row at its first pixel from the state it had there one row up: the change of that window — 2r+1 values
//     in, 2r+1 out — is collected by the chain's own lanes into a small delta histogram and applied like one more column pair.
//   * Few distinct values: the tile's footprint often holds no more than 64 different values (synthetic content always; the later,
//     plateau-like stages of the chain on photographs often).  A 256-bit presence map per tile (written by the previous median of the
//     chain, or by k_median_presence) gives the values' RANKS; with <= 64 of them a lane is one rank, the cumulative bytes are the whole
//     histogram and a pixel costs two byte reads, an add, a subtract and a ballot.  Ranking is monotone, so the median's rank is the
//     rank of the median.
#include "kernels_prefilter.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

namespace poppy_hip {

namespace {

constexpr int kMedPad = 48;                                  // replicated side columns of the padded source (kernels_prefilter.hip)
constexpr int kMcWaves = 8;                                  // chains per workgroup
constexpr int kMcMaxRun = 16;                                // pixels a chain takes per row (a tile is 128 columns wide)
constexpr int kMcMaxR = 44;                                  // ksize 89
constexpr int kMcCols = kMcWaves * kMcMaxRun + 2 * kMcMaxR;  // column tables per workgroup (224)
// dword strides of a column's two tables: odd, so that a thread per column (the row step) meets 32 different banks
constexpr int kFineStride = 65, kCumStride = 17;
constexpr int kOffFine = 0;                                          // [col][64] dwords — see the three forms below
constexpr int kOffCum = kOffFine + kMcCols * kFineStride;            // [col][64] bytes
constexpr int kOffDFine = kOffCum + kMcCols * kCumStride;            // per chain: the change of the counts of its first window, every byte biased by 128
constexpr int kOffRowIn = kOffDFine + kMcWaves * 64;                 // the row step's entering / leaving value (or rank) of every column
constexpr int kOffRowOut = kOffRowIn + (kMcCols + 3) / 4;
constexpr int kOffRank = kOffRowOut + (kMcCols + 3) / 4;             // value -> rank (256 bytes), rank -> value (256 bytes)
constexpr int kOffInv = kOffRank + 64;
constexpr int kOffPres = kOffInv + 64;                               // the footprint's presence bits, the output tile's
constexpr int kOffPresOut = kOffPres + 8;
constexpr int kOffOutRow = kOffPresOut + 8;                          // the tile's current output row (bytes), stored by the waves that step no columns
constexpr int kMcLdsWords = kOffOutRow + (kMcWaves * kMcMaxRun + 3) / 4;
static_assert(kMcCols <= 256, "the column threads are waves 0..3; waves 4..7 store the rows");
static_assert(kMcLdsWords * 4 <= 79 * 1024, "two workgroups per compute unit, whatever the allocation granule");

// The three forms of a tile (chosen by the number of different values its footprint holds):
//   kValues : all 256 values.  Column tables: fine[col][l] = the counts of the values 4l .. 4l+3 (a byte each), cum[col][l] = the values below
//             4(l+1).  A lane holds C (window values below 4(l+1)) and its four counts as two packed pairs.
//   kRank64 : at most 64 values, a lane is one RANK.  Column table: cum[col][l] = the values of rank <= l.  A lane holds C only.
//   kRank128: at most 128 values, a lane is two ranks.  Column table (in the fine table's place): two bytes per lane, the values of rank
//             <= 2l and <= 2l+1.  A lane holds both counts packed; the median's rank is popcount(ballot(low <= half)) + popcount(ballot(high <= half)).
enum { kValues = 0, kRank64 = 1, kRank128 = 2 };

constexpr uint32_t kBias4 = 0x80808080u;

// a loop whose index is a compile-time constant in the body (LDS offsets and lane selects become immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// bytes (0, 2) / (1, 3) of a dword as two 16-bit counts: full-rate instructions (v_perm_b32 issues at half rate on gfx950)
__device__ __forceinline__ uint32_t even_bytes(uint32_t x) { return x & 0x00ff00ffu; }
__device__ __forceinline__ uint32_t odd_bytes(uint32_t x) { return (x >> 8) & 0x00ff00ffu; }
__device__ __forceinline__ uint32_t spread2(uint32_t x) { return (x & 0xffu) | ((x & 0xff00u) << 8); }      // bytes 0, 1 -> the two 16-bit halves

// v_writelane_b32 (this compiler has no builtin for the instruction): lane kSel of each word becomes the wave-uniform value beside it.  The
// lane select is an immediate (with two scalar operands the instruction would need it in M0); the values come from v_readlane / the scalar
// unit as DATA operands — the instruction's one listed hazard is a vector-written lane SELECT.
template <int kSel>
__device__ __forceinline__ void write_lane1_imm(int& a, int va) { asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(a) : "s"(va), "i"(kSel)); }
template <int kSel>
__device__ __forceinline__ void write_lane3_imm(int& a, int va, int& b, int vb, int& c, int vc) {
    asm volatile("v_writelane_b32 %0, %3, %6\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6" : "+v"(a), "+v"(b), "+v"(c) : "s"(va), "s"(vb), "s"(vc), "i"(kSel));
}

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

// the bytes [lo, hi) of a cumulative table change by +1 / -1 (a thread owns its column's table; counts stay within 0..255: a leaving value was counted)
__device__ __forceinline__ void range_step(uint32_t* __restrict__ cum, int lo, int hi, bool add) {
    for (int d = lo >> 2; d <= (hi - 1) >> 2; ++d) {
        const int j0 = max(lo - 4 * d, 0), j1 = min(hi - 4 * d, 4);
        const uint32_t m = (0x01010101u >> (8 * (4 - (j1 - j0)))) << (8 * j0);
        if (add) atomicAdd(&cum[d], m); else atomicSub(&cum[d], m);
    }
}
// One value in, one value out of a column's tables.  ia / ib: fine index of the entering / leaving value (the value, or its rank).
template <int kForm>
__device__ __forceinline__ void tables_step(uint32_t* __restrict__ fine, uint32_t* __restrict__ cum, int ia, int ib) {
    if (ia == ib) return;
    if (kForm == kValues) {
        atomicAdd(&fine[ia >> 2], 1u << (8 * (ia & 3)));
        atomicSub(&fine[ib >> 2], 1u << (8 * (ib & 3)));
    }
    // a cumulative entry counts the values of index <= its own: entering adds one from la on, leaving removes one from lb on
    const int la = kForm == kValues ? ia >> 2 : ia, lb = kForm == kValues ? ib >> 2 : ib;
    if (la != lb) range_step(kForm == kRank128 ? fine : cum, min(la, lb), max(la, lb), la < lb);
}

template <int kForm>
__device__ __forceinline__ void median_cols_body(uint32_t* __restrict__ lds, const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst,
                                                 uint8_t* __restrict__ padded_out, int W, int H, int ksize, int run, int rows, int dbg) {
    constexpr bool kRanks = kForm != kValues;
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const uint16_t* const lds16 = reinterpret_cast<const uint16_t*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = ksize >> 1, half = (ksize * ksize) >> 1;
    const int Cw = kMcWaves * run, ncols = Cw + 2 * r;
    const int X0 = blockIdx.x * Cw, Y0 = blockIdx.y * rows, Yend = min(Y0 + rows, H);
    const int Wp = W + 2 * kMedPad;
    const uint8_t* const rank8 = lds8 + kOffRank * 4;
    const uint8_t* const inv8 = lds8 + kOffInv * 4;
    const bool col_thread = tid < ncols;
    const int gx = min(max(X0 - r + tid, 0), W - 1);
    const uint8_t* const colp = srcp + kMedPad + gx;                  // this thread's image column (threads past ncols: unused)
    auto row_of = [&](int yy) { return (size_t)min(max(yy, 0), H - 1) * Wp; };
    uint32_t* const my_fine = lds + kOffFine + tid * kFineStride;
    uint32_t* const my_cum = lds + kOffCum + tid * kCumStride;
    // ---- the column tables of the first window row: the counts of rows Y0 - r .. Y0 + r of every column, then their cumulative form ----
    if (col_thread && !(dbg & 1)) {
        for (int rr0 = -r; rr0 <= r; rr0 += 16) {                     // sixteen rows' loads in flight, then their sixteen additions
            int idx[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) idx[j] = colp[row_of(Y0 + min(rr0 + j, r))];
            if (kRanks) {
#pragma unroll
                for (int j = 0; j < 16; ++j) idx[j] = rank8[idx[j]];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (rr0 + j <= r) atomicAdd(&my_fine[idx[j] >> 2], 1u << (8 * (idx[j] & 3)));
        }
        uint32_t below = 0;
        if (kForm == kValues) {
            for (int d = 0; d < 16; ++d) {
                uint32_t w4 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    below = __builtin_amdgcn_udot4(my_fine[4 * d + j], 0x01010101u, below, false);
                    w4 |= below << (8 * j);
                }
                my_cum[d] = w4;
            }
        } else {                                                      // counts by rank -> cumulative by rank (kRank128: in the table's own place)
            uint32_t* const to = kForm == kRank128 ? my_fine : my_cum;
            for (int d = 0; d < (kForm == kRank128 ? 32 : 16); ++d) {
                const uint32_t x = my_fine[d];
                const uint32_t b0 = below + (x & 255u), b1 = b0 + ((x >> 8) & 255u), b2 = b1 + ((x >> 16) & 255u), b3 = b2 + (x >> 24);
                to[d] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                below = b3;
            }
        }
    }
    __syncthreads();
    // ---- this wave's chain: the window's counts at its first pixel of row Y0 ----
    const int c_first = wv * run;                                     // local column of the window's left edge at the chain's first pixel
    uint32_t C0 = 0, HE0 = 0, HO0 = 0;                                // (kRank128: C0 = both counts, packed)
    for (int j = 0; j <= 2 * r && !(dbg & 16); ++j) {
        const int col = c_first + j;
        if (kForm == kRank128) C0 += spread2(lds16[(kOffFine + col * kFineStride) * 2 + lane]);
        else C0 += lds8[(kOffCum + col * kCumStride) * 4 + lane];
        if (kForm == kValues) {
            const uint32_t x = lds[kOffFine + col * kFineStride + lane];
            HE0 += even_bytes(x);
            HO0 += odd_bytes(x);
        }
    }
    uint32_t* const d_fine = lds + kOffDFine + wv * 64;
    int prev_m = -1;                                                  // (storing threads) the value stored one row up: its presence bit is set
    for (int y = Y0; y < Yend; ++y) {
        const bool more = y + 1 < Yend;
        int nv_in = 0, nv_out = 0;
        if (more && col_thread) { nv_in = colp[row_of(y + r + 1)]; nv_out = colp[row_of(y - r)]; }      // the row step's values, asked for early
        // ---- the run: `run` pixels of row y ----
        uint32_t C = C0, HE = HE0, HO = HO0;
        int res0 = 0, res1 = 0, res2 = 0;
        // A step to pixel i: column c_first + i + 2r enters, column c_first + i - 1 leaves.  Packed counts: odd steps add 128 + (in - out) per count,
        // even steps subtract 128 - (in - out): a byte-wise difference without borrows, and the 128s cancel every second step (an odd pixel's search
        // sees every packed count 128 too high).  The pixel index is a compile-time constant: the columns' LDS offsets and the result's lane are
        // immediates; the columns of the next step are asked for before the current pixel is searched.
        struct ColPair { uint32_t ha, hs, ca, cs; };
        const int in0 = c_first + 2 * r, out0 = c_first - 1;          // columns of step 0 (step i: + i)
        const uint8_t* const cum_in = lds8 + (kOffCum + in0 * kCumStride) * 4 + lane, * const cum_out = lds8 + (kOffCum + out0 * kCumStride) * 4 + lane;
        const uint32_t* const fine_in = lds + kOffFine + in0 * kFineStride + lane, * const fine_out = lds + kOffFine + out0 * kFineStride + lane;
        const uint16_t* const two_in = lds16 + (kOffFine + in0 * kFineStride) * 2 + lane, * const two_out = lds16 + (kOffFine + out0 * kFineStride) * 2 + lane;
        auto fetch = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            ColPair k;
            k.ha = k.hs = k.ca = k.cs = 0;
            if (kForm == kRank128) { k.ha = two_in[i * kFineStride * 2]; k.hs = two_out[i * kFineStride * 2]; }
            else { k.ca = cum_in[i * kCumStride * 4]; k.cs = cum_out[i * kCumStride * 4]; }
            if (kForm == kValues) { k.ha = fine_in[i * kFineStride]; k.hs = fine_out[i * kFineStride]; }
            return k;
        };
        auto apply = [&](const ColPair& k, bool odd) {
            if (kForm == kValues) {
                if (odd) { const uint32_t z = (k.ha | kBias4) - k.hs; HE += even_bytes(z); HO += odd_bytes(z); }
                else     { const uint32_t z = (k.hs | kBias4) - k.ha; HE -= even_bytes(z); HO -= odd_bytes(z); }
            }
            if (kForm == kRank128) {
                if (odd) C += spread2((k.ha | 0x8080u) - k.hs);
                else     C -= spread2((k.hs | 0x8080u) - k.ha);
            } else C += k.ca - k.cs;
        };
        auto search = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr bool odd = (i & 1) != 0;
            if (kForm == kRank64) write_lane1_imm<i>(res0, __popcll(__ballot(C <= (uint32_t)half)));
            else if (kForm == kRank128) {
                const uint32_t thr = half + (odd ? 128 : 0);
                write_lane1_imm<i>(res0, __popcll(__ballot((C & 0xffffu) <= thr)) + __popcll(__ballot(C <= ((thr << 16) | 0xffffu))));
            } else {                                                  // what the finish needs of the median's lane: the count below it, its four counts
                const int L = __popcll(__ballot(C <= (uint32_t)half));   // lanes wholly below the median; C[63] = ksize^2 > half, so L <= 63
                const int below = __builtin_amdgcn_readlane((int)C, max(L - 1, 0));
                write_lane3_imm<i>(res0, (L ? below : 0) | (L << 16), res1, __builtin_amdgcn_readlane((int)HE, L), res2, __builtin_amdgcn_readlane((int)HO, L));
            }
        };
        if (!(dbg & 2)) {
            ColPair nx = fetch(std::integral_constant<int, 1>{});     // (past the run's end a fetch reads the next column of the table: unused)
            search(std::integral_constant<int, 0>{});
            static_for<1, kMcMaxRun>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const ColPair cur = nx;
                nx = fetch(std::integral_constant<int, i + 1>{});
                apply(cur, (i & 1) != 0);
                search(ic);
            });
        }
        // ---- lanes 0 .. run-1 finish their pixel; the row goes to LDS, waves 4..7 store it (the column threads then never wait for a store) ----
        if (lane < run) {
            int m;
            if (kRanks) m = inv8[res0];
            else {
                const int bias = (lane & 1) ? 128 : 0;
                const int L = res0 >> 16, t = half - (res0 & 0xffff);            // half - (window values below the lane's four)
                const int c0 = (res1 & 0xffff) - bias, c1 = c0 + (res2 & 0xffff) - bias, c2 = c1 + (int)((uint32_t)res1 >> 16) - bias;
                m = 4 * L + (c0 <= t) + (c1 <= t) + (c2 <= t);
            }
            lds8[kOffOutRow * 4 + wv * run + lane] = (uint8_t)m;
        }
        __syncthreads();                                              // every chain has read the columns of row y; the output row is complete
        if (tid >= 256 && tid - 256 < Cw) {
            const int x = X0 + tid - 256, m = lds8[kOffOutRow * 4 + tid - 256];
            if (x < W) {
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
            const int ia = kRanks ? rank8[nv_in] : nv_in, ib = kRanks ? rank8[nv_out] : nv_out;
            tables_step<kForm>(my_fine, my_cum, ia, ib);
            lds8[kOffRowIn * 4 + tid] = (uint8_t)ia;
            lds8[kOffRowOut * 4 + tid] = (uint8_t)ib;
        }
        __syncthreads();
        // ---- the chain's first window one row down: the changes of its 2r+1 columns, collected by the wave and applied like a column pair.
        // Only plain counts are collected (two atomics per column that changed); their running sum over the lanes is the cumulative change.
        // Neighbouring columns with the same pair of values — the edges of plateaus — are one atomic pair of their number: atomics of many
        // lanes on ONE address cost a pass each.
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
                const unsigned long long later = (stop >> lane) >> 1;
                const int len = later ? __ffsll((long long)later) : 64 - lane;
                atomicAdd(&d_fine[ia >> 2], (uint32_t)len << (8 * (ia & 3)));
                atomicSub(&d_fine[ib >> 2], (uint32_t)len << (8 * (ib & 3)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // (one wave's LDS operations execute in order; this orders the compiler)
        if (kForm == kValues) {
            const uint32_t z = d_fine[lane];                          // four counts' changes, each + 128
            C0 += (uint32_t)wave_prefix_sum((int)__builtin_amdgcn_udot4(z, 0x01010101u, 0u, false) - 512);
            HE0 = HE0 + even_bytes(z) - 0x00800080u;
            HO0 = HO0 + odd_bytes(z) - 0x00800080u;
            d_fine[lane] = kBias4;
        } else if (kForm == kRank64) {
            C0 += (uint32_t)wave_prefix_sum((int)reinterpret_cast<uint8_t*>(d_fine)[lane] - 128);
            if (lane < 16) d_fine[lane] = kBias4;
        } else {
            const int d2 = reinterpret_cast<uint16_t*>(d_fine)[lane];
            const int de = (d2 & 0xff) - 128, dsum = de + (d2 >> 8) - 128;
            const int incl = wave_prefix_sum(dsum);
            C0 = (((C0 & 0xffffu) + (uint32_t)(incl - dsum + de)) & 0xffffu) | (((C0 >> 16) + (uint32_t)incl) << 16);
            if (lane < 32) d_fine[lane] = kBias4;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// force: 0 = by the footprint's values, 1 = kValues always, 2 = no kRank64 (tests: kRank128 on few-valued content)
__global__ void __launch_bounds__(kMcWaves * 64) k_median_cols(const uint8_t* __restrict__ srcp, uint8_t* __restrict__ dst, uint8_t* __restrict__ padded_out,
                                                               int W, int H, int ksize, int run, int rows,
                                                               const uint32_t* __restrict__ pres_in, uint32_t* __restrict__ pres_out, int force, int dbg) {
    __shared__ uint32_t lds[kMcLdsWords];
    uint8_t* const lds8 = reinterpret_cast<uint8_t*>(lds);
    const int tid = threadIdx.x;
    const int r = ksize >> 1, Cw = kMcWaves * run;
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
    for (int k = tid; k < kOffDFine; k += kMcWaves * 64) lds[k] = 0;                       // column tables
    for (int k = kOffDFine + tid; k < kOffRowIn; k += kMcWaves * 64) lds[k] = kBias4;      // delta tables
    __syncthreads();
    int distinct = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) distinct += __popc(lds[kOffPres + k]);
    const int form = (!pres_in || force == 1 || distinct > 128) ? kValues : (distinct > 64 || force == 2) ? kRank128 : kRank64;
    if (form != kValues && tid < 256) {                               // value -> rank, rank -> value
        int rk = 0;
        for (int k = 0; k < (tid >> 5); ++k) rk += __popc(lds[kOffPres + k]);
        const uint32_t mine = lds[kOffPres + (tid >> 5)];
        rk += __popc(mine & ((1u << (tid & 31)) - 1u));
        lds8[kOffRank * 4 + tid] = (uint8_t)min(rk, 127);
        if ((mine >> (tid & 31)) & 1u) lds8[kOffInv * 4 + rk] = (uint8_t)tid;
    }
    __syncthreads();
    if (form == kRank64) median_cols_body<kRank64>(lds, srcp, dst, padded_out, W, H, ksize, run, rows, dbg);
    else if (form == kRank128) median_cols_body<kRank128>(lds, srcp, dst, padded_out, W, H, ksize, run, rows, dbg);
    else median_cols_body<kValues>(lds, srcp, dst, padded_out, W, H, ksize, run, rows, dbg);
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
    g.run = kMcMaxRun;                                               // (compiled in: the run's loop is unrolled)
    g.tiles_x = (w + kMcWaves * kMcMaxRun - 1) / (kMcWaves * kMcMaxRun);
    // rows per tile: about two tiles per compute unit for the whole image (a tile pays 2r rows of column warm-up), at least 16
    static const int forced_rows = getenv("POPPY_MED_COLS_ROWS") ? atoi(getenv("POPPY_MED_COLS_ROWS")) : 0;
    const int target_y = std::max(1, 512 / g.tiles_x);
    g.rows = forced_rows > 0 ? forced_rows : std::max((h + target_y - 1) / target_y, 16);
    g.rows = std::min(g.rows, h);
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
    static const int dbg = getenv("POPPY_MED_COLS_SKIP") ? atoi(getenv("POPPY_MED_COLS_SKIP")) : 0;    // timing experiments: parts left out (wrong results)
    hipLaunchKernelGGL(k_median_cols, dim3(g.tiles_x, g.tiles_y), dim3(kMcWaves * 64), 0, s, padded_src, dst, padded_next, w, h, ksize, g.run, g.rows, pres_in,
                       pres_out, force, dbg);
}

}  // namespace poppy_hip
