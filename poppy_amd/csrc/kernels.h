// kernels.h — launchers of the HIP kernels behind libpoppy_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace poppy_hip {

struct PyrLevel {        // geometry of one pyramid level and of the pyrDown that PRODUCES it
    int w, h;            // size of this level
    size_t off3, off1;   // element offsets of this level inside the 3-channel / 1-channel pyramid buffers
    int pitch;           // pixels per row of this level in those buffers: w, or w rounded up to a multiple of 4 (level_pitch)
};
// Rows of the large levels begin on 16-byte boundaries whatever the width is: the kernels that carry most of a frame's bytes move 4 pixels
// (12 bytes of u8, 48 of float) per access.  Widths that are multiples of 4 have that for free (pitch == w: nothing changes for them); other
// widths get up to 3 pixels of padding per row in the frame slots' own buffers (warped images, pyramid levels) — never in what a caller hands
// in or gets back.  The small levels (the fused and the tail kernels' domain) stay tight.
constexpr size_t kPitchMinPixels = 150001;          // (= above kFuseMaxPixels)
// (Round 6 tried cache-line aligned rows — a pitch of a multiple of 128 pixels from a megapixel up — for the warp kernel's outputs: k_warp_bin takes 57 us at 3836 x 2160
// and at 3844 x 2160 where 3840, 3712 and 3584, whose rows are multiples of 128 BYTES in the tight sources too, take 44 - 46; with the outputs' rows aligned it still took 57:
// it is the gathers from the tight sources, not the stores.  tools/experiments/width_align.sh, odd_trace2.sh.)
inline int level_pitch(int w, int h) { return ((w & 3) && (size_t)w * h >= kPitchMinPixels) ? (w + 3) & ~3 : w; }
// Level 1 under a padded level 0 is padded too whatever its size (round 6): the level-0 pyrDown's 16-byte stores and the level-0 collapse's loads take it then
// (749 x 480: level 1 is 375 x 240 — tight, the level-0 pyrDown fell back to the per-element kernel, 12.7 us against 7.0 at 752); k_pyrdown2 reads it by its pitch.
inline int level1_pitch(int w1, int h1, int pitch0, int w0) { return ((w1 & 3) && pitch0 != w0) ? (w1 + 3) & ~3 : level_pitch(w1, h1); }

// hipMemcpy2DAsync, except that rows which are tight on both sides go as ONE linear copy: a 2-D copy whose row length is no multiple of 4 bytes
// takes a slow path of the runtime (639 x 480 x 3: 4 ms against 0.03; 1918 x 1080 x 3: 9 ms)
inline hipError_t copy_rows_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t s) {
    if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * height, kind, s);
    return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, s);
}

// an integer measurement switch that exists only in a -DPOPPY_EXPERIMENTS build (kernels_prefilter.h: poppy_experiment_env); 0 in the shipped library
inline int poppy_experiment_env_i(const char* name) {
#ifdef POPPY_EXPERIMENTS
    const char* v = getenv(name);
    return v ? atoi(v) : 0;
#else
    (void)name;
    return 0;
#endif
}
// Wave-priority stagger (pyramid_device.h: stagger_priority): the kernel argument for kernel `bit` of POPPY_STAGGER
int stagger_flag(int bit);

// --- once per pair ---------------------------------------------------------------------------
// m2 = 1 - gray(gabor2)   (src/algo.cpp:250-252)
void launch_gray_inv(const float* gabor2, float* m2, int n_px, hipStream_t s);

// --- once per frame --------------------------------------------------------------------------
// triangle-id map: exact fillConvexPoly raster of every triangle, later index wins (atomicMax).
// A map value is id_base + triangle + 1, id_base = frame tag << kIdTagShift: tags grow from frame to frame, so this
// frame's values beat whatever older frames left behind and the map needs no clearing in between; a reader takes
// value - id_base when that is below 2^kIdTagShift and 0 ("no triangle") otherwise (decode_id, warp_device.h).  id_base 0
// on a zeroed map is the plain map.
// `work` = (triangle, row-chunk) pairs, kRasterChunkRows rows per chunk, built by the host per frame
constexpr int kRasterChunkRows = 16;
// `edges` = one RasterTri (frame_plan.h) per triangle: the fill-edge segments decided by the host plan
constexpr int kIdTagShift = 20;                 // up to 2^20 - 1 triangles
constexpr int kIdTagMax = 2047;                 // values stay positive: atomicMax on int32
void launch_raster(const int* tri_xy, const void* edges, const int* work, int n_work, int32_t* triMap, int w, int h, uint32_t id_base, hipStream_t s);

// copies `bytes` (rounded up to 16) from pinned, device-mapped host memory to device memory with a kernel
void launch_upload(const void* host_mapped, void* dst, size_t bytes, hipStream_t s);

// fused create_map + remap of both sources (src/algo.cpp:230-238): triMap + inverse matrices -> trImg1/2.
// The kernel visits every pixel once, so it carries a per-pixel rider that would otherwise be a launch of its own:
//   m2 / mask  lbmask = clamp((1-mr) - m2*mr) in double with one rounding (src/algo.cpp:254-257); skipped when m2 is null.
struct WarpExtras {
    uint32_t id_base = 0;                       // see launch_raster
    const float* m2 = nullptr;
    float* mask = nullptr;
    double alpha = 0, beta = 0;
    int out_pitch = 0;                          // pixels per row of tr1 / tr2 and of `mask` (0 = w; PyrLevel::pitch of level 0).  m2 and the sources are tight.
};
// t0 / t1 (optional) are attached to the dispatch: the kernel's own begin / end timestamps.
void launch_warp(int32_t* triMap, const float* inv1, const float* inv2, const uint8_t* c1, const uint8_t* c2,
                 uint8_t* tr1, uint8_t* tr2, int w, int h, const WarpExtras& ex, hipStream_t s,
                 hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// The same warp from per-triangle records (frame_plan.h: pack_warp_records), for frames whose matrices the host admitted
// and geometries warp_fast_geometry() accepts (kernels_warp_fast.hip).  Bit-identical output, about half the instructions.
bool warp_fast_geometry(int w, int h);
bool warp_bin_geometry(int w, int h);       // what launch_warp_bin takes: any width from 8 up (rows of the outputs: WarpExtras::out_pitch, a multiple of 4)
void launch_warp_fast(int32_t* triMap, const float* records, int n_records, const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2,
                      int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// paint_triangles + create_map + remap without an id map in HBM (kernels_warp_bin.hip).  launch_tile_expand turns the host plan's
// fill-edge tables (RasterTri), outline segments (OutlineSeg) and per-tile triangle lists into one byte per pixel — the number of the
// last covering entry of the pixel's tile — plus the tile's warp records in fixed slots, and depends on the plan only (it runs on the
// plan-upload stream); launch_warp_bin warps from those.  Same geometries as launch_warp_fast;
// tile_w = warp_bin_tile_width(w, h) is also what the host bins with.
int warp_bin_tile_width(int w, int h);
size_t warp_bin_entry_bytes();
int warp_bin_max_tile_entries();                                   // a tile's list may not be longer (ids are bytes)
size_t warp_bin_data_bytes(size_t n_tiles, size_t max_entries);    // size of the tile_data allocation
void launch_tile_expand(const float* records, const void* raster_tris, const void* outline, const int* tile_off, const uint16_t* tile_tris,
                        void* tile_data, int tile_w, int w, int h, hipStream_t s);
void launch_warp_bin(const float* records, const void* tile_data, size_t tile_data_bytes, const int* tile_off, int tile_w,
                     const uint8_t* c1, const uint8_t* c2, uint8_t* tr1, uint8_t* tr2,
                     int w, int h, const WarpExtras& ex, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);

// one Gaussian-pyramid reduction step for L, R (3 channels) and the mask (1 channel) in one launch.
// level 0 of L/R is the u8 warped image (converted on the fly), deeper levels are float.
// mask_ab (level 0 only, frames whose geometry pyr_level0_vec_ok admits): srcM is the pair's m2 field and mask_ab points to this
// frame's (alpha, beta) in device memory — the kernel computes lbmask = clamp(alpha + m2 * beta) on the values it loads.
// sp / mp / dp: pixels per row of srcL / srcR, of srcM and of the destination level (0 = the level's width)
void launch_pyrdown(const void* srcL, const void* srcR, const float* srcM, bool src_u8,
                    float* dstL, float* dstR, float* dstM, int sw, int sh, hipStream_t s, const double* mask_ab = nullptr, int sp = 0, int mp = 0, int dp = 0);

// one collapse step: B_i = pyrUp(B_{i+1}) + mix(G_i - pyrUp(G_{i+1}))   (src/blend.hpp:58-77)
void launch_collapse(const void* gL, const void* gR, bool g_u8, const float* gM,
                     const float* nL, const float* nR, const float* nB, float* outB,
                     int w, int h, int nw, int nh, hipStream_t s, const double* mask_ab = nullptr, int gp = 0, int mp = 0, int np = 0);      // gp: gL / gR / outB, mp: gM, np: n*
bool pyr_level0_vec_ok(int w, int h);
// lbmask = clamp(alpha + m2 * beta) as an array (debug fetches of frames that did not materialise it)
void launch_lbmask(const float* m2, const double* mask_ab, float* dst, size_t n, hipStream_t s);

// wide-access forms (kernels_pyramid_vec.hip); return false when the level's geometry does not allow them
bool launch_pyrdown_vec(const void* srcL, const void* srcR, const float* srcM, bool src_u8,
                        float* dstL, float* dstR, float* dstM, int sw, int sh, hipStream_t s, const double* mask_ab = nullptr, int sp = 0, int mp = 0, int dp = 0);
bool launch_collapse_vec(const void* gL, const void* gR, bool g_u8, const float* gM, const float* nL, const float* nR, const float* nB,
                         float* outB, int w, int h, int nw, int nh, hipStream_t s, const double* mask_ab = nullptr, int gp = 0, int mp = 0, int np = 0);

// Two levels per launch for levels that are launch-latency bound (kernels_pyramid_fused.hip): the workgroup that owns a
// tile of the second level computes the part of the intermediate level it needs into LDS itself.
constexpr size_t kFuseMaxPixels = 150000;          // first level of the pair: 480 x 270 and below
bool pyrdown2_eligible(int sw, int sh);
// A (sw x sh) -> B -> C for L, R (3 channels) and the mask (1 channel); B and C are both written.
void launch_pyrdown2(const float* aL, const float* aR, const float* aM, float* bL, float* bR, float* bM,
                     float* cL, float* cR, float* cM, int sw, int sh, hipStream_t s, int src_pitch = 0);      // src_pitch: pixels per row of A's buffers (0 = sw)
bool collapse2_eligible(int w, int h, int w1, int h1, int w2, int h2);
// blended level k (w x h) from blended level k+2: g* = Gaussian level k, m* = level k+1, n* = level k+2 (nB blended).
// The blended level k+1 only exists in LDS.
void launch_collapse2(const float* gL, const float* gR, const float* gM, const float* mL, const float* mR, const float* mM,
                      const float* nL, const float* nR, const float* nB, float* outB,
                      int w, int h, int w1, int h1, int w2, int h2, hipStream_t s);

// The way up through SEVERAL small levels in one launch (kernels_pyramid_cone.hip, round 6): blended level `levels[0]` from the blended level
// `levels[n]` (the tail kernel's, or mix_top's), 2 <= n <= kConeMaxLevels; the blended levels in between exist in LDS only, rebuilt by every
// workgroup for the cone under its own 64 x 16 tile.  All four pyramid buffers are addressed through the levels' offsets and pitches.
constexpr int kConeMaxLevels = 6;
constexpr size_t kConeMaxPixels = 600000;          // output level: 960 x 540 and below (level 1 at 1080p, level 2 at 4K)
constexpr int kConeTileX = 64, kConeTileY = 16;
struct ConeLevel { int w, h, pitch; unsigned off3, off1; };
struct ConeArgs { int n; ConeLevel lv[kConeMaxLevels + 1]; };
bool collapse_cone_eligible(const PyrLevel* levels, int n);
void launch_collapse_cone(const float* pyrL, const float* pyrR, const float* pyrM, float* pyrB, const PyrLevel* levels, int n, hipStream_t s);

// blended smallest level = L*m + R*(1-m), for pyramids whose coarsest level does not fit the tail kernel's LDS
void launch_mix_top(const float* l, const float* r, const float* m, float* out, int n_px, hipStream_t s);

// All remaining (small) levels in ONE workgroup (kernels_pyramid_tail.hip): reductions from level `first` down to the last level,
// the smallest-level mix and the collapse back up to level `first`; reads L/R/M of level `first`, writes B of level `first`.
// The level geometry is fixed for the life of a pair, so the host writes one 16-byte tap descriptor per output of every
// multi-pixel level step once (build_pyr_tail_plan); the kernel stages them in LDS beside the levels.
constexpr int kPyrTailMaxWide = 12;                // multi-pixel level steps (the tail starts at a few hundred pixels)
struct PyrTailDown { int c3, c1, desc, so3, so1, do3, do1, stride3, stride1; };   // outputs per 3-channel plane / of the mask, first descriptor, source / destination level offsets (floats, relative to level `first`), source row strides
struct PyrTailUp { int cnt, desc, co3, co1, no3, nstride; };                       // outputs, first descriptor, offsets of the output level (3-channel, mask) and of the coarser level, its row stride
struct PyrTailArgs {
    int n_wide, n3, n1, n_desc;                    // multi-pixel steps; 3-channel / 1-channel floats of levels first..last; descriptors
    int first_c3, first_c1;                        // elements of level `first`
    int nl, one_o3, one_o1;                        // single-pixel levels: number of reductions from the first of them (-1: the last level has more than one pixel), its offsets
    int top_o3, top_o1, top_c3;                    // the last level (used when nl < 0)
    unsigned long long g_off3, g_off1;             // level `first` inside the pyramid buffers
    PyrTailDown down[kPyrTailMaxWide];
    PyrTailUp up[kPyrTailMaxWide];
};
struct PyrTailPlan {
    bool ok = false;                               // false: geometry outside the descriptor fields or LDS; use the per-level kernels
    PyrTailArgs args;
    size_t lds_bytes = 0;
    std::vector<uint32_t> desc;                    // 4 words per descriptor, uploaded once per pair
};
PyrTailPlan build_pyr_tail_plan(const PyrLevel* levels, int first, int last);
bool prepare_pyr_tail(size_t lds_bytes);           // raises the kernel's dynamic-LDS limit; call outside stream capture
void launch_pyr_tail(const float* pyrL, const float* pyrR, const float* pyrM, float* pyrB, const void* d_desc, const PyrTailArgs& args,
                     size_t lds_bytes, hipStream_t s);

// unsharp_mask(lapBlend, 1, amount, 0.3) + convertTo(CV_8U, 255)  (src/util.cpp:113-148, src/algo.cpp:263-265)
// d_amount (device, may be null): when set, the tile kernel reads the amount from there instead of the argument, so
// that the launch can sit in a captured graph while the value changes per frame.
// `done` (optional): an event that completes with the launch's last kernel; for the tile kernel it rides on the dispatch
// itself instead of being a packet of its own behind it.
// src_pitch: pixels per row of src (0 = w; the outputs are tight).
void launch_unsharp(const float* src, float* tmpRow, float* diff, uint8_t* out_u8, float* out_f32_or_null,
                    int w, int h, float amount, const float* d_amount, float threshold, hipStream_t s, hipEvent_t done = nullptr, int src_pitch = 0);

// The streaming form (kernels_unsharp_stream.hip: a wave per 60-pixel column strip, rows kept in registers); launch_unsharp uses it
// for frames of 4 Mpx and more among the geometries it takes (w >= 64, h >= 16; any width and source pitch since round 6) unless POPPY_UNSHARP_TILE is set;
// POPPY_UNSHARP_STREAM forces it for all of them.  norm2_min: see launch_unsharp.
bool unsharp_stream_takes(int w, int h);
bool unsharp_stream_eligible(int w, int h);
void launch_unsharp_stream(const float* src, uint8_t* out_u8, float* out_f32_or_null, int w, int h, float amount, const float* d_amount,
                           double norm2_min, hipStream_t s, hipEvent_t done, int src_pitch = 0);

// u8 cross-dissolve fallback (src/poppy.hpp:129)
void launch_dissolve(const uint8_t* a, const uint8_t* b, uint8_t* dst, size_t n, float wa, float wb, hipStream_t s);

}  // namespace poppy_hip
