// frame_plan.h — host-side planning of one morph frame (the latency-bound, order-dependent part of
// poppy::morph_images, src/algo.cpp:184-228): point interpolation, Delaunay triangulation with the
// reference's triangle ORDER, integer triangle corners and the two inverse affine matrices per
// triangle.  The per-pixel work that consumes this plan runs in the HIP kernels.
#pragma once
#include <cstdint>
#include <vector>

namespace poppy_hip {

struct P2f { float x, y; };

// Fill edges of one triangle, precomputed from FillConvexPoly's two-chain state machine (drawing.cpp:1164-1252):
// chain i (0: vertices imin, imin+1, ...; 1: imin, imin+2, ...) has at most two segments; segment k of chain i is
// slot i*2+k and holds the first row it is valid for, the 16.16 position on that row and the per-row increment.
// Rows [ymin, ystop) are painted.  The device evaluates a row with one multiply-add per chain.
struct RasterTri {
    int ymin, ystop, n0, n1;
    int ybeg[4];
    long long ex[4], edx[4];
};
static_assert(sizeof(RasterTri) == 96, "RasterTri is shared with the device");

// One outline segment of a triangle as Line() walks it (drawing.cpp:80-297): clipped to the image (clipLine, in its double
// arithmetic) and swapped so that x increases.  Step k = 0 .. major puts a pixel at (x0 + k, y0 +- m_k) — or, when steep,
// (x0 + m_k, y0 +- k) — with m_k = max(0, ceil((2 minor k - major) / (2 major))).  flags: bit 0 steep, bit 1 the y direction is
// negative; major < 0 marks a segment that lies outside the image altogether.
struct OutlineSeg { int x0, y0, major, minor_flags; };     // minor_flags = minor | flags << 24
static_assert(sizeof(OutlineSeg) == 16, "OutlineSeg is shared with the device");

struct FramePlan {
    std::vector<P2f> morphed;        // n   (after clip_points)
    std::vector<int> idx3;           // T*3 indices into the point sets
    std::vector<int> tri_xy;         // T*6 truncated morphed corners (x0,y0,x1,y1,x2,y2)
    std::vector<float> M1, M2;       // T*9 forward matrices (diagnostics)
    std::vector<float> inv1, inv2;   // T*9 inverse matrices, what create_map actually uses
    std::vector<RasterTri> raster;   // T fill-edge tables
    std::vector<int> work;           // raster work list: (triangle, -1) = outline, (triangle, row chunk) = fill rows
    std::vector<OutlineSeg> outline; // T*3 segments: (v2,v0), (v0,v1), (v1,v2) — the order fillConvexPoly draws them in
    // triangles per warp tile (kernels_warp_bin.hip): tile_off[tile] .. tile_off[tile + 1] indexes tile_tris, ascending ids
    std::vector<int> tile_off;
    std::vector<uint16_t> tile_tris;
    int tile_w = 0, tile_h = 0;
    bool bins_ok = false;            // build_tile_bins succeeded for (tile_w, tile_h)
    int max_tile_entries = 0;        // longest tile list
    int n_tris = 0;
};
// Bins the triangles of `plan` into tile_w x tile_h tiles of the w x h image (every tile a triangle's bounding box touches).
// Returns false when a triangle index does not fit 16 bits or the list outgrows max_entries (the caller then takes the
// id-map path).
bool build_tile_bins(FramePlan& plan, int w, int h, int tile_w, int tile_h, size_t max_entries);
constexpr int kPlanRasterRows = 16;  // rows of one triangle per raster work item

// Returns 0, or -3 (POPPY_E_RANGE) when a point is outside [0,w)x[0,h) where Subdiv2D::insert throws.
int plan_frame(int w, int h, const std::vector<P2f>& src1, const std::vector<P2f>& src2,
               double shape_ratio, FramePlan& plan);

// Per-triangle records of the fast warp kernel (kernels_warp_fast.hip): (T+1) x 20 floats, record 0 = identity, record
// i+1 = triangle i with both inverse matrices interleaved as
//   {h0,h3}a {h1,h4}a {h2,h5}a {h0,h3}b {h1,h4}b {h2,h5}b {h6a,h6b} {h7a,h7b} {h8a,h8b} pad pad      (a = inv1, b = inv2)
// An all-zero matrix (singular triangle, lapack.cpp:1044-1045) gets h8 = 1e-5f, which is what create_map substitutes for
// its z = 0 (algo.cpp:166-167).  Returns false when some matrix could take the kernel's bare division sequence outside the
// range where it equals IEEE division (non-finite or > 2^40 entries, or a denominator h6*x + h7*y + h8 that can leave
// [2^-20, 2^20] in magnitude or change sign over the pixels that use the record: the bounding box of the triangle's integer
// corners `tri_xy`, or the whole image without them); such frames use the general kernel.
constexpr int kWarpRecordFloats = 20;
bool pack_warp_records(const float* inv1, const float* inv2, int n_tris, int w, int h, float* records, const int* tri_xy = nullptr);

void clip_points_ref(std::vector<P2f>& pts, int cols, int rows);   // src/util.cpp:453-460
void unique_points_ref(const std::vector<P2f>& pts, std::vector<P2f>& out);   // make_uniq, src/util.cpp:541-548: first occurrences, input order
bool invert3x3(const float* m, float* out);                        // OCV/core/src/lapack.cpp:965-993

}  // namespace poppy_hip
