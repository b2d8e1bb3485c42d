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

struct FramePlan {
    std::vector<P2f> morphed;        // n   (after clip_points)
    std::vector<int> idx3;           // T*3 indices into the point sets
    std::vector<int> tri_xy;         // T*6 truncated morphed corners (x0,y0,x1,y1,x2,y2)
    std::vector<float> M1, M2;       // T*9 forward matrices (diagnostics)
    std::vector<float> inv1, inv2;   // T*9 inverse matrices, what create_map actually uses
    std::vector<RasterTri> raster;   // T fill-edge tables
    std::vector<int> work;           // raster work list: (triangle, -1) = outline, (triangle, row chunk) = fill rows
    int n_tris = 0;
};
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
// [2^-20, 2^20] in magnitude or change sign over the image); such frames use the general kernel.
constexpr int kWarpRecordFloats = 20;
bool pack_warp_records(const float* inv1, const float* inv2, int n_tris, int w, int h, float* records);

void clip_points_ref(std::vector<P2f>& pts, int cols, int rows);   // src/util.cpp:453-460
void unique_points_ref(const std::vector<P2f>& pts, std::vector<P2f>& out);   // make_uniq, src/util.cpp:541-548: first occurrences, input order
bool invert3x3(const float* m, float* out);                        // OCV/core/src/lapack.cpp:965-993

}  // namespace poppy_hip
