// frame_plan.h — host-side planning of one morph frame (the latency-bound, order-dependent part of
// poppy::morph_images, src/algo.cpp:184-228): point interpolation, Delaunay triangulation with the
// reference's triangle ORDER, integer triangle corners and the two inverse affine matrices per
// triangle.  The per-pixel work that consumes this plan runs in the HIP kernels.
#pragma once
#include <cstdint>
#include <vector>

namespace poppy_hip {

struct P2f { float x, y; };

struct FramePlan {
    std::vector<P2f> morphed;        // n   (after clip_points)
    std::vector<int> idx3;           // T*3 indices into the point sets
    std::vector<int> tri_xy;         // T*6 truncated morphed corners (x0,y0,x1,y1,x2,y2)
    std::vector<float> M1, M2;       // T*9 forward matrices (diagnostics)
    std::vector<float> inv1, inv2;   // T*9 inverse matrices, what create_map actually uses
    int n_tris = 0;
};

// Returns 0, or -3 (POPPY_E_RANGE) when a point is outside [0,w)x[0,h) where Subdiv2D::insert throws.
int plan_frame(int w, int h, const std::vector<P2f>& src1, const std::vector<P2f>& src2,
               double shape_ratio, FramePlan& plan);

void clip_points_ref(std::vector<P2f>& pts, int cols, int rows);   // src/util.cpp:453-460
bool invert3x3(const float* m, float* out);                        // OCV/core/src/lapack.cpp:965-993

}  // namespace poppy_hip
