// auto_align.h — Poppy's auto-align (Settings::enable_auto_align): the second image and its points are moved
// onto the first image's points by alternating translation, Procrustes and rotation searches until none improves the
// morph distance (Matcher::autoAlign, src/matcher.cpp:133-244; Transformer, src/transformer.cpp; Procrustes, src/procrustes.cpp).
//
// Split:  * the image (corrected2) stays on the GPU: every step's cv::warpAffine is one kernel (k_warp_affine, exact
//           fixed-point arithmetic of OCV/imgproc/src/imgwarp.cpp:2155-2290), the "keep / undo" of the search is a device copy;
//         * the searches score their candidates (1080 rotation angles; the shifts of the translation search, a batch at a
//           time) on the GPU: one wave per candidate replays morph_distance's greedy claims and its ordered float sum, one
//           lane per candidate the O(N^2) offset sum (kernels_align.hip); the hull areas (Sklansky scan + Douglas-Peucker,
//           ~30 us each) and the long-double combination run on host threads (point_match.h);
//         * Procrustes / perspective fit on ~500 points: host, double and float exactly as OpenCV evaluates them.
#pragma once
#include "point_match.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace poppy_hip {

// cv::warpAffine(src, dst, M (2x3 forward map), src.size(), INTER_LINEAR, BORDER_CONSTANT, 0) on device u8x3 images (contiguous rows).
// d_tables: device scratch of at least 2 * (w + h) ints.  Returns false on a HIP error.
bool warp_affine_device(const uint8_t* d_src, uint8_t* d_dst, int w, int h, const double M[6], int* d_tables, hipStream_t s);
// Candidate scoring (kernels_align.hip): for n_cand versions of the second point set (n_cand x n points, device), per
// candidate the float sum of the greedy pairs' distances, the number of pairs and the raw inner offset sum.
constexpr int kAlignMaxPoints = 4096;             // the pairing kernel keeps 16 bytes per point in LDS
void launch_candidate_scores(const float* d_p1, const float* d_sets, int n, int n_cand, float* d_total, int* d_npairs, float* d_inner, hipStream_t s);
void rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6]);      // getRotationMatrix2D

struct ProcrustesFit { float rotation[4]; float scale, error; std::vector<P2f> yprime; };
void procrustes_fit(const std::vector<P2f>& X, const std::vector<P2f>& Y, ProcrustesFit& R);   // Procrustes(true, false)::procrustes
void perspective_from_4(const P2f* src, const P2f* dst, double M[9]);                           // getPerspectiveTransform (LU)
void perspective_points(std::vector<P2f>& pts, const double M[9]);                              // perspectiveTransform

class AutoAligner {
public:
    ~AutoAligner() { release(); }
    // d_img: corrected2 on the device (w*h*3, updated in place); p2 updated in place.  Returns 0 or a negative status; err set.
    int run(uint8_t* d_img, int w, int h, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s, double* final_distance);
    // single steps (tests): 0 retranslate, 1 reprocrustes, 2 rerotate; returns the step's distance through *dist
    int step(int which, uint8_t* d_img, int w, int h, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s, double* dist);
    void release();
    std::string err;
private:
    int ensure(int w, int h);
    bool warp_in_place(uint8_t* d_img, const double M[6], hipStream_t s);
    double retranslate(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s);
    double rerotate(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s);
    double reprocrustes(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s);
    uint8_t *d_tmp = nullptr, *d_last = nullptr;
    int* d_tables = nullptr;
    unsigned char* d_score = nullptr;              // first set, candidate sets and per-candidate results of a search
    size_t d_score_bytes = 0;
    // morph_distance(p1, set) for every candidate set (n_cand x n points, contiguous): O(N^2) parts on the GPU, hull areas on host threads
    bool score_sets(const std::vector<P2f>& p1, const std::vector<P2f>& sets, int n_cand, hipStream_t s, std::vector<double>& out);
    int W = 0, H = 0;
    bool failed = false;
};

}  // namespace poppy_hip
