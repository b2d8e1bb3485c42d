// orb_detect.h — ORB::detect driver (host) over the kernels of kernels_orb.hip.
#pragma once
#include "kernels_orb.h"
#include "worker.h"
#include <chrono>
#include <string>
#include <vector>

namespace poppy_hip {

struct OrbKeyPoint { float x, y, size, angle, response; int octave, class_id; };   // cv::KeyPoint field order

class OrbDetector {
public:
    ~OrbDetector() { release(); }
    // ORB::create(nfeatures)->detect(gray): host image in, keypoints (order significant) out.  <0 on error (see err).
    int detect(const uint8_t* gray, size_t stride, int w, int h, int nfeatures, hipStream_t s, std::vector<OrbKeyPoint>& out,
               bool gray_on_device = false);   // gray_on_device: `gray` is a device pointer of this GPU (no host round trip)
    // The same in two halves.  detect_begin: everything that does not depend on nfeatures — pyramid, FAST scores, non-maximum suppression, the
    // raster-ordered candidates on the host (returns after they have arrived; 0 or <0 on error).  detect_finish: quotas, retainBest, Harris,
    // retainBest, angles (orb.cpp:803-959).  The pair set-up knows nfeatures only when BOTH images' chains are through (src/extractor.cpp:40-45).
    int detect_begin(const uint8_t* gray, size_t stride, int w, int h, hipStream_t s, bool gray_on_device = false);
    int detect_finish(int nfeatures, hipStream_t s, std::vector<OrbKeyPoint>& out);
    // ORB::compute: 32 bytes per keypoint (kps7 rows in cv::KeyPoint field order); returns n or <0
    int describe(const uint8_t* gray, size_t stride, int w, int h, const float* kps7, int n, hipStream_t s, uint8_t* desc_out);
    // BFMatcher(NORM_HAMMING).match on 32-byte descriptors: out3 rows (queryIdx, trainIdx, distance)
    int hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, hipStream_t s, int* out3);
    int hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, hipStream_t s, int* out4);   // rows (idx0, d0, idx1, d1)
    void release();
    std::string err;

    // state of the last detect() call, reused by describe(): atlas on the device + (level, x, y) per keypoint
    OrbLevelSet S{};
    uint8_t *d_img = nullptr, *d_atlas = nullptr, *d_blur = nullptr, *d_scores = nullptr, *d_desc = nullptr;
    int *d_counters = nullptr, *d_cand = nullptr, *d_kp = nullptr;
    void* d_nms = nullptr;                   // scratch of the ordered compaction (kernels_orb.h: fast_nms_scratch_bytes)
    float* d_val = nullptr;
    int *h_cand = nullptr, *h_kp = nullptr;
    float* h_val = nullptr;
    std::vector<int> last_levels;
    size_t atlas_bytes = 0;
    int W = 0, H = 0, cap = 0, kp_cap = 0;
    static constexpr int kCandHeader = 16;   // ints in front of the candidate lists (device and host): the levels' counts
    size_t last_total = 0;                   // candidates of the last image (the next fetch's guess)

private:
    hipError_t prepare(int w, int h);
    hipError_t grow_candidates(int new_cap);
    hipError_t grow_keypoints(int n);
    Worker helper_, helper2_;                // take level 1 and levels 2..7 of the first retainBest while the caller does level 0
    std::chrono::steady_clock::time_point t_begin_{};
    double ms_fast_ = 0, ms_cand_ = 0;
    size_t level_base_[kOrbLevels] = {};     // where each level's list starts among the fetched candidates
};

}  // namespace poppy_hip
