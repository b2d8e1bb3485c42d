// kernels_orb.h — launchers of the ORB / matcher kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace poppy_hip {

constexpr int kOrbBorder = 32;     // max(edgeThreshold 31, ceil(15*sqrt 2) = 22, 4) + 1   (orb.cpp:985-989)
constexpr int kOrbLevels = 8;

struct OrbLevel {
    int w, h;               // level size without border
    size_t stride;          // row pitch of the padded level = w + 64
    size_t offset;          // byte offset of the padded level inside the atlas
    size_t score_offset;    // byte offset of the (unpadded) FAST score plane
    float scale;            // 1.2^level as float
};
struct OrbLevelSet { int n; OrbLevel lv[kOrbLevels]; };

void launch_orb_pyramid(const uint8_t* d_img, int w, int h, size_t stride, uint8_t* atlas, const OrbLevelSet& S, hipStream_t s);
// FAST score + non-maximum suppression; the survivors of every level land in `cand` as (x | y << 16, score) pairs IN RASTER ORDER (FAST's own
// emission order), their number in counters[level].  scratch: fast_nms_scratch_bytes(S) bytes of device memory.
size_t fast_nms_scratch_bytes(const OrbLevelSet& S);
void launch_fast(const uint8_t* atlas, const OrbLevelSet& S, uint8_t* scores, int threshold, int edge, int* counters, int* cand, int cap, void* scratch, hipStream_t s);
void launch_harris(const uint8_t* atlas, const OrbLevelSet& S, const int* kp, int n, float* resp, hipStream_t s);
void launch_ic_angle(const uint8_t* atlas, const OrbLevelSet& S, const int* kp, int n, float* angle, hipStream_t s);

// rBRIEF (WTA_K = 2): blurred copy of every level, then 32 bytes per keypoint.  kp = (level, x, y) in level coordinates.
void launch_orb_blur(const uint8_t* atlas, uint8_t* blurred, const OrbLevelSet& S, hipStream_t s);
// cos_sin: (cos, sin) of each keypoint angle, computed by the host with the reference's libm call
void launch_orb_describe(const uint8_t* blurred, const OrbLevelSet& S, const int* kp, const float* cos_sin, int n, uint8_t* desc, hipStream_t s);

// brute-force Hamming 1-NN: out[i] = (best train index, distance); ties -> lowest train index
void launch_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out2, hipStream_t s);
void launch_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out4, hipStream_t s);   // (idx0, d0, idx1, d1)

}  // namespace poppy_hip
