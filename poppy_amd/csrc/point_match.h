// point_match.h — Poppy's keypoint pairing (greedy nearest neighbour on POSITIONS + statistical threshold).
//
// This stage is a strictly sequential claim process over ~500 points (each claim changes what the next
// point may take) followed by double / long-double statistics whose libm results (hypotf, hypotl, sqrt, pow)
// must equal the reference's bit for bit; it costs < 1 ms once per pair.  It therefore runs on the host with
// the same libm, exactly like the order-defining std::nth_element of ORB.
#pragma once
#include "frame_plan.h"
#include <vector>

namespace poppy_hip {

struct PointPair { double dist; P2f a, b; };

long hypotf_selfcheck(long n, unsigned long long seed);                         // mismatches between hyp() and libm hypotf
void greedy_pairs(const std::vector<P2f>& src1, const std::vector<P2f>& src2, std::vector<PointPair>& pairs);   // make_distance_map
void drop_out_of_image(std::vector<P2f>& p1, std::vector<P2f>& p2, int cols, int rows);                         // filter_invalid_points + resize
double morph_distance_ref(const std::vector<P2f>& p1, const std::vector<P2f>& p2, int w, int h);               // morph_distance
// morph_distance in parts, for callers that compute the O(N^2) sums elsewhere (auto_align.cpp scores rotation candidates on the GPU)
double hull_area_of(const std::vector<P2f>& pts);                                 // |contourArea(approxPolyDP(convexHull(pts), 0.001))|
float inner_offset_sum(const std::vector<P2f>& a, const std::vector<P2f>& b);     // float sum over i, j of (a[i].x - b[j].x) + (a[i].y - b[j].y)
double morph_distance_combine(float total, size_t n_pairs, float inner1_sum, float inner2_sum, size_t n1, size_t n2,
                              double area1, double area2, int w, int h);
void match_and_prepare(std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h, double tolerance, double initial_morph_dist);
// the two steps sharing one greedy pairing (the reference pairs the same sets twice: morph_distance, then match)
class Worker;
// `helper`: a persistent thread for the O(n^2) offset sums that run beside the pairing (null: a std::thread is made for them)
double morph_distance_pairs(const std::vector<P2f>& p1, const std::vector<P2f>& p2, int w, int h, std::vector<PointPair>& pairs, Worker* helper = nullptr);
void match_and_prepare_from(const std::vector<PointPair>& pairs, std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h,
                            double tolerance, double initial_morph_dist);

void ratio_symmetry(const int* knn12, int n1, const int* knn21, int n2, float ratio, std::vector<int>& out3);      // experiments.hpp:14-144
void add_image_corners(std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h);                              // add_corners

}  // namespace poppy_hip
