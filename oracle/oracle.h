// oracle/ — CPU restatement of the reference's morph hot path.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
// The product (poppy_amd/, include/) never includes, links or calls anything in here.
//
// Every function restates, in plain scalar C++ with no OpenCV, the arithmetic of the
// reference routine it cites (paths relative to /root/reference; OCV = third/opencv-4.6.0/modules).
// Parity pins: tests/test_oracle_golden.py checks each stage against fixtures captured from the
// compiled reference (tests/golden/manifest.json -> provenance) and against the known-answer
// data OpenCV's own tests hold (Subdiv2D 65 pts -> 105 triangles).
//
// Build: make -C oracle   (g++ -O2 -ffp-contract=off, no -march: float ops must round like the
// SSE3-baseline reference build — one rounding per multiply and per add, never fused).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace oracle {

struct Pt { float x, y; };
struct IPt { int x, y; };

template <typename T> struct Image {
    int w = 0, h = 0, c = 1;
    std::vector<T> d;
    Image() {}
    Image(int w_, int h_, int c_ = 1) : w(w_), h(h_), c(c_), d((size_t)w_ * h_ * c_) {}
    T* row(int y) { return d.data() + (size_t)y * w * c; }
    const T* row(int y) const { return d.data() + (size_t)y * w * c; }
    size_t size() const { return d.size(); }
};
typedef Image<uint8_t> ImageU8;
typedef Image<float> ImageF;
typedef Image<int32_t> ImageI;

// ---- rounding helpers (x86 semantics) ------------------------------------------------------
int cv_round(double v);          // cvRound(double): cvtsd2si, round-half-even, "indefinite" INT_MIN when out of range
int cv_round_f(float v);         // cvRound(float):  cvtss2si
int border_reflect101(int p, int len);

// ---- geometry.cpp ---------------------------------------------------------------------------
void clip_points(std::vector<Pt>& pts, int cols, int rows);
void make_uniq(const std::vector<Pt>& pts, std::vector<Pt>& out);
void morph_points(const std::vector<Pt>& a, const std::vector<Pt>& b, std::vector<Pt>& out, float s);

// Incremental Delaunay triangulation with the quad-edge bookkeeping of cv::Subdiv2D, so that
// the triangle LIST ORDER equals Subdiv2D::getTriangleList().  Returns false if a point lies
// outside [0,w) x [0,h) (the reference throws there).
bool delaunay_triangles(int w, int h, const std::vector<Pt>& pts, std::vector<float>& tri6);

void triangle_indices(const std::vector<float>& tri6, const std::vector<Pt>& points, std::vector<int>& idx3);
void triangle_int_points(const std::vector<int>& idx3, const std::vector<Pt>& points, std::vector<IPt>& out);

// fillConvexPoly(img, 3 pts, Scalar(value), LINE_8, shift 0) on a 32SC1 image
void fill_triangle(ImageI& img, const IPt* v, int32_t value);
void paint_triangles(ImageI& img, const std::vector<IPt>& tris);
void fill_convex(ImageI& img, const IPt* v, int npts, int32_t value);        // cv::fillConvexPoly, LINE_8, shift 0

bool invert33(const float* m, float* out);          // cv::invert 3x3 CV_32F (double cofactors)
void solve_homography(const IPt* src1, const IPt* src2, float* H);
void morph_homography(const float* H, float ratio, float* M1, float* M2);

// ---- warp.cpp --------------------------------------------------------------------------------
void create_map(const ImageI& triMap, const std::vector<float>& mats, ImageF& mapx, ImageF& mapy);
void remap_bilinear(const ImageU8& src, const ImageF& mapx, const ImageF& mapy, ImageU8& dst);
const int16_t* bilinear_tab();   // [1024][4]

// ---- blend.cpp -------------------------------------------------------------------------------
void u8_to_f32(const ImageU8& src, ImageF& dst);            // convertTo(CV_32F, 1/255)
void f32_to_u8(const ImageF& src, ImageU8& dst);            // convertTo(CV_8U, 255)
void blend_mask(const ImageF& gabor2, double maskRatio, ImageF& mask);
void pyr_down(const ImageF& src, ImageF& dst);
void pyr_up(const ImageF& src, ImageF& dst, int dw, int dh);
void laplacian_blend(const ImageF& l, const ImageF& r, const ImageF& mask, int levels, ImageF& out);
void gaussian_blur_f32(const ImageF& src, const float* k, int ksize, ImageF& dst);
void median3_f32(const ImageF& src, ImageF& dst);
void unsharp_mask(const ImageF& src, float radius, float amount, float threshold, ImageF& dst,
                  ImageF* blurOut = nullptr, ImageF* medOut = nullptr);
extern const float kGauss9Sigma1[9];

// ---- frame.cpp -------------------------------------------------------------------------------
struct FrameDebug {
    std::vector<Pt> morphed, uniq;
    std::vector<float> tri6, H, M1, M2;
    std::vector<int> idx3;
    std::vector<IPt> triMorph;
    ImageI triMap;
    ImageF mapx1, mapy1, mapx2, mapy2, lbmask, lapBlend, unsharp;
    ImageU8 trImg1, trImg2;
};
// morph_images() (src/algo.cpp:178-273).  Returns 0 on success, <0 if the reference would throw.
int morph_images(const ImageU8& c1, const ImageU8& c2, const ImageF& gabor2,
                 const std::vector<Pt>& pts1, const std::vector<Pt>& pts2,
                 double shapeRatio, double maskRatio, int levels,
                 ImageU8& out, std::vector<Pt>& morphedPoints, FrameDebug* dbg = nullptr);

// ---- orb.cpp ---------------------------------------------------------------------------------
// ORB::create(nfeatures)->detect(image): n x 7 floats (x, y, size, angle, response, octave, class_id), order significant
int orb_detect(const ImageU8& image, int nfeatures, std::vector<float>& kps, std::vector<float>* fastLevel0 = nullptr);
void hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, std::vector<int>& out);
// descriptor-matching sketch of src/experiments.hpp:14-144 (BFMatcher::knnMatch k=2, ratioTest, symmetryTest)
void hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, std::vector<int>& out4);
void ratio_test(const int* knn4, int n, float ratio, std::vector<int>& keep);
void symmetry_test(const int* knn12, const int* keep12, int n1, const int* knn21, const int* keep21, int n2, std::vector<int>& out3);
// ORB::compute (WTA_K 2): n x 32 bytes.  trig_mode 0 = cosf/sinf, 1 = (float)cos(double) (diagnostic only)
int orb_describe(const ImageU8& image, const std::vector<float>& kps7, std::vector<uint8_t>& desc, int trig_mode = 0);

// ---- match.cpp -------------------------------------------------------------------------------
struct DistPair { double d; Pt a, b; };
void make_distance_map(const std::vector<Pt>& p1, const std::vector<Pt>& p2, std::vector<DistPair>& out);   // ascending, stable
void filter_invalid_points(std::vector<Pt>& p1, std::vector<Pt>& p2, int cols, int rows);
double morph_distance(const std::vector<Pt>& p1, const std::vector<Pt>& p2, int w, int h);
void match_prepare(std::vector<Pt>& s1, std::vector<Pt>& s2, int w, int h, double tolerance, double initialMorphDist);

// ---- align.cpp: auto-align (src/matcher.cpp:133-244, src/transformer.cpp, src/procrustes.cpp) -------------------
void warp_affine(const ImageU8& src, const double Mfwd[6], ImageU8& dst);      // cv::warpAffine, INTER_LINEAR, BORDER_CONSTANT 0
void rotation_matrix(float cx, float cy, double angle_deg, double scale, double M[6]);
void translate_image(const ImageU8& src, float tx, float ty, ImageU8& dst);
void rotate_image(const ImageU8& src, float cx, float cy, double angle_deg, ImageU8& dst);
void rotate_points(std::vector<Pt>& pts, Pt center, double ang_deg);
void mean2(const std::vector<Pt>& p, double mu[2]);
void sum_squares2(const std::vector<Pt>& p, double ss[2]);
void gemm_at_b(const std::vector<Pt>& X, const std::vector<Pt>& Y, float A[4]);
void svd2(const float A[4], float w[2], float U[4], float Vt[4]);
void transform2(const std::vector<Pt>& src, const float m[4], std::vector<Pt>& dst);
struct ProcrustesResult { float rotation[4]; float scale, error; std::vector<Pt> yprime; float translation[2]; };
void procrustes(const std::vector<Pt>& X, const std::vector<Pt>& Y, ProcrustesResult& R);
void perspective_from_4(const Pt* src, const Pt* dst, double M[9]);
void perspective_points(std::vector<Pt>& pts, const double m[9]);
double retranslate(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h);
double rerotate(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h);
double reprocrustes(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h);
void auto_align(ImageU8& corrected2, std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h);

// ---- prefilter.cpp: Extractor::foreground (src/extractor.cpp:136-229) -------------------------
void bgr_to_gray_u8(const ImageU8& bgr, ImageU8& gray);
struct Mog2 {                                   // BackgroundSubtractorMOG2(500, 16, true), one channel
    static const int kModes = 5;
    int w, h, nframes;
    std::vector<float> weight, variance, mean;  // [pixel][mode]
    std::vector<uint8_t> used;
    Mog2(int w, int h);
    void apply(const ImageU8& img, ImageU8& mask);
};
void accumulate_scaled_u8(ImageU8& acc, const ImageU8& flow, double scale);
void median_blur_u8(const ImageU8& src, int ksize, ImageU8& dst);
extern const int kGauss23Sigma1Fx[23];
void gaussian_blur23_u8(const ImageU8& src, ImageU8& dst);
// stages (optional): flow0, acc0, then per iteration med, flow, acc, blur
void foreground_mask(const ImageU8& grey, ImageU8& fgMask, std::vector<ImageU8>* stages = nullptr);
float cv_log32f(float x);
void equalize_hist(const ImageU8& src, ImageU8& dst);
struct ForegroundDebug {
    ImageU8 grey, masked;
    std::vector<ImageU8> stages;
    float ln20 = 0;
    ImageF lin, logged, final_mask;
};
void foreground(const ImageU8& bgr, ImageU8& fg, ForegroundDebug* dbg = nullptr);

void gabor_bank(int ks, double sigma, double lambd, double gamma, double psi, std::vector<float>& bank);
void gabor_filter_direct(const ImageF& src, int ks, const std::vector<float>& bank, ImageF& dst);   // tolerance comparator
void orb_unsharp_gray(const ImageU8& gf, ImageF& us);

void gaussian_blur_fx_u8(const ImageU8& src, int ksize, double sigma, ImageU8& dst);
// ---- detail.cpp ------------------------------------------------------------------------------
double dft_detail2(const ImageU8& gray);                                     // src/experiments.hpp:305-318 (cv::dft restated bit for bit)
void radial_mask(int width, int height, ImageF& out);                        // draw_radial_gradiant as Extractor::foreground uses it, src/draw.cpp:21-38
void set_radial_mask(bool on);                                               // Settings::enable_radial_mask for foreground()
void radial_gradient(int width, int height, ImageF& out);                    // draw_radial_gradiant2, src/draw.cpp:40-59
void orb_input_image(const ImageU8& good_features, ImageU8& g);              // Extractor::keypoints up to the detector, src/extractor.cpp:50-76
void gaussian_taps_fx(int n, double sigma, std::vector<int>& taps);          // 8.8 fixed-point taps of the 8-bit GaussianBlur (sigma <= 0: OpenCV's defaults)
void convex_hull_points(const std::vector<Pt>& pts, std::vector<Pt>& hull);  // cv::convexHull(points, hull): counter-clockwise, points returned
void dissolve_u8(const ImageU8& img1, const ImageU8& img2, double phase, ImageU8& out);   // img2*phase + img1*(1-phase), src/poppy.hpp:129   // GaussianBlur on 8 bit, fixed-point path
void blur_margin(const ImageU8& src, int union_w, int union_h, ImageU8& dst);            // src/util.cpp:574-602

}  // namespace oracle
