// oracle/align.cpp — CPU restatement of Poppy's auto-align (TEST INFRASTRUCTURE, never linked into the product).
//
//   Matcher::autoAlign            src/matcher.cpp:133-244
//   Transformer::retranslate      src/transformer.cpp:99-194     translate      :21-25
//   Transformer::rerotate         src/transformer.cpp:196-217    rotate / rotate_points :27-56
//   Transformer::reprocrustes     src/transformer.cpp:260-269
//   Procrustes::procrustes        src/procrustes.cpp:52-114 (use_scaling = true, best_reflection = false)
// and the OpenCV 4.6.0 routines they call, each with the arithmetic of the SSE3-baseline build (OCV = third/opencv-4.6.0/modules):
//   warpAffine (INTER_LINEAR, BORDER_CONSTANT 0)   OCV/imgproc/src/imgwarp.cpp:2155-2290,2540-2640 + remapBilinear
//   getRotationMatrix2D                            OCV/imgproc/src/imgwarp.cpp:3238-3251
//   getPerspectiveTransform + solve(DECOMP_LU)     OCV/imgproc/src/imgwarp.cpp:3277-3304, OCV/core/src/matrix_decomp.cpp:15-70
//   perspectiveTransform                           OCV/core/src/matmul.simd.hpp:1822-1844
//   mean / sum (float -> double, SIMD lane order)  OCV/core/src/mean.dispatch.cpp:121-180, sum.simd.hpp:256-330
//   gemm (GEMMSingleMul<float,double>)             OCV/core/src/matmul.simd.hpp:179-370
//   SVDecomp (JacobiSVDImpl_<float>)               OCV/core/src/lapack.cpp:412-588,1455-1527
//   transform (2 channels)                         OCV/core/src/matmul.dispatch.cpp transform(), matmul.simd.hpp:1394-1407,1703-1715
//   convertTo with scale, Mat +/- Scalar           OCV/core/src/convert_scale.simd.hpp, matrix_expressions.cpp:1293-1350
#include "oracle.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>

namespace oracle {

// ---- images ------------------------------------------------------------------------------------------------------
void warp_affine(const ImageU8& src, const double Mfwd[6], ImageU8& dst) {
    double M[6];
    memcpy(M, Mfwd, sizeof(M));
    {   // invert the forward map (imgwarp.cpp:2622-2631)
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        double b1 = -M[0] * M[2] - M[1] * M[5];
        double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    const int W = src.w, H = src.h, C = src.c;
    ImageU8 out(W, H, C);
    const int16_t* tab = bilinear_tab();
    std::vector<int> adelta(W), bdelta(W);
    for (int x = 0; x < W; ++x) {
        adelta[x] = cv_round(M[0] * x * 1024);
        bdelta[x] = cv_round(M[3] * x * 1024);
    }
    for (int y = 0; y < H; ++y) {
        const int X0 = cv_round((M[1] * y + M[2]) * 1024) + 16;
        const int Y0 = cv_round((M[4] * y + M[5]) * 1024) + 16;
        for (int x = 0; x < W; ++x) {
            const int X = (X0 + adelta[x]) >> 5, Y = (Y0 + bdelta[x]) >> 5;
            int ix = X >> 5, iy = Y >> 5;
            ix = ix > SHRT_MAX ? SHRT_MAX : ix < SHRT_MIN ? SHRT_MIN : ix;
            iy = iy > SHRT_MAX ? SHRT_MAX : iy < SHRT_MIN ? SHRT_MIN : iy;
            const int16_t* w = tab + ((Y & 31) * 32 + (X & 31)) * 4;
            uint8_t* D = &out.d[((size_t)y * W + x) * C];
            const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W;
            const bool y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
            for (int k = 0; k < C; ++k) {
                const int v00 = (x0 && y0) ? src.d[((size_t)iy * W + ix) * C + k] : 0;
                const int v01 = (x1 && y0) ? src.d[((size_t)iy * W + ix + 1) * C + k] : 0;
                const int v10 = (x0 && y1) ? src.d[((size_t)(iy + 1) * W + ix) * C + k] : 0;
                const int v11 = (x1 && y1) ? src.d[((size_t)(iy + 1) * W + ix + 1) * C + k] : 0;
                const int r = (v00 * w[0] + v01 * w[1] + v10 * w[2] + v11 * w[3] + (1 << 14)) >> 15;
                D[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
    }
    dst = out;
}

void rotation_matrix(float cx, float cy, double angle_deg, double scale, double M[6]) {
    const double a = angle_deg * (M_PI / 180);
    const double alpha = std::cos(a) * scale, beta = std::sin(a) * scale;
    M[0] = alpha; M[1] = beta; M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta; M[4] = alpha; M[5] = beta * cx + (1 - alpha) * cy;
}

void translate_image(const ImageU8& src, float tx, float ty, ImageU8& dst) {       // Transformer::translate
    const double M[6] = {1, 0, (double)tx, 0, 1, (double)ty};
    warp_affine(src, M, dst);
}
void rotate_image(const ImageU8& src, float cx, float cy, double angle_deg, ImageU8& dst) {   // Transformer::rotate
    double M[6];
    rotation_matrix(cx, cy, angle_deg, 1.0, M);
    warp_affine(src, M, dst);
}

// ---- points ------------------------------------------------------------------------------------------------------
static Pt rotate_point(Pt p, double ang_deg) {
    const double rad = ang_deg * M_PI / 180.0;
    Pt o;
    o.x = (float)(std::cos(rad) * p.x - std::sin(rad) * p.y);
    o.y = (float)(std::sin(rad) * p.x + std::cos(rad) * p.y);
    return o;
}
void rotate_points(std::vector<Pt>& pts, Pt center, double ang_deg) {
    for (Pt& p : pts) {
        Pt q = rotate_point(Pt{p.x - center.x, p.y - center.y}, ang_deg);
        p = Pt{q.x + center.x, q.y + center.y};
    }
}

// cv::sum of n interleaved 2-channel floats into double[2] (sum.simd.hpp:256-330: two 2-lane double accumulators fed 8 floats a turn)
static void sum2(const float* f, int n, double s[2]) {
    const int len = n * 2;
    int x = 0;
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    for (; x <= len - 8; x += 8) {
        a0 += (double)f[x] + (double)f[x + 4];
        a1 += (double)f[x + 1] + (double)f[x + 5];
        b0 += (double)f[x + 2] + (double)f[x + 6];
        b1 += (double)f[x + 3] + (double)f[x + 7];
    }
    s[0] += a0; s[1] += a1; s[0] += b0; s[1] += b1;
    double s0 = s[0], s1 = s[1];
    for (int i = x / 2; i < n; ++i) { s0 += f[2 * i]; s1 += f[2 * i + 1]; }
    s[0] = s0; s[1] = s1;
}

void mean2(const std::vector<Pt>& p, double mu[2]) {
    double s[2] = {0, 0};
    sum2((const float*)p.data(), (int)p.size(), s);
    const double r = p.empty() ? 0 : 1. / (double)p.size();
    mu[0] = s[0] * r; mu[1] = s[1] * r;
}

void sum_squares2(const std::vector<Pt>& p, double ss[2]) {
    std::vector<float> sq(p.size() * 2);
    for (size_t i = 0; i < p.size(); ++i) { sq[2 * i] = p[i].x * p[i].x; sq[2 * i + 1] = p[i].y * p[i].y; }
    ss[0] = ss[1] = 0;
    sum2(sq.data(), (int)p.size(), ss);
}

// X^T * Y for two n x 2 float matrices (GEMM_1_T through GEMMSingleMul: sequential double dot products)
void gemm_at_b(const std::vector<Pt>& X, const std::vector<Pt>& Y, float A[4]) {
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
            double s = 0;
            for (size_t k = 0; k < X.size(); ++k) s += (double)(r ? X[k].y : X[k].x) * (double)(c ? Y[k].y : Y[k].x);
            A[r * 2 + c] = (float)(s * 1.0);
        }
}

// SVDecomp of a 2 x 2 float matrix: w[2], U (2x2), Vt (2x2)
void svd2(const float A[4], float w[2], float U[4], float Vt[4]) {
    const int m = 2, n = 2;
    float At[4] = {A[0], A[2], A[1], A[3]};          // transpose(src, temp_a)
    float V[4] = {1, 0, 0, 1};
    double W[2];
    const float eps = FLT_EPSILON * 2;
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { float t = At[i * 2 + k]; sd += (double)t * t; }
        W[i] = sd;
    }
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        {
            const int i = 0, j = 1;
            float *Ai = At + i * 2, *Aj = At + j * 2;
            double a = W[i], p = 0, b = W[j];
            for (int k = 0; k < m; ++k) p += (double)Ai[k] * Aj[k];
            if (!(std::abs(p) <= eps * std::sqrt((double)a * b))) {
                p *= 2;
                double beta = a - b, gamma = hypot((double)p, beta);
                float c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = (float)std::sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)std::sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }
                a = b = 0;
                for (int k = 0; k < m; ++k) {
                    float t0 = c * Ai[k] + s * Aj[k];
                    float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                float *Vi = V + i * 2, *Vj = V + j * 2;
                for (int k = 0; k < n; ++k) {
                    float t0 = c * Vi[k] + s * Vj[k];
                    float t1 = -s * Vi[k] + c * Vj[k];
                    Vi[k] = t0; Vj[k] = t1;
                }
            }
        }
        if (!changed) break;
    }
    for (int i = 0; i < n; ++i) {
        double sd = 0;
        for (int k = 0; k < m; ++k) { float t = At[i * 2 + k]; sd += (double)t * t; }
        W[i] = std::sqrt(sd);
    }
    if (W[0] < W[1]) {
        std::swap(W[0], W[1]);
        for (int k = 0; k < 2; ++k) { std::swap(At[k], At[2 + k]); std::swap(V[k], V[2 + k]); }
    }
    w[0] = (float)W[0]; w[1] = (float)W[1];
    // left singular vectors: rows of At scaled by 1/W; a (near) zero singular value gets a vector built from OpenCV's RNG
    uint64_t rng = 0x12345678;
    auto rng_next = [&rng]() { rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32); return (unsigned)rng; };
    const double minval = FLT_MIN;
    for (int i = 0; i < n; ++i) {
        double sd = W[i];
        for (int ii = 0; ii < 100 && sd <= minval; ++ii) {
            const float val0 = (float)(1. / m);
            for (int k = 0; k < m; ++k) At[i * 2 + k] = (rng_next() & 256) != 0 ? val0 : -val0;
            for (int iter = 0; iter < 2; ++iter)
                for (int j = 0; j < i; ++j) {
                    sd = 0;
                    for (int k = 0; k < m; ++k) sd += At[i * 2 + k] * At[j * 2 + k];
                    float asum = 0;
                    for (int k = 0; k < m; ++k) {
                        float t = (float)(At[i * 2 + k] - sd * At[j * 2 + k]);
                        At[i * 2 + k] = t;
                        asum += std::abs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < m; ++k) At[i * 2 + k] *= asum;
                }
            sd = 0;
            for (int k = 0; k < m; ++k) { float t = At[i * 2 + k]; sd += (double)t * t; }
            sd = std::sqrt(sd);
        }
        const float s = (float)(sd > minval ? 1 / sd : 0.);
        for (int k = 0; k < m; ++k) At[i * 2 + k] *= s;
    }
    U[0] = At[0]; U[1] = At[2]; U[2] = At[1]; U[3] = At[3];      // transpose(temp_u, _u)
    memcpy(Vt, V, sizeof(V));
}

// cv::transform of 2-channel floats by a 2 x 2 float matrix m (row major)
void transform2(const std::vector<Pt>& src, const float m[4], std::vector<Pt>& dst) {
    const bool diag = !(std::fabs((double)m[1]) > FLT_EPSILON) && !(std::fabs((double)m[2]) > FLT_EPSILON);
    dst.resize(src.size());
    for (size_t i = 0; i < src.size(); ++i) {
        const float v0 = src[i].x, v1 = src[i].y;
        if (diag) dst[i] = Pt{m[0] * v0 + 0.f, m[3] * v1 + 0.f};
        else dst[i] = Pt{m[0] * v0 + m[1] * v1 + 0.f, m[2] * v0 + m[3] * v1 + 0.f};
    }
}

static void center_and_normalise(const std::vector<Pt>& P, double mu[2], std::vector<Pt>& P0, float& ss, float& norm) {
    mean2(P, mu);
    const float mx = (float)mu[0], my = (float)mu[1];
    P0.resize(P.size());
    for (size_t i = 0; i < P.size(); ++i) P0[i] = Pt{P[i].x - mx, P[i].y - my};
    double s2[2];
    sum_squares2(P0, s2);
    ss = (float)(s2[0] + s2[1]);
    norm = sqrtf(ss);
    const float a = (float)(1. / (double)norm);
    for (Pt& p : P0) p = Pt{p.x * a + 0.f, p.y * a + 0.f};
}

void procrustes(const std::vector<Pt>& X, const std::vector<Pt>& Y, ProcrustesResult& R) {
    double mu_x[2], mu_y[2];
    std::vector<Pt> X0, Y0;
    float ss_X, norm_X, ss_Y, norm_Y;
    center_and_normalise(X, mu_x, X0, ss_X, norm_X);
    center_and_normalise(Y, mu_y, Y0, ss_Y, norm_Y);
    float A[4], s[2], U[4], Vt[4];
    gemm_at_b(X0, Y0, A);
    svd2(A, s, U, Vt);
    float V[4] = {Vt[0], Vt[2], Vt[1], Vt[3]};
    auto v_ut = [&](float out[4]) {                      // V * U.t() (GEMM_2_T: row dot row in double)
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                double acc = 0;
                acc += (double)V[i * 2] * (double)U[j * 2];
                acc += (double)V[i * 2 + 1] * (double)U[j * 2 + 1];
                out[i * 2 + j] = (float)((acc + 0. + 0. + 0.) * 1.0);
            }
    };
    v_ut(R.rotation);
    const double det = (double)R.rotation[0] * R.rotation[3] - (double)R.rotation[1] * R.rotation[2];
    if (det < 0) {
        V[1] = V[1] * -1.f + 0.f; V[3] = V[3] * -1.f + 0.f;
        s[1] = s[1] * -1.f + 0.f;
        v_ut(R.rotation);
    }
    const float rt[4] = {R.rotation[0], R.rotation[2], R.rotation[1], R.rotation[3]};
    std::vector<Pt> rotated;
    transform2(Y0, rt, rotated);
    double tr = 0;
    tr += s[0]; tr += s[1];
    const float trace_TA = (float)tr;
    R.scale = trace_TA * norm_X / norm_Y;
    R.error = 1 - trace_TA * trace_TA;
    const float f = norm_X * trace_TA;
    const float mx = (float)mu_x[0], my = (float)mu_x[1];
    R.yprime.resize(rotated.size());
    for (size_t i = 0; i < rotated.size(); ++i) R.yprime[i] = Pt{(rotated[i].x * f + 0.f) + mx, (rotated[i].y * f + 0.f) + my};
    // translation = mu_x - scale * mu_y * rotation: the expression folds into ONE gemm(mu_y, rotation, -scale, mu_x, 1)
    // (MatOp_GEMM::subtract), which takes the 2-wide small-matrix path: float dot, then double alpha/beta combine
    const float ax = (float)mu_y[0], ay = (float)mu_y[1];
    for (int j = 0; j < 2; ++j) {
        const float t = ax * R.rotation[j] + ay * R.rotation[2 + j];
        R.translation[j] = (float)(t * -(double)R.scale + (j ? my : mx) * 1.0);
    }
}

// getPerspectiveTransform of the first four point pairs (LU with partial pivoting in double); M[8] = 1
void perspective_from_4(const Pt* src, const Pt* dst, double M[9]) {
    double a[8][8], b[8];
    for (int i = 0; i < 4; ++i) {
        a[i][0] = a[i + 4][3] = src[i].x;
        a[i][1] = a[i + 4][4] = src[i].y;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = -src[i].x * dst[i].x;
        a[i][7] = -src[i].y * dst[i].x;
        a[i + 4][6] = -src[i].x * dst[i].y;
        a[i + 4][7] = -src[i].y * dst[i].y;
        b[i] = dst[i].x;
        b[i + 4] = dst[i].y;
    }
    const int m = 8;
    bool ok = true;
    for (int i = 0; i < m && ok; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (std::abs(a[j][i]) > std::abs(a[k][i])) k = j;
        if (std::abs(a[k][i]) < DBL_EPSILON * 100) { ok = false; break; }
        if (k != i) {
            for (int j = i; j < m; ++j) std::swap(a[i][j], a[k][j]);
            std::swap(b[i], b[k]);
        }
        const double d = -1 / a[i][i];
        for (int j = i + 1; j < m; ++j) {
            const double alpha = a[j][i] * d;
            for (int kk = i + 1; kk < m; ++kk) a[j][kk] += alpha * a[i][kk];
            b[j] += alpha * b[i];
        }
    }
    if (ok) {
        for (int i = m - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < m; ++k) s -= a[i][k] * b[k];
            b[i] = s / a[i][i];
        }
        memcpy(M, b, sizeof(b));
    } else {
        for (int i = 0; i < 8; ++i) M[i] = 0;
    }
    M[8] = 1.;
}

void perspective_points(std::vector<Pt>& pts, const double m[9]) {
    for (Pt& p : pts) {
        const float x = p.x, y = p.y;
        double w = x * m[6] + y * m[7] + m[8];
        if (std::fabs(w) > FLT_EPSILON) {
            w = 1. / w;
            p = Pt{(float)((x * m[0] + y * m[1] + m[2]) * w), (float)((x * m[3] + y * m[4] + m[5]) * w)};
        } else p = Pt{0, 0};
    }
}

// ---- Transformer ---------------------------------------------------------------------------------------------------
double retranslate(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h) {
    auto shifted = [&](float dx, float dy) {
        std::vector<Pt> o(p2.size());
        for (size_t i = 0; i < p2.size(); ++i) o[i] = Pt{p2[i].x + dx, p2[i].y + dy};
        return o;
    };
    double mdCurrent = morph_distance(p1, p2, w, h);
    const double mdLeft = morph_distance(p1, shifted(-1, 0), w, h), mdRight = morph_distance(p1, shifted(1, 0), w, h);
    const double mdTop = morph_distance(p1, shifted(0, -1), w, h), mdBottom = morph_distance(p1, shifted(0, 1), w, h);
    long xchange = 0, ychange = 0;
    if (mdLeft < mdCurrent) xchange = -1; else if (mdRight < mdCurrent) xchange = +1;
    if (mdTop < mdCurrent) ychange = -1; else if (mdBottom < mdCurrent) ychange = +1;
    long xProgress = 1, yProgress = 1;
    auto walk = [&](long dx, long dy, long* px, long* py) {
        double last = mdCurrent;
        std::vector<Pt> tmp = p2;
        for (;;) {
            for (Pt& p : tmp) { if (dx) p.x += dx; if (dy) p.y += dy; }
            const double md = morph_distance(p1, tmp, w, h);
            if (md > last) break;
            mdCurrent = last = md;
            if (px) ++*px;
            if (py) ++*py;
        }
    };
    if (xchange != 0 && ychange != 0) walk(xchange, ychange, &xProgress, &yProgress);
    else {
        if (xchange != 0) walk(xchange, 0, &xProgress, nullptr);
        if (ychange != 0) walk(0, ychange, nullptr, &yProgress);
    }
    const float rx = (float)(xchange * xProgress), ry = (float)(ychange * yProgress);
    ImageU8 out;
    translate_image(corrected2, rx, ry, out);
    corrected2 = out;
    for (Pt& p : p2) { p.x += rx; p.y += ry; }
    return morph_distance(p1, p2, w, h);
}

double rerotate(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h) {
    Pt center{0, 0};                                          // average(): Point2f accumulation, then /= n  (src/util.cpp)
    for (const Pt& p : p2) { center.x += p.x; center.y += p.y; }
    center.x /= p2.size(); center.y /= p2.size();
    double lowest = std::numeric_limits<double>::max(), selected = 0;
    for (size_t i = 0; i < 1080; ++i) {
        std::vector<Pt> tmp = p2;
        rotate_points(tmp, center, i / 3.0);
        const double md = morph_distance(p1, tmp, w, h);
        if (md < lowest) { lowest = md; selected = i / 3.0; }
    }
    ImageU8 out;
    rotate_image(corrected2, center.x, center.y, -selected, out);
    corrected2 = out;
    rotate_points(p2, center, selected);
    return lowest;
}

double reprocrustes(ImageU8& corrected2, const std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h) {
    ProcrustesResult R;
    procrustes(p1, p2, R);
    double M[9];
    perspective_from_4(p2.data(), R.yprime.data(), M);
    perspective_points(p2, M);
    ImageU8 out;
    warp_affine(corrected2, M, out);                       // perspectiveMat.pop_back(): its first two rows
    corrected2 = out;
    return morph_distance(p1, p2, w, h);
}

void auto_align(ImageU8& corrected2, std::vector<Pt>& p1, std::vector<Pt>& p2, int w, int h) {
    const double initialDist = morph_distance(p1, p2, w, h);
    double lastDistTrans = initialDist, distTrans = initialDist, lastDistRot = initialDist, distRot = initialDist;
    double lastDistProcr = initialDist, distProcr = initialDist;
    ImageU8 lastC2;
    std::vector<Pt> last2;
    bool progress;
    do {
        progress = false;
        do {
            lastDistTrans = distTrans; lastC2 = corrected2; last2 = p2;
            distTrans = retranslate(corrected2, p1, p2, w, h);
            if (distTrans < lastDistTrans) progress = true;
        } while (distTrans < lastDistTrans);
        if (distTrans >= lastDistTrans) { corrected2 = lastC2; p2 = last2; distProcr = lastDistTrans; } else distProcr = distTrans;
        do {
            lastDistProcr = distProcr; lastC2 = corrected2; last2 = p2;
            distProcr = reprocrustes(corrected2, p1, p2, w, h);
            if (distProcr < lastDistProcr) progress = true;
        } while (lastDistProcr > distProcr);
        if (distProcr >= lastDistProcr) { corrected2 = lastC2; p2 = last2; distRot = lastDistProcr; } else distRot = distProcr;
        do {
            lastDistRot = distRot; lastC2 = corrected2; last2 = p2;
            distRot = rerotate(corrected2, p1, p2, w, h);
            if (distRot < lastDistRot) progress = true;
        } while (distRot < lastDistRot);
        if (distRot >= lastDistRot) { corrected2 = lastC2; p2 = last2; distTrans = lastDistRot; } else distTrans = distRot;
    } while (progress);
}

}  // namespace oracle
