// auto_align.cpp — see auto_align.h.  Reference routines, OCV = third/opencv-4.6.0/modules:
//   Matcher::autoAlign src/matcher.cpp:133-244; Transformer::retranslate / rerotate / reprocrustes src/transformer.cpp:99-217,260-269
//   Procrustes::procrustes src/procrustes.cpp:52-114
//   cv::mean / cv::sum (float data, double lanes) OCV/core/src/mean.dispatch.cpp:121-180, sum.simd.hpp:256-330
//   gemm OCV/core/src/matmul.simd.hpp:179-370; SVDecomp OCV/core/src/lapack.cpp:412-588,1455-1527; transform matmul.simd.hpp:1394-1407
//   getPerspectiveTransform OCV/imgproc/src/imgwarp.cpp:3277-3304 (LU: OCV/core/src/matrix_decomp.cpp:15-70);
//   perspectiveTransform OCV/core/src/matmul.simd.hpp:1822-1844; getRotationMatrix2D imgwarp.cpp:3238-3251
#include "auto_align.h"
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace poppy_hip {

void rotation_matrix_2d(float cx, float cy, double angle_deg, double scale, double M[6]) {
    const double a = angle_deg * (M_PI / 180);
    const double alpha = std::cos(a) * scale, beta = std::sin(a) * scale;
    M[0] = alpha; M[1] = beta; M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta; M[4] = alpha; M[5] = beta * cx + (1 - alpha) * cy;
}

namespace {

// cv::sum over n two-channel floats: on the SSE2 path two 2-lane double registers take 8 floats a turn, the rest goes one pair at a time
void channel_sums(const float* f, int n, double s[2]) {
    const int len = 2 * n;
    int x = 0;
    double lo0 = 0, lo1 = 0, hi0 = 0, hi1 = 0;
    for (; x <= len - 8; x += 8) {
        lo0 += (double)f[x] + (double)f[x + 4];     lo1 += (double)f[x + 1] + (double)f[x + 5];
        hi0 += (double)f[x + 2] + (double)f[x + 6]; hi1 += (double)f[x + 3] + (double)f[x + 7];
    }
    double s0 = 0, s1 = 0;
    s0 += lo0; s1 += lo1; s0 += hi0; s1 += hi1;
    for (int i = x / 2; i < n; ++i) { s0 += f[2 * i]; s1 += f[2 * i + 1]; }
    s[0] = s0; s[1] = s1;
}

// X - mean(X), then divided by its Frobenius norm (both through float images, as the Mat expressions evaluate them)
void centre_and_scale(const std::vector<P2f>& P, float mean_f[2], std::vector<P2f>& Q, float& norm) {
    const int n = (int)P.size();
    double s[2];
    channel_sums((const float*)P.data(), n, s);
    const double inv_n = n ? 1. / n : 0;
    mean_f[0] = (float)(s[0] * inv_n); mean_f[1] = (float)(s[1] * inv_n);
    Q.resize(n);
    std::vector<float> sq((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        Q[i] = P2f{P[i].x - mean_f[0], P[i].y - mean_f[1]};
        sq[2 * i] = Q[i].x * Q[i].x; sq[2 * i + 1] = Q[i].y * Q[i].y;
    }
    channel_sums(sq.data(), n, s);
    const float ss = (float)(s[0] + s[1]);
    norm = sqrtf(ss);
    const float k = (float)(1. / (double)norm);
    for (P2f& q : Q) q = P2f{q.x * k + 0.f, q.y * k + 0.f};
}

// singular value decomposition of a 2x2 float matrix by one-sided Jacobi rotations (OpenCV's JacobiSVD for m = n = 2)
void svd_2x2(const float A[4], float w[2], float U[4], float Vt[4]) {
    float At[2][2] = {{A[0], A[2]}, {A[1], A[3]}};
    float V[2][2] = {{1, 0}, {0, 1}};
    double W[2];
    const float eps = FLT_EPSILON * 2;
    for (int i = 0; i < 2; ++i) W[i] = (double)At[i][0] * At[i][0] + (double)At[i][1] * At[i][1];
    for (int iter = 0; iter < 30; ++iter) {
        double p = (double)At[0][0] * At[1][0];
        p += (double)At[0][1] * At[1][1];
        double a = W[0], b = W[1];
        if (std::abs(p) <= eps * std::sqrt(a * b)) break;
        p *= 2;
        const double beta = a - b, gamma = hypot(p, beta);
        float c, s;
        if (beta < 0) {
            const double delta = (gamma - beta) * 0.5;
            s = (float)std::sqrt(delta / gamma);
            c = (float)(p / (gamma * s * 2));
        } else {
            c = (float)std::sqrt((gamma + beta) / (gamma * 2));
            s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < 2; ++k) {
            const float t0 = c * At[0][k] + s * At[1][k], t1 = -s * At[0][k] + c * At[1][k];
            At[0][k] = t0; At[1][k] = t1;
            a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[0] = a; W[1] = b;
        for (int k = 0; k < 2; ++k) {
            const float t0 = c * V[0][k] + s * V[1][k], t1 = -s * V[0][k] + c * V[1][k];
            V[0][k] = t0; V[1][k] = t1;
        }
    }
    for (int i = 0; i < 2; ++i) W[i] = std::sqrt((double)At[i][0] * At[i][0] + (double)At[i][1] * At[i][1]);
    if (W[0] < W[1]) {
        std::swap(W[0], W[1]);
        for (int k = 0; k < 2; ++k) { std::swap(At[0][k], At[1][k]); std::swap(V[0][k], V[1][k]); }
    }
    w[0] = (float)W[0]; w[1] = (float)W[1];
    // rows of At / W are the left singular vectors; a vanishing singular value gets a vector from cv::RNG(0x12345678)
    uint64_t state = 0x12345678;
    auto next = [&state]() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; };
    for (int i = 0; i < 2; ++i) {
        double sd = W[i];
        for (int tries = 0; tries < 100 && sd <= FLT_MIN; ++tries) {
            for (int k = 0; k < 2; ++k) At[i][k] = (next() & 256) != 0 ? 0.5f : -0.5f;
            for (int pass = 0; pass < 2; ++pass)
                for (int j = 0; j < i; ++j) {
                    sd = 0;
                    for (int k = 0; k < 2; ++k) sd += At[i][k] * At[j][k];
                    float asum = 0;
                    for (int k = 0; k < 2; ++k) { const float t = (float)(At[i][k] - sd * At[j][k]); At[i][k] = t; asum += std::abs(t); }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (int k = 0; k < 2; ++k) At[i][k] *= asum;
                }
            sd = std::sqrt((double)At[i][0] * At[i][0] + (double)At[i][1] * At[i][1]);
        }
        const float r = (float)(sd > FLT_MIN ? 1 / sd : 0.);
        for (int k = 0; k < 2; ++k) At[i][k] *= r;
    }
    U[0] = At[0][0]; U[1] = At[1][0]; U[2] = At[0][1]; U[3] = At[1][1];
    Vt[0] = V[0][0]; Vt[1] = V[0][1]; Vt[2] = V[1][0]; Vt[3] = V[1][1];
}

}  // namespace

void procrustes_fit(const std::vector<P2f>& X, const std::vector<P2f>& Y, ProcrustesFit& R) {
    float mean_x[2], mean_y[2], norm_x, norm_y;
    std::vector<P2f> X0, Y0;
    centre_and_scale(X, mean_x, X0, norm_x);
    centre_and_scale(Y, mean_y, Y0, norm_y);
    float A[4];                                              // X0^T * Y0: double dot products in list order, rounded once
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 2; ++c) {
            double acc = 0;
            for (size_t k = 0; k < X0.size(); ++k) acc += (double)(r ? X0[k].y : X0[k].x) * (double)(c ? Y0[k].y : Y0[k].x);
            A[2 * r + c] = (float)acc;
        }
    float sv[2], U[4], Vt[4];
    svd_2x2(A, sv, U, Vt);
    float V[4] = {Vt[0], Vt[2], Vt[1], Vt[3]};
    auto v_times_ut = [&]() {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                double acc = (double)V[2 * i] * (double)U[2 * j];
                acc += (double)V[2 * i + 1] * (double)U[2 * j + 1];
                R.rotation[2 * i + j] = (float)acc;
            }
    };
    v_times_ut();
    if ((double)R.rotation[0] * R.rotation[3] - (double)R.rotation[1] * R.rotation[2] < 0) {      // no reflections
        V[1] = V[1] * -1.f + 0.f; V[3] = V[3] * -1.f + 0.f;
        sv[1] = sv[1] * -1.f + 0.f;
        v_times_ut();
    }
    // Y0 * rotation (cv::transform with rotation^T; a matrix diagonal within FLT_EPSILON takes the diagonal routine)
    const float m01 = R.rotation[2], m10 = R.rotation[1];
    const bool diagonal = !(std::fabs((double)m01) > FLT_EPSILON) && !(std::fabs((double)m10) > FLT_EPSILON);
    double trace = 0;
    trace += sv[0]; trace += sv[1];
    const float trace_f = (float)trace;
    R.scale = trace_f * norm_x / norm_y;
    R.error = 1 - trace_f * trace_f;
    const float gain = norm_x * trace_f;
    R.yprime.resize(Y0.size());
    for (size_t i = 0; i < Y0.size(); ++i) {
        const float v0 = Y0[i].x, v1 = Y0[i].y;
        const float rx = diagonal ? R.rotation[0] * v0 + 0.f : R.rotation[0] * v0 + m01 * v1 + 0.f;
        const float ry = diagonal ? R.rotation[3] * v1 + 0.f : m10 * v0 + R.rotation[3] * v1 + 0.f;
        R.yprime[i] = P2f{(rx * gain + 0.f) + mean_x[0], (ry * gain + 0.f) + mean_x[1]};
    }
}

void perspective_from_4(const P2f* src, const P2f* dst, double M[9]) {
    double a[8][8], b[8];
    for (int i = 0; i < 4; ++i) {
        a[i][0] = a[i + 4][3] = src[i].x;
        a[i][1] = a[i + 4][4] = src[i].y;
        a[i][2] = a[i + 4][5] = 1;
        a[i][3] = a[i][4] = a[i][5] = a[i + 4][0] = a[i + 4][1] = a[i + 4][2] = 0;
        a[i][6] = -src[i].x * dst[i].x; a[i][7] = -src[i].y * dst[i].x;
        a[i + 4][6] = -src[i].x * dst[i].y; a[i + 4][7] = -src[i].y * dst[i].y;
        b[i] = dst[i].x; b[i + 4] = dst[i].y;
    }
    bool regular = true;
    for (int i = 0; i < 8; ++i) {                       // LU with partial pivoting, right-hand side carried along
        int piv = i;
        for (int j = i + 1; j < 8; ++j)
            if (std::abs(a[j][i]) > std::abs(a[piv][i])) piv = j;
        if (std::abs(a[piv][i]) < DBL_EPSILON * 100) { regular = false; break; }
        if (piv != i) {
            for (int j = i; j < 8; ++j) std::swap(a[i][j], a[piv][j]);
            std::swap(b[i], b[piv]);
        }
        const double d = -1 / a[i][i];
        for (int j = i + 1; j < 8; ++j) {
            const double f = a[j][i] * d;
            for (int k = i + 1; k < 8; ++k) a[j][k] += f * a[i][k];
            b[j] += f * b[i];
        }
    }
    if (regular)
        for (int i = 7; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < 8; ++k) s -= a[i][k] * b[k];
            b[i] = s / a[i][i];
        }
    for (int i = 0; i < 8; ++i) M[i] = regular ? b[i] : 0.;
    M[8] = 1.;
}

void perspective_points(std::vector<P2f>& pts, const double m[9]) {
    for (P2f& p : pts) {
        const float x = p.x, y = p.y;
        double w = x * m[6] + y * m[7] + m[8];
        if (std::fabs(w) > FLT_EPSILON) {
            w = 1. / w;
            p = P2f{(float)((x * m[0] + y * m[1] + m[2]) * w), (float)((x * m[3] + y * m[4] + m[5]) * w)};
        } else p = P2f{0, 0};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
void AutoAligner::release() {
    if (d_tmp) (void)hipFree(d_tmp);
    if (d_last) (void)hipFree(d_last);
    if (d_tables) (void)hipFree(d_tables);
    if (d_score) (void)hipFree(d_score);
    d_tmp = d_last = nullptr; d_tables = nullptr; d_score = nullptr; d_score_bytes = 0; W = H = 0;
}

int AutoAligner::ensure(int w, int h) {
    if (w == W && h == H && d_tmp) return 0;
    release();
    const size_t bytes = (size_t)w * h * 3;
    if (hipMalloc((void**)&d_tmp, bytes) != hipSuccess || hipMalloc((void**)&d_last, bytes) != hipSuccess ||
        hipMalloc((void**)&d_tables, (size_t)2 * (w + h) * sizeof(int)) != hipSuccess) { err = "auto_align: hipMalloc failed"; release(); return -2; }
    W = w; H = h;
    return 0;
}

bool AutoAligner::warp_in_place(uint8_t* d_img, const double M[6], hipStream_t s) {      // cv::warpAffine clones an aliased source
    if (!warp_affine_device(d_img, d_tmp, W, H, M, d_tables, s) ||
        hipMemcpyAsync(d_img, d_tmp, (size_t)W * H * 3, hipMemcpyDeviceToDevice, s) != hipSuccess) { failed = true; err = "auto_align: warpAffine failed"; return false; }
    return true;
}

// morph_distance(p1, set_k) for n_cand candidate sets
bool AutoAligner::score_sets(const std::vector<P2f>& p1, const std::vector<P2f>& sets, int n_cand, hipStream_t s, std::vector<double>& out) {
    const int n = (int)p1.size();
    const size_t pts = (size_t)n * 8, all = pts * n_cand, res = (size_t)n_cand * 4;
    const size_t need = pts + all + 3 * res + 64;
    if (need > d_score_bytes) {
        if (d_score) (void)hipFree(d_score);
        d_score = nullptr; d_score_bytes = 0;
        if (hipMalloc((void**)&d_score, need) != hipSuccess) { failed = true; err = "auto_align: hipMalloc failed"; return false; }
        d_score_bytes = need;
    }
    float* d_p1 = (float*)d_score;
    float* d_sets = (float*)(d_score + pts);
    float* d_total = (float*)(d_score + pts + all);
    float* d_inner = d_total + n_cand;
    int* d_np = (int*)(d_inner + n_cand);
    std::vector<float> total(n_cand), inner(n_cand);
    std::vector<int> npairs(n_cand);
    bool ok = hipMemcpyAsync(d_p1, p1.data(), pts, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(d_sets, sets.data(), all, hipMemcpyHostToDevice, s) == hipSuccess;
    if (ok) {
        launch_candidate_scores(d_p1, d_sets, n, n_cand, d_total, d_np, d_inner, s);
        ok = hipMemcpyAsync(total.data(), d_total, res, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(inner.data(), d_inner, res, hipMemcpyDeviceToHost, s) == hipSuccess &&
             hipMemcpyAsync(npairs.data(), d_np, res, hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    // host, meanwhile: hull areas of the candidates (threads) and the candidate-independent parts
    const double area1 = hull_area_of(p1);
    const float inner1 = inner_offset_sum(p1, p1);
    std::vector<double> area2(n_cand);
    std::atomic<int> next{0};
    const int n_threads = std::max(1, std::min(std::min(32, n_cand), (int)std::thread::hardware_concurrency()));
    auto worker = [&]() {
        std::vector<P2f> t(n);
        for (int k; (k = next.fetch_add(1)) < n_cand;) {
            memcpy(t.data(), sets.data() + (size_t)k * n, pts);
            area2[k] = hull_area_of(t);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (std::thread& t : pool) t.join();
    if (!ok || hipStreamSynchronize(s) != hipSuccess) { failed = true; err = "auto_align: candidate scoring failed"; return false; }
    out.resize(n_cand);
    for (int k = 0; k < n_cand; ++k)
        out[k] = morph_distance_combine(total[k], (size_t)npairs[k], inner1, inner[k], p1.size(), p1.size(), area1, area2[k], W, H);
    return true;
}

double AutoAligner::retranslate(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s) {
    const int n = (int)p2.size();
    // One scoring call covers everything the search can ask for first: the point set as it is, its four unit shifts, and the first
    // kAhead steps of a walk in each of the eight directions (a step adds the offset to the previous step's float coordinates, as
    // the reference's loop does).  The reference's decisions are then replayed on those distances; a walk longer than kAhead
    // steps continues with further calls.
    constexpr int kAhead = 12, kDirs = 8;
    static const int dir_x[kDirs] = {-1, 1, 0, 0, -1, -1, 1, 1}, dir_y[kDirs] = {0, 0, -1, 1, -1, 1, -1, 1};
    std::vector<P2f> sets;
    sets.reserve((size_t)(5 + kDirs * kAhead) * n);
    auto append_shifted = [&](float dx, float dy) { for (const P2f& p : p2) sets.push_back(P2f{p.x + dx, p.y + dy}); };
    append_shifted(0, 0); append_shifted(-1, 0); append_shifted(1, 0); append_shifted(0, -1); append_shifted(0, 1);
    std::vector<std::vector<P2f>> walk_end(kDirs);
    for (int k = 0; k < kDirs; ++k) {
        std::vector<P2f> t = p2;
        for (int b = 0; b < kAhead; ++b) {
            for (P2f& p : t) { if (dir_x[k]) p.x += (long)dir_x[k]; if (dir_y[k]) p.y += (long)dir_y[k]; }
            sets.insert(sets.end(), t.begin(), t.end());
        }
        walk_end[k] = t;
    }
    std::vector<double> d;
    if (!score_sets(p1, sets, 5 + kDirs * kAhead, s, d)) return 0;
    double current = d[0];
    long xdir = 0, ydir = 0;
    if (d[1] < current) xdir = -1; else if (d[2] < current) xdir = +1;
    if (d[3] < current) ydir = -1; else if (d[4] < current) ydir = +1;
    long xsteps = 1, ysteps = 1;
    auto walk = [&](long dx, long dy, long* nx, long* ny) {          // keep stepping while the distance does not grow
        int k = 0;
        while (dir_x[k] != dx || dir_y[k] != dy) ++k;
        double last = current;
        for (int b = 0; b < kAhead; ++b) {
            const double v = d[5 + k * kAhead + b];
            if (v > last) return;
            current = last = v;
            if (nx) ++*nx;
            if (ny) ++*ny;
        }
        std::vector<P2f> t = walk_end[k], more;
        std::vector<double> dm;
        for (;;) {
            more.clear();
            for (int b = 0; b < kAhead; ++b) {
                for (P2f& p : t) { if (dx) p.x += dx; if (dy) p.y += dy; }
                more.insert(more.end(), t.begin(), t.end());
            }
            if (!score_sets(p1, more, kAhead, s, dm)) return;
            for (int b = 0; b < kAhead; ++b) {
                if (dm[b] > last) return;
                current = last = dm[b];
                if (nx) ++*nx;
                if (ny) ++*ny;
            }
        }
    };
    if (xdir != 0 && ydir != 0) walk(xdir, ydir, &xsteps, &ysteps);
    else {
        if (xdir != 0) walk(xdir, 0, &xsteps, nullptr);
        if (ydir != 0 && !failed) walk(0, ydir, nullptr, &ysteps);
    }
    if (failed) return 0;
    const float tx = (float)(xdir * xsteps), ty = (float)(ydir * ysteps);
    const double M[6] = {1, 0, (double)tx, 0, 1, (double)ty};
    if (!warp_in_place(d_img, M, s)) return 0;
    for (P2f& p : p2) { p.x += tx; p.y += ty; }
    if (!score_sets(p1, p2, 1, s, d)) return 0;
    return d[0];
}

static void rotate_about(std::vector<P2f>& pts, P2f c, double deg) {
    const double rad = deg * M_PI / 180.0;
    for (P2f& p : pts) {
        const float x = p.x - c.x, y = p.y - c.y;
        const float rx = (float)(std::cos(rad) * x - std::sin(rad) * y), ry = (float)(std::sin(rad) * x + std::cos(rad) * y);
        p = P2f{rx + c.x, ry + c.y};
    }
}

double AutoAligner::rerotate(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s) {
    P2f centre{0, 0};
    for (const P2f& p : p2) { centre.x += p.x; centre.y += p.y; }
    centre.x /= p2.size(); centre.y /= p2.size();
    constexpr int kAngles = 1080;
    const int n = (int)p2.size();
    std::vector<P2f> sets((size_t)kAngles * n);                 // the rotations use the host's libm, on threads
    {
        std::atomic<int> next{0};
        const int n_threads = std::max(1, std::min(32, (int)std::thread::hardware_concurrency()));
        auto worker = [&]() {
            std::vector<P2f> t;
            for (int i; (i = next.fetch_add(1)) < kAngles;) {
                t = p2;
                rotate_about(t, centre, i / 3.0);
                memcpy(sets.data() + (size_t)i * n, t.data(), (size_t)n * 8);
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (std::thread& t : pool) t.join();
    }
    std::vector<double> d;
    if (!score_sets(p1, sets, kAngles, s, d)) return 0;
    double lowest = std::numeric_limits<double>::max(), angle = 0;
    for (int i = 0; i < kAngles; ++i)
        if (d[i] < lowest) { lowest = d[i]; angle = i / 3.0; }
    double M[6];
    rotation_matrix_2d(centre.x, centre.y, -angle, 1.0, M);
    if (!warp_in_place(d_img, M, s)) return 0;
    rotate_about(p2, centre, angle);
    return lowest;
}

double AutoAligner::reprocrustes(uint8_t* d_img, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s) {
    ProcrustesFit fit;
    procrustes_fit(p1, p2, fit);
    double M[9];
    perspective_from_4(p2.data(), fit.yprime.data(), M);          // the reference fits the map to the first four pairs only
    perspective_points(p2, M);
    if (!warp_in_place(d_img, M, s)) return 0;                     // its first two rows, as an affine map
    std::vector<double> d;
    if (!score_sets(p1, p2, 1, s, d)) return 0;
    return d[0];
}

int AutoAligner::step(int which, uint8_t* d_img, int w, int h, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s, double* dist) {
    if (p1.size() != p2.size() || p1.size() < 4 || p1.size() > (size_t)kAlignMaxPoints) { err = "auto_align: needs 4 .. 4096 point pairs"; return -1; }
    if (int rc = ensure(w, h)) return rc;
    failed = false;
    const double d = which == 0 ? retranslate(d_img, p1, p2, s) : which == 1 ? reprocrustes(d_img, p1, p2, s) : rerotate(d_img, p1, p2, s);
    if (failed || hipStreamSynchronize(s) != hipSuccess) { if (err.empty()) err = "auto_align: device error"; return -2; }
    if (dist) *dist = d;
    return 0;
}

int AutoAligner::run(uint8_t* d_img, int w, int h, const std::vector<P2f>& p1, std::vector<P2f>& p2, hipStream_t s, double* final_distance) {
    if (p1.size() != p2.size() || p1.size() < 4 || p1.size() > (size_t)kAlignMaxPoints) { err = "auto_align: needs 4 .. 4096 point pairs"; return -1; }
    if (int rc = ensure(w, h)) return rc;
    failed = false;
    const size_t bytes = (size_t)w * h * 3;
    std::vector<P2f> kept;
    auto keep = [&]() { kept = p2; if (hipMemcpyAsync(d_last, d_img, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) failed = true; };
    auto undo = [&]() { p2 = kept; if (hipMemcpyAsync(d_img, d_last, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) failed = true; };
    const double initial = morph_distance_ref(p1, p2, w, h);
    double d_trans = initial, d_procr = initial, d_rot = initial, before;
    bool progress;
    do {
        progress = false;
        do { before = d_trans; keep(); d_trans = retranslate(d_img, p1, p2, s); if (d_trans < before) progress = true; } while (d_trans < before && !failed);
        undo(); d_procr = before;                                  // the last attempt did not improve: take it back
        do { before = d_procr; keep(); d_procr = reprocrustes(d_img, p1, p2, s); if (d_procr < before) progress = true; } while (d_procr < before && !failed);
        undo(); d_rot = before;
        do { before = d_rot; keep(); d_rot = rerotate(d_img, p1, p2, s); if (d_rot < before) progress = true; } while (d_rot < before && !failed);
        undo(); d_trans = before;
    } while (progress && !failed);
    if (failed || hipStreamSynchronize(s) != hipSuccess) { if (err.empty()) err = "auto_align: device error"; return -2; }
    if (final_distance) *final_distance = morph_distance_ref(p1, p2, w, h);
    return 0;
}

}  // namespace poppy_hip
