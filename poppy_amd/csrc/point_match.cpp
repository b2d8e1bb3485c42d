// point_match.cpp — Poppy's keypoint pairing on the host (see point_match.h for why it is host code).
//
//   src/util.cpp:351-383   make_distance_map: greedy first-come nearest neighbour, ascending by distance
//   src/util.cpp:385-431   morph_distance (hull areas, mean pair distance, signed coordinate sums; long double mix)
//   src/util.cpp:473-496   filter_invalid_points
//   src/matcher.cpp:246-332 match (statistical threshold) + prepare (add_corners, src/util.cpp:268-279)
//   OCV/imgproc/src/convhull.cpp:48-312, approx.cpp:476-671, shapedescr.cpp:308-338 (hull, Douglas-Peucker, area)
// The reference translation unit is `using namespace std`, so hypot(float,float) resolves to the float overload.
#include "point_match.h"
#include "worker.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cmath>
#include <limits>

namespace poppy_hip {

// hypotf for finite, non-zero floats IS this expression in glibc — sysdeps/ieee754/flt-32/e_hypotf.c returns
// (float) sqrt((double) x * x + (double) y * y) after peeling off infinities, NaNs and zeros (2.17-2.34), and the 2.35
// rewrite keeps the same double-precision square root with an added overflow check — so this is the library's own
// definition, not an approximation of it: the squares are exact in double, sum and root are each rounded once, one final
// rounding to float.  tools/micro/hypotf_check.c compares it with the installed libm on 4*10^8 inputs (0 differ on glibc
// 2.35; tests/test_host_plan.py repeats a 2*10^6-input sample on whatever libm the test box has).  The zero / infinity /
// NaN special cases give the same values through the formula (x = 0 -> sqrt(y*y) = |y| exactly; inf -> inf; NaN -> NaN) and
// image coordinates are finite.  The libm call costs several times this, and the matcher evaluates it ~10^6 times per pair;
// the candidate-scoring kernel (kernels_align.hip) uses the same expression with the device's correctly rounded f64 sqrt.
static inline float hyp(float x, float y) { return (float)std::sqrt((double)x * (double)x + (double)y * (double)y); }

long hypotf_selfcheck(long n, unsigned long long seed) {
    unsigned long long s = seed ? seed : 88172645463325252ull;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float x, y;
        switch (i & 3) {
        case 0: x = (float)((int)(s & 0xFFFFF) - 524288) / 256.f; y = (float)((int)((s >> 20) & 0xFFFFF) - 524288) / 256.f; break;   // image-scale coordinates
        case 1: x = (float)((int)(s & 0xFFFF) - 32768) / 8.f; y = (float)((int)((s >> 16) & 0xFFFF) - 32768) / 8.f; break;
        case 2: x = std::ldexp((float)(s & 0xFFFFFF), -(int)((s >> 24) & 31)); y = std::ldexp((float)((s >> 32) & 0xFFFFFF), -(int)((s >> 56) & 31)); break;
        default: x = (float)((int)(s & 0x3FFF) - 8192) / 4.f; y = (s >> 40) & 1 ? 0.f : (float)((int)((s >> 14) & 0x3FFF) - 8192) / 4.f; break;
        }
        if (::hypotf(x, y) != hyp(x, y)) ++bad;
    }
    return bad;
}

// make_distance_map (src/util.cpp:351-383): every point of the first set, in order, takes the nearest point of the second set that is still
// free — nearest by the FLOAT distance hypot(dx, dy), the first such point on ties — and the pairs are then ordered by that distance.
// The reference evaluates the distance to every free point (n^2 / 2 square roots, 0.29 ms of a 4.2 ms set-up at 530 points).  Here the
// squared distance s = dx^2 + dy^2 (the very double the distance is the rounded root of) decides: (float)sqrt(s) does not decrease with s, so
// the minimum distance is that of the smallest s, and the points that share it are those whose s lies within the rounding of a float — the
// first of them is found by evaluating the real expression on the few s below s_min (1 + 2^-20), no other root is taken.  Taken points are
// removed from the (order-preserving) candidate arrays instead of being skipped; (-1, -1), which the reference uses as its tombstone and
// therefore never pairs, is dropped up front.
// the squared distances of one point to the m candidates, and their minimum (the caller's inner loops; cloned for AVX2 hosts: the arithmetic is the same
// per element — double products and one sum, no contraction — only eight of them at a time)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
static double squared_distances(const float* __restrict__ x, const float* __restrict__ y, size_t m, float ax, float ay, double* __restrict__ sq) {
    for (size_t j = 0; j < m; ++j) {
        const float dx = x[j] - ax, dy = y[j] - ay;
        sq[j] = (double)dx * (double)dx + (double)dy * (double)dy;
    }
    // the minimum as the plain scan `s = sq[j] < s ? sq[j] : s` finds it: a NaN in front stays (the caller then takes the reference's loop as it stands),
    // NaNs elsewhere are passed over; four running minima only cut the scan's dependency chain
    if (!(sq[0] == sq[0])) return sq[0];
    double m0 = sq[0], m1 = m0, m2 = m0, m3 = m0;
    size_t j = 1;
    for (; j + 3 < m; j += 4) {
        m0 = sq[j] < m0 ? sq[j] : m0; m1 = sq[j + 1] < m1 ? sq[j + 1] : m1;
        m2 = sq[j + 2] < m2 ? sq[j + 2] : m2; m3 = sq[j + 3] < m3 ? sq[j + 3] : m3;
    }
    for (; j < m; ++j) m0 = sq[j] < m0 ? sq[j] : m0;
    m0 = m1 < m0 ? m1 : m0; m2 = m3 < m2 ? m3 : m2;
    return m2 < m0 ? m2 : m0;
}

void greedy_pairs(const std::vector<P2f>& src1, const std::vector<P2f>& src2, std::vector<PointPair>& pairs) {
    std::vector<float> X, Y;
    X.reserve(src2.size()); Y.reserve(src2.size());
    for (const P2f& p : src2) if (!(p.x == -1 && p.y == -1)) { X.push_back(p.x); Y.push_back(p.y); }
    std::vector<double> S(X.size());
    pairs.clear();
    pairs.reserve(src1.size());
    size_t m = X.size();
    for (const P2f& a : src1) {
        if (!m) break;                                          // (the reference goes on, finding nothing)
        const float ax = a.x, ay = a.y;
        float* const x = X.data(); float* const y = Y.data(); double* const sq = S.data();
        const double s_min = squared_distances(x, y, m, ax, ay, sq);
        size_t pick = 0;
        if (s_min == s_min) {                                   // (a NaN coordinate: fall through to the plain scan below)
            const float best = (float)std::sqrt(s_min);
            const double bound = s_min * (1.0 + 1.0 / 1048576.0);
            for (pick = 0; pick < m; ++pick)
                if (sq[pick] <= bound && (float)std::sqrt(sq[pick]) == best) break;
        } else pick = m;
        if (pick == m) {                                        // the reference's loop as it stands
            double best = std::numeric_limits<double>::max();
            for (size_t j = 0; j < m; ++j) { const double d = hyp(x[j] - ax, y[j] - ay); if (d < best) { best = d; pick = j; } }
            if (pick == m) continue;
        }
        const P2f b{x[pick], y[pick]};
        pairs.push_back(PointPair{(double)hyp(b.x - a.x, b.y - a.y), a, b});
        X.erase(X.begin() + pick); Y.erase(Y.begin() + pick);
        --m;
    }
    // multimap<double,...>: ascending keys, equal keys in insertion order
    std::stable_sort(pairs.begin(), pairs.end(), [](const PointPair& l, const PointPair& r) { return l.dist < r.dist; });
}

void drop_out_of_image(std::vector<P2f>& p1, std::vector<P2f>& p2, int cols, int rows) {
    auto bad = [&](const P2f& p) { return p.x < 0 || p.x > cols || p.y < 0 || p.y > rows; };
    for (size_t i = 0; i < p1.size();)
        if (bad(p1[i])) { p1.erase(p1.begin() + i); p2.erase(p2.begin() + i); } else ++i;
    for (size_t i = 0; i < p2.size();)
        if (bad(p2[i])) { p1.erase(p1.begin() + i); p2.erase(p2.begin() + i); } else ++i;
    if (p1.size() > p2.size()) p1.resize(p2.size()); else p2.resize(p1.size());
}

namespace {

inline int sign_of(double v) { return (v > 0) - (v < 0); }

// one monotone chain of Sklansky's scan over the x-sorted points
int chain(const std::vector<const P2f*>& a, int start, int end, int* stk, int nsign, int sign2) {
    const int step = end > start ? 1 : -1;
    int prev = start, cur = prev + step, next = cur + step, size = 3;
    if (start == end || (a[start]->x == a[end]->x && a[start]->y == a[end]->y)) { stk[0] = start; return 1; }
    stk[0] = prev; stk[1] = cur; stk[2] = next;
    end += step;
    while (next != end) {
        const float cy = a[cur]->y, ny = a[next]->y, by = ny - cy;
        if (sign_of(by) != nsign) {
            const float ax = a[cur]->x - a[prev]->x, bx = a[next]->x - a[cur]->x, ay = cy - a[prev]->y;
            const double turn = (double)ay * bx - (double)ax * by;
            if (sign_of(turn) == sign2 && (ax != 0 || ay != 0)) {
                prev = cur; cur = next; next += step; stk[size++] = next;
            } else if (prev == start) {
                cur = next; stk[1] = cur; next += step; stk[2] = next;
            } else {
                stk[size - 2] = next; cur = prev; prev = stk[size - 4]; --size;
            }
        } else { next += step; stk[size - 1] = next; }
    }
    return --size;
}

void hull_ccw(const std::vector<P2f>& pts, std::vector<P2f>& hull) {
    const int n = (int)pts.size();
    hull.clear();
    if (!n) return;
    std::vector<const P2f*> a(n);
    for (int i = 0; i < n; ++i) a[i] = &pts[i];
    std::sort(a.begin(), a.end(), [](const P2f* l, const P2f* r) {
        if (l->x != r->x) return l->x < r->x;
        if (l->y != r->y) return l->y < r->y;
        return l < r;
    });
    int lo = 0, hi = 0;
    for (int i = 1; i < n; ++i) { if (a[lo]->y > a[i]->y) lo = i; if (a[hi]->y < a[i]->y) hi = i; }
    std::vector<int> stk(n + 2), idx(n);
    int m = 0;
    const P2f* base = pts.data();
    if (a[0]->x == a[n - 1]->x && a[0]->y == a[n - 1]->y) idx[m++] = 0;
    else {
        int* s0 = stk.data(); int c0 = chain(a, 0, hi, s0, -1, 1);
        int* s1 = s0 + c0;    int c1 = chain(a, n - 1, hi, s1, -1, -1);
        std::swap(s0, s1); std::swap(c0, c1);
        for (int i = 0; i < c0 - 1; ++i) idx[m++] = (int)(a[s0[i]] - base);
        for (int i = c1 - 1; i > 0; --i) idx[m++] = (int)(a[s1[i]] - base);
        const int stop = c1 > 2 ? s1[1] : c0 > 2 ? s0[c0 - 2] : -1;
        int* b0 = stk.data(); int d0 = chain(a, 0, lo, b0, 1, -1);
        int* b1 = b0 + d0;    int d1 = chain(a, n - 1, lo, b1, 1, 1);
        if (stop >= 0) {
            const int chk = d0 > 2 ? b0[1] : d0 + d1 > 2 ? b1[2 - d0] : -1;
            if (chk == stop || (chk >= 0 && a[chk]->x == a[stop]->x && a[chk]->y == a[stop]->y)) { d0 = std::min(d0, 2); d1 = std::min(d1, 2); }
        }
        for (int i = 0; i < d0 - 1; ++i) idx[m++] = (int)(a[b0[i]] - base);
        for (int i = d1 - 1; i > 0; --i) idx[m++] = (int)(a[b1[i]] - base);
        if (m >= 3) {                    // rotate so the index sequence is monotone when it can be
            int imin = 0, imax = 0, lt = 0, i;
            for (i = 1; i < m; ++i) {
                lt += idx[i - 1] < idx[i];
                if (lt > 1 && lt <= i - 2) break;
                if (idx[i] < idx[imin]) imin = i;
                if (idx[i] > idx[imax]) imax = i;
            }
            const int gap = std::abs(imax - imin);
            if ((gap == 1 || gap == m - 1) && (lt <= 1 || lt >= m - 2)) {
                const int asc = (imax + 1) % m == imin;
                int i0 = asc ? imin : imax, j = i0;
                if (i0 > 0) {
                    int* tmp = stk.data();
                    for (i = 0; i < m; ++i) {
                        const int c = tmp[i] = idx[j], nj = j + 1 < m ? j + 1 : 0;
                        if (i < m - 1 && (asc != (c < idx[nj]))) break;
                        j = nj;
                    }
                    if (i == m) std::copy(tmp, tmp + m, idx.begin());
                }
            }
        }
    }
    for (int i = 0; i < m; ++i) hull.push_back(base[idx[i]]);
}

// approxPolyDP(hull, eps, closed = true)
void simplify_closed(const std::vector<P2f>& src, double eps, std::vector<P2f>& out) {
    int count = (int)src.size();
    out.clear();
    if (!count) return;
    struct Span { int s, e; };
    std::vector<Span> todo;
    std::vector<P2f> dst(count);
    int kept = 0, pos = 0;
    Span cur{0, 0}, right{0, 0};
    P2f a{-1000000.f, -1000000.f}, b{0, 0}, p{0, 0};
    bool flat = false;
    eps *= eps;
    auto next_src = [&](P2f& q, int& i) { q = src[i]; if (++i >= count) i = 0; };
    for (int it = 0; it < 3; ++it) {          // approximately the two farthest points
        double far = 0;
        pos = (pos + right.s) % count;
        next_src(a, pos);
        for (int j = 1; j < count; ++j) {
            next_src(p, pos);
            const double dx = p.x - a.x, dy = p.y - a.y, d = dx * dx + dy * dy;
            if (d > far) { far = d; right.s = j; }
        }
        flat = far <= eps;
    }
    if (!flat) {
        right.e = cur.s = pos % count;
        cur.e = right.s = (right.s + cur.s) % count;
        todo.push_back(right); todo.push_back(cur);
    } else dst[kept++] = a;
    while (!todo.empty()) {
        cur = todo.back(); todo.pop_back();
        b = src[cur.e];
        pos = cur.s;
        next_src(a, pos);
        if (pos != cur.e) {
            double far = 0;
            const double dx = b.x - a.x, dy = b.y - a.y;
            while (pos != cur.e) {
                next_src(p, pos);
                const double d = std::fabs((p.y - a.y) * dx - (p.x - a.x) * dy);
                if (d > far) { far = d; right.s = (pos + count - 1) % count; }
            }
            flat = far * far <= eps * (dx * dx + dy * dy);
        } else { flat = true; a = src[cur.s]; }
        if (flat) dst[kept++] = a;
        else { right.e = cur.e; cur.e = right.s; todo.push_back(right); todo.push_back(cur); }
    }
    count = kept;
    auto next_dst = [&](P2f& q, int& i) { q = dst[i]; if (++i >= count) i = 0; };
    pos = count - 1;
    next_dst(a, pos);
    int w = pos;
    next_dst(p, pos);
    for (int i = 0; i < count && kept > 2; ++i) {
        next_dst(b, pos);
        const double dx = b.x - a.x, dy = b.y - a.y;
        const double d = std::fabs((p.x - a.x) * dy - (p.y - a.y) * dx);
        const double along = (p.x - a.x) * (b.x - p.x) + (p.y - a.y) * (b.y - p.y);    // evaluated in float
        if (d * d <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && along >= 0) {
            --kept;
            dst[w] = a = b;
            if (++w >= count) w = 0;
            next_dst(p, pos);
            ++i;
            continue;
        }
        dst[w] = a = p;
        if (++w >= count) w = 0;
        p = b;
    }
    out.assign(dst.begin(), dst.begin() + kept);
}

double polygon_area(const std::vector<P2f>& c) {
    if (c.empty()) return 0.;
    double acc = 0;
    P2f prev = c.back();
    for (const P2f& q : c) { acc += (double)prev.x * q.y - (double)prev.y * q.x; prev = q; }
    return std::fabs(acc * 0.5);
}

double hull_area(const std::vector<P2f>& pts) {
    std::vector<P2f> h, c;
    hull_ccw(pts, h);
    simplify_closed(h, 0.001, c);
    return std::fabs(polygon_area(c));
}

}  // namespace

double hull_area_of(const std::vector<P2f>& pts) { return hull_area(pts); }

float inner_offset_sum(const std::vector<P2f>& a, const std::vector<P2f>& b) {
    float acc = 0;
    for (size_t i = 0; i < a.size(); ++i)
        for (size_t j = 0; j < a.size(); ++j) { const float vx = a[i].x - b[j].x, vy = a[i].y - b[j].y; acc += vx + vy; }
    return acc;
}

double morph_distance_combine(float total, size_t n_pairs, float inner1_sum, float inner2_sum, size_t n1, size_t n2,
                              double area1, double area2, int w, int h) {
    const long double width = w, height = h;
    const float in1 = (float)((inner1_sum / (n1 * n1)) / (width + height));
    const float in2 = (float)((inner2_sum / (n2 * n2)) / (width + height));
    const long double r = ((total / (n_pairs)) / hypotl(width, height)) + fabs(in1 - in2) + (fabs(area1 - area2) / (width * height)) / 3.0;
    return (double)r;
}

// Both O(n^2) offset sums of morph_distance at once: each is one chain of float additions in (i, j) order, bound by the
// adder's latency, so the two chains are interleaved.
static void inner_offset_sums(const std::vector<P2f>& p1, const std::vector<P2f>& p2, float* s11, float* s21) {   // sizes equal
    float a1 = 0, a2 = 0;
    const size_t n = p1.size();
    for (size_t i = 0; i < n; ++i) {
        const P2f u = p1[i], v = p2[i];
        for (size_t j = 0; j < n; ++j) {
            const float wx = p1[j].x, wy = p1[j].y;
            a1 += (u.x - wx) + (u.y - wy);
            a2 += (v.x - wx) + (v.y - wy);
        }
    }
    *s11 = a1; *s21 = a2;
}

// morph_distance, keeping the greedy pairing for the caller (match_and_prepare_from pairs the same two sets again)
double morph_distance_pairs(const std::vector<P2f>& p1, const std::vector<P2f>& p2, int w, int h, std::vector<PointPair>& pairs, Worker* helper) {
    float s11 = 0, s21 = 0;
    const bool same_n = p1.size() == p2.size();     // inner_offset_sum(p2, p1) indexes p1 with p2's count
    static const bool stage_times = getenv("POPPY_SETUP_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    double ms_sums = 0;
    std::thread sums;
    const bool beside = same_n && p1.size() >= 128;
    auto sums_job = [&]() { inner_offset_sums(p1, p2, &s11, &s21); ms_sums = ms(); };
    if (beside && helper) helper->run(sums_job);
    else if (beside) sums = std::thread(sums_job);
    greedy_pairs(p1, p2, pairs);
    const double ms_greedy = ms();
    float total = 0;
    for (const PointPair& e : pairs) total += hyp(e.b.x - e.a.x, e.b.y - e.a.y);
    const double a1 = hull_area(p1), a2 = hull_area(p2);
    const double ms_hull = ms();
    if (beside && helper) helper->wait();
    else if (sums.joinable()) sums.join();
    if (stage_times) fprintf(stderr, "  matcher, %zu points: greedy pairing %.3f, hulls %.3f ms (cumulative); the offset sums beside them %.3f ms\n", p1.size(), ms_greedy, ms_hull, ms_sums);
    if (beside) {}
    else if (same_n) inner_offset_sums(p1, p2, &s11, &s21);
    else { s11 = inner_offset_sum(p1, p1); s21 = inner_offset_sum(p2, p1); }
    // the second inner sum runs the second set against the FIRST, as in the reference
    return morph_distance_combine(total, pairs.size(), s11, s21, p1.size(), p2.size(), a1, a2, w, h);
}

double morph_distance_ref(const std::vector<P2f>& p1, const std::vector<P2f>& p2, int w, int h) {
    std::vector<PointPair> pairs;
    return morph_distance_pairs(p1, p2, w, h, pairs);
}

void match_and_prepare(std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h, double tolerance, double initial_morph_dist) {
    std::vector<PointPair> pairs;
    greedy_pairs(s1, s2, pairs);
    match_and_prepare_from(pairs, s1, s2, w, h, tolerance, initial_morph_dist);
}

void match_and_prepare_from(const std::vector<PointPair>& pairs, std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h,
                            double tolerance, double initial_morph_dist) {
    const size_t n = pairs.size();
    double sum = 0.0;
    for (const PointPair& e : pairs) sum += e.dist;
    const double mean = sum / n;
    double var = 0.0;
    for (const PointPair& e : pairs) var += pow(e.dist - mean, 2);
    const double deviation = sqrt(var / n), total = sum;
    const double density = total / (w * h), area = (w * h);
    s1.clear(); s2.clear();
    if (mean == 0) {
        for (const PointPair& e : pairs) { s1.push_back(e.a); s2.push_back(e.b); }
    } else {
        double thresh = 1;
        if (tolerance != 0)
            thresh = (area * (mean / deviation) * tolerance) / ((total * sqrt(density) * (1.0 / sqrt(initial_morph_dist))) / ((1 + sqrt(5)) / 2.0));
        for (const PointPair& e : pairs) {
            const double r = e.dist / thresh;
            if (r > 0.0 && r <= 1.0) { s1.push_back(e.a); s2.push_back(e.b); }
        }
        if (s1.empty() && n) { s1.push_back(pairs[0].a); s2.push_back(pairs[0].b); }
    }
    const float fw = (float)(w - 1), fh = (float)(h - 1);
    const P2f corners[4] = {{0, 0}, {fw, 0}, {0, fh}, {fw, fh}};
    for (const P2f& c : corners) { s1.push_back(c); s2.push_back(c); }
}

// Descriptor-matching sketch of src/experiments.hpp:14-144 on the 2-NN lists of both directions (rows idx0, d0, idx1, d1):
// ratioTest drops a query without two neighbours or with d0 / d1 > ratio (float division of the float-converted
// distances; 0/0 is NaN and "not >", so exact duplicates stay); symmetryTest keeps, for every surviving 1->2 match in
// query order, the first surviving 2->1 match that points back.  out rows: queryIdx, trainIdx, distance.
void ratio_symmetry(const int* knn12, int n1, const int* knn21, int n2, float ratio, std::vector<int>& out3) {
    auto keep = [ratio](const int* k) {
        if (k[0] < 0 || k[2] < 0) return false;
        return !((float)k[1] / (float)k[3] > ratio);
    };
    std::vector<char> k2(n2);
    for (int j = 0; j < n2; ++j) k2[j] = keep(knn21 + 4 * j);
    out3.clear();
    for (int i = 0; i < n1; ++i) {
        if (!keep(knn12 + 4 * i)) continue;
        for (int j = 0; j < n2; ++j) {
            if (!k2[j]) continue;
            if (i == knn21[4 * j] && j == knn12[4 * i]) {
                out3.push_back(i); out3.push_back(knn12[4 * i]); out3.push_back(knn12[4 * i + 1]);
                break;
            }
        }
    }
}

void add_image_corners(std::vector<P2f>& s1, std::vector<P2f>& s2, int w, int h) {      // src/util.cpp:268-279
    const float fw = (float)(w - 1), fh = (float)(h - 1);
    const P2f corners[4] = {{0, 0}, {fw, 0}, {0, fh}, {fw, fh}};
    for (const P2f& c : corners) { s1.push_back(c); s2.push_back(c); }
}

}  // namespace poppy_hip
