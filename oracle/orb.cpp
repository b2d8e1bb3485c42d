// oracle/orb.cpp — ORB keypoint detection (and the rBRIEF descriptor + Hamming matcher the north-star
// asks for although Poppy never calls them).  TEST INFRASTRUCTURE (see oracle.h).  Restates:
//   OCV/features2d/src/orb.cpp:130-177 (HarrisResponses), :181-215 (ICAngles), :219-285 (descriptor, WTA_K=2),
//   :653-656 (getScale), :784-959 (computeKeyPoints), :970-1127 (pyramid atlas: sizes, resize chain, border 32)
//   OCV/features2d/src/fast.cpp:57-292 + fast_score.cpp:50-211 (FAST-9/16, score, 3x3 NMS, raster emission)
//   OCV/features2d/src/keypoint.cpp:69-118 (retainBest = std::nth_element + std::partition, runByImageBorder)
//   OCV/imgproc/src/resize.cpp:345-398,619-760,853 (INTER_LINEAR_EXACT: 8.8 fixed-point coefficients)
//   OCV/core/src/mathfuncs_core.simd.hpp:34-70 (fastAtan2 polynomial)
//   OCV/core/src/batch_distance.cpp:103-110,199-262 (Hamming 1-NN, lowest train index wins ties)
#include "oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace oracle {

namespace {

struct Level {
    int w, h;            // level size without border
    float scale;
    ImageU8 img;         // (w + 64) x (h + 64), reflect-101 border of 32
    const uint8_t* at(int x, int y) const { return &img.d[(size_t)(y + 32) * img.w + (x + 32)]; }
};

const int kBorder = 32;

// resize(src, dst, INTER_LINEAR_EXACT), 8UC1
void resize_linear_exact(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep) {
    auto coeffs = [](int dsize, int ssize, std::vector<int>& ofs, std::vector<uint16_t>& c, int& dmin, int& dmax) {
        double inv = (double)dsize / ssize;
        double scale = 1.0 / inv;
        ofs.assign(dsize, 0); c.assign((size_t)dsize * 2, 0);
        dmin = 0; dmax = dsize;
        for (int v = 0; v < dsize; ++v) {
            double f = scale * ((double)v + 0.5) - 0.5;
            int iv = (int)std::floor(f);
            if (iv >= 0 && ssize > 1) {
                if (iv < ssize - 1) {
                    ofs[v] = iv;
                    double frac = f - (double)iv;
                    uint16_t c1 = frac < 0 ? 0 : (uint16_t)cv_round(frac * 256.0);
                    c[2 * v + 1] = c1;
                    c[2 * v] = 256 > c1 ? (uint16_t)(256 - c1) : 0;
                } else { ofs[v] = ssize - 1; dmax = std::min(dmax, v); }
            } else dmin = std::max(dmin, v + 1);
        }
    };
    std::vector<int> xo, yo; std::vector<uint16_t> xc, yc;
    int xmin, xmax, ymin, ymax;
    coeffs(dw, sw, xo, xc, xmin, xmax);
    coeffs(dh, sh, yo, yc, ymin, ymax);
    auto hline = [&](int sy, std::vector<uint16_t>& out) {
        const uint8_t* s = src + (size_t)sy * sstep;
        out.resize(dw);
        int i = 0;
        for (; i < xmin; ++i) out[i] = (uint16_t)(s[0] << 8);
        for (; i < xmax; ++i) out[i] = (uint16_t)(xc[2 * i] * s[xo[i]] + xc[2 * i + 1] * s[xo[i] + 1]);
        for (; i < dw; ++i) out[i] = (uint16_t)(s[xo[dw - 1]] << 8);
    };
    std::vector<uint16_t> r0, r1;
    for (int dy = 0; dy < dh; ++dy) {
        uint8_t* d = dst + (size_t)dy * dstep;
        if (dy < ymin) {
            hline(0, r0);
            for (int i = 0; i < dw; ++i) d[i] = (uint8_t)((r0[i] + 128) >> 8);
        } else if (dy >= ymax) {
            hline(sh - 1, r0);
            for (int i = 0; i < dw; ++i) d[i] = (uint8_t)((r0[i] + 128) >> 8);
        } else {
            hline(yo[dy], r0); hline(yo[dy] + 1, r1);
            uint32_t m0 = yc[2 * dy], m1 = yc[2 * dy + 1];
            for (int i = 0; i < dw; ++i) d[i] = (uint8_t)((r0[i] * m0 + r1[i] * m1 + (1u << 15)) >> 16);
        }
    }
}

void fill_border(Level& L) {
    int W = L.img.w;
    for (int y = -kBorder; y < L.h + kBorder; ++y)
        for (int x = -kBorder; x < L.w + kBorder; ++x) {
            if (x >= 0 && x < L.w && y >= 0 && y < L.h) continue;
            int sx = border_reflect101(x, L.w), sy = border_reflect101(y, L.h);
            L.img.d[(size_t)(y + kBorder) * W + (x + kBorder)] = *L.at(sx, sy);
        }
}

void build_pyramid(const ImageU8& image, int nlevels, std::vector<Level>& lv) {
    const double scaleFactor = (double)1.2f;            // ORB::create(float scaleFactor = 1.2f) stored as double
    lv.resize(nlevels);
    for (int l = 0; l < nlevels; ++l) {
        Level& L = lv[l];
        L.scale = (float)std::pow(scaleFactor, (double)l);
        float inv = 1.0f / L.scale;
        L.w = cv_round_f(image.w * inv); L.h = cv_round_f(image.h * inv);
        L.img = ImageU8(L.w + 2 * kBorder, L.h + 2 * kBorder, 1);
        if (l == 0) {
            for (int y = 0; y < L.h; ++y) memcpy((void*)L.at(0, y), image.row(y), L.w);
        } else if (l == 1) {                             // prevImg is still the input image for level 1
            resize_linear_exact(image.d.data(), image.w, image.h, image.w, (uint8_t*)L.at(0, 0), L.w, L.h, L.img.w);
        } else {
            const Level& P = lv[l - 1];
            resize_linear_exact(P.at(0, 0), P.w, P.h, P.img.w, (uint8_t*)L.at(0, 0), L.w, L.h, L.img.w);
        }
        fill_border(L);
    }
}

const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// FAST-9/16 corner score: max threshold at which the pixel is still a corner; 0 when it is none at t
int fast_score(const Level& L, int x, int y, int t) {
    int v = *L.at(x, y);
    int d[25];
    for (int k = 0; k < 25; ++k) d[k] = v - *L.at(x + kRing[k & 15][0], y + kRing[k & 15][1]);
    int best_min = -1000, best_max = 1000;
    for (int k = 0; k < 16; ++k) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; ++j) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        best_min = std::max(best_min, mn);      // darker arc: all (v - ring) > t
        best_max = std::min(best_max, mx);      // brighter arc: all (v - ring) < -t
    }
    int s = std::max(best_min, -best_max) - 1;
    bool corner = best_min > t || -best_max > t;
    return corner ? s : 0;
}

struct Cand { float x, y, response; int octave; float size, angle; };

void fast_nms(const Level& L, int threshold, std::vector<Cand>& out) {
    out.clear();
    int W = L.w, H = L.h;
    if (W < 7 || H < 7) return;
    ImageU8 score(W, H, 1);
    std::fill(score.d.begin(), score.d.end(), 0);
    for (int y = 3; y < H - 3; ++y)
        for (int x = 3; x < W - 3; ++x) score.d[(size_t)y * W + x] = (uint8_t)fast_score(L, x, y, threshold);
    for (int y = 3; y < H - 3; ++y)
        for (int x = 3; x < W - 3; ++x) {
            int s = score.d[(size_t)y * W + x];
            // a corner always has score >= threshold; with threshold 0 a zero score can still be a corner, but
            // NMS needs score > neighbours >= 0, so zero never survives
            if (s == 0) continue;
            const uint8_t* p = &score.d[(size_t)y * W + x];
            if (s > p[-1] && s > p[1] && s > p[-W - 1] && s > p[-W] && s > p[-W + 1] && s > p[W - 1] && s > p[W] && s > p[W + 1])
                out.push_back({(float)x, (float)y, (float)s, 0, 7.f, -1.f});
        }
}

struct RespGreater { bool operator()(const Cand& a, const Cand& b) const { return a.response > b.response; } };

void retain_best(std::vector<Cand>& k, int n) {
    if (n >= 0 && k.size() > (size_t)n) {
        if (n == 0) { k.clear(); return; }
        std::nth_element(k.begin(), k.begin() + n - 1, k.end(), RespGreater());
        float amb = k[n - 1].response;
        auto new_end = std::partition(k.begin() + n, k.end(), [amb](const Cand& c) { return c.response >= amb; });
        k.resize(new_end - k.begin());
    }
}

float harris(const Level& L, int x0, int y0) {
    const int bs = 7, r = bs / 2;
    const float harris_k = 0.04f;
    float scale = 1.f / ((1 << 2) * bs * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
            int x = x0 - r + j, y = y0 - r + i;
            auto P = [&](int dx, int dy) { return (int)*L.at(x + dx, y + dy); };
            int Ix = (P(1, 0) - P(-1, 0)) * 2 + (P(1, -1) - P(-1, -1)) + (P(1, 1) - P(-1, 1));
            int Iy = (P(0, 1) - P(0, -1)) * 2 + (P(-1, 1) - P(-1, -1)) + (P(1, 1) - P(1, -1));
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI), p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI), p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON); c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON); c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

void make_umax(int* umax /*[17]*/) {
    const int hp = 15;
    int vmax = (int)std::floor(hp * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(hp * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)hp * hp - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

float ic_angle(const Level& L, int x0, int y0, const int* umax) {
    const int hk = 15;
    int m01 = 0, m10 = 0;
    for (int u = -hk; u <= hk; ++u) m10 += u * *L.at(x0 + u, y0);
    for (int v = 1; v <= hk; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int p = *L.at(x0 + u, y0 + v), m = *L.at(x0 + u, y0 - v);
            vsum += p - m;
            m10 += u * (p + m);
        }
        m01 += v * vsum;
    }
    return fast_atan2((float)m01, (float)m10);
}

}  // namespace

// keypoints out: n x 7 floats (x, y, size, angle, response, octave, class_id) like the golden dumps
int orb_detect(const ImageU8& image, int nfeatures, std::vector<float>& kps, std::vector<float>* fastLevel0) {
    const int nlevels = 8, edge = 31, patch = 31, fastThreshold = 20;
    std::vector<Level> lv;
    build_pyramid(image, nlevels, lv);

    std::vector<int> perLevel(nlevels);
    float factor = (float)(1.0 / (double)1.2f);
    float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) { perLevel[l] = cv_round_f(desired); sum += perLevel[l]; desired *= factor; }
    perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);

    int umax[17]; make_umax(umax);
    std::vector<Cand> all, k;
    std::vector<int> counters(nlevels);
    for (int l = 0; l < nlevels; ++l) {
        fast_nms(lv[l], fastThreshold, k);
        if (l == 0 && fastLevel0) { fastLevel0->clear(); for (auto& c : k) { fastLevel0->push_back(c.x); fastLevel0->push_back(c.y); fastLevel0->push_back(c.response); } }
        // runByImageBorder
        if (lv[l].h <= edge * 2 || lv[l].w <= edge * 2) k.clear();
        else k.erase(std::remove_if(k.begin(), k.end(), [&](const Cand& c) {
                         int x = (int)c.x, y = (int)c.y;       // Rect::contains(Point2f -> int truncation)
                         return !(x >= edge && x < lv[l].w - edge && y >= edge && y < lv[l].h - edge); }), k.end());
        retain_best(k, 2 * perLevel[l]);
        counters[l] = (int)k.size();
        for (auto& c : k) { c.octave = l; c.size = patch * lv[l].scale; }
        all.insert(all.end(), k.begin(), k.end());
    }
    if (all.empty()) { kps.clear(); return 0; }
    for (auto& c : all) c.response = harris(lv[c.octave], cv_round_f(c.x), cv_round_f(c.y));
    std::vector<Cand> fin;
    size_t off = 0;
    for (int l = 0; l < nlevels; ++l) {
        k.assign(all.begin() + off, all.begin() + off + counters[l]);
        off += counters[l];
        retain_best(k, perLevel[l]);
        fin.insert(fin.end(), k.begin(), k.end());
    }
    kps.clear();
    for (auto& c : fin) {
        c.angle = ic_angle(lv[c.octave], cv_round_f(c.x), cv_round_f(c.y), umax);
        float s = lv[c.octave].scale;
        float px = c.x * s, py = c.y * s;
        kps.insert(kps.end(), {px, py, c.size, c.angle, c.response, (float)c.octave, -1.f});
    }
    return (int)fin.size();
}

// ORB::compute(image, keypoints, descriptors) with the default WTA_K = 2, patch 31: 32 bytes per keypoint.
// kps7 as produced by orb_detect (octave in column 5).  Every level is first blurred IN PLACE inside the
// padded atlas by GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) (orb.cpp:1188): the level is a submatrix and
// the border type is not ISOLATED, so the fixed-point 8-bit path is skipped (smooth.dispatch.cpp:646) and the
// generic separable filter runs in float (row: uchar->float RowFilter; column: symmetric, cvRound to uchar).
static const int8_t kPattern[1024] = {
#include "orb_pattern.inc"
};
static const float kGauss7Sigma2[7] = {0x1.1f5f62p-4f, 0x1.0c70fcp-3f, 0x1.869472p-3f, 0x1.ba95c0p-3f, 0x1.869472p-3f, 0x1.0c70fcp-3f, 0x1.1f5f62p-4f};

static void blur_level(const Level& L, ImageU8& out) {
    const int W = L.w, H = L.h, P = L.img.w;
    std::vector<float> tmp((size_t)(H + 6) * W);
    for (int y = -3; y < H + 3; ++y)
        for (int x = 0; x < W; ++x) {
            const uint8_t* s = L.at(x, y);
            float acc = kGauss7Sigma2[0] * (float)s[-3];
            for (int k = 1; k < 7; ++k) acc += kGauss7Sigma2[k] * (float)s[k - 3];
            tmp[(size_t)(y + 3) * W + x] = acc;
        }
    out = L.img;                                     // border keeps the unblurred pixels
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float* c = &tmp[(size_t)(y + 3) * W + x];
            float acc = kGauss7Sigma2[3] * c[0] + 0.f;
            for (int k = 1; k <= 3; ++k) acc += kGauss7Sigma2[3 + k] * (c[(size_t)k * W] + c[-(ptrdiff_t)k * W]);
            int v = cv_round_f(acc);
            out.d[(size_t)(y + kBorder) * P + (x + kBorder)] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

int orb_describe(const ImageU8& image, const std::vector<float>& kps7, std::vector<uint8_t>& desc, int trig_mode) {
    int n = (int)kps7.size() / 7, nlevels = 0;
    for (int i = 0; i < n; ++i) nlevels = std::max(nlevels, (int)kps7[i * 7 + 5] + 1);
    desc.assign((size_t)n * 32, 0);
    if (!n) return 0;
    std::vector<Level> lv;
    build_pyramid(image, nlevels, lv);
    std::vector<ImageU8> blurred(nlevels);
    for (int l = 0; l < nlevels; ++l) blur_level(lv[l], blurred[l]);
    for (int j = 0; j < n; ++j) {
        const float* k = &kps7[(size_t)j * 7];
        const int oct = (int)k[5];
        const Level& L = lv[oct];
        const int P = L.img.w;
        float scale = 1.f / L.scale;
        float angle = k[3];
        angle *= (float)(M_PI / 180.f);
        float a, b;
        if (trig_mode == 0) { a = cosf(angle); b = sinf(angle); } else { a = (float)cos((double)angle); b = (float)sin((double)angle); }
        int cx = cv_round_f(k[0] * scale), cy = cv_round_f(k[1] * scale);
        const uint8_t* center = &blurred[oct].d[(size_t)(cy + kBorder) * P + (cx + kBorder)];
        const int8_t* pat = kPattern;
        for (int i = 0; i < 32; ++i, pat += 32) {
            int val = 0;
            for (int bit = 0; bit < 8; ++bit) {
                int t[2];
                for (int q = 0; q < 2; ++q) {
                    float px = pat[bit * 4 + q * 2], py = pat[bit * 4 + q * 2 + 1];
                    float x = px * a - py * b, y = px * b + py * a;
                    t[q] = center[cv_round_f(y) * P + cv_round_f(x)];
                }
                val |= (t[0] < t[1]) << bit;
            }
            desc[(size_t)j * 32 + i] = (uint8_t)val;
        }
    }
    return n;
}

// BFMatcher(NORM_HAMMING).match: for every query the train descriptor with the smallest Hamming distance,
// lowest train index on ties.  out: nq x 3 ints (queryIdx, trainIdx, distance)
void hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, std::vector<int>& out) {
    out.clear();
    for (int i = 0; i < nq; ++i) {
        int best = -1, bd = INT32_MAX;
        for (int j = 0; j < nt; ++j) {
            int d = 0;
            for (int b = 0; b < bytes; ++b) d += __builtin_popcount((unsigned)(q[(size_t)i * bytes + b] ^ t[(size_t)j * bytes + b]));
            if (d < bd) { bd = d; best = j; }
        }
        if (best >= 0) { out.push_back(i); out.push_back(best); out.push_back(bd); }
    }
}

// BFMatcher(NORM_HAMMING).knnMatch(k = 2): sorted insertion with strict comparisons, so among equal distances the
// lower train index comes first (OCV/core/src/batch_distance.cpp:225-248; K = min(2, nt), :286).
// out: nq x 4 ints (trainIdx0, distance0, trainIdx1, distance1), -1 where there is no neighbour
void hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, std::vector<int>& out) {
    out.assign((size_t)nq * 4, -1);
    const int K = nt < 2 ? nt : 2;
    for (int i = 0; i < nq; ++i) {
        int dist[2] = {INT32_MAX, INT32_MAX}, idx[2] = {-1, -1};
        for (int j = 0; j < nt && K > 0; ++j) {
            int d = 0;
            for (int b = 0; b < bytes; ++b) d += __builtin_popcount((unsigned)(q[(size_t)i * bytes + b] ^ t[(size_t)j * bytes + b]));
            if (d < dist[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && dist[k] > d; --k) { idx[k + 1] = idx[k]; dist[k + 1] = dist[k]; }
                idx[k + 1] = j; dist[k + 1] = d;
            }
        }
        for (int k = 0; k < K; ++k)
            if (idx[k] >= 0) { out[(size_t)i * 4 + 2 * k] = idx[k]; out[(size_t)i * 4 + 2 * k + 1] = dist[k]; }
    }
}

// ratioTest (src/experiments.hpp:14-37): a query keeps its match when it has two neighbours and
// distance0 / distance1 (float division of the float-converted distances) is not > ratio.  0/0 = NaN is "not >": kept.
void ratio_test(const int* knn4, int n, float ratio, std::vector<int>& keep) {
    keep.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        if (knn4[4 * i] < 0 || knn4[4 * i + 2] < 0) continue;
        const float d0 = (float)knn4[4 * i + 1], d1 = (float)knn4[4 * i + 3];
        keep[i] = !(d0 / d1 > ratio);
    }
}

// symmetryTest (src/experiments.hpp:114-144): for every surviving 1->2 match, in query order, the first surviving
// 2->1 match that points back.  out: n x 3 ints (queryIdx, trainIdx, distance)
void symmetry_test(const int* knn12, const int* keep12, int n1, const int* knn21, const int* keep21, int n2, std::vector<int>& out) {
    out.clear();
    for (int i = 0; i < n1; ++i) {
        if (!keep12[i]) continue;
        for (int j = 0; j < n2; ++j) {
            if (!keep21[j]) continue;
            if (i == knn21[4 * j] && j == knn12[4 * i]) {
                out.push_back(i); out.push_back(knn12[4 * i]); out.push_back(knn12[4 * i + 1]);
                break;
            }
        }
    }
}

}  // namespace oracle
