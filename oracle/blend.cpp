// oracle/blend.cpp — float conversion, blend mask, Gaussian pyramids, Laplacian blend, unsharp mask.
// TEST INFRASTRUCTURE.  Restates:
//   src/algo.cpp:245-265 (l/r conversion, mask, blend, sharpen, u8), src/blend.hpp:11-91, src/util.cpp:113-148
//   OCV/core/src/convert_scale.simd.hpp:92-118 (cvtScale: v*a + b in float)
//   OCV/imgproc/src/color_rgb.simd.hpp:594-643 (RGB2Gray<float>, BGR order: b*0.114f + g*0.587f + r*0.299f)
//   OCV/core/src/matrix_expressions.cpp:369-400,1293-1350 + arithm.simd.hpp:1703-1719 (mask MatExpr lowering)
//   OCV/imgproc/src/pyramids.cpp:344-402,503-521,700-719 (SIMD bodies), :745-900 (pyrDown_), :903-1005 (pyrUp_)
//   OCV/imgproc/src/filter.simd.hpp:1625-1730,2446-2488 (row filter), :2716-2760 (symmetric column filter)
//   OCV/imgproc/src/median_blur.simd.hpp:676-735 (3x3 sorting network, replicated borders)
//   OCV/core/include/opencv2/core/matx.hpp:929-932 (norm(Vec3f): sqrt of a DOUBLE sum of squares)
// The SIMD-body / scalar-tail split of the SSE (4-lane) reference build changes the association of the
// float sums, so the split points are reproduced exactly.
#include "oracle.h"
#include <algorithm>
#include <cmath>

namespace oracle {

// getGaussianKernel(9, 1.0, CV_32F) of the reference build (values captured from the compiled
// reference by oracle/golden_gen; see tests/golden p_prims: gauss_k9_s1)
const float kGauss9Sigma1[9] = {
    0x1.18a9c4p-13f, 0x1.22724cp-8f, 0x1.ba4b9ap-5f, 0x1.ef8ebap-3f, 0x1.9884a4p-2f,
    0x1.ef8ebap-3f, 0x1.ba4b9ap-5f, 0x1.22724cp-8f, 0x1.18a9c4p-13f};

void u8_to_f32(const ImageU8& src, ImageF& dst) {
    dst = ImageF(src.w, src.h, src.c);
    const float a = (float)(1.0 / 255.0), b = 0.f;
    for (size_t i = 0; i < src.d.size(); ++i) dst.d[i] = (float)src.d[i] * a + b;
}

void f32_to_u8(const ImageF& src, ImageU8& dst) {
    dst = ImageU8(src.w, src.h, src.c);
    for (size_t i = 0; i < src.d.size(); ++i) {
        int v = cv_round_f(src.d[i] * 255.f + 0.f);
        dst.d[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

void blend_mask(const ImageF& gabor2, double maskRatio, ImageF& mask) {
    int W = gabor2.w, H = gabor2.h;
    mask = ImageF(W, H, 1);
    const float cb = 0.114f, cg = 0.587f, cr = 0.299f;
    // addWeighted on CV_32F runs in DOUBLE with double scalars and rounds once to float
    // (arithm.simd.hpp:1808 add_weighted_loop_d, scalar_loader_n<float,double> :1160-1216)
    const double alpha = 1.0 - maskRatio, beta = -maskRatio;
    for (size_t i = 0; i < (size_t)W * H; ++i) {
        const float* s = &gabor2.d[i * 3];
        float gray = s[0] * cb + s[1] * cg + s[2] * cr;
        float m2 = 1.f - gray;
        float v = (float)(1.0 * alpha + ((double)m2 * beta + 0.0));   // ones*(1-mr) - m2*mr: one addWeighted pass
        if (v < 0) v = 0.f;
        if (v > 1) v = 1.f;
        mask.d[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
void pyr_down(const ImageF& src, ImageF& dst) {
    const int cn = src.c, sw = src.w, sh = src.h;
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dst = ImageF(dw, dh, cn);
    int width0 = std::min((sw - 2 - 1) / 2 + 1, dw);            // pixels handled without border lookup
    std::vector<int> tabL(cn * 7), tabR(cn * 7);
    for (int x = 0; x <= 6; ++x) {
        int l = border_reflect101(x - 2, sw) * cn, r = border_reflect101(x + width0 * 2 - 2, sw) * cn;
        for (int k = 0; k < cn; ++k) { tabL[x * cn + k] = l + k; tabR[x * cn + k] = r + k; }
    }
    const int dwe = dw * cn, w0e = width0 * cn;
    // horizontal pass of every needed source row (cached by source row index)
    std::vector<std::vector<float>> hrow(sh);
    auto horiz = [&](int sy) -> const std::vector<float>& {
        std::vector<float>& row = hrow[sy];
        if (!row.empty()) return row;
        row.resize(dwe);
        const float* s = src.row(sy);
        int x = 0;
        for (; x < cn; ++x)
            row[x] = s[tabL[x + cn * 2]] * 6 + (s[tabL[x + cn]] + s[tabL[x + cn * 3]]) * 4 + s[tabL[x]] + s[tabL[x + cn * 4]];
        if (x != dwe) {
            // SIMD body: r2*6 + ((r1+r3)*4 + (r0+r4)); 4-lane vectors; cn==1 steps 4, cn==3 steps 3 (one pixel)
            int width = w0e - x;
            if (cn == 1) {
                int xv = 0;
                for (; xv <= width - 4; xv += 4)
                    for (int k = 0; k < 4; ++k) {
                        const float* p = s + (x + xv + k) * 2;
                        row[x + xv + k] = p[0] * 6.f + ((p[-1] + p[1]) * 4.f + (p[-2] + p[2]));
                    }
                x += xv;
            } else if (cn == 3) {
                int xv = 0;
                for (; xv <= width - 4; xv += 3)
                    for (int k = 0; k < 3; ++k) {
                        const float* p = s + (x + xv) * 2 + k;
                        row[x + xv + k] = p[0] * 6.f + ((p[-3] + p[3]) * 4.f + (p[-6] + p[6]));
                    }
                x += xv;
            }
            for (; x < w0e; ++x) {           // scalar tail: ((s0*6 + (s-1+s1)*4) + s-2) + s2
                const float* p = s + (x / cn) * 2 * cn + x % cn;
                row[x] = p[0] * 6 + (p[-cn] + p[cn]) * 4 + p[-2 * cn] + p[2 * cn];
            }
            for (int x_ = 0; x < dwe; ++x, ++x_)
                row[x] = s[tabR[x_ + cn * 2]] * 6 + (s[tabR[x_ + cn]] + s[tabR[x_ + cn * 3]]) * 4 + s[tabR[x_]] + s[tabR[x_ + cn * 4]];
        }
        return row;
    };
    const float scale = 1.f / 256;
    for (int y = 0; y < dh; ++y) {
        const float* r[5];
        for (int k = 0; k < 5; ++k) r[k] = horiz(border_reflect101(y * 2 - 2 + k, sh)).data();
        float* d = dst.row(y);
        int x = 0;
        for (; x <= dwe - 4; x += 4)
            for (int k = 0; k < 4; ++k) {
                int i = x + k;
                d[i] = ((r[1][i] + r[3][i] + r[2][i]) * 4.f + (r[0][i] + r[4][i] + (r[2][i] + r[2][i]))) * scale;
            }
        for (; x < dwe; ++x)
            d[x] = (r[2][x] * 6 + (r[1][x] + r[3][x]) * 4 + r[0][x] + r[4][x]) * scale;
    }
}

void pyr_up(const ImageF& src, ImageF& dst, int dw, int dh) {
    const int cn = src.c, sw = src.w, sh = src.h;
    dst = ImageF(dw, dh, cn);
    const int swe = sw * cn, dwe = dw * cn;
    const int rowlen = (2 * sw + 1) * cn;
    std::vector<std::vector<float>> hrow(sh);
    auto horiz = [&](int sy) -> const std::vector<float>& {
        std::vector<float>& row = hrow[sy];
        if (!row.empty()) return row;
        row.assign(rowlen, 0.f);
        const float* s = src.row(sy);
        if (swe == cn) {
            for (int x = 0; x < cn; ++x) row[x] = row[x + cn] = s[x] * 8;
            return row;
        }
        for (int x = 0; x < cn; ++x) {
            int dx = x;                                   // dtab[x] for the first pixel
            row[dx] = s[x] * 6 + s[x + cn] * 2;
            row[dx + cn] = (s[x] + s[x + cn]) * 4;
            int sx = swe - cn + x;
            dx = (sx / cn) * 2 * cn + sx % cn;
            row[dx] = s[sx - cn] + s[sx] * 7;
            row[dx + cn] = s[sx] * 8;
            if (dwe > swe * 2) row[(dw - 1) + x] = row[dx + cn];     // reference quirk (pyramids.cpp:962-965)
        }
        for (int x = cn; x < swe - cn; ++x) {
            int dx = (x / cn) * 2 * cn + x % cn;
            row[dx] = s[x - cn] + s[x] * 6 + s[x + cn];
            row[dx + cn] = (s[x] + s[x + cn]) * 4;
        }
        return row;
    };
    const float s64 = 1.f / 64;
    for (int y = 0; y < sh; ++y) {
        int ym = border_reflect101((y - 1) * 2, sh * 2) / 2, yp = border_reflect101((y + 1) * 2, sh * 2) / 2;
        const float *r0 = horiz(ym).data(), *r1 = horiz(y).data(), *r2 = horiz(yp).data();
        float* d0 = dst.row(y * 2);
        float* d1 = dst.row(std::min(y * 2 + 1, dh - 1));
        for (int x = 0; x < dwe; ++x) {
            float t1 = ((r1[x] + r2[x]) * 4) * s64;
            float t0 = (r0[x] + r1[x] * 6 + r2[x]) * s64;
            d1[x] = t1; d0[x] = t0;                       // d0 written last: wins when both alias
        }
    }
    if (dh > sh * 2) {
        const float* a = dst.row(sh * 2 - 2);
        float* b = dst.row(sh * 2);
        for (int x = 0; x < dwe; ++x) b[x] = a[x];
    }
}

// ------------------------------------------------------------------------------------------------
static void replicate3(const ImageF& m, ImageF& out) {
    out = ImageF(m.w, m.h, 3);
    for (size_t i = 0; i < (size_t)m.w * m.h; ++i) out.d[i * 3] = out.d[i * 3 + 1] = out.d[i * 3 + 2] = m.d[i];
}

void laplacian_blend(const ImageF& l, const ImageF& r, const ImageF& mask, int levels, ImageF& out) {
    std::vector<ImageF> lapL(levels), lapR(levels), mk(levels + 1);
    ImageF smallL, smallR;
    auto build = [&](const ImageF& img, std::vector<ImageF>& lap, ImageF& smallest) {
        ImageF cur = img;
        for (int i = 0; i < levels; ++i) {
            ImageF down, up;
            pyr_down(cur, down);
            pyr_up(down, up, cur.w, cur.h);
            lap[i] = ImageF(cur.w, cur.h, cur.c);
            for (size_t k = 0; k < cur.d.size(); ++k) lap[i].d[k] = cur.d[k] - up.d[k];
            cur = std::move(down);
        }
        smallest = cur;
    };
    build(l, lapL, smallL);
    build(r, lapR, smallR);
    ImageF curm = mask;
    replicate3(curm, mk[0]);
    for (int i = 1; i <= levels; ++i) {
        ImageF down;
        pyr_down(curm, down);          // target size equals the default (w+1)/2 in every case
        replicate3(down, mk[i]);
        curm = std::move(down);
    }
    auto mix = [](const ImageF& a, const ImageF& b, const ImageF& m, ImageF& res) {
        res = ImageF(a.w, a.h, a.c);
        for (size_t k = 0; k < a.d.size(); ++k) {
            float A = a.d[k] * m.d[k];
            float anti = 1.f - m.d[k];
            float B = b.d[k] * anti;
            res.d[k] = A + B;
        }
    };
    ImageF cur;
    mix(smallL, smallR, mk[levels], cur);
    for (int i = levels - 1; i >= 0; --i) {
        ImageF lvl, up;
        mix(lapL[i], lapR[i], mk[i], lvl);
        pyr_up(cur, up, lvl.w, lvl.h);
        cur = ImageF(lvl.w, lvl.h, lvl.c);
        for (size_t k = 0; k < lvl.d.size(); ++k) cur.d[k] = up.d[k] + lvl.d[k];
    }
    out = std::move(cur);
}

// ------------------------------------------------------------------------------------------------
void gaussian_blur_f32(const ImageF& src, const float* k, int ksize, ImageF& dst) {
    const int W = src.w, H = src.h, C = src.c, r = ksize / 2;
    ImageF tmp(W, H, C);
    for (int y = 0; y < H; ++y) {
        const float* s = src.row(y);
        float* t = tmp.row(y);
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < C; ++c) {
                float acc = s[border_reflect101(x - r, W) * C + c] * k[0];
                for (int j = 1; j < ksize; ++j) acc = s[border_reflect101(x - r + j, W) * C + c] * k[j] + acc;
                t[x * C + c] = acc;
            }
    }
    dst = ImageF(W, H, C);
    for (int y = 0; y < H; ++y) {
        float* d = dst.row(y);
        const float* c0 = tmp.row(y);
        for (int i = 0; i < W * C; ++i) {
            float acc = k[r] * c0[i] + 0.f;
            for (int j = 1; j <= r; ++j)
                acc = k[r + j] * (tmp.row(border_reflect101(y + j, H))[i] + tmp.row(border_reflect101(y - j, H))[i]) + acc;
            d[i] = acc;
        }
    }
}

static inline void mnmx(float& a, float& b) {
    float t = a;
    a = (b < a) ? b : a;       // std::min(a, b)
    b = (b < t) ? t : b;       // std::max(b, t)
}

void median3_f32(const ImageF& src, ImageF& dst) {
    const int W = src.w, H = src.h, C = src.c;
    dst = ImageF(W, H, C);
    if (W == 1 || H == 1) {
        int len = W + H - 1;
        for (int i = 0; i < len; ++i)
            for (int c = 0; c < C; ++c) {
                float p0 = src.d[(size_t)(i > 0 ? i - 1 : i) * C + c], p1 = src.d[(size_t)i * C + c];
                float p2 = src.d[(size_t)(i < len - 1 ? i + 1 : i) * C + c];
                mnmx(p0, p1); mnmx(p1, p2); mnmx(p0, p1);
                dst.d[(size_t)i * C + c] = p1;
            }
        return;
    }
    for (int y = 0; y < H; ++y) {
        const float *r0 = src.row(std::max(y - 1, 0)), *r1 = src.row(y), *r2 = src.row(std::min(y + 1, H - 1));
        float* d = dst.row(y);
        for (int j = 0; j < W * C; ++j) {
            int j0 = j >= C ? j - C : j, j2 = j < W * C - C ? j + C : j;
            float p0 = r0[j0], p1 = r0[j], p2 = r0[j2], p3 = r1[j0], p4 = r1[j], p5 = r1[j2], p6 = r2[j0], p7 = r2[j], p8 = r2[j2];
            mnmx(p1, p2); mnmx(p4, p5); mnmx(p7, p8); mnmx(p0, p1);
            mnmx(p3, p4); mnmx(p6, p7); mnmx(p1, p2); mnmx(p4, p5);
            mnmx(p7, p8); mnmx(p0, p3); mnmx(p5, p8); mnmx(p4, p7);
            mnmx(p3, p6); mnmx(p1, p4); mnmx(p2, p5); mnmx(p4, p7);
            mnmx(p4, p2); mnmx(p6, p4); mnmx(p4, p2);
            d[j] = p4;
        }
    }
}

void unsharp_mask(const ImageF& src, float radius, float amount, float threshold, ImageF& dst, ImageF* blurOut, ImageF* medOut) {
    // GaussianBlur(Size(0,0), radius) on float: ksize = cvRound(radius*4*2 + 1) | 1 (smooth.dispatch.cpp:289): 9 taps for the
    // per-frame call (radius 1), 17 for the pre-ORB call (radius 2); taps = exp(-x^2 / 2 sigma^2) / sum in double, stored as float
    ImageF blurred, diff(src.w, src.h, src.c), med;
    if (radius == 1.f) gaussian_blur_f32(src, kGauss9Sigma1, 9, blurred);
    else {
        const int n = cv_round_f(radius * 4 * 2 + 1) | 1;
        std::vector<double> v(n); double sum = 0;
        for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; v[i] = std::exp(-(x * x) / (2.0 * radius * radius)); sum += v[i]; }
        std::vector<float> taps(n);
        for (int i = 0; i < n; ++i) taps[i] = (float)(v[i] / sum);
        gaussian_blur_f32(src, taps.data(), n, blurred);
    }
    for (size_t i = 0; i < src.d.size(); ++i) diff.d[i] = src.d[i] - blurred.d[i];
    median3_f32(diff, med);
    dst = src;
    for (size_t p = 0; p < (size_t)src.w * src.h; ++p) {
        const float* d = &med.d[p * 3];
        double s = (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2];
        if (std::sqrt(s) >= threshold)
            for (int c = 0; c < 3; ++c) dst.d[p * 3 + c] = src.d[p * 3 + c] + amount * d[c];
    }
    if (blurOut) *blurOut = std::move(blurred);
    if (medOut) *medOut = std::move(med);
}

}  // namespace oracle
