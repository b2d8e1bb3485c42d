// oracle/prefilter.cpp — CPU restatement of the first part of the pre-ORB filter chain: Extractor::foreground
// (src/extractor.cpp:136-229).  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
//   grey   = cvtColor(BGR2GRAY)                                   OCV/imgproc/src/color_rgb.simd.hpp:646-730
//   flow   = BackgroundSubtractorMOG2(500, 16, shadows).apply()   OCV/video/src/bgfg_gaussmix2.cpp:479-523,539-755,847-884
//   acc   += flow * (1/6)   (u8 convertTo with a float scale, then saturating add)
//                                                                 OCV/core/src/matrix_expressions.cpp:270-275,1330-1336
//   med    = medianBlur(last, 8i+1)   (exact median, replicated border)     OCV/imgproc/src/median_blur.simd.hpp
//   acc    = GaussianBlur(acc, 23x23, sigma 1)  (8-bit fixed-point path)    OCV/imgproc/src/smooth.dispatch.cpp:224-258,611-683,
//                                                                           smooth.simd.hpp:1136-1199,1780-1866
//   mask   = log(acc/255 * 19 + 1) / log(20);  fg = equalizeHist(u8(grey/255 * mask * 255))
//                                                                 OCV/core/src/mathfuncs_core.simd.hpp:683-752,
//                                                                 OCV/imgproc/src/histogram.cpp:3436-3493
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {

// ---- BGR -> grey, 8 bit: (b*3735 + g*19235 + r*9798 + 2^14) >> 15 --------------------------------------------------
void bgr_to_gray_u8(const ImageU8& bgr, ImageU8& gray) {
    gray = ImageU8(bgr.w, bgr.h, 1);
    const size_t n = (size_t)bgr.w * bgr.h;
    for (size_t i = 0; i < n; ++i) {
        const int b = bgr.d[3 * i], g = bgr.d[3 * i + 1], r = bgr.d[3 * i + 2];
        gray.d[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

// ---- MOG2, one channel, default parameters ---------------------------------------------------------------------------
Mog2::Mog2(int w_, int h_) : w(w_), h(h_), nframes(0) {
    const size_t n = (size_t)w * h;
    weight.assign(n * kModes, 0.f); variance.assign(n * kModes, 0.f); mean.assign(n * kModes, 0.f);
    used.assign(n, 0);
}

void Mog2::apply(const ImageU8& img, ImageU8& mask) {
    const float Tb = 16.f, TB = 0.9f, Tg = 9.f, varInit = 15.f, varMin = 4.f, varMax = 75.f, tau = 0.5f;
    const double fCT = 0.05f;
    ++nframes;
    const double learningRate = 1. / std::min(2 * nframes, 500);
    const float alphaT = (float)learningRate, alpha1 = 1.f - alphaT, prune = (float)(-learningRate * fCT);
    mask = ImageU8(w, h, 1);
    const size_t n = (size_t)w * h;
    for (size_t p = 0; p < n; ++p) {
        float* gw = &weight[p * kModes]; float* gv = &variance[p * kModes]; float* mu = &mean[p * kModes];
        const float data = (float)img.d[p];
        bool background = false, fitsPDF = false;
        int nmodes = used[p];
        float totalWeight = 0.f;
        for (int mode = 0; mode < nmodes; ++mode) {            // nmodes shrinks inside the loop when a mode is pruned, as in the reference
            float wgt = alpha1 * gw[mode] + prune;
            int swap_count = 0;
            if (!fitsPDF) {
                const float var = gv[mode];
                const float dD = mu[mode] - data;
                float dist2 = 0.f;
                dist2 += dD * dD;
                if (totalWeight < TB && dist2 < Tb * var) background = true;
                if (dist2 < Tg * var) {
                    fitsPDF = true;
                    wgt += alphaT;
                    const float k = alphaT / wgt;
                    mu[mode] -= k * dD;
                    float varnew = var + k * (dist2 - var);
                    varnew = std::max(varnew, varMin);
                    varnew = std::min(varnew, varMax);
                    gv[mode] = varnew;
                    for (int i = mode; i > 0; --i) {
                        if (wgt < gw[i - 1]) break;
                        ++swap_count;
                        std::swap(gw[i], gw[i - 1]); std::swap(gv[i], gv[i - 1]); std::swap(mu[i], mu[i - 1]);
                    }
                }
            }
            if (wgt < -prune) { wgt = 0.f; --nmodes; }
            gw[mode - swap_count] = wgt;
            totalWeight += wgt;
        }
        float invWeight = 0.f;
        if (std::fabs(totalWeight) > 1.1920928955078125e-7f) invWeight = 1.f / totalWeight;
        for (int mode = 0; mode < nmodes; ++mode) gw[mode] *= invWeight;
        if (!fitsPDF && alphaT > 0.f) {
            const int mode = nmodes == kModes ? kModes - 1 : nmodes++;
            if (nmodes == 1) gw[mode] = 1.f;
            else {
                gw[mode] = alphaT;
                for (int i = 0; i < nmodes - 1; ++i) gw[i] *= alpha1;
            }
            mu[mode] = data;
            gv[mode] = varInit;
            for (int i = nmodes - 1; i > 0; --i) {
                if (alphaT < gw[i - 1]) break;
                std::swap(gw[i], gw[i - 1]); std::swap(gv[i], gv[i - 1]); std::swap(mu[i], mu[i - 1]);
            }
        }
        used[p] = (uint8_t)nmodes;
        uint8_t out = 0;
        if (!background) {
            out = 255;
            float tWeight = 0.f;                     // detectShadowGMM
            for (int mode = 0; mode < nmodes; ++mode) {
                float numerator = 0.f, denominator = 0.f;
                numerator += data * mu[mode];
                denominator += mu[mode] * mu[mode];
                if (denominator == 0) break;
                if (numerator <= denominator && numerator >= tau * denominator) {
                    const float a = numerator / denominator;
                    float dist2a = 0.f;
                    const float dDs = a * mu[mode] - data;
                    dist2a += dDs * dDs;
                    if (dist2a < Tb * gv[mode] * a * a) { out = 127; break; }
                }
                tWeight += gw[mode];
                if (tWeight > TB) break;
            }
        }
        mask.d[p] = out;
    }
}

// ---- acc += saturate(round(flow * (float)(1/6))) ---------------------------------------------------------------------
void accumulate_scaled_u8(ImageU8& acc, const ImageU8& flow, double scale) {
    const float a = (float)scale;
    for (size_t i = 0; i < acc.d.size(); ++i) {
        int t = cv_round_f((float)flow.d[i] * a + 0.f);
        t = t < 0 ? 0 : t > 255 ? 255 : t;
        const int s = acc.d[i] + t;
        acc.d[i] = (uint8_t)(s > 255 ? 255 : s);
    }
}

// ---- medianBlur, 8 bit, odd ksize, replicated border: the exact median of the ksize x ksize window ----------------------
void median_blur_u8(const ImageU8& src, int ksize, ImageU8& dst) {
    dst = ImageU8(src.w, src.h, 1);
    if (ksize <= 1) { dst.d = src.d; return; }
    const int r = ksize / 2, w = src.w, h = src.h, half = (ksize * ksize) / 2;     // median = element number `half` (0-based)
    std::vector<int> rows(ksize);
    for (int y = 0; y < h; ++y) {
        for (int k = 0; k < ksize; ++k) rows[k] = std::min(std::max(y + k - r, 0), h - 1);
        int hist[256] = {0};
        auto add_col = [&](int cx, int delta) {
            const int x = std::min(std::max(cx, 0), w - 1);
            for (int k = 0; k < ksize; ++k) hist[src.d[(size_t)rows[k] * w + x]] += delta;
        };
        for (int cx = -r; cx <= r; ++cx) add_col(cx, 1);
        for (int x = 0; x < w; ++x) {
            int s = 0, v = 0;
            for (; v < 256; ++v) { s += hist[v]; if (s > half) break; }
            dst.d[(size_t)y * w + x] = (uint8_t)v;
            add_col(x - r, -1);
            add_col(x + r + 1, 1);
        }
    }
}

// ---- GaussianBlur 23x23 sigma 1 on 8 bit: taps in 8.8 fixed point with error diffusion, exact integer sums ---------------
// (getGaussianKernelFixedPoint_ED: round(k_i*256 + carried error), centre = 256 - rest.)  No saturation can occur: the taps
// sum to 256, so the horizontal sum is <= 255*256 and fits 16 bits.
const int kGauss23Sigma1Fx[23] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 14, 62, 102, 62, 14, 1, 0, 0, 0, 0, 0, 0, 0, 0};

void gaussian_blur23_u8(const ImageU8& src, ImageU8& dst) {
    const int w = src.w, h = src.h, n = 23, r = 11;
    std::vector<uint16_t> hbuf((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < n; ++k)
                if (kGauss23Sigma1Fx[k]) s += (uint32_t)kGauss23Sigma1Fx[k] * src.d[(size_t)y * w + border_reflect101(x + k - r, w)];
            hbuf[(size_t)y * w + x] = (uint16_t)s;
        }
    dst = ImageU8(w, h, 1);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < n; ++k)
                if (kGauss23Sigma1Fx[k]) s += (uint32_t)kGauss23Sigma1Fx[k] * hbuf[(size_t)border_reflect101(y + k - r, h) * w + x];
            const uint32_t v = (s + (1u << 15)) >> 16;
            dst.d[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
        }
}

void foreground_mask(const ImageU8& grey, ImageU8& fgMask, std::vector<ImageU8>* stages) {
    fgMask = ImageU8(grey.w, grey.h, 1);
    ImageU8 last = grey, med, flow, blur;
    Mog2 bs(grey.w, grey.h);
    bs.apply(grey, flow);
    accumulate_scaled_u8(fgMask, flow, 1.0 / (12 / 2.0));
    if (stages) { stages->push_back(flow); stages->push_back(fgMask); }
    for (int i = 0; i < 12; ++i) {
        median_blur_u8(last, i * 8 + 1, med);
        bs.apply(med, flow);
        accumulate_scaled_u8(fgMask, flow, 1.0 / (12 / 2.0));
        if (stages) { stages->push_back(med); stages->push_back(flow); stages->push_back(fgMask); }
        gaussian_blur23_u8(fgMask, blur);
        fgMask = blur;
        if (stages) stages->push_back(fgMask);
        last = med;
    }
}

// ---- cv::log on float: 256-entry table of (ln(1 + i/256), 1/(1 + i/256)) + cubic ------------------------------------------
static const float* log_tab() {
    static float tab[512];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 255; ++i) {
            const long double t = 1.0L + (long double)i / 256.0L;
            tab[2 * i] = (float)(double)logl(t);            // the reference table holds these as doubles, used as floats
            tab[2 * i + 1] = (float)(double)(1.0L / t);
        }
        // the last interval is expanded around 2 instead: (ln 2, 1/2), with x0 shifted by -1/512 (OCV/core/src/mathfuncs.cpp:2151-2408)
        tab[510] = (float)0.69314718055994530941723212145818; tab[511] = 0.5f;
        init = true;
    }
    return tab;
}

float cv_log32f(float x) {
    const float* tab = log_tab();
    const float A0 = 0.3333333333333333333333333f, A1 = -0.5f, A2 = 1.f;
    const float ln2 = (float)0.69314718055994530941723212145818;
    int32_t i0;
    memcpy(&i0, &x, 4);
    const int32_t bi = (i0 & ((1 << 15) - 1)) | (127 << 23);
    float bf;
    memcpy(&bf, &bi, 4);
    const int idx = (i0 >> 14) & 510;
    const float y0 = (float)(((i0 >> 23) & 0xff) - 127) * ln2 + tab[idx];
    const float x0 = (bf - 1.f) * tab[idx + 1] + (idx == 510 ? -1.f / 512 : 0.f);
    return ((A0 * x0 + A1) * x0 + A2) * x0 + y0;
}

void equalize_hist(const ImageU8& src, ImageU8& dst) {
    dst = ImageU8(src.w, src.h, 1);
    int hist[256] = {0};
    for (uint8_t v : src.d) ++hist[v];
    int i = 0;
    while (!hist[i]) ++i;
    const int total = (int)src.d.size();
    if (hist[i] == total) { std::fill(dst.d.begin(), dst.d.end(), (uint8_t)i); return; }
    const float scale = (256 - 1.f) / (total - hist[i]);
    int lut[256] = {0};
    int sum = 0;
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        int v = cv_round_f((float)sum * scale);
        lut[i] = v < 0 ? 0 : v > 255 ? 255 : v;
    }
    for (size_t p = 0; p < src.d.size(); ++p) dst.d[p] = (uint8_t)lut[src.d[p]];
}

// draw_radial_gradiant (src/draw.cpp:21-38) as Extractor::foreground uses it with Settings::enable_radial_mask (src/extractor.cpp:178-185):
// float image of pow(sin(sin(d * pi/2) * pi/2), 12), d = distance to cv::Point(cols / 2.0, rows / 2.0) (the doubles truncate to int) over
// hypot(cols / 2, rows / 2); normalize(0, 255, NORM_MINMAX, CV_8U) = convertTo(CV_8U, scale, shift) with scale = 255 / (max - min) and
// shift = -min * scale as doubles, applied as floats (cvt_32f: v * a + b, unfused, cvRound, saturate); bitwise_not; convertTo(CV_32F, 1/255).
// PARITY UNPINNED: no reference-run fixture exists for this option (the survey-stage OpenCV build is gone from the container); every
// primitive used here is the one pinned by the draw_radial_gradiant2 fixture in detail.cpp.
void radial_mask(int width, int height, ImageF& out) {
    const int ccx = (int)(width / 2.0), ccy = (int)(height / 2.0);
    const double max_dist = std::hypot(width / 2.0, height / 2.0);
    std::vector<float> g((size_t)width * height);
    float mn = INFINITY, mx = -INFINITY;
    for (int row = 0; row < height; ++row)
        for (int col = 0; col < width; ++col) {
            const double dist = std::hypot((double)(ccx - col), (double)(ccy - row)) / max_dist;
            const float v = (float)std::pow(std::sin(std::sin(dist * (M_PI / 2)) * (M_PI / 2)), 12);
            g[(size_t)row * width + col] = v;
            mn = std::min(mn, v); mx = std::max(mx, v);
        }
    const double smin = mn, smax = mx;
    const double scale = 255.0 * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0), shift = 0.0 - smin * scale;
    const float fs = (float)scale, fb = (float)shift;
    out = ImageF(width, height, 1);
    for (size_t i = 0; i < g.size(); ++i) {
        int v = cv_round_f(g[i] * fs + fb);
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        const uint8_t inv = (uint8_t)~(uint8_t)v;
        out.d[i] = (float)inv * (float)(1.0 / 255.0) + 0.f;
    }
}

static bool g_radial_mask = false;                              // Settings::enable_radial_mask of the restatement
void set_radial_mask(bool on) { g_radial_mask = on; }

void foreground(const ImageU8& bgr, ImageU8& fg, ForegroundDebug* dbg) {
    ImageU8 grey, fgMask;
    bgr_to_gray_u8(bgr, grey);
    foreground_mask(grey, fgMask, dbg ? &dbg->stages : nullptr);
    ImageF greyF, maskF;
    u8_to_f32(grey, greyF);
    u8_to_f32(fgMask, maskF);
    if (g_radial_mask) {                                        // multiply(fgMaskFloat, radialMaskFloat, finalMaskFloat), src/extractor.cpp:195-197
        ImageF radial;
        radial_mask(maskF.w, maskF.h, radial);
        for (size_t i = 0; i < maskF.d.size(); ++i) maskF.d[i] = maskF.d[i] * radial.d[i];
    }
    const float ln20 = cv_log32f(20.f);
    ImageF lin(maskF.w, maskF.h, 1), logged(maskF.w, maskF.h, 1), fin(maskF.w, maskF.h, 1), masked(maskF.w, maskF.h, 1);
    for (size_t i = 0; i < maskF.d.size(); ++i) {
        lin.d[i] = maskF.d[i] * 19.f + 1.f;                  // convertTo(CV_32F, 19, 1)
        logged.d[i] = cv_log32f(lin.d[i]);
        fin.d[i] = logged.d[i] / ln20;
        masked.d[i] = greyF.d[i] * fin.d[i];
    }
    ImageU8 m8;
    f32_to_u8(masked, m8);
    equalize_hist(m8, fg);
    if (dbg) { dbg->grey = grey; dbg->ln20 = ln20; dbg->lin = lin; dbg->logged = logged; dbg->final_mask = fin; dbg->masked = m8; }
}

// ---- Extractor::keypoints up to the detector (src/extractor.cpp:50-76), gabor_filter (src/util.cpp:40-60) -----------------------
// getGaborKernel (OCV/imgproc/src/gabor.cpp:50-95), theta_i = i * float(180 / 16) used as radians
void gabor_bank(int ks, double sigma, double lambd, double gamma, double psi, std::vector<float>& bank) {
    bank.resize((size_t)16 * ks * ks);
    const float step = (float)(180 / 16);
    for (int a = 0; a < 16; ++a) {
        const double theta = (double)(a * step), sx = sigma, sy = sigma / gamma, c = std::cos(theta), s = std::sin(theta);
        const int m = ks / 2;
        const double ex = -0.5 / (sx * sx), ey = -0.5 / (sy * sy), cscale = M_PI * 2 / lambd;
        for (int y = -m; y <= m; ++y)
            for (int x = -m; x <= m; ++x) {
                const double xr = x * c + y * s, yr = -x * s + y * c;
                bank[(size_t)a * ks * ks + (m - y) * ks + (m - x)] = (float)(1 * std::exp(ex * xr * xr + ey * yr * yr) * std::cos(cscale * xr + psi));
            }
    }
}

// mean over the bank of clamp(correlate(src, kernel), 0, 1).  The reference evaluates the correlation through OpenCV's DFT-based
// filter2D; this is the direct sum in double, rounded once: the comparator for tolerance checks, not a bit-exact restatement.
void gabor_filter_direct(const ImageF& src, int ks, const std::vector<float>& bank, ImageF& dst) {
    dst = ImageF(src.w, src.h, src.c);
    const int m = ks / 2, w = src.w, h = src.h, cn = src.c;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int ch = 0; ch < cn; ++ch) {
                float sum = 0.f;
                for (int a = 0; a < 16; ++a) {
                    double acc = 0;
                    const float* k = &bank[(size_t)a * ks * ks];
                    for (int dy = 0; dy < ks; ++dy) {
                        const float* row = src.row(border_reflect101(y + dy - m, h));
                        for (int dx = 0; dx < ks; ++dx) acc += (double)k[dy * ks + dx] * row[(size_t)border_reflect101(x + dx - m, w) * cn + ch];
                    }
                    float p = (float)acc;
                    p = p > 1.f ? 1.f : p; p = p < 0.f ? 0.f : p;
                    sum += p;
                }
                dst.d[((size_t)y * w + x) * cn + ch] = sum * 0.0625f;
            }
}

// us = grey(unsharp_mask(triple(gf / 255), 2, 6, 0.1)): exact
void orb_unsharp_gray(const ImageU8& gf, ImageF& us) {
    ImageF f1, trip(gf.w, gf.h, 3), um;
    u8_to_f32(gf, f1);
    for (size_t i = 0; i < f1.d.size(); ++i) trip.d[3 * i] = trip.d[3 * i + 1] = trip.d[3 * i + 2] = f1.d[i];
    unsharp_mask(trip, 2.f, 6.f, 0.1f, um);
    us = ImageF(gf.w, gf.h, 1);
    for (size_t i = 0; i < us.d.size(); ++i) {
        const float b = um.d[3 * i], g = um.d[3 * i + 1], r = um.d[3 * i + 2];
        us.d[i] = r * 0.299f + (g * 0.587f + b * 0.114f);          // color_rgb.simd.hpp:630,638
    }
}

// ---- blur_margin (src/util.cpp:574-602): pad into the union canvas, blur the four margin strips ------------------------------
// 8-bit GaussianBlur, any odd ksize: taps in 8.8 fixed point with error diffusion (smooth.dispatch.cpp:224-258), exact integer
// sums, (sum + 2^15) >> 16 at the end (smooth.simd.hpp:1136-1199,1780-1866); a dimension of size 1 is not filtered (:626-631).
// getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED (smooth.dispatch.cpp:81-258): for sigma <= 0 the fixed tables
// for n = 1, 3, 5, 7, 9 (all multiples of 1/256, so their fixed-point form is exact), else sigma = n*0.15 + 0.35;
// t_i = exp(x^2 * (-0.125 / sigma^2)) over the odd integers x = 2i - (n - 1), normalised by 1 / (2*sum + 1).  The reference
// evaluates this in softfloat; libm's exp agrees to the last bit or differs by one ulp, which the rounding to 8 fractional
// bits absorbs unless a tap sits within 2^-45 of a rounding boundary (pinned by OpenCV's own vectors,
// imgproc/test/test_smooth_bitexact.cpp:14-27, tests/test_oracle_known_answers.py).
void gaussian_taps_fx(int n, double sigma, std::vector<int>& taps) {
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7 || n == 9)) {
        static const int t1[] = {256}, t3[] = {64, 128, 64}, t5[] = {16, 64, 96, 64, 16}, t7[] = {8, 28, 56, 72, 56, 28, 8},
                         t9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
        const int* t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : n == 7 ? t7 : t9;
        taps.assign(t, t + n);
        return;
    }
    const double sg = sigma > 0 ? sigma : n * 0.15 + 0.35;
    const double scale = -0.125 / (sg * sg);
    const int half = (n - 1) / 2;
    std::vector<double> v(half); double sum = 0;
    for (int i = 0, x = 1 - n; i < half; ++i, x += 2) { v[i] = std::exp((double)(x * x) * scale); sum += v[i]; }
    sum = sum * 2 + 1;
    const double mul = 1.0 / sum;
    taps.assign(n, 0);
    double err = 0; int tot = 0;
    for (int i = 0; i < half; ++i) {
        const double adj = (v[i] * mul) * 256 + err;
        const int v0 = cv_round(adj);
        err = adj - v0;
        taps[i] = taps[n - 1 - i] = v0; tot += v0;
    }
    taps[n / 2] = 256 - 2 * tot;
}

void gaussian_blur_fx_u8(const ImageU8& src, int ksize, double sigma, ImageU8& dst) {
    const int w = src.w, h = src.h, cn = src.c;
    std::vector<int> kx, ky, one(1, 256);
    gaussian_taps_fx(ksize, sigma, kx);
    ky = kx;
    if (w == 1) kx = one;
    if (h == 1) ky = one;
    const int nx = (int)kx.size(), ny = (int)ky.size();
    std::vector<uint32_t> hbuf((size_t)w * h * cn);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < cn; ++c) {
                uint32_t s = 0;
                for (int k = 0; k < nx; ++k)
                    if (kx[k]) s += (uint32_t)kx[k] * src.d[((size_t)y * w + border_reflect101(x + k - nx / 2, w)) * cn + c];
                hbuf[((size_t)y * w + x) * cn + c] = s;
            }
    dst = ImageU8(w, h, cn);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < cn; ++c) {
                uint32_t s = 0;
                for (int k = 0; k < ny; ++k)
                    if (ky[k]) s += (uint32_t)ky[k] * hbuf[((size_t)border_reflect101(y + k - ny / 2, h) * w + x) * cn + c];
                const uint32_t v = (s + (1u << 15)) >> 16;
                dst.d[((size_t)y * w + x) * cn + c] = (uint8_t)(v > 255 ? 255 : v);
            }
}

void blur_margin(const ImageU8& src, int uw, int uh, ImageU8& dst) {
    ImageU8 U(uw, uh, 3);
    const double margin = (src.w + src.h) / 100.0;
    double dx = std::fabs((double)(src.w - uw)) / 2.0, dy = std::fabs((double)(src.h - uh)) / 2.0;
    const int rx = (int)dx, ry = (int)dy;
    for (int y = 0; y < src.h; ++y) memcpy(&U.d[((size_t)(ry + y) * uw + rx) * 3], src.row(y), (size_t)src.w * 3);
    dx = (dx == 0 ? 1.3 : dx + margin);
    dy = (dy == 0 ? 1.3 : dy + margin);
    const int rects[4][4] = {{0, 0, (int)dx, uh}, {(int)(uw - dx), 0, (int)dx, uh}, {0, 0, uw, (int)dy}, {0, (int)(uh - dy), uw, (int)dy}};
    dst = U;
    for (const auto& r : rects) {                       // all four strips are cut from the unblurred canvas; written in this order
        ImageU8 strip(r[2], r[3], 3), blurred;
        for (int y = 0; y < r[3]; ++y) memcpy(strip.row(y), &U.d[((size_t)(r[1] + y) * uw + r[0]) * 3], (size_t)r[2] * 3);
        gaussian_blur_fx_u8(strip, 127, 6, blurred);
        for (int y = 0; y < r[3]; ++y) memcpy(&dst.d[((size_t)(r[1] + y) * uw + r[0]) * 3], blurred.row(y), (size_t)r[2] * 3);
    }
}

}  // namespace oracle
