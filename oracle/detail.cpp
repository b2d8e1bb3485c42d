// oracle/detail.cpp — dft_detail2, the radial gradient and the ORB input image.  TEST INFRASTRUCTURE.
//
//   src/experiments.hpp:267-303  dft_spectrum: zero-pad to getOptimalDFTSize, complex cv::dft, magnitude, +1, cv::log, crop to
//                                even sizes, quadrant swap, normalize(0, 1, NORM_MINMAX)
//   src/experiments.hpp:305-318  dft_detail2: RMS over spectrum.at<uchar>(r, c), r < rows, c < cols — i.e. over the first `cols`
//                                raw BYTES of every float row
//   src/extractor.cpp:40-45      detail = 255 / max(d1, d2), nfeatures = int(max_keypoints * detail)
//   src/extractor.cpp:50-76      ORB input: unsharp(sigma 2) grey x Gabor mean x radial gradient -> u8 -> equalizeHist
//   src/draw.cpp:40-59           draw_radial_gradiant2
//   OCV/core/src/dxt.cpp:158-400 (DFTFactorize / DFTInit: factor order, permutation, twiddle recurrence),
//                :645-727,835-1130 (DFT<float>: SSE3 radix-4, radix-2, radix-3, radix-5), :3150-3400 (row pass, then column pass)
//   OCV/core/src/mathfuncs_core.simd.hpp (magnitude32f: sqrt(x*x + y*y) in float), norm.cpp:1384-1397 (normalize)
//
// Because the RMS is taken over raw float bytes, every float of the spectrum has to come out with the reference's exact
// bits: same factor order, same permutation, same twiddles (a double recurrence rounded to float) and the same order of
// operations inside each butterfly.  The structure here — a pass list and one routine per radix working on a gathered
// line — is this file's own; only the arithmetic per butterfly is dictated.
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {

namespace {

struct Cx { float re, im; };

inline Cx mulc(Cx a, Cx w) { return Cx{a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re}; }

const double kPow2Roots[32][2] = {
#include "dft_pow2_roots.inc"
};

// One 1-D transform of length n = 2^a * 3^b * 5^c.
struct LinePlan {
    int n = 1;
    int pow2 = 1;                     // the power-of-two factor (1 when n is odd)
    std::vector<int> odd;             // the odd prime factors in the order the passes run: descending
    std::vector<int> perm;            // output slot i is loaded from input perm[i]
    std::vector<Cx> tw;               // tw[k] = exp(-2 pi i k / n), from the reference's recurrence
};

int reverse_bits(int v, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((v >> b) & 1) << (bits - 1 - b);
    return r;
}

bool make_plan(int n, LinePlan& p) {
    p.n = n;
    p.pow2 = n & -n;                                   // lowest set bit = the whole power of two
    int rest = n / p.pow2;
    std::vector<int> asc;
    for (int f = 3; rest > 1; f += 2)
        while (rest % f == 0) { asc.push_back(f); rest /= f; }
    p.odd.assign(asc.rbegin(), asc.rend());            // the reference finds them ascending and then reverses the odd part
    for (int f : p.odd) if (f != 3 && f != 5) return false;       // getOptimalDFTSize never yields anything else
    // The digits of an output index, least significant first, run over the factors in PASS order (power of two first);
    // the source index weighs digit k with the product of all LATER factors (digit reversal), and the power-of-two digit
    // is bit-reversed inside its own range.
    std::vector<int> fac;
    if (p.pow2 > 1) fac.push_back(p.pow2);
    fac.insert(fac.end(), p.odd.begin(), p.odd.end());
    int bits = 0;
    while ((1 << bits) < p.pow2) ++bits;
    p.perm.resize(n);
    for (int i = 0; i < n; ++i) {
        int r = i, weight = n, src = 0;
        for (size_t k = 0; k < fac.size(); ++k) {
            int d = r % fac[k];
            r /= fac[k];
            weight /= fac[k];
            if (k == 0 && p.pow2 > 1) d = reverse_bits(d, bits);
            src += d * weight;
        }
        p.perm[i] = src;
    }
    // Twiddles: w_1 = (cos, -sin)(2 pi / n) — from the constant table when n is a power of two, else sin from libm and
    // cos = sqrt(1 - sin^2) — then w_{k+1} = w_k * w_1 in double, each stored rounded to float; the upper half is the mirror.
    double c1, s1;
    if ((n & (n - 1)) == 0) {
        int m = 0;
        while ((1 << m) < n) ++m;
        c1 = kPow2Roots[m][0]; s1 = -kPow2Roots[m][1];
    } else {
        s1 = std::sin(-M_PI * 2 / n);
        c1 = std::sqrt(1. - s1 * s1);
    }
    p.tw.assign(n, Cx{0.f, 0.f});
    p.tw[0] = Cx{1.f, 0.f};
    const int half = (n + 1) / 2;
    if ((n & 1) == 0) p.tw[half] = Cx{-1.f, 0.f};
    double wr = c1, wi = s1;
    for (int k = 1; k < half; ++k) {
        p.tw[k] = Cx{(float)wr, (float)wi};
        p.tw[n - k] = Cx{(float)wr, (float)-wi};
        const double t = wr * c1 - wi * s1;
        wi = wr * s1 + wi * c1;
        wr = t;
    }
    return true;
}

// the butterflies: `span` = distance between the inputs of one butterfly, `step` = twiddle stride of this pass
void pass_radix4(Cx* v, int n, int span, int step, const Cx* tw) {
    for (int base = 0; base < n; base += 4 * span)
        for (int j = 0; j < span; ++j) {
            Cx* q = v + base + j;
            Cx x0 = q[0], x1 = q[span], x2 = q[2 * span], x3 = q[3 * span];
            if (j) {
                const int d = j * step;
                x1 = mulc(x1, tw[2 * d]);              // the reference pairs the second input with w^2 and the third with w^1
                x3 = mulc(x3, tw[3 * d]);
                x2 = mulc(x2, tw[d]);
            }
            const Cx a = {x0.re + x1.re, x0.im + x1.im}, b = {x2.re + x3.re, x2.im + x3.im};
            const Cx c = {x0.re - x1.re, x0.im - x1.im}, e = {x2.re - x3.re, x2.im - x3.im};
            q[0] = Cx{a.re + b.re, a.im + b.im};
            q[span] = Cx{c.re + e.im, c.im - e.re};
            q[2 * span] = Cx{a.re - b.re, a.im - b.im};
            q[3 * span] = Cx{c.re - e.im, c.im + e.re};
        }
}

void pass_radix2(Cx* v, int n, int span, int step, const Cx* tw) {
    for (int base = 0; base < n; base += 2 * span)
        for (int j = 0; j < span; ++j) {
            Cx* q = v + base + j;
            const Cx x0 = q[0];
            Cx x1 = q[span];
            if (j) { const Cx w = tw[j * step]; x1 = Cx{x1.re * w.re - x1.im * w.im, x1.im * w.re + x1.re * w.im}; }
            q[0] = Cx{x0.re + x1.re, x0.im + x1.im};
            q[span] = Cx{x0.re - x1.re, x0.im - x1.im};
        }
}

void pass_radix3(Cx* v, int n, int span, int step, const Cx* tw) {
    const float s120 = (float)0.86602540378443864676372317075294;
    for (int base = 0; base < n; base += 3 * span)
        for (int j = 0; j < span; ++j) {
            Cx* q = v + base + j;
            Cx a = q[span], b = q[2 * span];
            if (j) { a = mulc(a, tw[j * step]); b = mulc(b, tw[2 * j * step]); }
            const float r1 = a.re + b.re, i1 = a.im + b.im;
            const float r2 = s120 * (a.im - b.im), i2 = s120 * (b.re - a.re);
            float r0 = q[0].re, i0 = q[0].im;
            q[0] = Cx{r0 + r1, i0 + i1};
            r0 -= 0.5f * r1; i0 -= 0.5f * i1;
            q[span] = Cx{r0 + r2, i0 + i2};
            q[2 * span] = Cx{r0 - r2, i0 - i2};
        }
}

void pass_radix5(Cx* v, int n, int span, int step, const Cx* tw) {
    const float k2 = (float)0.559016994374947424102293417182819, k3 = (float)-0.951056516295153572116439333379382;
    const float k4 = (float)-1.538841768587626701285145288018455, k5 = (float)0.363271264002680442947733378740309;
    for (int base = 0; base < n; base += 5 * span)
        for (int j = 0; j < span; ++j) {
            Cx* q = v + base + j;
            const int d = j * step;                     // multiplied by the twiddle even when j = 0 (w = 1)
            const Cx b1 = mulc(q[span], tw[d]), b4 = mulc(q[4 * span], tw[4 * d]);
            const Cx b3 = mulc(q[3 * span], tw[3 * d]), b2 = mulc(q[2 * span], tw[2 * d]);
            float r1 = b1.re + b4.re, i1 = b1.im + b4.im;
            float r3 = b1.re - b4.re, i3 = b1.im - b4.im;
            float r2 = b3.re + b2.re, i2 = b3.im + b2.im;
            float r4 = b3.re - b2.re, i4 = b3.im - b2.im;
            float r0 = q[0].re, i0 = q[0].im;
            float r5 = r1 + r2, i5 = i1 + i2;
            q[0] = Cx{r0 + r5, i0 + i5};
            r0 -= 0.25f * r5; i0 -= 0.25f * i5;
            r1 = k2 * (r1 - r2); i1 = k2 * (i1 - i2);
            r2 = -k3 * (i3 + i4); i2 = k3 * (r3 + r4);
            i3 *= -k5; r3 *= k5;
            i4 *= -k4; r4 *= k4;
            r5 = r2 + i3; i5 = i2 + r3;
            r2 -= i4; i2 -= r4;
            r3 = r0 + r1; i3 = i0 + i1;
            r0 -= r1; i0 -= i1;
            q[span] = Cx{r3 + r2, i3 + i2};
            q[4 * span] = Cx{r3 - r2, i3 - i2};
            q[2 * span] = Cx{r0 + r5, i0 + i5};
            q[3 * span] = Cx{r0 - r5, i0 - i5};
        }
}

void transform_line(const LinePlan& p, const Cx* in, Cx* out) {
    const int n = p.n;
    for (int i = 0; i < n; ++i) out[i] = in[p.perm[i]];
    int done = 1;                                       // length of the sub-transforms finished so far
    while (done * 4 <= p.pow2) { pass_radix4(out, n, done, n / (done * 4), p.tw.data()); done *= 4; }
    if (done < p.pow2) { pass_radix2(out, n, done, n / (done * 2), p.tw.data()); done *= 2; }
    for (int f : p.odd) {
        if (f == 3) pass_radix3(out, n, done, n / (done * 3), p.tw.data());
        else pass_radix5(out, n, done, n / (done * 5), p.tw.data());
        done *= f;
    }
}

int optimal_dft_size(int n) {                           // getOptimalDFTSize: the next 2^a 3^b 5^c
    for (int m = n;; ++m) {
        int t = m;
        for (int f : {2, 3, 5}) while (t % f == 0) t /= f;
        if (t == 1) return m;
    }
}

}  // namespace

double dft_detail2(const ImageU8& gray) {
    const int W = gray.w, H = gray.h, N = optimal_dft_size(W), M = optimal_dft_size(H);
    LinePlan rows, cols;
    if (!make_plan(N, rows) || !make_plan(M, cols)) return -1;
    std::vector<Cx> a((size_t)N * M, Cx{0.f, 0.f}), b((size_t)N * M);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) a[(size_t)y * N + x].re = (float)gray.d[(size_t)y * W + x];
    for (int y = 0; y < M; ++y) transform_line(rows, &a[(size_t)y * N], &b[(size_t)y * N]);
    std::vector<Cx> cin(M), cout(M);
    for (int x = 0; x < N; ++x) {
        for (int y = 0; y < M; ++y) cin[y] = b[(size_t)y * N + x];
        transform_line(cols, cin.data(), cout.data());
        for (int y = 0; y < M; ++y) a[(size_t)y * N + x] = cout[y];
    }
    const int Nc = N & -2, Mc = M & -2;
    std::vector<float> mag((size_t)Nc * Mc);
    float lo = 0, hi = 0;
    for (int y = 0; y < Mc; ++y)
        for (int x = 0; x < Nc; ++x) {
            const Cx c = a[(size_t)y * N + x];
            float m = std::sqrt(c.re * c.re + c.im * c.im);        // float sqrt of the float sum (IEEE: correctly rounded)
            m = m + 1.f;
            m = cv_log32f(m);
            mag[(size_t)y * Nc + x] = m;
            if ((x | y) == 0) lo = hi = m;
            lo = std::min(lo, m); hi = std::max(hi, m);
        }
    // normalize(.., 0, 1, NORM_MINMAX) to CV_32F: the scale is rounded to float first, the shift is built from the rounded scale
    double scale = (1.0 - 0.0) * ((double)hi - (double)lo > 2.220446049250313e-16 ? 1. / ((double)hi - (double)lo) : 0);
    scale = (float)scale;
    const float fs = (float)scale, fb = (float)((float)0.0 - (float)((double)lo * scale));
    const int cx = Nc / 2, cy = Mc / 2;
    double pow_sum = 0;
    for (int r = 0; r < Mc; ++r)
        for (int cbyte = 0; cbyte < Nc; ++cbyte) {      // byte cbyte of row r of the quadrant-swapped, normalised image
            const int fcol = cbyte >> 2;
            const float v = mag[(size_t)((r + cy) % Mc) * Nc + (fcol + cx) % Nc] * fs + fb;
            uint32_t u;
            memcpy(&u, &v, 4);
            const double byte = (double)((u >> (8 * (cbyte & 3))) & 255u);
            pow_sum += byte * byte;
        }
    return std::sqrt(pow_sum / ((double)Nc * Mc));
}

// draw_radial_gradiant2 (src/draw.cpp:40-59)
void radial_gradient(int width, int height, ImageF& out) {
    const int ccx = (int)(width / 2.0), ccy = (int)(height / 2.0);             // cv::Point(double, double) truncates
    const double max_dist = std::hypot(width / 2.0, height / 2.0);
    ImageF g(width, height, 1);
    for (int row = 0; row < height; ++row)
        for (int col = 0; col < width; ++col) {
            const double dist = std::hypot((double)(ccx - col), (double)(ccy - row)) / max_dist;
            g.d[(size_t)row * width + col] = (float)std::pow(std::sin(std::sin(dist * (M_PI / 2)) * (M_PI / 2)), 32);
        }
    ImageU8 g8;
    f32_to_u8(g, g8);                                                           // convertTo(CV_8U, 255.0)
    int mn = 255, mx = 0;
    for (uint8_t v : g8.d) { mn = std::min<int>(mn, v); mx = std::max<int>(mx, v); }
    const double scale = 255.0 * (mx - mn > 2.220446049250313e-16 ? 1. / (mx - mn) : 0), shift = 0.0 - mn * scale;
    const float fs = (float)scale, fb = (float)shift;
    out = ImageF(width, height, 1);
    for (size_t i = 0; i < g8.d.size(); ++i) {
        int v = cv_round_f((float)g8.d[i] * fs + fb);                            // normalize -> convertTo(u8, scale, shift)
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        const uint8_t inv = (uint8_t)~(uint8_t)v;                                // bitwise_not
        out.d[i] = (float)inv * (float)(1.0 / 255.0) + 0.f;                      // convertTo(CV_32F, 1/255)
    }
}

// Extractor::keypoints' image chain for one goodFeatures image (src/extractor.cpp:50-76)
void orb_input_image(const ImageU8& good_features, ImageU8& g) {
    ImageF us, gb, radial, prod(good_features.w, good_features.h, 1);
    orb_unsharp_gray(good_features, us);
    std::vector<float> bank;
    gabor_bank(31, 5, 2, 0.04, M_PI / 4, bank);
    gabor_filter_direct(us, 31, bank, gb);
    radial_gradient(good_features.w, good_features.h, radial);
    for (size_t i = 0; i < prod.d.size(); ++i) prod.d[i] = (gb.d[i] * us.d[i]) * radial.d[i];      // multiply, multiply
    ImageU8 q;
    f32_to_u8(prod, q);
    equalize_hist(q, g);
}

}  // namespace oracle
