// foreground2.cpp — host side of the rest of the pre-ORB chain (ForegroundFilter::orb_inputs / gabor_field / detail):
// Extractor::keypoints up to the detector call (src/extractor.cpp:33-78), gabor_filter(corrected2 / 255) (src/poppy.hpp:119-122,
// src/util.cpp:40-60) and dft_detail2 (src/experiments.hpp:267-318).  Kernel tables that the reference computes with libm
// in double (Gaussian taps, Gabor kernels, the radial gradient) are computed here the same way and uploaded.
#include "foreground.h"
#include "kernels_prefilter.h"
#include "dft_exact.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace poppy_hip {

#define F2_CHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } \
    } while (0)

// cv::getGaussianKernel(n, sigma, CV_32F) for n > 7 (OCV/imgproc/src/smooth.dispatch.cpp:81-221): exp(-x^2 / 2 sigma^2)
// normalised by the sum, evaluated in double, stored as float (checked against the reference's taps, tests/golden a_*: gauss17)
static void gaussian_taps(int n, double sigma, std::vector<float>& out) {
    std::vector<double> v(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; v[i] = std::exp(-(x * x) / (2 * sigma * sigma)); sum += v[i]; }
    out.resize(n);
    for (int i = 0; i < n; ++i) out[i] = (float)(v[i] / sum);
}

// cv::getGaborKernel(Size(ks, ks), sigma, theta, lambd, gamma, psi, CV_32F)  (OCV/imgproc/src/gabor.cpp:50-95)
static void gabor_kernel(int ks, double sigma, double theta, double lambd, double gamma, double psi, float* out) {
    const double sigma_x = sigma, sigma_y = sigma / gamma;
    const double c = std::cos(theta), s = std::sin(theta);
    const int xmax = ks / 2, ymax = ks / 2, xmin = -xmax, ymin = -ymax;
    const double scale = 1, ex = -0.5 / (sigma_x * sigma_x), ey = -0.5 / (sigma_y * sigma_y), cscale = M_PI * 2 / lambd;
    for (int y = ymin; y <= ymax; ++y)
        for (int x = xmin; x <= xmax; ++x) {
            const double xr = x * c + y * s, yr = -x * s + y * c;
            const double v = scale * std::exp(ex * xr * xr + ey * yr * yr) * std::cos(cscale * xr + psi);
            out[(ymax - y) * ks + (xmax - x)] = (float)v;
        }
}
// gabor_filter's bank: theta_i = i * float(180 / numAngles) used as radians (src/util.cpp:40-47)
void gabor_bank(int ks, double sigma, double lambd, double gamma, double psi, std::vector<float>& bank) {
    bank.resize((size_t)16 * ks * ks);
    const float step = (float)(180 / 16);
    for (int i = 0; i < 16; ++i) gabor_kernel(ks, sigma, (double)(i * step), lambd, gamma, psi, &bank[(size_t)i * ks * ks]);
}

// draw_radial_gradiant (src/draw.cpp:21-38) + convertTo(CV_32F, 1/255) (src/extractor.cpp:181-183): pow(sin(sin(d pi/2) pi/2), 12) as float ->
// normalize(0, 255, NORM_MINMAX, CV_8U), i.e. convertTo(CV_8U, 255 / (max - min), -min * scale) with both doubles applied as floats
// (convert_scale.simd.hpp cvt_32f: v * a + b unfused, cvRound, saturate) -> bitwise_not -> / 255.  Rows on host threads (libm pow / sin).
// No reference-run fixture pins this option; the primitives are those of draw_radial_gradiant2 below, which one does.
void radial_mask(int width, int height, std::vector<float>& out) {
    const int ccx = (int)(width / 2.0), ccy = (int)(height / 2.0);
    const double maxDist = std::hypot(width / 2.0, height / 2.0);
    std::vector<float> g((size_t)width * height);
    const int nt = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
    std::vector<float> mns(nt, INFINITY), mxs(nt, -INFINITY);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
            float mn = INFINITY, mx = -INFINITY;
            for (int row = t; row < height; row += nt)
                for (int col = 0; col < width; ++col) {
                    const double dist = std::hypot((double)(ccx - col), (double)(ccy - row)) / maxDist;
                    const float v = (float)std::pow(std::sin(std::sin(dist * (M_PI / 2)) * (M_PI / 2)), 12);
                    g[(size_t)row * width + col] = v;
                    mn = std::min(mn, v); mx = std::max(mx, v);
                }
            mns[t] = mn; mxs[t] = mx;
        });
    for (auto& t : th) t.join();
    const double smin = *std::min_element(mns.begin(), mns.end()), smax = *std::max_element(mxs.begin(), mxs.end());
    const double scale = (255.0 - 0.0) * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0), shift = 0.0 - smin * scale;
    const float fs = (float)scale, fb = (float)shift;
    out.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        int v = (int)std::nearbyintf(g[i] * fs + fb);
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        const uint8_t inv = (uint8_t)~(uint8_t)v;
        out[i] = (float)inv * (float)(1.0 / 255.0) + 0.f;
    }
}

// draw_radial_gradiant2 (src/draw.cpp:40-59): pow(sin(sin(d pi/2) pi/2), 32) -> u8 -> min-max normalise -> invert -> / 255
void radial_gradient(int width, int height, std::vector<float>& out) {
    // cv::Point center(width / 2.0, height / 2.0): the doubles convert to the int members by truncation
    const int ccx = (int)(width / 2.0), ccy = (int)(height / 2.0);
    const double maxDist = std::hypot(width / 2.0, height / 2.0);
    std::vector<uint8_t> g8((size_t)width * height);
    int mn = 255, mx = 0;
    for (int row = 0; row < height; ++row)
        for (int col = 0; col < width; ++col) {
            const double dist = std::hypot((double)(ccx - col), (double)(ccy - row)) / maxDist;
            const float f = (float)std::pow(std::sin(std::sin(dist * (M_PI / 2)) * (M_PI / 2)), 32);
            const float t = f * 255.f + 0.f;
            int v = (int)std::nearbyintf(t);
            v = v < 0 ? 0 : v > 255 ? 255 : v;
            g8[(size_t)row * width + col] = (uint8_t)v;
            mn = std::min(mn, v); mx = std::max(mx, v);
        }
    // cv::normalize(grad, grad, 0, 255, NORM_MINMAX) on 8 bit: convertTo(u8, scale, shift) in float
    const double scale = (255.0 - 0.0) * (mx - mn > 2.220446049250313e-16 ? 1. / (mx - mn) : 0), shift = 0.0 - mn * scale;
    const float fs = (float)scale, fb = (float)shift;
    out.resize(g8.size());
    for (size_t i = 0; i < g8.size(); ++i) {
        int v = (int)std::nearbyintf((float)g8[i] * fs + fb);
        v = v < 0 ? 0 : v > 255 ? 255 : v;
        const uint8_t inv = (uint8_t)~(uint8_t)v;                              // bitwise_not
        out[i] = (float)inv * (float)(1.0 / 255.0) + 0.f;
    }
}

int ForegroundFilter::ensure2(int w, int h) {
    if (w == W2 && h == H2) return 0;
    void* bufs[] = {f_a, f_b, f_c, f_d, radial, bank31, bank13, fft31, fft13, doubt_list31, doubt_list13, taps17, spec, mag, minmax, powsum, g_tmp, g_out[0], g_out[1], c3_in, c3_out};
    for (void* b : bufs) if (b) (void)hipFree(b);
    f_a = f_b = f_c = f_d = radial = taps17 = mag = c3_in = c3_out = nullptr; bank31 = bank13 = fft31 = fft13 = nullptr; doubt_list31 = doubt_list13 = nullptr; spec = nullptr;
    minmax = nullptr; powsum = nullptr; g_tmp = g_out[0] = g_out[1] = nullptr;
    for (void* b : {(void*)spec_tmp, (void*)spec_out, (void*)d_itab[0], (void*)d_itab[1], (void*)d_wave[0], (void*)d_wave[1]}) if (b) (void)hipFree(b);
    spec_tmp = spec_out = nullptr; d_itab[0] = d_itab[1] = nullptr; d_wave[0] = d_wave[1] = nullptr;
    const size_t P = (size_t)w * h;
    F2_CHK(hipMalloc((void**)&f_a, P * 4)); F2_CHK(hipMalloc((void**)&f_b, P * 4)); F2_CHK(hipMalloc((void**)&f_c, P * 4)); F2_CHK(hipMalloc((void**)&f_d, P * 4));
    F2_CHK(hipMalloc((void**)&radial, P * 4)); F2_CHK(hipMalloc((void**)&g_tmp, P)); F2_CHK(hipMalloc((void**)&g_out[0], P)); F2_CHK(hipMalloc((void**)&g_out[1], P));
    F2_CHK(hipMalloc((void**)&c3_in, P * 12)); F2_CHK(hipMalloc((void**)&c3_out, P * 12));
    std::vector<float> t17, b31, b13, rad;
    gaussian_taps(17, 2.0, t17);
    gabor_bank(31, 5, 2, 0.04, M_PI / 4, b31);                // Extractor::keypoints (src/extractor.cpp:63-64)
    gabor_bank(13, 5, 10, 0.04, M_PI / 4, b13);               // gabor_filter defaults (src/util.hpp:95)
    radial_gradient(w, h, rad);
    // the float taps, widened (exact) and regrouped [tap][orientation]: the 16 weights of a tap are one 128-byte scalar fetch
    auto by_tap = [](const std::vector<float>& b, int ks) {
        std::vector<double> o(b.size());
        const int n = ks * ks;
        for (int k = 0; k < 16; ++k) for (int t = 0; t < n; ++t) o[(size_t)t * 16 + k] = b[(size_t)k * n + t];
        return o;
    };
    const std::vector<double> b31d = by_tap(b31, 31), b13d = by_tap(b13, 13);
    F2_CHK(hipMalloc((void**)&taps17, 17 * 4)); F2_CHK(hipMalloc((void**)&bank31, b31d.size() * 8)); F2_CHK(hipMalloc((void**)&bank13, b13d.size() * 8));
    F2_CHK(hipMemcpy(taps17, t17.data(), 17 * 4, hipMemcpyHostToDevice));
    F2_CHK(hipMemcpy(bank31, b31d.data(), b31d.size() * 8, hipMemcpyHostToDevice));
    F2_CHK(hipMemcpy(bank13, b13d.data(), b13d.size() * 8, hipMemcpyHostToDevice));
    // the FFT form of the banks (POPPY_GABOR_DIRECT keeps the direct double sums)
    static const bool direct_only = getenv("POPPY_GABOR_DIRECT") != nullptr;
    if (!direct_only) {
        static const std::vector<double> t31 = gabor_fft_tables(b31, 31), t13 = gabor_fft_tables(b13, 13);      // geometry-independent: once per process
        if (!gabor_fft_prepare()) { err = "gabor_fft_prepare failed"; return -2; }
        F2_CHK(hipMalloc((void**)&fft31, t31.size() * 8)); F2_CHK(hipMalloc((void**)&fft13, t13.size() * 8));
        F2_CHK(hipMalloc((void**)&doubt_list31, (2 + P) * 4)); F2_CHK(hipMalloc((void**)&doubt_list13, (2 + P * 3) * 4));
        // on the device each pair's spectrum lies transposed ([column][row]): the kernel holds the patch spectrum as row `lane`, columns 8 g + i
        auto transposed = [](const std::vector<double>& t) {
            std::vector<double> o(t.size());
            for (int j = 0; j < 8; ++j) for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c)
                for (int k = 0; k < 2; ++k) o[(((size_t)j * 64 + c) * 64 + r) * 2 + k] = t[(((size_t)j * 64 + r) * 64 + c) * 2 + k];
            return o;
        };
        static const std::vector<double> d31 = transposed(t31), d13 = transposed(t13);
        F2_CHK(hipMemcpy(fft31, d31.data(), d31.size() * 8, hipMemcpyHostToDevice));
        F2_CHK(hipMemcpy(fft13, d13.data(), d13.size() * 8, hipMemcpyHostToDevice));
    }
    F2_CHK(hipMemcpy(radial, rad.data(), P * 4, hipMemcpyHostToDevice));
    dftN = dft_optimal_size(w); dftM = dft_optimal_size(h);
    F2_CHK(hipMalloc((void**)&spec, (size_t)dftN * dftM * 8)); F2_CHK(hipMalloc((void**)&mag, (size_t)dftN * dftM * 4));
    F2_CHK(hipMalloc((void**)&minmax, 8)); F2_CHK(hipMalloc((void**)&powsum, 8));
    F2_CHK(hipMalloc((void**)&spec_tmp, (size_t)dftN * dftM * 8)); F2_CHK(hipMalloc((void**)&spec_out, (size_t)dftN * dftM * 8));
    for (int d = 0; d < 2; ++d) {                             // 0: rows (length N), 1: columns (length M)
        DftPlanHost ph;
        dft_make_plan(d ? dftM : dftN, ph);
        if (ph.nf > 16) { err = "DFT size with too many factors"; return -2; }
        F2_CHK(hipMalloc((void**)&d_itab[d], ph.itab.size() * 4)); F2_CHK(hipMalloc((void**)&d_wave[d], ph.wave.size() * 4));
        F2_CHK(hipMemcpy(d_itab[d], ph.itab.data(), ph.itab.size() * 4, hipMemcpyHostToDevice));
        F2_CHK(hipMemcpy(d_wave[d], ph.wave.data(), ph.wave.size() * 4, hipMemcpyHostToDevice));
        dft_n[d] = ph.n; dft_nf[d] = ph.nf;
        for (int k = 0; k < 16; ++k) dft_factors[d][k] = k < ph.nf ? ph.factors[k] : 0;
    }
    W2 = w; H2 = h;
    return 0;
}

void ForegroundFilter::release2() {
    void* bufs[] = {f_a, f_b, f_c, f_d, radial, bank31, bank13, fft31, fft13, doubt_list31, doubt_list13, taps17, spec, mag, minmax, powsum, g_tmp, g_out[0], g_out[1], c3_in, c3_out};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (void* b : {(void*)spec_tmp, (void*)spec_out, (void*)d_itab[0], (void*)d_itab[1], (void*)d_wave[0], (void*)d_wave[1]}) if (b) (void)hipFree(b);
    W2 = H2 = 0;
}

// dft_detail2(goodFeatures): RMS of the first `cols` bytes of every row of the normalised, centred log spectrum
// dft_detail2 in two halves: detail_begin queues everything on the stream (no host round trip: the normalisation's scale and shift are formed on the
// device, the sum of squares lands in pinned memory behind an event) so that the caller can queue the ORB input's kernels behind it at once;
// detail_end waits for the event and finishes the value.
int ForegroundFilter::detail_begin(const uint8_t* d_gf, int w, int h, hipStream_t s) {
    if (ensure(w, h) || ensure2(w, h)) return -2;
    const int N = dftN, M = dftM, Nc = N & -2, Mc = M & -2;
    launch_pad_complex(d_gf, (float2*)spec, w, h, N, M, minmax, powsum, s);
    DftPlanDev pr, pc;
    pr.n = dft_n[0]; pr.nf = dft_nf[0]; pr.itab = d_itab[0]; pr.wave = (const float2*)d_wave[0];
    pc.n = dft_n[1]; pc.nf = dft_nf[1]; pc.itab = d_itab[1]; pc.wave = (const float2*)d_wave[1];
    for (int k = 0; k < 16; ++k) { pr.factors[k] = dft_factors[0][k]; pc.factors[k] = dft_factors[1][k]; }
    launch_dft2d_exact((const float2*)spec, (float2*)spec_tmp, (float2*)spec_out, N, M, pr, pc, s);
    launch_spectrum_log((const float2*)spec_out, mag, logtab, minmax, N, M, Nc, Mc, s);
    launch_spectrum_bytes(mag, minmax, Nc, Mc, powsum, s);
    if (!detail_ev) F2_CHK(hipEventCreateWithFlags(&detail_ev, hipEventDisableTiming));
    F2_CHK(hipMemcpyAsync(h_easy + 2, powsum, 8, hipMemcpyDeviceToHost, s));      // (pinned: 16 bytes, the first word belongs to the medians' count)
    F2_CHK(hipEventRecord(detail_ev, s));
    detail_px = (double)Nc * (double)Mc;
    return 0;
}
int ForegroundFilter::detail_end(double* out) {
    if (!detail_ev) { err = "detail_end without detail_begin"; return -1; }
    F2_CHK(hipEventSynchronize(detail_ev));
    unsigned long long ps;
    memcpy(&ps, h_easy + 2, 8);
    *out = std::sqrt((double)ps / detail_px);
    return 0;
}
int ForegroundFilter::detail(const uint8_t* d_gf, int w, int h, hipStream_t s, double* out) {
    const int rc = detail_begin(d_gf, w, h, s);
    return rc ? rc : detail_end(out);
}

// the ORB input image of Extractor::keypoints for one goodFeatures image (device in, device out); which = 0 / 1 selects the output buffer
const uint8_t* ForegroundFilter::orb_input(const uint8_t* d_gf, int w, int h, int which, hipStream_t s, float* h_us, float* h_gb) {
    if (ensure(w, h) || ensure2(w, h)) return nullptr;
    const int n = w * h;
    launch_unsharp1_gray(d_gf, f_a, f_b, f_c, f_d, taps17, w, h, s);       // f_d = us (grey of the unsharp-masked triple)
    if (fft31 && !gabor_direct) launch_gabor_fft31(f_d, fft31, bank31, doubt_list31, f_a, w, h, s);                 // f_a = gabor mean
    else launch_gabor_bank31(f_d, bank31, f_a, w, h, s);
    if (h_us) (void)hipMemcpyAsync(h_us, f_d, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    if (h_gb) (void)hipMemcpyAsync(h_gb, f_a, (size_t)n * 4, hipMemcpyDeviceToHost, s);
    launch_orb_input(f_a, f_d, radial, g_tmp, hist, lut, g_out[which & 1], n, s);
    if (hipGetLastError() != hipSuccess) { err = "orb_input launch failed"; return nullptr; }
    return g_out[which & 1];
}

// gabor_filter(corrected2 / 255) with the default arguments: 3-channel float field (device in u8 BGR packed, device out f32x3)
// err_out: a caller on a SECOND thread (the pair set-up queues gabor2 from the first image's thread while the second image's thread uses this object's
// chain) passes its own string; the buffers must then be in place already (prepare / prepare2) and no member is written.
const float* ForegroundFilter::gabor_field(const uint8_t* d_bgr_packed, int w, int h, hipStream_t s, std::string* err_out) {
    if (err_out) {
        if (w != W || h != H || w != W2 || h != H2) { *err_out = "gabor_field: buffers not prepared for this size"; return nullptr; }
    } else if (ensure(w, h) || ensure2(w, h)) return nullptr;
    launch_u8_to_f32(d_bgr_packed, c3_in, w * h * 3, s);
    if (fft13 && !gabor_direct) launch_gabor_fft13_c3(c3_in, fft13, bank13, doubt_list13, c3_out, w, h, s);
    else launch_gabor_bank13_c3(c3_in, bank13, c3_out, w, h, s);
    if (hipGetLastError() != hipSuccess) { (err_out ? *err_out : err) = "gabor_field launch failed"; return nullptr; }
    return c3_out;
}

}  // namespace poppy_hip
