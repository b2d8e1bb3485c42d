// foreground.h — host side of Extractor::foreground on the GPU (src/extractor.cpp:136-229): buffers and the sequence
// of launches for one image.  Used by the C ABI (poppy_hip_foreground).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace poppy_hip {

struct ForegroundDebugOut {          // host pointers, each may be null
    uint8_t* grey = nullptr;         // w*h
    uint8_t* stages = nullptr;       // 50 planes of w*h: flow0, acc0, then 12 x (med, flow, acc, blur)
    float* floats = nullptr;         // 3 planes of w*h: lin, logged, finalMask
    uint8_t* masked = nullptr;       // w*h
};

// draw_radial_gradiant2 (src/draw.cpp:40-59), host
void radial_gradient(int width, int height, std::vector<float>& out);
// the 16 kernels of gabor_filter's bank, [orientation][ks * ks] floats (src/util.cpp:31-61, OCV/imgproc/src/gabor.cpp:50-95)
void gabor_bank(int ks, double sigma, double lambd, double gamma, double psi, std::vector<float>& bank);
// draw_radial_gradiant as Extractor::foreground uses it under Settings::enable_radial_mask (src/draw.cpp:21-38, src/extractor.cpp:178-185), host
void radial_mask(int width, int height, std::vector<float>& out);

class ForegroundFilter {
public:
    ~ForegroundFilter();
    // bgr: host image (8UC3, `stride` bytes per row); out: host w*h bytes (goodFeatures).  Returns 0, -1 (arguments) or -2 (device).
    int run(const uint8_t* bgr, size_t stride, int w, int h, hipStream_t s, uint8_t* out, const ForegroundDebugOut* dbg);
    // same, from / to device memory (bgr device pointer with `stride`), result left in device memory and returned
    const uint8_t* run_device(const uint8_t* d_bgr, size_t stride, int w, int h, hipStream_t s, const ForegroundDebugOut* dbg);
    // --- part 2 (foreground2.cpp): everything between goodFeatures and the ORB input, gabor2, dft_detail2 ---------------
    int detail(const uint8_t* d_gf, int w, int h, hipStream_t s, double* out);
    int detail_begin(const uint8_t* d_gf, int w, int h, hipStream_t s);      // the same in two halves: everything queued, no host round trip ...
    int detail_end(double* out);                                             // ... and the value, once the stream has passed it
    const uint8_t* orb_input(const uint8_t* d_gf, int w, int h, int which, hipStream_t s, float* h_us = nullptr, float* h_gb = nullptr);
    const float* gabor_field(const uint8_t* d_bgr_packed, int w, int h, hipStream_t s, std::string* err_out = nullptr);
    uint8_t* bgr_staging() { return d_bgr; }
    // medianBlur(src, ksize) on a host image by one of the two kernels (diagnostics / tests): form 1 = a lane per column (k_median_u8), 2 = column
    // histograms with the presence map, 3 = column histograms, every tile on all 256 values, 4 = with the map but no tile on 64 ranks; 0 = what the chain
    // would take for this ksize
    int median(const uint8_t* src, int w, int h, int ksize, int form, hipStream_t s, uint8_t* dst);         // w*h*3 bytes once ensure() ran
    int prepare(int w, int h) { return ensure(w, h); }
    int prepare2(int w, int h) { return ensure2(w, h); }      // the buffers of part 2 (a caller that uses the object from two threads allocates first)
    std::string err;

    hipEvent_t medians_done = nullptr;   // when set, run_device records it on its stream behind the last median (the pair set-up starts gabor2 there)
    // Which kernel the chain's medians take from median_cols_min_ksize() on: 1 = column histograms (images of few values per neighbourhood), 0 = a lane per
    // column, -1 = ask the device (a presence pass over the grey image and a 4-byte read-back).  Set per image by the caller that holds host pixels
    // (median_cols_hint_from_host); reset to -1 by run_device.  POPPY_MED_COLS_FORCE = 0 / 1 overrides.  Every choice gives the same bytes.
    int median_cols_hint = -1;
    bool radial_mask_on = false;         // Settings::enable_radial_mask (src/extractor.cpp:178-197): set before the first run of a geometry or any time after
    bool gabor_direct = false;           // run the Gabor banks as direct double sums even when the FFT spectra exist (tests compare the two)
private:
    int ensure(int w, int h);
    int ensure2(int w, int h);
    void release();
    void release2();
    int W2 = 0, H2 = 0, dftN = 0, dftM = 0;
    float *f_a = nullptr, *f_b = nullptr, *f_c = nullptr, *f_d = nullptr, *radial = nullptr, *taps17 = nullptr;
    double *bank31 = nullptr, *bank13 = nullptr;
    double *fft31 = nullptr, *fft13 = nullptr;      // the banks' paired kernel spectra (kernels_gabor_fft.hip), null: direct sums only
    unsigned *doubt_list31 = nullptr, *doubt_list13 = nullptr;   // pixels the FFT form of a bank hands to the direct sums (2 + P, 2 + 3 P words); one per bank: gabor2 runs on another stream beside the image's own chain
    float *mag = nullptr, *c3_in = nullptr, *c3_out = nullptr;
    void* spec = nullptr;
    unsigned* minmax = nullptr;
    unsigned long long* powsum = nullptr;
    uint8_t *g_tmp = nullptr, *g_out[2] = {nullptr, nullptr};
    void *spec_tmp = nullptr, *spec_out = nullptr;
    int* d_itab[2] = {nullptr, nullptr};
    float* d_wave[2] = {nullptr, nullptr};
    int dft_n[2] = {0, 0}, dft_nf[2] = {0, 0}, dft_factors[2][16] = {};
    int W = 0, H = 0;
    uint8_t *d_bgr = nullptr, *grey = nullptr, *meds = nullptr, *acc[2] = {nullptr, nullptr}, *flows = nullptr;
    uint8_t *masked = nullptr, *out = nullptr, *lut = nullptr;
    uint16_t* tmp16 = nullptr;
    uint8_t* padded = nullptr;           // median source with replicated side columns
    float *logtab = nullptr, *dbgf = nullptr, *radial12 = nullptr;      // radial12: the radial mask of the current geometry, built on first use
    unsigned* hist = nullptr;
    uint32_t* med_pres = nullptr;        // two presence maps (kernels_median_cols.hip): the current median's source and result; then the counter of easy tiles
    uint32_t* h_easy = nullptr;          // pinned: that counter read back
    hipEvent_t easy_ev = nullptr;        // ... when this event has passed
    hipEvent_t detail_ev = nullptr;      // dft_detail2's sum of squares has landed in h_easy + 2
    double detail_px = 0;
    bool prepared = false;
};

}  // namespace poppy_hip
