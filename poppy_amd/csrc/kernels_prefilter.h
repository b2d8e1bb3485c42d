// kernels_prefilter.h — launch wrappers of kernels_prefilter.hip (Extractor::foreground, src/extractor.cpp:136-229).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>

// Switches that can change a RESULT BIT (parts of a kernel left out for timing, the Gabor transform without its exactness hand-over) exist only
// in a build made with -DPOPPY_EXPERIMENTS (python -m poppy_amd.build --experiments): the shipped library ignores these variables, so a stray
// environment variable on a user's box cannot break parity.  The tuning variables read with plain getenv never change a result (include/poppy_hip.h lists them).
inline const char* poppy_experiment_env(const char* name) {
#ifdef POPPY_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace poppy_hip {

// cvtColor(BGR2GRAY) on 8-bit pixels; `stride` = bytes per source row
void launch_bgr2gray(const uint8_t* bgr, size_t stride, uint8_t* gray, int w, int h, hipStream_t s);

// One BackgroundSubtractorMOG2::apply on a 1-channel 8-bit image plus the accumulation fgMask += flow * acc_scale.
// Model: gw, gv, gm = weight, variance, mean as [5][n_px] floats, used = [n_px] bytes, all zero before the first frame.
// alphaT = (float)(1 / min(2 * nframes, 500)), prune = (float)(-alphaT_double * 0.05f).
constexpr int kMog2Steps = 13;         // Extractor::foreground: one apply on the grey image, twelve on its progressive medians
// flows: steps planes of n_px mask bytes (0 / 127 / 255); alphaT / prune: per-step learning rate and -rate * 0.05 (CT)
void launch_mog2_all(const uint8_t* const* imgs, const float* alphaT, const float* prune, int steps, uint8_t* flows, int n_px, hipStream_t s);
void launch_acc_flow(uint8_t* acc, const uint8_t* flow, int n_px, float acc_scale, hipStream_t s, bool first = false);   // first: acc counts as zeros and is only written   // acc += sat(round(flow * scale))

// medianBlur(src, dst, ksize) for odd ksize >= 3 on a 1-channel 8-bit image (replicated border, exact median)
bool prepare_median_u8();               // raises the kernel's LDS limit once; call outside stream capture
// padded_tmp: median_padded_bytes(w, h) bytes of scratch (the source with replicated side columns)
size_t median_padded_bytes(int w, int h);
void launch_median_u8(const uint8_t* src, uint8_t* padded_tmp, uint8_t* dst, int w, int h, int ksize, hipStream_t s);
// The same in two parts, for chains of medians: launch_pad_cols makes the padded copy of a source; launch_median_padded filters a padded source into
// dst (tight rows) and, when padded_next is not null, ALSO writes the result as the padded source of the next median (one launch per link of the chain).
void launch_pad_cols(const uint8_t* src, uint8_t* padded, int w, int h, hipStream_t s);
void launch_median_padded(const uint8_t* padded_src, uint8_t* dst, uint8_t* padded_next, int w, int h, int ksize, hipStream_t s);
// The same filter by column histograms (kernels_median_cols.hip; the reference's own scheme for these windows, median_blur.simd.hpp:84-346): the cost
// of a pixel does not depend on ksize.  The image is cut into tiles of median_cols_geom(); pres_in (may be null) = the 256-bit presence map of every
// tile of the SOURCE (median_presence_words() dwords: launch_median_presence, or the pres_out of the median that made the source): tiles whose
// footprint holds at most 64 different values are filtered on the values' ranks, at a third of the cost.  pres_out (may be null): the map of dst.
struct MedianColsGeom { int run, rows, tiles_x, tiles_y; };
MedianColsGeom median_cols_geom(int w, int h);
size_t median_presence_words(int w, int h);
// easy_or_null: a device counter (zeroed by the caller) of the tiles that hold few different values: nine tenths of an image's tiles -> the column-histogram form pays
void launch_median_presence(const uint8_t* src_tight, uint32_t* pres, int w, int h, uint32_t* easy_or_null, hipStream_t s);
int median_cols_hint_from_host(const uint8_t* bgr, size_t stride, int w, int h);      // the same decision from a sparse sample of a host image (1 / 0)
void launch_median_cols(const uint8_t* padded_src, uint8_t* dst, uint8_t* padded_next, int w, int h, int ksize, const uint32_t* pres_in, uint32_t* pres_out,
                        int force, hipStream_t s);             // force: 0 = by content, 1 = every tile by windows of ranks, 2 = no tile with one count per lane; + 4 / 8: see k_median_cols
int median_cols_min_ksize();            // windows from this size on take the column-histogram form (POPPY_MED_COLS_MIN overrides) on images of few values per tile
int median_cols_min_ksize_hard();       // ... on every other image (POPPY_MED_COLS_MIN_HARD overrides)

// GaussianBlur(src, dst, 23x23, sigma 1) on 8 bit (the fixed-point path); tmp = w*h uint16
void launch_gauss23_u8(const uint8_t* src, uint16_t* tmp, uint8_t* dst, int w, int h, hipStream_t s);
// dst = GaussianBlur23(acc + sat(round(flow * scale))) in one launch (acc is not modified)
void launch_acc_gauss23(const uint8_t* acc, const uint8_t* flow, uint8_t* dst, int w, int h, float acc_scale, hipStream_t s);
// S = 1, 2, 3, 4 or 6 of those steps in one launch (flows: the first step's mask plane, the following ones `plane` bytes apart), for the
// geometries acc_gauss23_fused_takes() accepts (width divisible by 4, at least 256 x 64; 4-byte aligned planes)
bool acc_gauss23_fused_takes(int w, int h);
void launch_acc_gauss23_fused(const uint8_t* acc, const uint8_t* flows, size_t plane, int steps, uint8_t* dst, int w, int h, float acc_scale, hipStream_t s);

// mask = log(fg/255*19 + 1)/log(20); masked = u8(grey/255 * mask * 255); out = equalizeHist(masked).
// d_logtab: 512 floats (see foreground.cpp); hist: 256 unsigned; lut: 256 bytes; dbg (optional): 3*n_px floats lin, logged, mask;
// radial (optional, n_px floats): the radial mask the foreground mask is multiplied with first (Settings::enable_radial_mask).
void launch_fg_tail(const uint8_t* grey, const uint8_t* fg, const float* d_logtab, const float* radial_or_null, uint8_t* masked, unsigned* hist, uint8_t* lut,
                    uint8_t* out, float* dbg_or_null, int n_px, hipStream_t s);

// GaussianBlur (8-bit fixed-point path, n taps in 8.8) of the strip (rx, ry, rw, rh) of a 3-channel canvas of width cw, read from
// `canvas`, written to the same place in `out`; tmp: rw*rh*3 uint32
void launch_strip_blur(const uint8_t* canvas, uint8_t* out, int cw, uint32_t* tmp, const int* d_taps, int n, int rx, int ry, int rw, int rh, hipStream_t s);

// equalizeHist given the 256-bin histogram of src (lut: 256 bytes of scratch)
void launch_equalize_from_hist(const uint8_t* src, const unsigned* hist, uint8_t* lut, uint8_t* out, int n_px, hipStream_t s);

// ---- part 2 (kernels_prefilter2.hip) ----------------------------------------------------------------------------------
// grey(unsharp_mask(triple(gf / 255), 2, 6, 0.1)) on one channel; f32, tmp, diff, out: w*h floats; d_taps17: getGaussianKernel(17, 2)
void launch_unsharp1_gray(const uint8_t* gf, float* f32, float* tmp, float* diff, float* out, const float* d_taps17, int w, int h, hipStream_t s);
// gabor_filter(src, dst, 16 angles, ksize, ...): bank = [16][ksize*ksize] kernel taps (the float taps widened to double);
// 31x31 on one channel, 13x13 on three
void launch_gabor_bank31(const float* src, const double* d_bank, float* dst, int w, int h, hipStream_t s);
void launch_gabor_bank13_c3(const float* src, const double* d_bank, float* dst, int w, int h, hipStream_t s);
// The same banks by tiled 64 x 64 double-precision FFTs (kernels_gabor_fft.hip).  gabor_fft_tables: the paired, conjugated kernel
// spectra of a bank ([orientation][ks * ks] floats), 8 x 4096 complex doubles, computed on the host; gabor_fft_prepare: per device.
std::vector<double> gabor_fft_tables(const std::vector<float>& bank, int ks);
// d_list (2 + w * h * channels words): the pixels with a plane value too close to a float rounding boundary for the transform to decide; the
// launch re-forms them as the direct sums (d_bank: the direct kernels' table), so both forms give the same bits.
bool gabor_fft_prepare();
bool gabor_fft_doubt(unsigned long long out[3]);
void launch_gabor_fft31(const float* src, const double* d_tables, const double* d_bank, unsigned* d_list, float* dst, int w, int h, hipStream_t s);
void launch_gabor_fft13_c3(const float* src, const double* d_tables, const double* d_bank, unsigned* d_list, float* dst, int w, int h, hipStream_t s);
void launch_u8_to_f32(const uint8_t* src, float* dst, int n, hipStream_t s);
// out = equalizeHist(u8(gb * us * radial * 255))
void launch_orb_input(const float* gb, const float* us, const float* radial, uint8_t* tmp_u8, unsigned* hist, uint8_t* lut, uint8_t* out,
                      int n_px, hipStream_t s);
// dft_detail2 around the 2-D DFT: zero-padded complex input; log-magnitude + min/max (as order-preserving uints) over the
// even-cropped nc x mc window; sum of squared bytes of the first nc bytes of every row of the swapped, normalised image
// cv::dft's complex float transform (exact operation order, dft_exact.h builds the tables); 2-D forward: rows then columns
struct DftPlanDev { int n, nf; int factors[16]; const int* itab; const float2* wave; };
void launch_dft2d_exact(const float2* src, float2* tmp, float2* dst, int n, int m, const DftPlanDev& rows, const DftPlanDev& cols, hipStream_t s);
void launch_pad_complex(const uint8_t* src, float2* dst, int w, int h, int n, int m, unsigned* minmax, unsigned long long* powsum, hipStream_t s);   // also resets the two reductions
void launch_spectrum_log(const float2* spec, float* mag, const float* d_logtab, unsigned* minmax, int n, int m, int nc, int mc, hipStream_t s);
void launch_spectrum_bytes(const float* mag, const unsigned* minmax, int nc, int mc, unsigned long long* powsum, hipStream_t s);

}  // namespace poppy_hip
