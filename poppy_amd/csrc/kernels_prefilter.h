// kernels_prefilter.h — launch wrappers of kernels_prefilter.hip (Extractor::foreground, src/extractor.cpp:136-229).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace poppy_hip {

// cvtColor(BGR2GRAY) on 8-bit pixels; `stride` = bytes per source row
void launch_bgr2gray(const uint8_t* bgr, size_t stride, uint8_t* gray, int w, int h, hipStream_t s);

// One BackgroundSubtractorMOG2::apply on a 1-channel 8-bit image plus the accumulation fgMask += flow * acc_scale.
// Model: gw, gv, gm = weight, variance, mean as [5][n_px] floats, used = [n_px] bytes, all zero before the first frame.
// alphaT = (float)(1 / min(2 * nframes, 500)), prune = (float)(-alphaT_double * 0.05f).
void launch_mog2(const uint8_t* img, float* gw, float* gv, float* gm, uint8_t* used, uint8_t* flow_or_null, uint8_t* acc,
                 int n_px, float alphaT, float prune, float acc_scale, hipStream_t s);

// medianBlur(src, dst, ksize) for odd ksize >= 3 on a 1-channel 8-bit image (replicated border, exact median)
bool prepare_median_u8();               // raises the kernel's LDS limit once; call outside stream capture
// padded_tmp: median_padded_bytes(w, h) bytes of scratch (the source with replicated side columns)
size_t median_padded_bytes(int w, int h);
void launch_median_u8(const uint8_t* src, uint8_t* padded_tmp, uint8_t* dst, int w, int h, int ksize, hipStream_t s);

// GaussianBlur(src, dst, 23x23, sigma 1) on 8 bit (the fixed-point path); tmp = w*h uint16
void launch_gauss23_u8(const uint8_t* src, uint16_t* tmp, uint8_t* dst, int w, int h, hipStream_t s);

// mask = log(fg/255*19 + 1)/log(20); masked = u8(grey/255 * mask * 255); out = equalizeHist(masked).
// d_logtab: 512 floats (see foreground.cpp); hist: 256 unsigned; lut: 256 bytes; dbg (optional): 3*n_px floats lin, logged, mask.
void launch_fg_tail(const uint8_t* grey, const uint8_t* fg, const float* d_logtab, uint8_t* masked, unsigned* hist, uint8_t* lut,
                    uint8_t* out, float* dbg_or_null, int n_px, hipStream_t s);

}  // namespace poppy_hip
