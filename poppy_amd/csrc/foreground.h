// foreground.h — host side of Extractor::foreground on the GPU (src/extractor.cpp:136-229): buffers and the sequence
// of launches for one image.  Used by the C ABI (poppy_hip_foreground).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>

namespace poppy_hip {

struct ForegroundDebugOut {          // host pointers, each may be null
    uint8_t* grey = nullptr;         // w*h
    uint8_t* stages = nullptr;       // 50 planes of w*h: flow0, acc0, then 12 x (med, flow, acc, blur)
    float* floats = nullptr;         // 3 planes of w*h: lin, logged, finalMask
    uint8_t* masked = nullptr;       // w*h
};

class ForegroundFilter {
public:
    ~ForegroundFilter();
    // bgr: host image (8UC3, `stride` bytes per row); out: host w*h bytes (goodFeatures).  Returns 0, -1 (arguments) or -2 (device).
    int run(const uint8_t* bgr, size_t stride, int w, int h, hipStream_t s, uint8_t* out, const ForegroundDebugOut* dbg);
    // same, from / to device memory (bgr device pointer with `stride`), result left in device memory and returned
    const uint8_t* run_device(const uint8_t* d_bgr, size_t stride, int w, int h, hipStream_t s, const ForegroundDebugOut* dbg);
    std::string err;

private:
    int ensure(int w, int h);
    void release();
    int W = 0, H = 0;
    uint8_t *d_bgr = nullptr, *grey = nullptr, *img[2] = {nullptr, nullptr}, *acc[2] = {nullptr, nullptr}, *flow = nullptr;
    uint8_t *used = nullptr, *masked = nullptr, *out = nullptr, *lut = nullptr;
    uint16_t* tmp16 = nullptr;
    uint8_t* padded = nullptr;           // median source with replicated side columns
    float *gw = nullptr, *gv = nullptr, *gm = nullptr, *logtab = nullptr, *dbgf = nullptr;
    unsigned* hist = nullptr;
    bool prepared = false;
};

}  // namespace poppy_hip
