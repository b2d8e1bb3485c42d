// warp_device.h — per-pixel pieces of create_map + remap shared by the warp kernels.
//   src/algo.cpp:146-176 (create_map), OCV/imgproc/src/imgwarp.cpp:1197-1234,721-731,808-852 (remap, INTER_LINEAR,
//   BORDER_CONSTANT 0), :213-287 (BilinearTab_i)
#pragma once
#include "pyramid_device.h"

namespace poppy_hip {

// id-map value -> triangle + 1, or 0 when the value is not of this frame (kernels.h: launch_raster)
__device__ __forceinline__ int decode_id(uint32_t raw, uint32_t id_base) {
    const uint32_t d = raw - id_base;
    return d < (1u << 20) ? (int)d : 0;
}

__device__ __forceinline__ void bilinear_weights(int fx, int fy, int& w00, int& w01, int& w10, int& w11) {
    // BilinearTab_i: saturate_cast<short>((1-fy/32)(1-fx/32)*32768) etc.  All products are exact; only the
    // very first entry saturates (32768 -> 32767) and its deficit goes to tap [1][1] (imgwarp.cpp:251-267).
    // operands are < 2^24: 24-bit multiplies are full rate on CDNA, 32-bit ones quarter rate
    w00 = __mul24(32 - fx, 32 - fy) << 5; w01 = __mul24(fx, 32 - fy) << 5;
    w10 = __mul24(32 - fx, fy) << 5;      w11 = __mul24(fx, fy) << 5;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
}

// remapBilinear on one pixel: (ix, iy) = integer part already saturated to short, (fx, fy) = 5-bit fractions
__device__ __forceinline__ void sample3_fixed(const uint8_t* __restrict__ src, int W, int H, int ix, int iy, int fx, int fy, uint8_t* out) {
    int w00, w01, w10, w11;
    bilinear_weights(fx, fy, w00, w01, w10, w11);
    bool x0 = (unsigned)ix < (unsigned)W, x1 = (unsigned)(ix + 1) < (unsigned)W;
    bool y0 = (unsigned)iy < (unsigned)H, y1 = (unsigned)(iy + 1) < (unsigned)H;
    const uint8_t* p00 = src + ((long long)iy * W + ix) * 3;
    const uint8_t* p10 = p00 + (size_t)W * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int v00 = (x0 && y0) ? p00[k] : 0, v01 = (x1 && y0) ? p00[3 + k] : 0;
        int v10 = (x0 && y1) ? p10[k] : 0, v11 = (x1 && y1) ? p10[3 + k] : 0;
        int acc = __mul24(v00, w00) + __mul24(v01, w01) + __mul24(v10, w10) + __mul24(v11, w11);
        out[k] = sat_u8((acc + (1 << 14)) >> 15);
    }
}

__device__ __forceinline__ void sample3(const uint8_t* __restrict__ src, int W, int H, float mx, float my, uint8_t* out) {
    int sx = cv_round_x86(mx * 32.f), sy = cv_round_x86(my * 32.f);
    int ix = sx >> 5, iy = sy >> 5;
    ix = max(-32768, min(32767, ix)); iy = max(-32768, min(32767, iy));
    sample3_fixed(src, W, H, ix, iy, sx & 31, sy & 31, out);
}

__device__ __forceinline__ void map_point(const float* __restrict__ h, int x, int y, float& mx, float& my) {
    float fx = (float)x, fy = (float)y;
    float z = h[6] * fx + h[7] * fy + h[8];
    if (z == 0.f) z = 0.00001f;
    mx = __fdiv_rn(h[0] * fx + h[1] * fy + h[2], z);
    my = __fdiv_rn(h[3] * fx + h[4] * fy + h[5], z);
}


}  // namespace poppy_hip
