// kernels_align.hip — cv::warpAffine(u8x3, INTER_LINEAR, BORDER_CONSTANT 0) for the auto-align steps (auto_align.h).
// OCV/imgproc/src/imgwarp.cpp:2155-2290 (WarpAffineInvoker: 10-bit fixed-point coordinates, 5-bit fractions) feeding
// remapBilinear (:721-731,808-852).  The per-column / per-row coordinate terms are the reference's saturate_cast<int> of
// double products; the host computes them with the same libm-free double arithmetic and uploads 2 * (w + h) ints, the
// kernel is integer only: memory-bound gather, one thread per pixel.
#include "auto_align.h"
#include "warp_device.h"
#include <cmath>
#include <vector>

namespace poppy_hip {

__global__ void __launch_bounds__(256) k_warp_affine(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int W, int H,
                                                     const int* __restrict__ adelta, const int* __restrict__ bdelta,
                                                     const int* __restrict__ x0row, const int* __restrict__ y0row) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int X = (int)((unsigned)x0row[y] + (unsigned)adelta[x]) >> 5;      // the SIMD adds wrap
    const int Y = (int)((unsigned)y0row[y] + (unsigned)bdelta[x]) >> 5;
    const int ix = max(-32768, min(32767, X >> 5)), iy = max(-32768, min(32767, Y >> 5));
    uint8_t o[3];
    sample3_fixed(src, W, H, ix, iy, X & 31, Y & 31, o);
    uint8_t* d = dst + ((size_t)y * W + x) * 3;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
}

static int round_x86(double v) {                  // cvRound(double) = cvtsd2si: nearest even, "indefinite" when out of range
    const double r = std::nearbyint(v);
    return (r >= -2147483648.0 && r <= 2147483647.0) ? (int)r : INT_MIN;
}

bool warp_affine_device(const uint8_t* d_src, uint8_t* d_dst, int w, int h, const double Mfwd[6], int* d_tables, hipStream_t s) {
    double M[6] = {Mfwd[0], Mfwd[1], Mfwd[2], Mfwd[3], Mfwd[4], Mfwd[5]};
    {   // dst -> src map (imgwarp.cpp:2622-2631)
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D;
        M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5];
        const double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
    }
    std::vector<int> t((size_t)2 * (w + h));
    int *ad = t.data(), *bd = ad + w, *x0 = bd + w, *y0 = x0 + h;
    for (int x = 0; x < w; ++x) { ad[x] = round_x86(M[0] * x * 1024); bd[x] = round_x86(M[3] * x * 1024); }
    for (int y = 0; y < h; ++y) {
        x0[y] = (int)((unsigned)round_x86((M[1] * y + M[2]) * 1024) + 16u);
        y0[y] = (int)((unsigned)round_x86((M[4] * y + M[5]) * 1024) + 16u);
    }
    // the table is tiny and the caller waits for each step's result anyway: a plain synchronous copy keeps the staging buffer simple
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    if (hipMemcpy(d_tables, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return false;
    hipLaunchKernelGGL(k_warp_affine, dim3((w + 255) / 256, h), dim3(256), 0, s, d_src, d_dst, w, h,
                       d_tables, d_tables + w, d_tables + 2 * w, d_tables + 2 * w + h);
    return hipGetLastError() == hipSuccess;
}

}  // namespace poppy_hip
