// oracle/warp.cpp — per-pixel map construction and fixed-point bilinear remap.  TEST INFRASTRUCTURE.
// Restates:
//   src/algo.cpp:146-176 (create_map: invert every matrix again, float math left-to-right, z==0 -> 1e-5)
//   OCV/imgproc/src/imgwarp.cpp:146-150,190-197,213-287 (BilinearTab_i incl. the entry-0 fix-up quirk),
//   :1197-1234 (float map -> cvRound(v*32): integer part saturate_cast<short>(>>5), 5+5 fraction bits),
//   :311-318 (FixedPtCast<int,uchar,15>), :721-731 and :808-852 (bilinear taps, BORDER_CONSTANT value 0)
#include "oracle.h"
#include <cmath>
#include <climits>

namespace oracle {

static int16_t g_tab[1024][4];
static bool g_tab_ready = false;

const int16_t* bilinear_tab() {
    if (!g_tab_ready) {
        for (int fy = 0; fy < 32; ++fy)
            for (int fx = 0; fx < 32; ++fx) {
                float ty[2] = {1.f - fy * (1.f / 32), fy * (1.f / 32)};
                float tx[2] = {1.f - fx * (1.f / 32), fx * (1.f / 32)};
                int16_t* t = g_tab[fy * 32 + fx];
                int sum = 0;
                for (int k1 = 0; k1 < 2; ++k1)
                    for (int k2 = 0; k2 < 2; ++k2) {
                        int v = cv_round_f(ty[k1] * tx[k2] * 32768.f);
                        v = v > SHRT_MAX ? SHRT_MAX : v < SHRT_MIN ? SHRT_MIN : v;
                        t[k1 * 2 + k2] = (int16_t)v;
                        sum += v;
                    }
                // imgwarp.cpp:251-267: the min/max search of the fix-up runs over k1,k2 in {1,2}, i.e. it
                // only ever looks at tap [1][1] and at not-yet-written (zero) taps of the NEXT entry, so the
                // whole deficit lands on tap [1][1].  Only entry 0 saturates (32768 -> 32767): {32767,0,0,1}.
                if (sum != 32768) t[3] = (int16_t)(t[3] - (sum - 32768));
            }
        g_tab_ready = true;
    }
    return &g_tab[0][0];
}

void create_map(const ImageI& triMap, const std::vector<float>& mats, ImageF& mapx, ImageF& mapy) {
    int W = triMap.w, H = triMap.h;
    mapx = ImageF(W, H); mapy = ImageF(W, H);
    size_t nt = mats.size() / 9;
    std::vector<float> inv(mats.size());
    for (size_t t = 0; t < nt; ++t) invert33(&mats[t * 9], &inv[t * 9]);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int idx = triMap.d[(size_t)y * W + x] - 1;
            float mx, my;
            if (idx >= 0) {
                const float* h = &inv[(size_t)idx * 9];
                float z = h[6] * x + h[7] * y + h[8];       // int -> float conversions, then float ops
                if (z == 0) z = 0.00001;
                mx = (h[0] * x + h[1] * y + h[2]) / z;
                my = (h[3] * x + h[4] * y + h[5]) / z;
            } else { mx = (float)x; my = (float)y; }
            mapx.d[(size_t)y * W + x] = mx;
            mapy.d[(size_t)y * W + x] = my;
        }
}

void remap_bilinear(const ImageU8& src, const ImageF& mapx, const ImageF& mapy, ImageU8& dst) {
    const int16_t* tab = bilinear_tab();
    int W = mapx.w, H = mapx.h, C = src.c, SW = src.w, SH = src.h;
    dst = ImageU8(W, H, C);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int sx = cv_round_f(mapx.d[(size_t)y * W + x] * 32.f);
            int sy = cv_round_f(mapy.d[(size_t)y * W + x] * 32.f);
            int a = (sy & 31) * 32 + (sx & 31);
            int ix = sx >> 5, iy = sy >> 5;
            ix = ix > SHRT_MAX ? SHRT_MAX : ix < SHRT_MIN ? SHRT_MIN : ix;
            iy = iy > SHRT_MAX ? SHRT_MAX : iy < SHRT_MIN ? SHRT_MIN : iy;
            const int16_t* w = tab + a * 4;
            uint8_t* D = &dst.d[((size_t)y * W + x) * C];
            bool x0 = ix >= 0 && ix < SW, x1 = ix + 1 >= 0 && ix + 1 < SW;
            bool y0 = iy >= 0 && iy < SH, y1 = iy + 1 >= 0 && iy + 1 < SH;
            for (int k = 0; k < C; ++k) {
                int v00 = (x0 && y0) ? src.d[((size_t)iy * SW + ix) * C + k] : 0;
                int v01 = (x1 && y0) ? src.d[((size_t)iy * SW + ix + 1) * C + k] : 0;
                int v10 = (x0 && y1) ? src.d[((size_t)(iy + 1) * SW + ix) * C + k] : 0;
                int v11 = (x1 && y1) ? src.d[((size_t)(iy + 1) * SW + ix + 1) * C + k] : 0;
                int acc = v00 * w[0] + v01 * w[1] + v10 * w[2] + v11 * w[3];
                int r = (acc + (1 << 14)) >> 15;
                D[k] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
            }
        }
}

}  // namespace oracle
