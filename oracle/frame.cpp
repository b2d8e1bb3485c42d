// oracle/frame.cpp — one morph frame = poppy::morph_images (src/algo.cpp:178-273).  TEST INFRASTRUCTURE.
#include "oracle.h"
#include <cmath>

namespace oracle {

int morph_images(const ImageU8& c1, const ImageU8& c2, const ImageF& gabor2,
                 const std::vector<Pt>& pts1, const std::vector<Pt>& pts2,
                 double shapeRatio, double maskRatio, int levels,
                 ImageU8& out, std::vector<Pt>& morphedPoints, FrameDebug* dbg) {
    const int W = c1.w, H = c1.h;
    std::vector<Pt> s1 = pts1, s2 = pts2, uniq;
    clip_points(s1, W, H);                       // algo.cpp:185,191 (subDiv1/2 only feed the GUI overlay)
    clip_points(s2, W, H);
    morph_points(s1, s2, morphedPoints, (float)shapeRatio);      // float parameter (algo.cpp:50)
    clip_points(morphedPoints, W, H);
    make_uniq(morphedPoints, uniq);

    std::vector<float> tri6;
    if (!delaunay_triangles(W, H, uniq, tri6)) return -1;
    std::vector<int> idx3;
    triangle_indices(tri6, morphedPoints, idx3);
    std::vector<IPt> t1, t2, tm;
    triangle_int_points(idx3, s1, t1);
    triangle_int_points(idx3, s2, t2);
    triangle_int_points(idx3, morphedPoints, tm);

    ImageI triMap(W, H);
    paint_triangles(triMap, tm);

    size_t nt = idx3.size() / 3;
    std::vector<float> Hm(nt * 9), M1(nt * 9), M2(nt * 9);
    for (size_t t = 0; t < nt; ++t) {
        solve_homography(&t1[t * 3], &t2[t * 3], &Hm[t * 9]);
        morph_homography(&Hm[t * 9], (float)shapeRatio, &M1[t * 9], &M2[t * 9]);   // float blend_ratio (algo.cpp:128)
    }

    ImageF mx1, my1, mx2, my2;
    ImageU8 tr1, tr2;
    create_map(triMap, M1, mx1, my1);
    remap_bilinear(c1, mx1, my1, tr1);
    create_map(triMap, M2, mx2, my2);
    remap_bilinear(c2, mx2, my2, tr2);

    ImageF l, r, mask, lap, sharp;
    u8_to_f32(tr1, l);
    u8_to_f32(tr2, r);
    blend_mask(gabor2, maskRatio, mask);
    laplacian_blend(l, r, mask, levels, lap);
    double amount = std::sin(maskRatio * M_PI);
    unsharp_mask(lap, 1.f, (float)(1.0 - amount), (float)0.3, sharp);
    f32_to_u8(sharp, out);

    if (dbg) {
        dbg->morphed = morphedPoints; dbg->uniq = uniq; dbg->tri6 = tri6; dbg->idx3 = idx3; dbg->triMorph = tm;
        dbg->H = Hm; dbg->M1 = M1; dbg->M2 = M2; dbg->triMap = triMap;
        dbg->mapx1 = mx1; dbg->mapy1 = my1; dbg->mapx2 = mx2; dbg->mapy2 = my2;
        dbg->trImg1 = tr1; dbg->trImg2 = tr2; dbg->lbmask = mask; dbg->lapBlend = lap; dbg->unsharp = sharp;
    }
    return 0;
}

// The no-match fallback  Mat blend = ((img2 * phase) + (img1 * (1.0 - phase)))  (src/poppy.hpp:129).  The MatExpr sum of two
// scaled 8-bit images is one addWeighted(img2, phase, img1, 1 - phase, 0) (matrix_expressions.cpp MatOp_AddEx); for 8-bit
// inputs the weights are narrowed to float and a pixel is  a*alpha + (b*beta + gamma)  in float with a rounding after every
// operation (v_fma is unfused in the SSE3-baseline build), then cvRound + saturate (arithm.simd.hpp:1705-1755).
void dissolve_u8(const ImageU8& img1, const ImageU8& img2, double phase, ImageU8& out) {
    const float alpha = (float)phase, beta = (float)(1.0 - phase);
    out = ImageU8(img1.w, img1.h, img1.c);
    const size_t n = (size_t)img1.w * img1.h * img1.c;
    for (size_t i = 0; i < n; ++i) {
        const float t = (float)img1.d[i] * beta + 0.f;
        const float v = (float)img2.d[i] * alpha + t;
        const int r = cv_round_f(v);
        out.d[i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
    }
}

}  // namespace oracle
