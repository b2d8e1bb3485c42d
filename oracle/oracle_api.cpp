// oracle/oracle_api.cpp — flat C entry points so tests/ and bench.py's cpu_baseline leg can drive
// the oracle through ctypes.  TEST INFRASTRUCTURE; nothing in the product links this.
#include "oracle.h"
#include <cstring>

using namespace oracle;

static ImageU8 wrap_u8(const uint8_t* p, int w, int h, int c) { ImageU8 m(w, h, c); memcpy(m.d.data(), p, m.d.size()); return m; }
static ImageF wrap_f(const float* p, int w, int h, int c) { ImageF m(w, h, c); memcpy(m.d.data(), p, m.d.size() * 4); return m; }
static std::vector<Pt> wrap_pts(const float* p, int n) { std::vector<Pt> v(n); if (n) memcpy(v.data(), p, (size_t)n * 8); return v; }
template <typename T> static void put(T* dst, const std::vector<T>& v) { if (dst && !v.empty()) memcpy(dst, v.data(), v.size() * sizeof(T)); }
static void put_img(float* dst, const ImageF& m) { if (dst) memcpy(dst, m.d.data(), m.d.size() * 4); }
static void put_img(uint8_t* dst, const ImageU8& m) { if (dst) memcpy(dst, m.d.data(), m.d.size()); }

extern "C" {

int orc_round_f(float v) { return cv_round_f(v); }
int orc_round_d(double v) { return cv_round(v); }

void orc_clip_points(float* pts, int n, int cols, int rows) {
    auto v = wrap_pts(pts, n); clip_points(v, cols, rows); memcpy(pts, v.data(), (size_t)n * 8);
}
int orc_make_uniq(const float* pts, int n, float* out) {
    std::vector<Pt> o; make_uniq(wrap_pts(pts, n), o); put((Pt*)out, o); return (int)o.size();
}
void orc_morph_points(const float* a, const float* b, int n, float s, float* out) {
    std::vector<Pt> o; morph_points(wrap_pts(a, n), wrap_pts(b, n), o, s); put((Pt*)out, o);
}
int orc_delaunay(int w, int h, const float* pts, int n, float* tri6, int maxTris) {
    std::vector<float> t;
    if (!delaunay_triangles(w, h, wrap_pts(pts, n), t)) return -1;
    int nt = (int)t.size() / 6;
    if (nt > maxTris) return -2;
    put(tri6, t);
    return nt;
}
int orc_triangle_indices(const float* tri6, int nt, const float* pts, int n, int* idx3) {
    std::vector<float> t(tri6, tri6 + (size_t)nt * 6); std::vector<int> o;
    triangle_indices(t, wrap_pts(pts, n), o); put(idx3, o); return (int)o.size() / 3;
}
void orc_triangle_int_points(const int* idx3, int nt, const float* pts, int n, int* out) {
    std::vector<int> id(idx3, idx3 + (size_t)nt * 3); std::vector<IPt> o;
    triangle_int_points(id, wrap_pts(pts, n), o); if (!o.empty()) memcpy(out, o.data(), o.size() * 8);
}
void orc_paint_triangles(int w, int h, const int* tris, int nt, int32_t* map) {
    ImageI img(w, h);
    std::vector<IPt> t((size_t)nt * 3); if (nt) memcpy(t.data(), tris, t.size() * 8);
    paint_triangles(img, t);
    memcpy(map, img.d.data(), img.d.size() * 4);
}
int orc_invert33(const float* m, float* out) { return invert33(m, out) ? 1 : 0; }
void orc_homographies(const int* t1, const int* t2, int nt, float ratio, float* H, float* M1, float* M2) {
    for (int t = 0; t < nt; ++t) {
        solve_homography((const IPt*)t1 + t * 3, (const IPt*)t2 + t * 3, H + t * 9);
        morph_homography(H + t * 9, ratio, M1 + t * 9, M2 + t * 9);
    }
}
void orc_create_map(const int32_t* triMap, int w, int h, const float* mats, int nt, float* mapx, float* mapy) {
    ImageI tm(w, h); memcpy(tm.d.data(), triMap, tm.d.size() * 4);
    std::vector<float> m(mats, mats + (size_t)nt * 9);
    ImageF mx, my; create_map(tm, m, mx, my); put_img(mapx, mx); put_img(mapy, my);
}
void orc_remap(const uint8_t* src, int sw, int sh, int c, const float* mapx, const float* mapy, int w, int h, uint8_t* dst) {
    ImageU8 d; remap_bilinear(wrap_u8(src, sw, sh, c), wrap_f(mapx, w, h, 1), wrap_f(mapy, w, h, 1), d); put_img(dst, d);
}
void orc_bilinear_tab(int16_t* out) { memcpy(out, bilinear_tab(), 1024 * 4 * 2); }
void orc_u8_to_f32(const uint8_t* s, int n, float* d) { ImageF o; u8_to_f32(wrap_u8(s, n, 1, 1), o); put_img(d, o); }
void orc_f32_to_u8(const float* s, int n, uint8_t* d) { ImageU8 o; f32_to_u8(wrap_f(s, n, 1, 1), o); put_img(d, o); }
void orc_blend_mask(const float* gabor2, int w, int h, double maskRatio, float* mask) {
    ImageF m; blend_mask(wrap_f(gabor2, w, h, 3), maskRatio, m); put_img(mask, m);
}
void orc_pyr_down(const float* s, int w, int h, int c, float* d) { ImageF o; pyr_down(wrap_f(s, w, h, c), o); put_img(d, o); }
void orc_pyr_up(const float* s, int w, int h, int c, int dw, int dh, float* d) { ImageF o; pyr_up(wrap_f(s, w, h, c), o, dw, dh); put_img(d, o); }
void orc_laplacian_blend(const float* l, const float* r, const float* mask, int w, int h, int levels, float* out) {
    ImageF o; laplacian_blend(wrap_f(l, w, h, 3), wrap_f(r, w, h, 3), wrap_f(mask, w, h, 1), levels, o); put_img(out, o);
}
void orc_unsharp(const float* s, int w, int h, float amount, float threshold, float* out, float* blur, float* med) {
    ImageF o, b, m; unsharp_mask(wrap_f(s, w, h, 3), 1.f, amount, threshold, o, &b, &m);
    put_img(out, o); put_img(blur, b); put_img(med, m);
}

// Whole frame.  Optional debug outputs may be null.  tri buffers must hold maxTris entries.
int orc_morph_images(const uint8_t* c1, const uint8_t* c2, const float* gabor2, int w, int h,
                     const float* p1, const float* p2, int n, double shape, double mask, int levels,
                     uint8_t* out, float* morphedPts,
                     int maxTris, int* nTris, float* tri6, int* idx3, float* Hm, float* M1, float* M2,
                     int32_t* triMap, float* mapx1, float* mapy1, float* mapx2, float* mapy2,
                     uint8_t* trImg1, uint8_t* trImg2, float* lbmask, float* lapBlend, float* unsharpOut) {
    FrameDebug dbg; ImageU8 o; std::vector<Pt> mp;
    int rc = morph_images(wrap_u8(c1, w, h, 3), wrap_u8(c2, w, h, 3), wrap_f(gabor2, w, h, 3),
                          wrap_pts(p1, n), wrap_pts(p2, n), shape, mask, levels, o, mp, &dbg);
    if (rc) return rc;
    put_img(out, o); put((Pt*)morphedPts, mp);
    int nt = (int)dbg.idx3.size() / 3;
    if (nTris) *nTris = nt;
    if (nt <= maxTris) { put(idx3, dbg.idx3); put(Hm, dbg.H); put(M1, dbg.M1); put(M2, dbg.M2); }
    if ((int)dbg.tri6.size() / 6 <= maxTris) put(tri6, dbg.tri6);
    if (triMap) memcpy(triMap, dbg.triMap.d.data(), dbg.triMap.d.size() * 4);
    put_img(mapx1, dbg.mapx1); put_img(mapy1, dbg.mapy1); put_img(mapx2, dbg.mapx2); put_img(mapy2, dbg.mapy2);
    put_img(trImg1, dbg.trImg1); put_img(trImg2, dbg.trImg2);
    put_img(lbmask, dbg.lbmask); put_img(lapBlend, dbg.lapBlend); put_img(unsharpOut, dbg.unsharp);
    return 0;
}

int orc_orb_detect(const uint8_t* img, int w, int h, int nfeatures, float* kps7, int maxKps, float* fast3, int maxFast, int* nFast) {
    std::vector<float> k, f;
    int n = orb_detect(wrap_u8(img, w, h, 1), nfeatures, k, &f);
    if (n > maxKps) return -1;
    put(kps7, k);
    if (nFast) *nFast = (int)f.size() / 3;
    if (fast3 && (int)f.size() / 3 <= maxFast) put(fast3, f);
    return n;
}
int orc_orb_describe(const uint8_t* img, int w, int h, const float* kps7, int n, uint8_t* desc, int trig_mode) {
    std::vector<float> k(kps7, kps7 + (size_t)n * 7); std::vector<uint8_t> d;
    int m = orb_describe(wrap_u8(img, w, h, 1), k, d, trig_mode); put(desc, d); return m;
}
int orc_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, int* out4) {
    std::vector<int> o; hamming_knn2(q, nq, t, nt, bytes, o); put(out4, o); return nq;
}
int orc_ratio_symmetry(const int* knn12, int n1, const int* knn21, int n2, float ratio, int* keep12, int* keep21, int* out3) {
    std::vector<int> k1, k2, o;
    ratio_test(knn12, n1, ratio, k1); ratio_test(knn21, n2, ratio, k2);
    symmetry_test(knn12, k1.data(), n1, knn21, k2.data(), n2, o);
    if (keep12) put(keep12, k1);
    if (keep21) put(keep21, k2);
    put(out3, o);
    return (int)o.size() / 3;
}
int orc_hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int bytes, int* out3) {
    std::vector<int> o; hamming_match(q, nq, t, nt, bytes, o); put(out3, o); return (int)o.size() / 3;
}

int orc_distance_map(const float* p1, const float* p2, int n, double* out5) {
    std::vector<DistPair> dm; make_distance_map(wrap_pts(p1, n), wrap_pts(p2, n), dm);
    for (size_t i = 0; i < dm.size(); ++i) { double* o = out5 + i * 5; o[0] = dm[i].d; o[1] = dm[i].a.x; o[2] = dm[i].a.y; o[3] = dm[i].b.x; o[4] = dm[i].b.y; }
    return (int)dm.size();
}
int orc_filter_invalid(float* p1, float* p2, int n, int cols, int rows) {
    auto a = wrap_pts(p1, n), b = wrap_pts(p2, n); filter_invalid_points(a, b, cols, rows);
    if (a.size() > b.size()) a.resize(b.size()); else b.resize(a.size());
    put((Pt*)p1, a); put((Pt*)p2, b); return (int)a.size();
}
double orc_morph_distance(const float* p1, const float* p2, int n, int w, int h) { return morph_distance(wrap_pts(p1, n), wrap_pts(p2, n), w, h); }
int orc_match_prepare(const float* p1, const float* p2, int n, int w, int h, double tol, double imd, float* o1, float* o2) {
    auto a = wrap_pts(p1, n), b = wrap_pts(p2, n); match_prepare(a, b, w, h, tol, imd); put((Pt*)o1, a); put((Pt*)o2, b); return (int)a.size();
}

// Extractor::foreground for one BGR image.  `stages` (optional): 50 u8 images of w*h in the order flow0, acc0, then 12 x
// (med, flow, acc, blur); `floats` (optional): lin, logged, finalMask (w*h each); misc: grey, masked (u8), ln20.
void orc_foreground(const uint8_t* bgr, int w, int h, uint8_t* fg, uint8_t* grey, uint8_t* stages, float* floats, uint8_t* masked, float* ln20) {
    ForegroundDebug dbg; ImageU8 out;
    foreground(wrap_u8(bgr, w, h, 3), out, &dbg);
    put_img(fg, out); put_img(grey, dbg.grey); put_img(masked, dbg.masked);
    const size_t n = (size_t)w * h;
    if (stages) for (size_t k = 0; k < dbg.stages.size(); ++k) memcpy(stages + k * n, dbg.stages[k].d.data(), n);
    if (floats) { memcpy(floats, dbg.lin.d.data(), n * 4); memcpy(floats + n, dbg.logged.d.data(), n * 4); memcpy(floats + 2 * n, dbg.final_mask.d.data(), n * 4); }
    if (ln20) *ln20 = dbg.ln20;
}
void orc_median_blur_u8(const uint8_t* s, int w, int h, int ksize, uint8_t* d) { ImageU8 o; median_blur_u8(wrap_u8(s, w, h, 1), ksize, o); put_img(d, o); }
void orc_gaussian_blur23_u8(const uint8_t* s, int w, int h, uint8_t* d) { ImageU8 o; gaussian_blur23_u8(wrap_u8(s, w, h, 1), o); put_img(d, o); }
void orc_equalize_hist(const uint8_t* s, int w, int h, uint8_t* d) { ImageU8 o; equalize_hist(wrap_u8(s, w, h, 1), o); put_img(d, o); }
float orc_log32f(float x) { return cv_log32f(x); }

void orc_orb_unsharp_gray(const uint8_t* gf, int w, int h, float* us) { ImageF o; orb_unsharp_gray(wrap_u8(gf, w, h, 1), o); put_img(us, o); }
void orc_gabor_bank(int ks, double sigma, double lambd, double gamma, double psi, float* out) {
    std::vector<float> b; gabor_bank(ks, sigma, lambd, gamma, psi, b); memcpy(out, b.data(), b.size() * 4);
}
void orc_gabor_filter_direct(const float* src, int w, int h, int c, int ks, const float* bank, float* dst) {
    std::vector<float> b(bank, bank + (size_t)16 * ks * ks); ImageF o;
    gabor_filter_direct(wrap_f(src, w, h, c), ks, b, o); put_img(dst, o);
}

double orc_dft_detail2(const uint8_t* gray, int w, int h) { return dft_detail2(wrap_u8(gray, w, h, 1)); }
void orc_radial_gradient(int w, int h, float* out) { ImageF o; radial_gradient(w, h, o); put_img(out, o); }
void orc_radial_mask(int w, int h, float* out) { ImageF o; radial_mask(w, h, o); put_img(out, o); }
void orc_set_radial_mask(int on) { set_radial_mask(on != 0); }
void orc_orb_input(const uint8_t* gf, int w, int h, uint8_t* g) { ImageU8 o; orb_input_image(wrap_u8(gf, w, h, 1), o); put_img(g, o); }
void orc_gaussian_taps_fx(int n, double sigma, int* out) { std::vector<int> t; gaussian_taps_fx(n, sigma, t); memcpy(out, t.data(), t.size() * sizeof(int)); }
void orc_dissolve(const uint8_t* a, const uint8_t* b, int w, int h, int c, double phase, uint8_t* d) {
    ImageU8 o; dissolve_u8(wrap_u8(a, w, h, c), wrap_u8(b, w, h, c), phase, o); put_img(d, o);
}
void orc_fill_convex(int w, int h, const int* xy, int npts, int32_t value, int32_t* map) {      // map is updated in place
    ImageI img(w, h);
    memcpy(img.d.data(), map, (size_t)w * h * 4);
    std::vector<IPt> v(npts);
    for (int i = 0; i < npts; ++i) v[i] = IPt{xy[2 * i], xy[2 * i + 1]};
    fill_convex(img, v.data(), npts, value);
    memcpy(map, img.d.data(), (size_t)w * h * 4);
}
int orc_convex_hull(const float* pts, int n, float* hull) {
    std::vector<Pt> h; convex_hull_points(wrap_pts(pts, n), h);
    memcpy(hull, h.data(), h.size() * sizeof(Pt));
    return (int)h.size();
}
void orc_blur_margin(const uint8_t* src, int w, int h, int uw, int uh, uint8_t* dst) { ImageU8 o; blur_margin(wrap_u8(src, w, h, 3), uw, uh, o); put_img(dst, o); }

// ---- auto-align ----------------------------------------------------------------------------------------------
void orc_warp_affine(const uint8_t* src, int w, int h, int c, const double* M6, uint8_t* dst) {
    ImageU8 o; warp_affine(wrap_u8(src, w, h, c), M6, o); memcpy(dst, o.d.data(), o.d.size());
}
void orc_rotation_matrix(float cx, float cy, double angle, double scale, double* M6) { rotation_matrix(cx, cy, angle, scale, M6); }
void orc_align_prims(const float* x, const float* y, int n, double* mean2_, double* sumsq2, float* gemm4, const float* svd_in4, float* svd_w2,
                     float* svd_u4, float* svd_vt4, const float* tr_m4, float* tr_out, double* persp9, float* persp_pts) {
    auto X = wrap_pts(x, n), Y = wrap_pts(y, n);
    mean2(X, mean2_); sum_squares2(X, sumsq2); gemm_at_b(X, Y, gemm4);
    svd2(svd_in4, svd_w2, svd_u4, svd_vt4);
    std::vector<Pt> t; transform2(X, tr_m4, t); put((Pt*)tr_out, t);
    perspective_from_4(X.data(), Y.data(), persp9);
    auto q = X; perspective_points(q, persp9); put((Pt*)persp_pts, q);
}
void orc_procrustes(const float* x, const float* y, int n, float* rot4, float* scalars2, float* yprime, float* trans2) {
    ProcrustesResult R; procrustes(wrap_pts(x, n), wrap_pts(y, n), R);
    memcpy(rot4, R.rotation, 16); scalars2[0] = R.scale; scalars2[1] = R.error; put((Pt*)yprime, R.yprime); memcpy(trans2, R.translation, 8);
}
// which: 0 retranslate, 1 reprocrustes, 2 rerotate, 3 autoAlign.  img (w*h*3) and p2 are updated in place; p1 only by 3 (never changes).
double orc_align_step(int which, uint8_t* img, int w, int h, const float* p1, float* p2, int n) {
    ImageU8 im = wrap_u8(img, w, h, 3);
    auto a = wrap_pts(p1, n), b = wrap_pts(p2, n);
    double d = 0;
    if (which == 0) d = retranslate(im, a, b, w, h);
    else if (which == 1) d = reprocrustes(im, a, b, w, h);
    else if (which == 2) d = rerotate(im, a, b, w, h);
    else { auto_align(im, a, b, w, h); d = morph_distance(a, b, w, h); }
    memcpy(img, im.d.data(), im.d.size());
    put((Pt*)p2, b);
    return d;
}

}  // extern "C"
