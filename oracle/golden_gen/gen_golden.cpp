// Golden-vector generator — CONTAINER-ONLY TOOL, never built or run on the GPU box,
// never linked into the product or into oracle/liboracle.so.
//
// It links the reference's own translation units (compiled from where they lie
// under /root/reference/src) against the vendored OpenCV 4.6.0 that the survey
// stage built and dumps the inputs/outputs of every
// stage boundary of the morph hot path as raw little-endian arrays.  pack.py then
// turns the dumps into the committed fixtures under tests/golden/.
//
// The sequence inside dump_bstage() calls the reference's public helpers
// (algo.hpp) one by one so intermediate values can be written out, and then calls
// the real poppy::morph_images() on the same inputs and aborts unless both final
// frames are byte-identical — so the dumped intermediates are those of the real path.
//
// usage: gen_golden <mode> <case_dir>
//   inputs  are read  from <case_dir>/in/NAME.DTYPE.SHAPE.bin
//   outputs are written to <case_dir>/out/NAME.DTYPE.SHAPE.bin
#include <random>
#include <vector>
#include <sstream>
#include <fstream>
#include <complex>
#include <opencv2/opencv.hpp>
#include "face.hpp"
#define private public      // Matcher::initialMorphDist_ must be set to test match() in isolation
#include "matcher.hpp"
#undef private
#include "util.hpp"
#include "algo.hpp"
#include "blend.hpp"
#include "draw.hpp"
#include "extractor.hpp"
#include "settings.hpp"
#include "transformer.hpp"
#include "procrustes.hpp"
namespace poppy {   // defined in experiments.hpp, emitted by extractor.o
double dft_detail2(const cv::Mat& src, cv::Mat& dst);
int ratioTest(std::vector<std::vector<cv::DMatch>>& matches);
void symmetryTest(const std::vector<std::vector<cv::DMatch>>& matches1, const std::vector<std::vector<cv::DMatch>>& matches2,
                  std::vector<cv::DMatch>& symMatches);
}
#include "poppy.hpp"
#include <opencv2/imgproc.hpp>
#include <opencv2/features2d.hpp>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <dirent.h>
#include <sys/stat.h>

// (Poppy's util.cpp references cv::namedWindow / imshow / waitKey from show_image() / wait_key(), never reached with show_gui = false: the
// symbols come from the vendored highgui module, built without a window back end — build_ocv.sh.  Until round 4 three no-op stand-ins stood here.)

using namespace cv;
using namespace poppy;
using std::string;
using std::vector;

static string g_in, g_out;

static string shape_str(const vector<int>& shp) {
    std::ostringstream o;
    for (size_t i = 0; i < shp.size(); ++i) { if (i) o << "x"; o << shp[i]; }
    return o.str();
}

static void write_raw(const string& name, const char* dtype, const vector<int>& shp, const void* p, size_t bytes) {
    string fn = g_out + "/" + name + "." + dtype + "." + shape_str(shp) + ".bin";
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) { perror(fn.c_str()); exit(1); }
    if (bytes) fwrite(p, 1, bytes, f);
    fclose(f);
}

static const char* depth_name(int depth) {
    switch (depth) {
    case CV_8U: return "u8"; case CV_16S: return "i16"; case CV_16U: return "u16";
    case CV_32S: return "i32"; case CV_32F: return "f32"; case CV_64F: return "f64";
    }
    return "unk";
}

static void dump_mat(const string& name, const Mat& m_) {
    Mat m = m_.isContinuous() ? m_ : m_.clone();
    vector<int> shp = { m.rows, m.cols };
    if (m.channels() > 1) shp.push_back(m.channels());
    write_raw(name, depth_name(m.depth()), shp, m.data, m.total() * m.elemSize());
}

static void dump_pts(const string& name, const vector<Point2f>& pts) {
    write_raw(name, "f32", { (int)pts.size(), 2 }, pts.data(), pts.size() * sizeof(Point2f));
}

static void dump_f64(const string& name, const vector<double>& v) {
    write_raw(name, "f64", { (int)v.size() }, v.data(), v.size() * sizeof(double));
}

// find input file by prefix "NAME." and parse its shape
static bool find_input(const string& name, string& path, string& dtype, vector<int>& shp) {
    DIR* d = opendir(g_in.c_str());
    if (!d) return false;
    bool ok = false;
    while (dirent* e = readdir(d)) {
        string fn = e->d_name;
        if (fn.rfind(name + ".", 0) != 0) continue;
        // NAME.DTYPE.SHAPE.bin
        size_t p1 = name.size() + 1, p2 = fn.find('.', p1), p3 = fn.find('.', p2 + 1);
        if (p2 == string::npos || p3 == string::npos) continue;
        dtype = fn.substr(p1, p2 - p1);
        string s = fn.substr(p2 + 1, p3 - p2 - 1);
        shp.clear();
        std::stringstream ss(s); string tok;
        while (std::getline(ss, tok, 'x')) shp.push_back(atoi(tok.c_str()));
        path = g_in + "/" + fn;
        ok = true;
        break;
    }
    closedir(d);
    return ok;
}

static Mat read_mat(const string& name) {
    string path, dtype; vector<int> shp;
    if (!find_input(name, path, dtype, shp)) { fprintf(stderr, "missing input %s\n", name.c_str()); exit(1); }
    int depth = dtype == "u8" ? CV_8U : dtype == "f32" ? CV_32F : dtype == "i32" ? CV_32S : dtype == "f64" ? CV_64F : -1;
    int cn = shp.size() > 2 ? shp[2] : 1;
    int rows = shp[0], cols = shp.size() > 1 ? shp[1] : 1;
    Mat m(rows, cols, CV_MAKETYPE(depth, cn));
    FILE* f = fopen(path.c_str(), "rb");
    size_t n = fread(m.data, 1, m.total() * m.elemSize(), f);
    fclose(f);
    if (n != m.total() * m.elemSize()) { fprintf(stderr, "short read %s\n", path.c_str()); exit(1); }
    return m;
}

static vector<Point2f> read_pts(const string& name) {
    Mat m = read_mat(name);
    vector<Point2f> v(m.rows);
    for (int i = 0; i < m.rows; ++i) v[i] = Point2f(m.at<float>(i, 0), m.at<float>(i, 1));
    return v;
}

static vector<double> read_f64(const string& name) {
    Mat m = read_mat(name);
    vector<double> v(m.total());
    for (size_t i = 0; i < v.size(); ++i) v[i] = ((double*)m.data)[i];
    return v;
}

static void dump_mats33(const string& name, const vector<Mat>& ms) {
    vector<float> buf(ms.size() * 9);
    for (size_t i = 0; i < ms.size(); ++i)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) buf[i * 9 + r * 3 + c] = ms[i].at<float>(r, c);
    write_raw(name, "f32", { (int)ms.size(), 3, 3 }, buf.data(), buf.size() * 4);
}

// ---------------------------------------------------------------------------------------------
// B stage: one morph_images() call per (shape, mask) ratio, with intermediates.
// ---------------------------------------------------------------------------------------------
static void dump_bstage() {
    Mat c1 = read_mat("c1"), c2 = read_mat("c2"), gabor2 = read_mat("gabor2");
    vector<Point2f> pts1 = read_pts("pts1"), pts2 = read_pts("pts2");
    vector<double> ratios = read_f64("ratios");   // pairs (shape, mask)
    int levels = (int)read_f64("levels")[0];
    Settings::instance().pyramid_levels = levels;
    int w = c1.cols, h = c1.rows;

    for (size_t k = 0; k + 1 < ratios.size(); k += 2) {
        double shapeRatio = ratios[k], maskRatio = ratios[k + 1];
        string pf = "f" + std::to_string(k / 2) + "_";
        vector<Point2f> s1 = pts1, s2 = pts2, morphed, uniqMorph;

        clip_points(s1, w, h);
        clip_points(s2, w, h);
        morph_points(s1, s2, morphed, shapeRatio);
        clip_points(morphed, w, h);
        make_uniq(morphed, uniqMorph);
        dump_pts(pf + "morphedPoints", morphed);
        dump_pts(pf + "uniqMorph", uniqMorph);

        Subdiv2D sd(Rect(0, 0, w, h));
        sd.insert(uniqMorph);
        vector<Vec6f> tl;
        sd.getTriangleList(tl);
        write_raw(pf + "triangleList", "f32", { (int)tl.size(), 6 }, tl.data(), tl.size() * sizeof(Vec6f));

        vector<Vec3i> tri;
        get_triangle_indices(sd, morphed, tri);
        write_raw(pf + "triIdx", "i32", { (int)tri.size(), 3 }, tri.data(), tri.size() * sizeof(Vec3i));

        vector<vector<Point>> t1, t2, tm;
        make_triangler_points(tri, s1, t1);
        make_triangler_points(tri, s2, t2);
        make_triangler_points(tri, morphed, tm);
        {
            vector<int> buf;
            for (auto& t : tm) for (auto& p : t) { buf.push_back(p.x); buf.push_back(p.y); }
            write_raw(pf + "triMorphInt", "i32", { (int)tm.size(), 3, 2 }, buf.data(), buf.size() * 4);
        }

        Mat triMap = Mat::zeros(Size(w, h), CV_32SC1);
        paint_triangles(triMap, tm);
        dump_mat(pf + "triMap", triMap);

        vector<Mat> hm, m1, m2;
        solve_homography(t1, t2, hm);
        morph_homography(hm, m1, m2, shapeRatio);
        dump_mats33(pf + "H", hm);
        dump_mats33(pf + "M1", m1);
        dump_mats33(pf + "M2", m2);

        Mat mx1, my1, mx2, my2, tr1, tr2;
        create_map(triMap, m1, mx1, my1);
        remap(c1, tr1, mx1, my1, INTER_LINEAR);
        create_map(triMap, m2, mx2, my2);
        remap(c2, tr2, mx2, my2, INTER_LINEAR);
        dump_mat(pf + "mapx1", mx1); dump_mat(pf + "mapy1", my1);
        dump_mat(pf + "mapx2", mx2); dump_mat(pf + "mapy2", my2);
        dump_mat(pf + "trImg1", tr1); dump_mat(pf + "trImg2", tr2);

        Mat_<Vec3f> l, r;
        tr1.convertTo(l, CV_32F, 1.0 / 255.0);
        tr2.convertTo(r, CV_32F, 1.0 / 255.0);
        dump_mat(pf + "l", l);

        Mat mk;
        cvtColor(gabor2, mk, COLOR_BGR2GRAY);
        mk = 1.0 - mk;
        Mat ones = Mat::ones(mk.size(), mk.type());
        Mat lbmask = (ones * (1.0 - maskRatio)) - (mk * maskRatio);
        lbmask.setTo(0.0, lbmask < 0);
        lbmask.setTo(1.0, lbmask > 1);
        dump_mat(pf + "lbmask", lbmask);

        // pyramid primitives on the first two levels (pin pyrDown / pyrUp themselves)
        {
            Mat d0, u0, d1, u1, md0;
            pyrDown(l, d0); pyrUp(d0, u0, l.size());
            pyrDown(d0, d1); pyrUp(d1, u1, d0.size());
            pyrDown(lbmask, md0);
            dump_mat(pf + "pyrDown0", d0); dump_mat(pf + "pyrUp0", u0);
            dump_mat(pf + "pyrDown1", d1); dump_mat(pf + "pyrUp1", u1);
            dump_mat(pf + "maskDown0", md0);
        }

        LaplacianBlending lb(l, r, lbmask, levels);
        Mat_<Vec3f> lap = lb.blend();
        dump_mat(pf + "lapBlend", lap);

        double amount = sin(maskRatio * M_PI);
        {   // unsharp primitives
            Mat blurred, um;
            GaussianBlur(lap, blurred, Size(0, 0), 1.0f);
            subtract(lap, blurred, um);
            dump_mat(pf + "usBlur", blurred);
            medianBlur(um, um, 3);
            dump_mat(pf + "usMedian", um);
        }
        Mat us = unsharp_mask(lap, 1, 1.0 - amount, 0.3);
        dump_mat(pf + "unsharp", us);
        Mat dst;
        us.convertTo(dst, CV_8U, 255);
        dump_mat(pf + "frame", dst);

        // the real thing, same inputs
        Mat gf1, gf2, real;
        vector<Point2f> realPts;
        morph_images(c1, c2, c1, c2, gabor2, gf1, gf2, real, Mat(), realPts, pts1, pts2, shapeRatio, maskRatio, 0.0);
        if (real.size() != dst.size() || norm(real, dst, NORM_INF) != 0 || realPts != morphed) {
            fprintf(stderr, "FATAL: staged sequence differs from morph_images() at ratio %zu\n", k / 2);
            exit(2);
        }
        fprintf(stderr, "bstage ratio %zu (%g,%g): %zu tris, ok\n", k / 2, shapeRatio, maskRatio, tri.size());
    }
}

// ---------------------------------------------------------------------------------------------
// ORB detect / describe / Hamming match on given u8 images.
// ---------------------------------------------------------------------------------------------
static void dump_kps(const string& name, const vector<KeyPoint>& kps) {
    vector<float> buf;
    for (auto& k : kps) { buf.push_back(k.pt.x); buf.push_back(k.pt.y); buf.push_back(k.size); buf.push_back(k.angle);
                          buf.push_back(k.response); buf.push_back((float)k.octave); buf.push_back((float)k.class_id); }
    write_raw(name, "f32", { (int)kps.size(), 7 }, buf.data(), buf.size() * 4);
}

static void dump_orb() {
    Mat g1 = read_mat("g1"), g2 = read_mat("g2");
    vector<double> nf = read_f64("nfeatures");
    {   // primitives
        vector<KeyPoint> fk;
        FAST(g1, fk, 20, true);
        dump_kps("fast_g1", fk);
        Mat r1;
        resize(g1, r1, Size(cvRound(g1.cols / 1.2f), cvRound(g1.rows / 1.2f)), 0, 0, INTER_LINEAR_EXACT);
        dump_mat("resize_g1", r1);
        Mat gb;
        GaussianBlur(g1, gb, Size(7, 7), 2, 2, BORDER_REFLECT_101);
        dump_mat("gauss7_g1", gb);
    }
    for (size_t i = 0; i < nf.size(); ++i) {
        string pf = "n" + std::to_string((int)nf[i]) + "_";
        Ptr<ORB> orb = ORB::create((int)nf[i]);
        vector<KeyPoint> k1, k2;
        orb->detect(g1, k1);
        orb->detect(g2, k2);
        dump_kps(pf + "kp1", k1);
        dump_kps(pf + "kp2", k2);
        Mat d1, d2;
        vector<KeyPoint> k1c = k1, k2c = k2;
        orb->compute(g1, k1c, d1);
        orb->compute(g2, k2c, d2);
        if (k1c.size() != k1.size() || k2c.size() != k2.size()) { fprintf(stderr, "FATAL: compute dropped keypoints\n"); exit(2); }
        dump_mat(pf + "desc1", d1);
        dump_mat(pf + "desc2", d2);
        BFMatcher bf(NORM_HAMMING);
        vector<DMatch> ms;
        bf.match(d1, d2, ms);
        vector<int> buf;
        for (auto& m : ms) { buf.push_back(m.queryIdx); buf.push_back(m.trainIdx); buf.push_back((int)m.distance); }
        write_raw(pf + "bfmatch", "i32", { (int)ms.size(), 3 }, buf.data(), buf.size() * 4);
        // the descriptor-matching sketch of src/experiments.hpp:14-144: 2-NN both ways, ratio test, symmetry test
        vector<vector<DMatch>> m12, m21;
        bf.knnMatch(d1, d2, m12, 2);
        bf.knnMatch(d2, d1, m21, 2);
        auto dump_knn = [&](const string& name, const vector<vector<DMatch>>& mm) {
            vector<int> kb;
            for (auto& v : mm) for (int k = 0; k < 2; ++k) {
                kb.push_back(k < (int)v.size() ? v[k].trainIdx : -1);
                kb.push_back(k < (int)v.size() ? (int)v[k].distance : -1);
            }
            write_raw(name, "i32", { (int)mm.size(), 4 }, kb.data(), kb.size() * 4);
        };
        dump_knn(pf + "knn12", m12);
        dump_knn(pf + "knn21", m21);
        int rm1 = poppy::ratioTest(m12), rm2 = poppy::ratioTest(m21);
        vector<int> keep1, keep2;
        for (auto& v : m12) keep1.push_back(v.size() > 1 ? 1 : 0);
        for (auto& v : m21) keep2.push_back(v.size() > 1 ? 1 : 0);
        write_raw(pf + "ratio12", "i32", { (int)keep1.size() }, keep1.data(), keep1.size() * 4);
        write_raw(pf + "ratio21", "i32", { (int)keep2.size() }, keep2.data(), keep2.size() * 4);
        vector<DMatch> sym;
        poppy::symmetryTest(m12, m21, sym);
        vector<int> sb;
        for (auto& m : sym) { sb.push_back(m.queryIdx); sb.push_back(m.trainIdx); sb.push_back((int)m.distance); }
        if (sb.empty()) sb.assign(3, 0);
        write_raw(pf + "sym", "i32", { (int)sym.size(), 3 }, sb.data(), sym.size() * 12);
        fprintf(stderr, "  knn: ratio test removed %d / %d, %zu symmetric matches\n", rm1, rm2, sym.size());
        fprintf(stderr, "orb nfeatures=%d: %zu / %zu keypoints\n", (int)nf[i], k1.size(), k2.size());
    }
}

// ---------------------------------------------------------------------------------------------
// Point matcher (make_distance_map / morph_distance / Matcher::match / prepare) on given point sets.
// ---------------------------------------------------------------------------------------------
static void dump_match() {
    vector<Point2f> p1 = read_pts("pts1"), p2 = read_pts("pts2");
    vector<double> cfg = read_f64("cfg");   // w, h, tolerance
    int w = (int)cfg[0], h = (int)cfg[1];
    Settings::instance().match_tolerance = cfg[2];

    auto dm = make_distance_map(p1, p2);
    {
        vector<double> buf;
        for (auto& e : dm) { buf.push_back(e.first); buf.push_back(e.second.first.x); buf.push_back(e.second.first.y);
                             buf.push_back(e.second.second.x); buf.push_back(e.second.second.y); }
        write_raw("distanceMap", "f64", { (int)dm.size(), 5 }, buf.data(), buf.size() * 8);
    }
    vector<Point2f> f1 = p1, f2 = p2;
    filter_invalid_points(f1, f2, w, h);
    if (f1.size() > f2.size()) f1.resize(f2.size()); else f2.resize(f1.size());
    dump_pts("filtered1", f1); dump_pts("filtered2", f2);
    double md = (double)morph_distance(f1, f2, w, h);
    dump_f64("initialMorphDist", { md });

    Mat img1(h, w, CV_8UC3, Scalar(0, 0, 0)), img2(h, w, CV_8UC3, Scalar(0, 0, 0));
    Features ft1, ft2;
    Matcher m(img1, img2, ft1, ft2);
    m.initialMorphDist_ = md;
    Mat cc1 = img1.clone(), cc2 = img2.clone();
    vector<Point2f> q1 = f1, q2 = f2;
    m.prepare(cc1, cc2, q1, q2);
    dump_pts("prepared1", q1); dump_pts("prepared2", q2);

    // morph()'s printed distance after clip/uniq (poppy.hpp:142-159)
    vector<Point2f> u1, u2;
    clip_points(q1, w, h); make_uniq(q1, u1);
    clip_points(q2, w, h); make_uniq(q2, u2);
    if (u1.size() > u2.size()) u1.resize(u2.size()); else u2.resize(u1.size());
    dump_f64("finalMorphDist", { (double)morph_distance(u1, u2, w, h) });
    fprintf(stderr, "match: %zu in, %zu filtered, %zu prepared\n", p1.size(), f1.size(), q1.size());
}

// ---------------------------------------------------------------------------------------------
// Full A stage + whole morph() on a BGR pair (goldens for the "next" rows and end-to-end frames).
// ---------------------------------------------------------------------------------------------
struct CollectWriter { vector<Mat> frames; void write(Mat& m) { frames.push_back(m.clone()); } };

static void dump_astage() {
    Mat a = read_mat("img1"), b = read_mat("img2");
    vector<double> cfg = read_f64("cfg");   // nframes, phase, levels
    int nframes = (int)cfg[0]; double phase = cfg[1]; int levels = (int)cfg[2];
    const int flags = cfg.size() > 3 ? (int)cfg[3] : 0;    // bit 0: Settings::enable_auto_align, bit 1: Settings::enable_radial_mask
    const bool autoAlign = (flags & 1) != 0, radialMask = (flags & 2) != 0;
    poppy::init(false, nframes, 1.0, autoAlign, radialMask, false, false, false, 30, levels, "FFV1", false, 8);

    Extractor ex(a, b);
    if (radialMask) {                          // the mask Extractor::foreground multiplies in (src/extractor.cpp:178-185), by the same calls
        Mat radial = Mat::ones(a.rows, a.cols, CV_32F), radialMaskFloat;
        draw_radial_gradiant(radial);
        radial.convertTo(radialMaskFloat, CV_32F, 1.0 / 255.0);
        dump_mat("radialMask", radialMaskFloat);
    }
    auto gf = ex.prepareFeatures();
    dump_mat("goodFeatures1", gf.first); dump_mat("goodFeatures2", gf.second);

    // Extractor::keypoints() (src/extractor.cpp:33-83) step by step, to expose the ORB inputs
    Mat dft1, dft2;
    double d1 = dft_detail2(gf.first, dft1), d2 = dft_detail2(gf.second, dft2);
    double detail = 255.0 / std::max(d1, d2);
    int nfeatures = (int)(Settings::instance().max_keypoints * detail);
    dump_f64("detail", { d1, d2, detail, (double)nfeatures });
    Mat g[2];
    for (int i = 0; i < 2; ++i) {
        Mat trip, us, gg;
        triple_channel(i ? gf.second : gf.first, trip);
        trip.convertTo(trip, CV_32F, 1.0 / 255.0);
        us = unsharp_mask(trip, 2, 6, 0.1);
        cvtColor(us, us, COLOR_BGR2GRAY);
        Mat radial = draw_radial_gradiant2(us.cols, us.rows);
        gabor_filter(us, gg, 16, 31, 5, 2, 0.04, CV_PI / 4);
        if (i == 0) { dump_mat("us1", us); dump_mat("gb1", gg); }
        multiply(gg, us, gg);
        multiply(gg, radial, gg);
        gg.convertTo(gg, CV_8U, 255.0);
        equalizeHist(gg, gg);
        g[i] = gg;
        if (i == 0) dump_mat("radial", radial);
    }
    dump_mat("g1", g[0]); dump_mat("g2", g[1]);
    {
        Mat k17 = getGaussianKernel(17, 2, CV_32F);          // unsharp_mask(.., 2, ..) -> GaussianBlur(Size(0,0), 2) on float
        dump_mat("gauss17", k17);
        Mat gk0 = getGaborKernel(Size(31, 31), 5, 0 * 11.0, 2, 0.04, CV_PI / 4, CV_32F);
        Mat gk5 = getGaborKernel(Size(31, 31), 5, 5 * 11.0, 2, 0.04, CV_PI / 4, CV_32F);
        Mat gs3 = getGaborKernel(Size(13, 13), 5, 3 * 11.0, 10, 0.04, CV_PI / 4, CV_32F);
        dump_mat("gaborK31_0", gk0); dump_mat("gaborK31_5", gk5); dump_mat("gaborK13_3", gs3);
    }
    auto kps = ex.keypoints();
    {
        Ptr<ORB> orb = ORB::create(nfeatures);
        vector<KeyPoint> k1, k2;
        orb->detect(g[0], k1); orb->detect(g[1], k2);
        bool same = k1.size() == kps.first.size() && k2.size() == kps.second.size();
        for (size_t i = 0; same && i < k1.size(); ++i) same = k1[i].pt == kps.first[i].pt;
        for (size_t i = 0; same && i < k2.size(); ++i) same = k2[i].pt == kps.second[i].pt;
        if (!same) { fprintf(stderr, "FATAL: staged keypoints() differs from Extractor::keypoints()\n"); exit(2); }
    }
    dump_kps("kp1", kps.first); dump_kps("kp2", kps.second);

    Features ft1, ft2;
    Mat t1 = a.clone(), t2 = b.clone();
    Matcher matcher(t1, t2, ft1, ft2);
    Mat c1, c2;
    vector<Point2f> s1, s2;
    matcher.find(c1, c2, s1, s2);
    dump_pts("found1", s1); dump_pts("found2", s2);
    dump_f64("initialMorphDist", { matcher.initialMorphDist_ });
    if (autoAlign) dump_mat("corrected2", c2);
    Mat c2f, gabor2;
    c2.convertTo(c2f, CV_32F, 1.0 / 255);
    gabor_filter(c2f, gabor2);
    dump_mat("gabor2", gabor2);
    if (!s1.empty()) {
        matcher.prepare(c1, c2, s1, s2);
        dump_pts("prepared1", s1); dump_pts("prepared2", s2);
        // the value poppy::morph prints before --distance exits (src/poppy.hpp:142-159), from the same reference helpers
        vector<Point2f> q1 = s1, q2 = s2, u1, u2;
        clip_points(q1, a.cols, a.rows); make_uniq(q1, u1);
        clip_points(q2, a.cols, a.rows); make_uniq(q2, u2);
        if (u1.size() > u2.size()) u1.resize(u2.size()); else u2.resize(u1.size());
        dump_f64("printedMorphDist", { morph_distance(u1, u2, a.cols, a.rows) });
    }

    // whole call
    Mat cc1, cc2; CollectWriter out;
    poppy::morph(a, b, cc1, cc2, phase, false, out);
    for (size_t i = 0; i < out.frames.size(); ++i) dump_mat("frame" + std::to_string(i), out.frames[i]);
    // extra single-frame phase-mode calls: init(numberOfFrames = 1), morph(phase = t)  (what one frame of the sharded job is)
    for (size_t k = 4; k < cfg.size(); ++k) {
        poppy::init(false, 1, 1.0, autoAlign, radialMask, false, false, false, 30, levels, "FFV1", false, 8);
        Mat e1, e2; CollectWriter eo;
        poppy::morph(a, b, e1, e2, cfg[k], false, eo);
        if (eo.frames.size() != 1) { fprintf(stderr, "FATAL: phase call wrote %zu frames\n", eo.frames.size()); exit(2); }
        dump_mat("phase" + std::to_string(k - 4) + "_frame", eo.frames[0]);
    }
    fprintf(stderr, "astage: nfeatures=%d kps=%zu/%zu found=%zu frames=%zu\n", nfeatures, kps.first.size(), kps.second.size(), s1.size(), out.frames.size());
}

// ---------------------------------------------------------------------------------------------
// Known-answer style primitives on random inputs (used to pin single routines in isolation).
// ---------------------------------------------------------------------------------------------
static void dump_prims() {
    Mat pts = read_mat("subdiv_pts");          // n x 2 f32, inside rect
    vector<double> rc = read_f64("subdiv_rect"); // w, h
    Subdiv2D sd(Rect(0, 0, (int)rc[0], (int)rc[1]));
    for (int i = 0; i < pts.rows; ++i) sd.insert(Point2f(pts.at<float>(i, 0), pts.at<float>(i, 1)));
    vector<Vec6f> tl; sd.getTriangleList(tl);
    write_raw("subdiv_tris", "f32", { (int)tl.size(), 6 }, tl.data(), tl.size() * sizeof(Vec6f));

    Mat polys = read_mat("polys");             // n x 6 i32
    Mat img = Mat::zeros((int)rc[1], (int)rc[0], CV_32SC1);
    for (int i = 0; i < polys.rows; ++i) {
        vector<Point> p = { Point(polys.at<int>(i, 0), polys.at<int>(i, 1)), Point(polys.at<int>(i, 2), polys.at<int>(i, 3)), Point(polys.at<int>(i, 4), polys.at<int>(i, 5)) };
        fillConvexPoly(img, p, Scalar(i + 1));
    }
    dump_mat("polys_map", img);

    Mat src = read_mat("remap_src"), mx = read_mat("remap_mx"), my = read_mat("remap_my"), dst;
    remap(src, dst, mx, my, INTER_LINEAR);
    dump_mat("remap_dst", dst);

    Mat mats = read_mat("mats33");             // n x 9 f32
    vector<float> inv(mats.total());
    for (int i = 0; i < mats.rows; ++i) {
        Mat m(3, 3, CV_32F, mats.ptr<float>(i));
        Mat iv = m.inv();
        memcpy(&inv[i * 9], iv.data, 36);
    }
    write_raw("mats33_inv", "f32", { mats.rows, 3, 3 }, inv.data(), inv.size() * 4);

    // Gaussian taps the hot path uses (soft-float getGaussianKernel): baked into oracle/ and the kernels
    dump_mat("gauss_k9_s1", getGaussianKernel(9, 1.0, CV_32F));
    dump_mat("gauss_k17_s2", getGaussianKernel(17, 2.0, CV_32F));
    dump_mat("gauss_k7_s2", getGaussianKernel(7, 2.0, CV_32F));
    dump_mat("gauss_k23_s1", getGaussianKernel(23, 1.0, CV_32F));
}

// ---- pre-ORB filter chain, first part: Extractor::foreground (src/extractor.cpp:136-229) ------------------------------
// Restates foregroundMask()/foreground() for ONE image with every intermediate written out, then checks the result
// against the real Extractor::foreground (both of its outputs, since it processes two images the same way).
static void dump_fstage() {
    Mat a = read_mat("img1");
    poppy::init(false, 60, 1.0, false, false, false, false, false, 30, 64, "FFV1", false, 8);
    Mat grey;
    cvtColor(a, grey, COLOR_BGR2GRAY);
    dump_mat("grey", grey);

    const size_t iterations = 12;
    Mat fgMask = Mat::zeros(grey.rows, grey.cols, grey.type());
    Mat last = grey.clone();
    Mat fgMaskBlur, med, flow;
    auto bs = createBackgroundSubtractorMOG2();
    bs->apply(grey, flow);
    dump_mat("flow0", flow);
    fgMask += (flow * (1.0 / (iterations / 2.0)));
    dump_mat("acc0", fgMask);
    for (size_t i = 0; i < 12; ++i) {
        medianBlur(last, med, i * 8 + 1);
        bs->apply(med, flow);
        fgMask += (flow * (1.0 / (iterations / 2.0)));
        Mat acc = fgMask.clone();
        GaussianBlur(fgMask, fgMaskBlur, { 23, 23 }, 1);
        fgMask = fgMaskBlur.clone();
        last = med.clone();
        char nm[32];
        snprintf(nm, sizeof nm, "med%zu", i + 1); dump_mat(nm, med);
        snprintf(nm, sizeof nm, "flow%zu", i + 1); dump_mat(nm, flow);
        snprintf(nm, sizeof nm, "acc%zu", i + 1); dump_mat(nm, acc);
        snprintf(nm, sizeof nm, "blur%zu", i + 1); dump_mat(nm, fgMask);
    }
    Mat greyF, maskF, finalMask;
    grey.convertTo(greyF, CV_32F, 1.0 / 255.0);
    fgMask.convertTo(maskF, CV_32F, 1.0 / 255.0);
    maskF.copyTo(finalMask);
    int logBase = 20;
    Mat logMask(finalMask.size(), CV_32F);
    logMask = Scalar::all(logBase);
    log(logMask, logMask);
    dump_mat("log20", logMask(Rect(0, 0, 1, 1)));
    Mat lin = finalMask * (logBase - 1.0) + 1.0;
    dump_mat("lin", lin);
    log(lin, finalMask);
    dump_mat("logged", finalMask);
    divide(finalMask, logMask, finalMask);
    dump_mat("finalMask", finalMask);
    Mat masked;
    multiply(greyF, finalMask, masked);
    masked.convertTo(masked, CV_8U, 255.0);
    dump_mat("masked", masked);
    Mat fg;
    equalizeHist(masked, fg);
    dump_mat("foreground", fg);

    Extractor ex(a, a);
    Mat f1, f2;
    ex.foreground(f1, f2);
    if (countNonZero(f1 != fg) || countNonZero(f2 != fg)) { fprintf(stderr, "FATAL: staged foreground differs from Extractor::foreground\n"); exit(2); }
    fprintf(stderr, "fstage ok (staged == Extractor::foreground)\n");
}

// dft_detail2 (src/experiments.hpp:305-318) on a grey image: the value and a few raw spectrum samples
static void dump_detail() {
    Mat g = read_mat("gray");
    Mat spec;
    double d = dft_detail2(g, spec);
    dump_f64("detail", { d });
    Mat head = spec(Rect(0, 0, std::min(spec.cols, 64), std::min(spec.rows, 8))).clone();
    dump_mat("spectrum_head", head);
    if (getenv("DUMP_FULL_SPECTRUM")) dump_mat("spectrum_full", spec);
    {   // the raw transform, as dft_spectrum computes it (src/experiments.hpp:267-279)
        Mat padded;
        int m = getOptimalDFTSize(g.rows), n = getOptimalDFTSize(g.cols);
        copyMakeBorder(g, padded, 0, m - g.rows, 0, n - g.cols, BORDER_CONSTANT, Scalar::all(0));
        Mat planes[] = {Mat_<float>(padded), Mat::zeros(padded.size(), CV_32F)};
        Mat complexI;
        merge(planes, 2, complexI);
        dft(complexI, complexI);
        dump_mat("dft", complexI);
        Mat row0;
        dft(Mat_<Vec2f>(complexI.size(), Vec2f(0, 0)), row0);    // (keeps the planner warm; not used)
        Mat in0;
        merge(planes, 2, in0);
        Mat r0 = in0.row(0).clone(), r0out;
        dft(r0, r0out, DFT_ROWS);
        dump_mat("dft_row0", r0out);
    }
}

// blur_margin (src/util.cpp:574-602): the CLI's padding of an image into the union canvas
static void dump_margin() {
    Mat a = read_mat("img");
    vector<double> cfg = read_f64("cfg");          // union width, height
    Mat out;
    blur_margin(a, Size((int)cfg[0], (int)cfg[1]), out);
    dump_mat("padded", out);
}

// Auto-align (Matcher::autoAlign, src/matcher.cpp:133-244) and its pieces (Transformer, src/transformer.cpp; Procrustes),
// each run on fresh copies of the same inputs, plus the OpenCV primitives they are made of.
static void dump_align() {
    Mat img2 = read_mat("img2");
    vector<Point2f> p1 = read_pts("pts1"), p2 = read_pts("pts2");
    vector<double> cfg = read_f64("cfg");
    int w = (int)cfg[0], h = (int)cfg[1];
    Mat aff = read_mat("aff");
    Transformer tr(w, h);
    {   // warpAffine as Transformer uses it
        Mat o;
        tr.translate(img2, o, Point2f(5, -3)); dump_mat("wa_t1", o);
        tr.translate(img2, o, Point2f(-40, 17)); dump_mat("wa_t2", o);
        Point2f c1(w / 3.f, h / 2.f), c2(w * 0.61f, h * 0.27f);
        dump_mat("rm1", getRotationMatrix2D(c1, 12.0 + 1.0 / 3.0, 1.0));
        dump_mat("rm2", getRotationMatrix2D(c2, -100.0 / 3.0, 1.0));
        tr.rotate(img2, o, c1, 12.0 + 1.0 / 3.0); dump_mat("wa_r1", o);
        tr.rotate(img2, o, c2, -100.0 / 3.0); dump_mat("wa_r2", o);
        warpAffine(img2, o, aff, img2.size()); dump_mat("wa_a1", o);
        Mat inplace = img2.clone();
        warpAffine(inplace, inplace, aff, inplace.size());
        if (cv::norm(inplace, o, NORM_INF) != 0) { fprintf(stderr, "FATAL: in-place warpAffine differs\n"); exit(2); }
    }
    {   // primitives of Procrustes on the raw point sets
        Mat X(p1), Y(p2);
        Scalar mu = cv::mean(X);
        dump_f64("prim_mean", { mu[0], mu[1] });
        Mat sq; pow(X, 2.0, sq);
        Scalar ss = sum(sq);
        dump_f64("prim_sumsq", { ss[0], ss[1] });
        Mat A = X.reshape(1).t() * Y.reshape(1);
        dump_mat("prim_gemm", A);
        Mat An = A / 1000.0;
        Mat U, s, Vt;
        SVDecomp(An, s, U, Vt);
        dump_mat("prim_svd_in", An); dump_mat("prim_svd_s", s); dump_mat("prim_svd_u", U); dump_mat("prim_svd_vt", Vt);
        Mat T; cv::transform(X, T, Vt);
        dump_mat("prim_transform", T.reshape(1));
        Mat P = getPerspectiveTransform(p1.data(), p2.data());
        dump_mat("prim_persp", P);
        vector<Point2f> q; perspectiveTransform(p1, q, P);
        dump_pts("prim_persp_pts", q);
    }
    dump_f64("md0", { (double)morph_distance(p1, p2, w, h) });
    {
        Procrustes procr(true, false);
        float err = procr.procrustes(p1, p2);
        dump_mat("pc_rotation", procr.rotation);
        dump_f64("pc_scalars", { (double)procr.scale, (double)procr.error, (double)err });
        dump_pts("pc_yprime", procr.yPrimeAsVector());
        dump_mat("pc_translation", procr.translation);
    }
    {
        Mat c2 = img2.clone(); vector<Point2f> a = p1, b = p2;
        double d = tr.retranslate(c2, a, b);
        dump_mat("rt_img", c2); dump_pts("rt_pts2", b); dump_f64("rt_dist", { d });
    }
    {
        Mat c2 = img2.clone(); vector<Point2f> a = p1, b = p2;
        double d = tr.reprocrustes(c2, a, b);
        dump_mat("rp_img", c2); dump_pts("rp_pts2", b); dump_f64("rp_dist", { d });
    }
    {
        Mat c2 = img2.clone(); vector<Point2f> a = p1, b = p2;
        double d = tr.rerotate(c2, a, b);
        dump_mat("rr_img", c2); dump_pts("rr_pts2", b); dump_f64("rr_dist", { d });
    }
    {
        Mat img1(h, w, CV_8UC3, Scalar(0, 0, 0));
        Features ft1, ft2;
        Matcher m(img1, img2, ft1, ft2);
        Mat c1 = img1.clone(), c2 = img2.clone(); vector<Point2f> a = p1, b = p2;
        m.autoAlign(c1, c2, a, b);
        dump_mat("aa_img", c2); dump_pts("aa_pts1", a); dump_pts("aa_pts2", b);
        dump_f64("aa_dist", { (double)morph_distance(a, b, w, h) });
    }
}

// img2*phase + img1*(1.0-phase) exactly as poppy::morph's no-match fallback writes it (src/poppy.hpp:125-134)
static void dump_dissolve() {
    Mat img1 = read_mat("img1"), img2 = read_mat("img2");
    vector<double> ph = read_f64("phases");
    for (size_t k = 0; k < ph.size(); ++k) {
        const double phase = ph[k];
        Mat blend = ((img2 * phase) + (img1 * (1.0 - phase)));
        dump_mat("blend" + std::to_string(k), blend);
    }
}

static void dump_logcheck() {      // cv::log / cv::magnitude on given floats (debugging aid for the oracle)
    Mat x = read_mat("x"), y;
    log(x, y);
    dump_mat("logx", y);
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <bstage|orb|match|astage|prims> <case_dir>\n", argv[0]); return 1; }
    string mode = argv[1], dir = argv[2];
    g_in = dir + "/in"; g_out = dir + "/out";
    mkdir(g_out.c_str(), 0755);
    setNumThreads(1);
    if (mode == "bstage") dump_bstage();
    else if (mode == "orb") dump_orb();
    else if (mode == "match") dump_match();
    else if (mode == "astage") dump_astage();
    else if (mode == "align") dump_align();
    else if (mode == "prims") dump_prims();
    else if (mode == "fstage") dump_fstage();
    else if (mode == "detail") dump_detail();
    else if (mode == "logcheck") dump_logcheck();
    else if (mode == "margin") dump_margin();
    else if (mode == "dissolve") dump_dissolve();
    else { fprintf(stderr, "unknown mode\n"); return 1; }
    return 0;
}
