// oracle/match.cpp — Poppy's point matcher (greedy nearest neighbour + statistical threshold) and the
// morph-distance metric it is normalised with.  TEST INFRASTRUCTURE (see oracle.h).  Restates:
//   src/util.cpp:251-279 (calculate_sum_mean_and_sd, add_corners), :351-383 (make_distance_map),
//   :385-431 (morph_distance), :473-496 (filter_invalid_points)
//   src/matcher.cpp:118-131 (find: filter + initialMorphDist_), :246-310 (match), :312-332 (prepare)
//   OCV/imgproc/src/convhull.cpp:48-312 (convexHull, Sklansky), approx.cpp:476-671 (approxPolyDP, closed),
//   OCV/imgproc/src/shapedescr.cpp:308-338 (contourArea)
#include "oracle.h"
#include <algorithm>
#include <cmath>
#include <map>

namespace oracle {

void make_distance_map(const std::vector<Pt>& p1, const std::vector<Pt>& p2, std::vector<DistPair>& out) {
    std::multimap<double, std::pair<Pt, Pt>> dm;
    std::vector<Pt> c1 = p1, c2 = p2;
    Pt nopoint{-1, -1};
    for (Pt& a : c1) {
        double best = std::numeric_limits<double>::max();
        Pt* closest = &nopoint;
        for (Pt& b : c2) {
            if (b.x == -1 && b.y == -1) continue;
            double d = hypotf(b.x - a.x, b.y - a.y);         // util.cpp is `using namespace std`: hypot(float,float) is the FLOAT overload
            if (d < best) { best = d; closest = &b; }
        }
        if (closest->x == -1 && closest->y == -1) continue;
        double d = hypotf(closest->x - a.x, closest->y - a.y);
        dm.insert({d, {a, *closest}});
        closest->x = -1; closest->y = -1;
    }
    out.clear();
    for (auto& e : dm) out.push_back({e.first, e.second.first, e.second.second});
}

void filter_invalid_points(std::vector<Pt>& p1, std::vector<Pt>& p2, int cols, int rows) {
    for (size_t i = 0; i < p1.size(); ++i) {
        const Pt& p = p1[i];
        if (p.x < 0 || p.x > cols || p.y < 0 || p.y > rows) { p1.erase(p1.begin() + i); p2.erase(p2.begin() + i); --i; }
    }
    for (size_t i = 0; i < p2.size(); ++i) {
        const Pt& p = p2[i];
        if (p.x < 0 || p.x > cols || p.y < 0 || p.y > rows) { p1.erase(p1.begin() + i); p2.erase(p2.begin() + i); --i; }
    }
}

namespace {

int sgn(double v) { return (v > 0) - (v < 0); }

int sklansky(const std::vector<const Pt*>& arr, int start, int end, int* stack, int nsign, int sign2) {
    int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int stacksize = 3;
    if (start == end || (arr[start]->x == arr[end]->x && arr[start]->y == arr[end]->y)) { stack[0] = start; return 1; }
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        float cury = arr[pcur]->y, nexty = arr[pnext]->y;
        float by = nexty - cury;
        if (sgn(by) != nsign) {
            float ax = arr[pcur]->x - arr[pprev]->x;
            float bx = arr[pnext]->x - arr[pcur]->x;
            float ay = cury - arr[pprev]->y;
            double convexity = (double)ay * bx - (double)ax * by;
            if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pcur = pnext; pnext += incr;
                stack[stacksize] = pnext; stacksize++;
            } else if (pprev == start) {
                pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
            } else {
                stack[stacksize - 2] = pnext;
                pcur = pprev; pprev = stack[stacksize - 4]; stacksize--;
            }
        } else { pnext += incr; stack[stacksize - 1] = pnext; }
    }
    return --stacksize;
}

// convexHull(points, hull) with default clockwise = false, returnPoints = true
void convex_hull(const std::vector<Pt>& pts, std::vector<Pt>& hull) {
    int total = (int)pts.size();
    hull.clear();
    if (!total) return;
    std::vector<const Pt*> ptr(total);
    for (int i = 0; i < total; ++i) ptr[i] = &pts[i];
    std::sort(ptr.begin(), ptr.end(), [](const Pt* a, const Pt* b) {
        if (a->x != b->x) return a->x < b->x;
        if (a->y != b->y) return a->y < b->y;
        return a < b;
    });
    int miny = 0, maxy = 0;
    for (int i = 1; i < total; ++i) {
        float y = ptr[i]->y;
        if (ptr[miny]->y > y) miny = i;
        if (ptr[maxy]->y < y) maxy = i;
    }
    std::vector<int> stackv(total + 2), hullbuf(total);
    int* stack = stackv.data();
    int nout = 0;
    const Pt* data0 = pts.data();
    if (ptr[0]->x == ptr[total - 1]->x && ptr[0]->y == ptr[total - 1]->y) hullbuf[nout++] = 0;
    else {
        int* tl = stack; int tlc = sklansky(ptr, 0, maxy, tl, -1, 1);
        int* tr = stack + tlc; int trc = sklansky(ptr, total - 1, maxy, tr, -1, -1);
        std::swap(tl, tr); std::swap(tlc, trc);                     // !clockwise
        for (int i = 0; i < tlc - 1; ++i) hullbuf[nout++] = (int)(ptr[tl[i]] - data0);
        for (int i = trc - 1; i > 0; --i) hullbuf[nout++] = (int)(ptr[tr[i]] - data0);
        int stop_idx = trc > 2 ? tr[1] : tlc > 2 ? tl[tlc - 2] : -1;

        int* bl = stack; int blc = sklansky(ptr, 0, miny, bl, 1, -1);
        int* br = stack + blc; int brc = sklansky(ptr, total - 1, miny, br, 1, 1);
        if (stop_idx >= 0) {
            int check_idx = blc > 2 ? bl[1] : blc + brc > 2 ? br[2 - blc] : -1;
            if (check_idx == stop_idx || (check_idx >= 0 && ptr[check_idx]->x == ptr[stop_idx]->x && ptr[check_idx]->y == ptr[stop_idx]->y)) {
                blc = std::min(blc, 2); brc = std::min(brc, 2);
            }
        }
        for (int i = 0; i < blc - 1; ++i) hullbuf[nout++] = (int)(ptr[bl[i]] - data0);
        for (int i = brc - 1; i > 0; --i) hullbuf[nout++] = (int)(ptr[br[i]] - data0);

        if (nout >= 3) {
            int min_idx = 0, max_idx = 0, lt = 0, i;
            for (i = 1; i < nout; ++i) {
                int idx = hullbuf[i];
                lt += hullbuf[i - 1] < idx;
                if (lt > 1 && lt <= i - 2) break;
                if (idx < hullbuf[min_idx]) min_idx = i;
                if (idx > hullbuf[max_idx]) max_idx = i;
            }
            int mmdist = std::abs(max_idx - min_idx);
            if ((mmdist == 1 || mmdist == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
                int ascending = (max_idx + 1) % nout == min_idx;
                int i0 = ascending ? min_idx : max_idx, j = i0;
                if (i0 > 0) {
                    for (i = 0; i < nout; ++i) {
                        int curr = stack[i] = hullbuf[j];
                        int nj = j + 1 < nout ? j + 1 : 0;
                        int nidx = hullbuf[nj];
                        if (i < nout - 1 && (ascending != (curr < nidx))) break;
                        j = nj;
                    }
                    if (i == nout) std::copy(stack, stack + nout, hullbuf.begin());
                }
            }
        }
    }
    for (int i = 0; i < nout; ++i) hull.push_back(data0[hullbuf[i]]);
}

// approxPolyDP(curve, out, eps, closed = true) for float points
void approx_poly_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& out) {
    int count = (int)src.size();
    out.clear();
    if (!count) return;
    struct Rng { int start, end; };
    std::vector<Rng> stack;
    std::vector<Pt> dst(count);
    int new_count = 0;
    Rng slice{0, 0}, right{0, 0};
    Pt start_pt{-1000000.f, -1000000.f}, end_pt{0, 0}, pt{0, 0};
    int pos = 0;
    bool le_eps = false;
    eps *= eps;
    auto read = [&](Pt& p, int& ps) { p = src[ps]; if (++ps >= count) ps = 0; };
    right.start = 0;
    for (int i = 0; i < 3; ++i) {
        double max_dist = 0;
        pos = (pos + right.start) % count;
        read(start_pt, pos);
        for (int j = 1; j < count; ++j) {
            read(pt, pos);
            double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    if (!le_eps) {
        right.end = slice.start = pos % count;
        slice.end = right.start = (right.start + slice.start) % count;
        stack.push_back(right); stack.push_back(slice);
    } else dst[new_count++] = start_pt;

    while (!stack.empty()) {
        slice = stack.back(); stack.pop_back();
        end_pt = src[slice.end];
        pos = slice.start;
        read(start_pt, pos);
        if (pos != slice.end) {
            double max_dist = 0;
            double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                read(pt, pos);
                double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else { le_eps = true; start_pt = src[slice.start]; }
        if (le_eps) dst[new_count++] = start_pt;
        else {
            right.end = slice.end;
            slice.end = right.start;
            stack.push_back(right); stack.push_back(slice);
        }
    }
    // clean-up of (almost) straight runs
    count = new_count;
    auto readd = [&](Pt& p, int& ps) { p = dst[ps]; if (++ps >= count) ps = 0; };
    pos = count - 1;
    readd(start_pt, pos);
    int wpos = pos;
    readd(pt, pos);
    for (int i = 0; i < count && new_count > 2; ++i) {
        readd(end_pt, pos);
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        double sip = (pt.x - start_pt.x) * (end_pt.x - pt.x) + (pt.y - start_pt.y) * (end_pt.y - pt.y);   // float arithmetic
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= count) wpos = 0;
            readd(pt, pos);
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= count) wpos = 0;
        pt = end_pt;
    }
    out.assign(dst.begin(), dst.begin() + new_count);
}

double contour_area(const std::vector<Pt>& c) {
    if (c.empty()) return 0.;
    double a = 0;
    Pt prev = c.back();
    for (const Pt& p : c) { a += (double)prev.x * p.y - (double)prev.y * p.x; prev = p; }
    return std::fabs(a * 0.5);
}

}  // namespace

void convex_hull_points(const std::vector<Pt>& pts, std::vector<Pt>& hull) { convex_hull(pts, hull); }

double morph_distance(const std::vector<Pt>& p1, const std::vector<Pt>& p2, int w, int h) {
    const long double width = w, height = h;
    std::vector<DistPair> dm;
    make_distance_map(p1, p2, dm);
    std::vector<Pt> hull1, hull2, c1, c2;
    convex_hull(p1, hull1); convex_hull(p2, hull2);
    approx_poly_closed(hull1, 0.001, c1); approx_poly_closed(hull2, 0.001, c2);
    double area1 = std::fabs(contour_area(c1)), area2 = std::fabs(contour_area(c2));

    float inner1 = 0;
    for (size_t i = 0; i < p1.size(); ++i)
        for (size_t j = 0; j < p1.size(); ++j) { float vx = p1[i].x - p1[j].x, vy = p1[i].y - p1[j].y; inner1 += vx + vy; }
    inner1 = (float)((inner1 / (p1.size() * p1.size())) / (width + height));
    float inner2 = 0;
    for (size_t i = 0; i < p2.size(); ++i)
        for (size_t j = 0; j < p2.size(); ++j) { float vx = p2[i].x - p1[j].x, vy = p2[i].y - p1[j].y; inner2 += vx + vy; }   // sic: srcPoints1[j]
    inner2 = (float)((inner2 / (p2.size() * p2.size())) / (width + height));

    float total = 0;
    for (const DistPair& e : dm) total += hypotf(e.b.x - e.a.x, e.b.y - e.a.y);    // float += float
    long double ret = ((total / (dm.size())) / hypotl(width, height)) + fabs(inner1 - inner2)
                      + (fabs(area1 - area2) / (width * height)) / 3.0;
    return (double)ret;
}

// Matcher::match + Matcher::prepare.  Returns the point sets after add_corners.
void match_prepare(std::vector<Pt>& s1, std::vector<Pt>& s2, int w, int h, double tolerance, double initialMorphDist) {
    std::vector<DistPair> dm;
    make_distance_map(s1, s2, dm);
    size_t n = dm.size();
    double sum = 0.0;
    for (auto& e : dm) sum += e.d;
    double mean = sum / n;
    double sd = 0.0;
    for (auto& e : dm) sd += pow(e.d - mean, 2);
    double deviation = sqrt(sd / n);
    double total = sum;
    double density = total / (w * h);
    double area = (w * h);
    s1.clear(); s2.clear();
    if (mean == 0) {
        for (auto& e : dm) { s1.push_back(e.a); s2.push_back(e.b); }
    } else {
        double thresh = 1;
        if (tolerance != 0)
            thresh = (area * (mean / deviation) * tolerance) / ((total * sqrt(density) * (1.0 / sqrt(initialMorphDist))) / ((1 + sqrt(5)) / 2.0));
        for (auto& e : dm) {
            double r = e.d / thresh;
            if (r > 0.0 && r <= 1.0) { s1.push_back(e.a); s2.push_back(e.b); }
        }
        if (s1.empty() && !dm.empty()) { s1.push_back(dm[0].a); s2.push_back(dm[0].b); }
    }
    if (s1.size() > s2.size()) s1.resize(s2.size()); else s2.resize(s1.size());
    float fw = (float)(w - 1), fh = (float)(h - 1);
    const Pt corners[4] = {{0, 0}, {fw, 0}, {0, fh}, {fw, fh}};
    for (const Pt& c : corners) { s1.push_back(c); s2.push_back(c); }
}

}  // namespace oracle
