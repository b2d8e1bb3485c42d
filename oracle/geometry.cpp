// oracle/geometry.cpp — point hygiene, Delaunay, triangle raster, per-triangle affine matrices.
// TEST INFRASTRUCTURE (see oracle.h).  Restates:
//   src/util.cpp:453-460 (clip_points), :541-548 (make_uniq, LessPointOp util.hpp:88-92)
//   src/algo.cpp:50-58 (morph_points), :60-81 (get_triangle_indices), :83-93 (make_triangler_points),
//   :95-106 (paint_triangles), :108-144 (solve_homography / morph_homography)
//   OCV/imgproc/src/subdivision2d.cpp:45-537,756-785 (Subdiv2D insert / getTriangleList)
//   OCV/imgproc/src/drawing.cpp:80-297,1093-1255 (clipLine, LineIterator, Line, FillConvexPoly)
//   OCV/core/src/lapack.cpp:760-763,965-993,1044 (3x3 inverse), matmul.simd.hpp:827-841 (3x3 gemm),
//   matmul.simd.hpp:1934-1948 (scaleAdd_32f), matrix_expressions.cpp:326-357,1293-1320,1677-1703
#include "oracle.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <climits>
#include <set>

namespace oracle {

int cv_round(double v) {
    if (!(v >= -2147483648.5 && v < 2147483647.5)) return INT_MIN;   // cvtsd2si "integer indefinite"
    return (int)std::nearbyint(v);                                    // default mode: half-to-even
}
int cv_round_f(float v) { return cv_round((double)v); }

int border_reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = len - 1 - (p - len) - 1;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

void clip_points(std::vector<Pt>& pts, int cols, int rows) {
    for (Pt& p : pts) {
        p.x = p.x > cols ? (float)(cols - 1) : p.x;   // note: x == cols survives (util.cpp:455)
        p.y = p.y > rows ? (float)(rows - 1) : p.y;
        p.x = p.x < 0 ? 0.f : p.x;
        p.y = p.y < 0 ? 0.f : p.y;
    }
}

namespace {
struct PtLess {
    bool operator()(const Pt& a, const Pt& b) const { return a.x < b.x || (a.x == b.x && a.y < b.y); }
};
}

void make_uniq(const std::vector<Pt>& pts, std::vector<Pt>& out) {
    std::set<Pt, PtLess> seen;
    for (const Pt& p : pts)
        if (seen.insert(p).second) out.push_back(p);
}

void morph_points(const std::vector<Pt>& a, const std::vector<Pt>& b, std::vector<Pt>& out, float s) {
    out.resize(a.size());
    for (size_t i = 0; i < a.size(); ++i) {
        // (1.0 - s) is double; products and the sum are double; one rounding on the store
        out[i].x = (float)((1.0 - s) * a[i].x + s * b[i].x);
        out[i].y = (float)((1.0 - s) * a[i].y + s * b[i].y);
    }
}

// ------------------------------------------------------------------------------------------------
// Delaunay.  Directed edge e = 4*q + r; rot(e,k) stays inside quad q.  onext[e] is the next edge
// counter-clockwise around org(e).  Slot 0 of both tables is a dummy, as in the reference, so that
// edge/vertex numbers — and with them the order of getTriangleList — come out identical.
// ------------------------------------------------------------------------------------------------
namespace {

struct Mesh {
    std::vector<int> onext, origin;          // 4 entries per quad
    struct Vert { float x, y; int first, kind; };
    std::vector<Vert> verts;
    int freeQuad = 0, freeVert = 0, recent = 0;
    float x0, y0, x1, y1;

    static int rot(int e, int k) { return (e & ~3) + ((e + k) & 3); }
    static int sym(int e) { return e ^ 2; }
    // generalised neighbour: step through the table at rotation a, then rotate the result by b
    int hop(int e, int a, int b) const { int t = onext[(e & ~3) + ((e + a) & 3)]; return rot(t, b); }
    int oprev(int e) const { return hop(e, 1, 1); }      // PREV_AROUND_ORG  0x11
    int dprev(int e) const { return hop(e, 3, 3); }      // PREV_AROUND_DST  0x33
    int lnext(int e) const { return hop(e, 3, 1); }      // NEXT_AROUND_LEFT 0x13
    int lprev(int e) const { return hop(e, 0, 2); }      // PREV_AROUND_LEFT 0x20
    int org(int e) const { return origin[e]; }
    int dst(int e) const { return origin[sym(e)]; }

    int alloc_quad() {
        if (freeQuad <= 0) {
            onext.insert(onext.end(), 4, 0);
            origin.insert(origin.end(), 4, 0);
            freeQuad = (int)(onext.size() / 4) - 1;
        }
        int e = freeQuad * 4;
        freeQuad = onext[e + 1];
        onext[e] = e; onext[e + 1] = e + 3; onext[e + 2] = e + 2; onext[e + 3] = e + 1;
        origin[e] = origin[e + 1] = origin[e + 2] = origin[e + 3] = 0;
        return e;
    }
    int alloc_vert(float x, float y) {
        if (freeVert == 0) {
            verts.push_back({0.f, 0.f, 0, -1});
            freeVert = (int)verts.size() - 1;
        }
        int v = freeVert;
        freeVert = verts[v].first;
        verts[v] = {x, y, 0, 0};
        return v;
    }
    void splice(int a, int b) {
        int an = onext[a], bn = onext[b];
        int ar = rot(an, 1), br = rot(bn, 1);
        std::swap(onext[a], onext[b]);
        std::swap(onext[ar], onext[br]);
    }
    void set_ends(int e, int o, int d) {
        origin[e] = o; origin[sym(e)] = d;
        verts[o].first = e; verts[d].first = sym(e);
    }
    int connect(int a, int b) {
        int e = alloc_quad();
        splice(e, lnext(a));
        splice(sym(e), b);
        set_ends(e, dst(a), org(b));
        return e;
    }
    void flip(int e) {
        int s = sym(e), a = oprev(e), b = oprev(s);
        splice(e, a); splice(s, b);
        set_ends(e, dst(a), dst(b));
        splice(e, lnext(a)); splice(s, lnext(b));
    }
    void drop(int e) {
        splice(e, oprev(e));
        int s = sym(e);
        splice(s, oprev(s));
        int q = e >> 2;
        onext[4 * q] = 0; onext[4 * q + 1] = freeQuad;
        freeQuad = q;
    }
    static double area2(float ax, float ay, float bx, float by, float cx, float cy) {
        return ((double)bx - ax) * ((double)cy - ay) - ((double)by - ay) * ((double)cx - ax);
    }
    int side(float px, float py, int e) const {     // >0: p is right of e
        const Vert& o = verts[org(e)]; const Vert& d = verts[dst(e)];
        double a = area2(px, py, d.x, d.y, o.x, o.y);
        return (a > 0) - (a < 0);
    }
    static int in_circle(const Vert& p, const Vert& a, const Vert& b, const Vert& c) {
        const double eps = FLT_EPSILON * 0.125;
        double v = ((double)a.x * a.x + (double)a.y * a.y) * area2(b.x, b.y, c.x, c.y, p.x, p.y);
        v -= ((double)b.x * b.x + (double)b.y * b.y) * area2(a.x, a.y, c.x, c.y, p.x, p.y);
        v += ((double)c.x * c.x + (double)c.y * c.y) * area2(a.x, a.y, b.x, b.y, p.x, p.y);
        v -= ((double)p.x * p.x + (double)p.y * p.y) * area2(a.x, a.y, b.x, b.y, c.x, c.y);
        return v > eps ? 1 : v < -eps ? -1 : 0;
    }

    void init(int w, int h) {
        float big = 3.f * std::max(w, h);
        x0 = 0.f; y0 = 0.f; x1 = (float)w; y1 = (float)h;
        onext.assign(4, 0); origin.assign(4, 0);
        verts.assign(1, {0.f, 0.f, 0, -1});
        freeQuad = 0; freeVert = 0;
        int A = alloc_vert(big, 0.f), B = alloc_vert(0.f, big), C = alloc_vert(-big, -big);
        int ab = alloc_quad(), bc = alloc_quad(), ca = alloc_quad();
        set_ends(ab, A, B); set_ends(bc, B, C); set_ends(ca, C, A);
        splice(ab, sym(ca)); splice(bc, sym(ab)); splice(ca, sym(bc));
        recent = ab;
    }

    enum { LOC_ERROR = -2, LOC_INSIDE = 0, LOC_VERTEX = 1, LOC_EDGE = 2 };
    int locate(float px, float py, int& edgeOut, int& vertOut) {
        int limit = (int)onext.size();
        int e = recent, where = LOC_ERROR, vert = 0;
        int rc = side(px, py, e);
        if (rc > 0) { e = sym(e); rc = -rc; }
        for (int i = 0; i < limit; ++i) {
            int on = onext[e], dp = dprev(e);
            int r_on = side(px, py, on), r_dp = side(px, py, dp);
            if (r_dp > 0) {
                if (r_on > 0 || (r_on == 0 && rc == 0)) { where = LOC_INSIDE; break; }
                rc = r_on; e = on;
            } else if (r_on > 0) {
                if (r_dp == 0 && rc == 0) { where = LOC_INSIDE; break; }
                rc = r_dp; e = dp;
            } else if (rc == 0 && side(verts[dst(on)].x, verts[dst(on)].y, e) >= 0) {
                e = sym(e);
            } else { rc = r_on; e = on; }
        }
        recent = e;
        if (where == LOC_INSIDE) {
            const Vert& o = verts[org(e)]; const Vert& d = verts[dst(e)];
            double t1 = std::fabs(px - o.x); t1 += std::fabs(py - o.y);
            double t2 = std::fabs(px - d.x); t2 += std::fabs(py - d.y);
            double t3 = std::fabs(o.x - d.x); t3 += std::fabs(o.y - d.y);
            if (t1 < FLT_EPSILON) { where = LOC_VERTEX; vert = org(e); e = 0; }
            else if (t2 < FLT_EPSILON) { where = LOC_VERTEX; vert = dst(e); e = 0; }
            else if ((t1 < t3 || t2 < t3) && std::fabs(area2(px, py, o.x, o.y, d.x, d.y)) < FLT_EPSILON) { where = LOC_EDGE; vert = 0; }
        }
        if (where == LOC_ERROR) { e = 0; vert = 0; }
        edgeOut = e; vertOut = vert;
        return where;
    }

    bool insert(float px, float py) {
        if (px < x0 || py < y0 || px >= x1 || py >= y1) return false;     // reference: CV_StsOutOfRange
        int cur = 0, vert = 0;
        int where = locate(px, py, cur, vert);
        if (where == LOC_ERROR) return false;
        if (where == LOC_VERTEX) return true;
        if (where == LOC_EDGE) {
            int doomed = cur;
            recent = cur = oprev(cur);
            drop(doomed);
        }
        int np = alloc_vert(px, py);
        int base = alloc_quad();
        int first = org(cur);
        set_ends(base, first, np);
        splice(base, cur);
        do {
            base = connect(cur, sym(base));
            cur = oprev(base);
        } while (dst(cur) != first);
        cur = oprev(base);
        int limit = (int)onext.size();
        for (int i = 0; i < limit; ++i) {
            int t = oprev(cur);
            int td = dst(t), co = org(cur), cd = dst(cur);
            if (side(verts[td].x, verts[td].y, cur) > 0 &&
                in_circle(verts[co], verts[td], verts[cd], verts[np]) < 0) {
                flip(cur);
                cur = oprev(cur);
            } else if (co == first) break;
            else cur = lprev(onext[cur]);
        }
        return true;
    }

    bool inside(int v) const {
        const Vert& p = verts[v];
        return x0 <= p.x && p.x < x0 + (x1 - x0) && y0 <= p.y && p.y < y0 + (y1 - y0);
    }
    void triangles(std::vector<float>& out) const {
        out.clear();
        int total = (int)onext.size();
        std::vector<char> used(total, 0);
        for (int i = 4; i < total; i += 2) {
            if (used[i]) continue;
            int ea = i, a = org(ea);
            if (!inside(a)) continue;
            int eb = lnext(ea), b = org(eb);
            if (!inside(b)) continue;
            int ec = lnext(eb), c = org(ec);
            if (!inside(c)) continue;
            used[ea] = used[eb] = used[ec] = 1;
            const Vert &A = verts[a], &B = verts[b], &C = verts[c];
            out.insert(out.end(), {A.x, A.y, B.x, B.y, C.x, C.y});
        }
    }
};

}  // namespace

bool delaunay_triangles(int w, int h, const std::vector<Pt>& pts, std::vector<float>& tri6) {
    Mesh m;
    m.init(w, h);
    for (const Pt& p : pts)
        if (!m.insert(p.x, p.y)) return false;
    m.triangles(tri6);
    return true;
}

void triangle_indices(const std::vector<float>& tri6, const std::vector<Pt>& points, std::vector<int>& idx3) {
    idx3.clear();
    size_t nt = tri6.size() / 6;
    for (size_t t = 0; t < nt; ++t) {
        int id[3];
        bool ok = true;
        for (int k = 0; k < 3 && ok; ++k) {
            float x = tri6[t * 6 + 2 * k], y = tri6[t * 6 + 2 * k + 1];
            size_t j = 0;
            while (j < points.size() && !(points[j].x == x && points[j].y == y)) ++j;   // std::find, first hit
            if (j == points.size()) ok = false; else id[k] = (int)j;
        }
        if (ok) idx3.insert(idx3.end(), id, id + 3);
    }
}

void triangle_int_points(const std::vector<int>& idx3, const std::vector<Pt>& points, std::vector<IPt>& out) {
    out.resize(idx3.size());
    for (size_t i = 0; i < idx3.size(); ++i)
        out[i] = {(int)points[idx3[i]].x, (int)points[idx3[i]].y};      // Point(float,float): truncation
}

// ------------------------------------------------------------------------------------------------
// Raster: outline (8-connected Bresenham after clipping) + scanline fill in 16.16 fixed point.
// ------------------------------------------------------------------------------------------------
namespace {

bool clip_segment(int64_t W, int64_t H, int64_t& x1, int64_t& y1, int64_t& x2, int64_t& y2) {
    if (W <= 0 || H <= 0) return false;
    int64_t right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (x2 - x1) / (y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (x2 - x1) / (y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (y2 - y1) / (x2 - x1));
                x1 = a; c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (y2 - y1) / (x2 - x1));
                x2 = a; c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

void draw_line8(ImageI& img, IPt p1, IPt p2, int32_t value) {
    int W = img.w, H = img.h;
    if ((unsigned)p1.x >= (unsigned)W || (unsigned)p2.x >= (unsigned)W ||
        (unsigned)p1.y >= (unsigned)H || (unsigned)p2.y >= (unsigned)H) {
        int64_t ax = p1.x, ay = p1.y, bx = p2.x, by = p2.y;
        if (!clip_segment(W, H, ax, ay, bx, by)) return;
        p1 = {(int)ax, (int)ay}; p2 = {(int)bx, (int)by};
    }
    int sx = 1, sy = 1;
    int dx = p2.x - p1.x, dy = p2.y - p1.y;
    if (dx < 0) { dx = -dx; dy = -dy; p1 = p2; }      // leftToRight: always walk with x increasing
    if (dy < 0) { dy = -dy; sy = -1; }
    bool steep = dy > dx;
    if (steep) { std::swap(dx, dy); std::swap(sx, sy); }
    // after the swap (sx, sy) are the steps along (major, minor) axes
    int err = dx - (dy + dy), plusDelta = dx + dx, minusDelta = -(dy + dy);
    int count = dx + 1;
    int x = p1.x, y = p1.y;
    // major step: every iteration; minor step: when err < 0
    int majX = steep ? 0 : sx, majY = steep ? sx : 0;       // note sx/sy were swapped with the deltas
    int minX = steep ? sy : 0, minY = steep ? 0 : sy;
    for (int i = 0; i < count; ++i) {
        img.d[(size_t)y * W + x] = value;
        bool neg = err < 0;
        err += minusDelta + (neg ? plusDelta : 0);
        x += majX + (neg ? minX : 0);
        y += majY + (neg ? minY : 0);
    }
}

}  // namespace

void fill_triangle(ImageI& img, const IPt* v, int32_t value) { fill_convex(img, v, 3, value); }

// FillConvexPoly for any convex polygon (drawing.cpp:1093-1255); Poppy only ever passes triangles, the general form exists so
// that OpenCV's own known-answer test (imgproc/test/test_drawing.cpp:432-462: a 4-point polygon) can pin this routine.
void fill_convex(ImageI& img, const IPt* v, int npts, int32_t value) {
    const int SHIFT = 16;
    const int64_t ONE = 1 << SHIFT;
    int W = img.w, H = img.h;
    struct Edge { int idx, di; int64_t x, dx; int ye; } edge[2];
    int imin = 0;
    int64_t xmin = v[0].x, xmax = v[0].x, ymin = v[0].y, ymax = v[0].y;
    IPt prev = v[npts - 1];
    for (int i = 0; i < npts; ++i) {
        IPt p = v[i];
        if (p.y < ymin) { ymin = p.y; imin = i; }
        ymax = std::max<int64_t>(ymax, p.y);
        xmax = std::max<int64_t>(xmax, p.x);
        xmin = std::min<int64_t>(xmin, p.x);
        draw_line8(img, prev, p, value);
        prev = p;
    }
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= W || (int)ymin >= H) return;
    ymax = std::min<int64_t>(ymax, H - 1);
    int y = (int)ymin, edges = npts;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (int i = 0; i < 2; ++i) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    int ty = v[idx].y;
                    if (ty > y) {
                        int64_t xs = (int64_t)v[idx0].x << SHIFT, xe = (int64_t)v[idx].x << SHIFT;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + (ONE >> 1)) >> SHIFT);
            int xx2 = (int)((edge[right].x + (ONE >> 1)) >> SHIFT);
            if (xx2 >= 0 && xx1 < W) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= W) xx2 = W - 1;
                int32_t* row = img.row(y);
                for (int x = xx1; x <= xx2; ++x) row[x] = value;
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

void paint_triangles(ImageI& img, const std::vector<IPt>& tris) {
    for (size_t t = 0; t < tris.size() / 3; ++t) fill_triangle(img, &tris[t * 3], (int32_t)(t + 1));
}

// ------------------------------------------------------------------------------------------------
// 3x3 algebra
// ------------------------------------------------------------------------------------------------
bool invert33(const float* m, float* out) {
#define M(r, c) m[(r) * 3 + (c)]
    double d = M(0, 0) * ((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) -
               M(0, 1) * ((double)M(1, 0) * M(2, 2) - (double)M(1, 2) * M(2, 0)) +
               M(0, 2) * ((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0));
    if (d == 0.) { for (int i = 0; i < 9; ++i) out[i] = 0.f; return false; }
    d = 1. / d;
    double t[9];
    t[0] = ((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) * d;
    t[1] = ((double)M(0, 2) * M(2, 1) - (double)M(0, 1) * M(2, 2)) * d;
    t[2] = ((double)M(0, 1) * M(1, 2) - (double)M(0, 2) * M(1, 1)) * d;
    t[3] = ((double)M(1, 2) * M(2, 0) - (double)M(1, 0) * M(2, 2)) * d;
    t[4] = ((double)M(0, 0) * M(2, 2) - (double)M(0, 2) * M(2, 0)) * d;
    t[5] = ((double)M(0, 2) * M(1, 0) - (double)M(0, 0) * M(1, 2)) * d;
    t[6] = ((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0)) * d;
    t[7] = ((double)M(0, 1) * M(2, 0) - (double)M(0, 0) * M(2, 1)) * d;
    t[8] = ((double)M(0, 0) * M(1, 1) - (double)M(0, 1) * M(1, 0)) * d;
#undef M
    for (int i = 0; i < 9; ++i) out[i] = (float)t[i];
    return true;
}

static void homog(const IPt* p, float* m) {   // 3 x npts(=3): rows x, y, 1
    for (int i = 0; i < 3; ++i) { m[i] = (float)p[i].x; m[3 + i] = (float)p[i].y; m[6 + i] = 1.f; }
}

void solve_homography(const IPt* src1, const IPt* src2, float* H) {
    float P1[9], P2[9], P1i[9];
    homog(src1, P1); homog(src2, P2);
    invert33(P1, P1i);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            float t = P2[r * 3] * P1i[c] + P2[r * 3 + 1] * P1i[3 + c] + P2[r * 3 + 2] * P1i[6 + c];
            H[r * 3 + c] = (float)(t * 1.0 + 0.f * 0.0);          // d = t*alpha + c*beta, alpha=1, beta=0
        }
}

void morph_homography(const float* H, float ratio, float* M1, float* M2) {
    float Hi[9];
    invert33(H, Hi);
    // M1 = eye*(1.0 - r) + H*r      -> scaleAdd(H, r, diag(float(1.0 - r)))     [cv::add when r == 1]
    // M2 = eye*r + Hinv*(1.0 - r)   -> scaleAdd(Hinv, 1.0 - r, diag(float(r)))  [cv::add when r == 0]
    double a1 = 1.0 - (double)ratio, b1 = (double)ratio;
    float d1 = (float)a1, s1 = (float)b1;
    double a2 = (double)ratio, b2 = 1.0 - (double)ratio;
    float d2 = (float)a2, s2 = (float)b2;
    for (int i = 0; i < 9; ++i) {
        float e1 = (i % 4 == 0) ? d1 : 0.f, e2 = (i % 4 == 0) ? d2 : 0.f;
        M1[i] = (b1 == 1.0) ? e1 + H[i] : H[i] * s1 + e1;
        M2[i] = (b2 == 1.0) ? e2 + Hi[i] : Hi[i] * s2 + e2;
    }
}

}  // namespace oracle
