// frame_plan.cpp — see frame_plan.h.  Host C++ (the reference's own code for this part is host C++).
//
// Behaviour follows, step for step, what the reference computes between the clip of the point sets
// and the matrices handed to create_map:
//   src/algo.cpp:184-209 clip / uniq / morph_points            src/util.cpp:453-460,541-548
//   src/algo.cpp:60-93   triangle -> point indices (first exact match), truncation to int
//   src/algo.cpp:108-144 H = P2 * inv(P1);  M1 = (1-r) I + r H;  M2 = r I + (1-r) inv(H)
//   src/algo.cpp:154-157 create_map inverts M1 / M2 once more
//   OCV/imgproc/src/subdivision2d.cpp:276-537,756-785 incremental Delaunay + triangle enumeration
// Unlike the reference, vertices carry the index of the input point they came from, so the
// O(T*N) coordinate search of get_triangle_indices is not needed; the result is the same because
// make_uniq keeps the FIRST occurrence of every coordinate pair.
#include "frame_plan.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <set>

namespace poppy_hip {

void clip_points_ref(std::vector<P2f>& pts, int cols, int rows) {
    for (auto& p : pts) {
        if (p.x > cols) p.x = (float)(cols - 1);
        if (p.y > rows) p.y = (float)(rows - 1);
        if (p.x < 0) p.x = 0.f;
        if (p.y < 0) p.y = 0.f;
    }
}

bool invert3x3(const float* a, float* out) {
    const double a00 = a[0], a01 = a[1], a02 = a[2], a10 = a[3], a11 = a[4], a12 = a[5], a20 = a[6], a21 = a[7], a22 = a[8];
    // the determinant multiplies FLOAT leading terms with double minors (lapack.cpp:761-763)
    double det = a[0] * (a11 * a22 - a12 * a21) - a[1] * (a10 * a22 - a12 * a20) + a[2] * (a10 * a21 - a11 * a20);
    if (det == 0.) {
        std::fill(out, out + 9, 0.f);
        return false;
    }
    const double s = 1. / det;
    out[0] = (float)((a11 * a22 - a12 * a21) * s);
    out[1] = (float)((a02 * a21 - a01 * a22) * s);
    out[2] = (float)((a01 * a12 - a02 * a11) * s);
    out[3] = (float)((a12 * a20 - a10 * a22) * s);
    out[4] = (float)((a00 * a22 - a02 * a20) * s);
    out[5] = (float)((a02 * a10 - a00 * a12) * s);
    out[6] = (float)((a10 * a21 - a11 * a20) * s);
    out[7] = (float)((a01 * a20 - a00 * a21) * s);
    out[8] = (float)((a00 * a11 - a01 * a10) * s);
    return true;
}

namespace {

// Quad-edge subdivision.  Edge handle = 4*quad + rotation.  Tables are laid out and recycled exactly
// like cv::Subdiv2D's so that handles — hence the enumeration order of triangles — coincide.
class Triangulator {
public:
    Triangulator(int w, int h) : W((float)w), H((float)h) {
        quads.reserve(1024);
        quads.push_back(Quad{{0, 0, 0, 0}, {0, 0, 0, 0}});
        nodes.push_back(Node{0.f, 0.f, 0, -1, -1});
        const float far = 3.f * std::max(w, h);
        int a = add_node(far, 0.f, -1), b = add_node(0.f, far, -1), c = add_node(-far, -far, -1);
        int ab = new_quad(), bc = new_quad(), ca = new_quad();
        attach(ab, a, b); attach(bc, b, c); attach(ca, c, a);
        splice(ab, ca ^ 2); splice(bc, ab ^ 2); splice(ca, bc ^ 2);
        last = ab;
    }

    // tag = index of the source point; false when outside the rectangle or the walk fails
    bool add(float x, float y, int tag) {
        if (x < 0.f || y < 0.f || x >= W || y >= H) return false;
        int e = 0;
        int where = find(x, y, e);
        if (where == kFail) return false;
        if (where == kOnVertex) return true;
        if (where == kOnEdge) {
            int dead = e;
            last = e = prev_org(e);
            release(dead);
        }
        int v = add_node(x, y, tag);
        int spoke = new_quad();
        int first = org(e);
        attach(spoke, first, v);
        splice(spoke, e);
        do {
            spoke = bridge(e, spoke ^ 2);
            e = prev_org(spoke);
        } while (dst(e) != first);
        e = prev_org(spoke);
        const int guard = (int)quads.size() * 4;
        for (int i = 0; i < guard; ++i) {
            int t = prev_org(e);
            int td = dst(t), eo = org(e), ed = dst(e);
            if (right_of(nodes[td].x, nodes[td].y, e) > 0 && incircle(nodes[eo], nodes[td], nodes[ed], nodes[v]) < 0) {
                swap_diagonal(e);
                e = prev_org(e);
            } else if (eo == first) {
                break;
            } else {
                e = prev_left(next_org(e));
            }
        }
        return true;
    }

    // triangles in enumeration order, as source-point tags (virtual/outer vertices filtered out)
    void triangles(std::vector<int>& tags) const {
        const int total = (int)quads.size() * 4;
        std::vector<uint8_t> seen(total, 0);
        for (int e = 4; e < total; e += 2) {
            if (seen[e]) continue;
            int a = org(e);
            if (!in_rect(a)) continue;
            int e2 = next_left(e), b = org(e2);
            if (!in_rect(b)) continue;
            int e3 = next_left(e2), c = org(e3);
            if (!in_rect(c)) continue;
            seen[e] = seen[e2] = seen[e3] = 1;
            // every in-rect vertex is a user point (the three outer ones lie far outside)
            tags.push_back(nodes[a].tag); tags.push_back(nodes[b].tag); tags.push_back(nodes[c].tag);
        }
    }

private:
    struct Quad { int nx[4]; int pt[4]; };
    struct Node { float x, y; int edge, state, tag; };
    enum { kFail = -2, kInside = 0, kOnVertex = 1, kOnEdge = 2 };
    std::vector<Quad> quads;
    std::vector<Node> nodes;
    int freeQuad = 0, freeNode = 0, last = 0;
    float W, H;

    int& nx(int e) { return quads[e >> 2].nx[e & 3]; }
    int nx(int e) const { return quads[e >> 2].nx[e & 3]; }
    static int turn(int e, int r) { return (e & ~3) | ((e + r) & 3); }
    int via(int e, int pre, int post) const { return turn(nx(turn(e, pre)), post); }
    int next_org(int e) const { return nx(e); }
    int prev_org(int e) const { return via(e, 1, 1); }
    int prev_dst(int e) const { return via(e, 3, 3); }
    int next_left(int e) const { return via(e, 3, 1); }
    int prev_left(int e) const { return via(e, 0, 2); }
    int org(int e) const { return quads[e >> 2].pt[e & 3]; }
    int dst(int e) const { return quads[e >> 2].pt[(e + 2) & 3]; }
    bool in_rect(int v) const { const Node& n = nodes[v]; return 0.f <= n.x && n.x < W && 0.f <= n.y && n.y < H; }

    int new_quad() {
        if (freeQuad <= 0) {
            quads.push_back(Quad{{0, 0, 0, 0}, {0, 0, 0, 0}});
            freeQuad = (int)quads.size() - 1;
        }
        int q = freeQuad;
        freeQuad = quads[q].nx[1];
        int e = q * 4;
        quads[q] = Quad{{e, e + 3, e + 2, e + 1}, {0, 0, 0, 0}};
        return e;
    }
    void release(int e) {
        splice(e, prev_org(e));
        splice(e ^ 2, prev_org(e ^ 2));
        int q = e >> 2;
        quads[q].nx[0] = 0;
        quads[q].nx[1] = freeQuad;
        freeQuad = q;
    }
    int add_node(float x, float y, int tag) {
        if (freeNode == 0) {
            nodes.push_back(Node{0.f, 0.f, 0, -1, -1});
            freeNode = (int)nodes.size() - 1;
        }
        int v = freeNode;
        freeNode = nodes[v].edge;
        nodes[v] = Node{x, y, 0, 0, tag};
        return v;
    }
    void attach(int e, int o, int d) {
        quads[e >> 2].pt[e & 3] = o;
        quads[e >> 2].pt[(e + 2) & 3] = d;
        nodes[o].edge = e;
        nodes[d].edge = e ^ 2;
    }
    void splice(int a, int b) {
        int ra = turn(nx(a), 1), rb = turn(nx(b), 1);
        std::swap(nx(a), nx(b));
        std::swap(nx(ra), nx(rb));
    }
    int bridge(int a, int b) {
        int e = new_quad();
        splice(e, next_left(a));
        splice(e ^ 2, b);
        attach(e, dst(a), org(b));
        return e;
    }
    void swap_diagonal(int e) {
        int s = e ^ 2, a = prev_org(e), b = prev_org(s);
        splice(e, a); splice(s, b);
        attach(e, dst(a), dst(b));
        splice(e, next_left(a)); splice(s, next_left(b));
    }
    static double cross(float ax, float ay, float bx, float by, float cx, float cy) {
        return ((double)bx - ax) * ((double)cy - ay) - ((double)by - ay) * ((double)cx - ax);
    }
    int right_of(float x, float y, int e) const {
        const Node &o = nodes[org(e)], &d = nodes[dst(e)];
        double c = cross(x, y, d.x, d.y, o.x, o.y);
        return (c > 0) - (c < 0);
    }
    static int incircle(const Node& p, const Node& a, const Node& b, const Node& c) {
        const double tol = FLT_EPSILON * 0.125;
        double v = ((double)a.x * a.x + (double)a.y * a.y) * cross(b.x, b.y, c.x, c.y, p.x, p.y);
        v -= ((double)b.x * b.x + (double)b.y * b.y) * cross(a.x, a.y, c.x, c.y, p.x, p.y);
        v += ((double)c.x * c.x + (double)c.y * c.y) * cross(a.x, a.y, b.x, b.y, p.x, p.y);
        v -= ((double)p.x * p.x + (double)p.y * p.y) * cross(a.x, a.y, b.x, b.y, c.x, c.y);
        return v > tol ? 1 : v < -tol ? -1 : 0;
    }
    int find(float x, float y, int& out) {
        const int guard = (int)quads.size() * 4;
        int e = last, verdict = kFail;
        int here = right_of(x, y, e);
        if (here > 0) { e ^= 2; here = -here; }
        for (int i = 0; i < guard; ++i) {
            int on = next_org(e), dp = prev_dst(e);
            int r_on = right_of(x, y, on), r_dp = right_of(x, y, dp);
            if (r_dp > 0) {
                if (r_on > 0 || (r_on == 0 && here == 0)) { verdict = kInside; break; }
                here = r_on; e = on;
            } else if (r_on > 0) {
                if (r_dp == 0 && here == 0) { verdict = kInside; break; }
                here = r_dp; e = dp;
            } else if (here == 0 && right_of(nodes[dst(on)].x, nodes[dst(on)].y, e) >= 0) {
                e ^= 2;
            } else {
                here = r_on; e = on;
            }
        }
        last = e;
        if (verdict == kInside) {
            const Node &o = nodes[org(e)], &d = nodes[dst(e)];
            double d_org = std::fabs(x - o.x); d_org += std::fabs(y - o.y);
            double d_dst = std::fabs(x - d.x); d_dst += std::fabs(y - d.y);
            double len = std::fabs(o.x - d.x); len += std::fabs(o.y - d.y);
            if (d_org < FLT_EPSILON || d_dst < FLT_EPSILON) { verdict = kOnVertex; e = 0; }
            else if ((d_org < len || d_dst < len) && std::fabs(cross(x, y, o.x, o.y, d.x, d.y)) < FLT_EPSILON) verdict = kOnEdge;
        }
        out = verdict == kFail ? 0 : e;
        return verdict;
    }
};

struct P2fLess {
    bool operator()(const P2f& a, const P2f& b) const { return a.x < b.x || (a.x == b.x && a.y < b.y); }
};

inline void homogeneous(const int* xy, float* m) {
    for (int i = 0; i < 3; ++i) { m[i] = (float)xy[2 * i]; m[3 + i] = (float)xy[2 * i + 1]; m[6 + i] = 1.f; }
}

}  // namespace

// Runs FillConvexPoly's edge walk (drawing.cpp:1164-1252) for every triangle and records what it decides: the rows
// painted and each chain's (first row, x at that row, dx per row) segments.  All divisions of the walk happen here.
static void build_raster(FramePlan& plan, int w, int h) {
    const int T = plan.n_tris;
    plan.raster.assign(T, RasterTri{});
    plan.work.clear();
    for (int t = 0; t < T; ++t) { plan.work.push_back(t); plan.work.push_back(-1); }      // outlines first
    for (int t = 0; t < T; ++t) {
        const int* v = &plan.tri_xy[(size_t)t * 6];
        const int vx[3] = {v[0], v[2], v[4]}, vy[3] = {v[1], v[3], v[5]};
        RasterTri& r = plan.raster[t];
        int imin = 0, xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
        for (int i = 0; i < 3; ++i) {
            if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
            ymax = std::max(ymax, vy[i]); xmax = std::max(xmax, vx[i]); xmin = std::min(xmin, vx[i]);
        }
        r.ymin = ymin; r.ystop = ymin;
        if (!(xmax < 0 || ymax < 0 || xmin >= w || ymin >= h)) {
            ymax = std::min(ymax, h - 1);
            int eidx[2] = {imin, imin}, eye[2] = {ymin, ymin}, nseg[2] = {0, 0};
            const int edi[2] = {1, 2};
            int edges = 3, y = ymin;
            for (; y <= ymax; ++y) {
                for (int i = 0; i < 2; ++i) {
                    if (y < eye[i]) continue;
                    int idx0 = eidx[i], idx = (idx0 + edi[i]) % 3;
                    for (; edges-- > 0;) {
                        const int ty = vy[idx];
                        if (ty > y) {
                            const long long xs = (long long)vx[idx0] << 16, xe = (long long)vx[idx] << 16;
                            const int k = i * 2 + std::min(nseg[i], 1);
                            r.ybeg[k] = y; r.ex[k] = xs;
                            r.edx[k] = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                            ++nseg[i];
                            eye[i] = ty; eidx[i] = idx;
                            break;
                        }
                        idx0 = idx; idx = (idx + edi[i]) % 3;
                    }
                }
                if (edges < 0) break;
            }
            r.ystop = y;
            r.n0 = nseg[0]; r.n1 = nseg[1];
        }
        const int rows = r.ystop - r.ymin;
        const int chunks = (rows + kPlanRasterRows - 1) / kPlanRasterRows;
        for (int k = 0; k < chunks; ++k) { plan.work.push_back(t); plan.work.push_back(k); }
    }
}

// clipLine (drawing.cpp:80-130) on 64-bit coordinates, with its double arithmetic and truncations
static bool clip_line_ref(long long W, long long H, long long& x1, long long& y1, long long& x2, long long& y2) {
    const long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a; c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a; c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// The three outline segments of every triangle in the form the fused warp kernel evaluates row by row.
static void build_outlines(FramePlan& plan, int w, int h) {
    const int T = plan.n_tris;
    plan.outline.assign((size_t)T * 3, OutlineSeg{0, 0, -1, 0});
    for (int t = 0; t < T; ++t) {
        const int* v = &plan.tri_xy[(size_t)t * 6];
        for (int e = 0; e < 3; ++e) {
            const int ia = e == 0 ? 2 : e - 1, ib = e == 0 ? 0 : e;
            long long ax = v[2 * ia], ay = v[2 * ia + 1], bx = v[2 * ib], by = v[2 * ib + 1];
            if ((unsigned long long)ax >= (unsigned long long)w || (unsigned long long)bx >= (unsigned long long)w ||
                (unsigned long long)ay >= (unsigned long long)h || (unsigned long long)by >= (unsigned long long)h) {
                if (!clip_line_ref(w, h, ax, ay, bx, by)) continue;
            }
            long long dx = bx - ax, dy = by - ay, x0 = ax, y0 = ay;
            if (dx < 0) { dx = -dx; dy = -dy; x0 = bx; y0 = by; }
            int flags = 0;
            if (dy < 0) { dy = -dy; flags |= 2; }
            if (dy > dx) flags |= 1;
            const long long major = (flags & 1) ? dy : dx, minor = (flags & 1) ? dx : dy;
            plan.outline[(size_t)t * 3 + e] = OutlineSeg{(int)x0, (int)y0, (int)major, (int)minor | (flags << 24)};
        }
    }
}

bool build_tile_bins(FramePlan& plan, int w, int h, int tile_w, int tile_h, size_t max_entries) {
    const int T = plan.n_tris;
    const int tiles_x = (w + tile_w - 1) / tile_w, tiles_y = (h + tile_h - 1) / tile_h;
    plan.tile_w = tile_w; plan.tile_h = tile_h; plan.bins_ok = false;
    plan.tile_off.assign((size_t)tiles_x * tiles_y + 1, 0);
    plan.tile_tris.clear();
    if (T > 65535) return false;
    struct Box { int tx0, tx1, ty0, ty1; };
    std::vector<Box> box(T);
    size_t total = 0;
    for (int t = 0; t < T; ++t) {
        const int* v = &plan.tri_xy[(size_t)t * 6];
        int x0 = std::min({v[0], v[2], v[4]}) - 1, x1 = std::max({v[0], v[2], v[4]}) + 1;
        int y0 = std::min({v[1], v[3], v[5]}) - 1, y1 = std::max({v[1], v[3], v[5]}) + 1;
        x0 = std::max(x0, 0); y0 = std::max(y0, 0); x1 = std::min(x1, w - 1); y1 = std::min(y1, h - 1);
        if (x1 < x0 || y1 < y0) { box[t] = Box{1, 0, 1, 0}; continue; }
        box[t] = Box{x0 / tile_w, x1 / tile_w, y0 / tile_h, y1 / tile_h};
        for (int ty = box[t].ty0; ty <= box[t].ty1; ++ty)
            for (int tx = box[t].tx0; tx <= box[t].tx1; ++tx) ++plan.tile_off[(size_t)ty * tiles_x + tx + 1];
        total += (size_t)(box[t].tx1 - box[t].tx0 + 1) * (box[t].ty1 - box[t].ty0 + 1);
        if (total > max_entries) return false;
    }
    plan.max_tile_entries = 0;
    for (size_t i = 1; i < plan.tile_off.size(); ++i) {
        plan.max_tile_entries = std::max(plan.max_tile_entries, plan.tile_off[i]);
        plan.tile_off[i] += plan.tile_off[i - 1];
    }
    plan.tile_tris.resize(total);
    std::vector<int> fill(plan.tile_off.begin(), plan.tile_off.end() - 1);
    for (int t = 0; t < T; ++t)                            // ascending t: every tile's list comes out in painter's order
        for (int ty = box[t].ty0; ty <= box[t].ty1; ++ty)
            for (int tx = box[t].tx0; tx <= box[t].tx1; ++tx) plan.tile_tris[fill[(size_t)ty * tiles_x + tx]++] = (uint16_t)t;
    plan.bins_ok = true;
    return true;
}

void unique_points_ref(const std::vector<P2f>& pts, std::vector<P2f>& out) {
    std::set<P2f, P2fLess> seen;
    out.clear();
    for (const P2f& p : pts)
        if (seen.insert(p).second) out.push_back(p);
}

int plan_frame(int w, int h, const std::vector<P2f>& src1, const std::vector<P2f>& src2, double shape_ratio, FramePlan& plan) {
    const size_t n = src1.size();
    std::vector<P2f> a = src1, b = src2;
    clip_points_ref(a, w, h);
    clip_points_ref(b, w, h);
    const float s = (float)shape_ratio;                       // morph_points takes a float (algo.cpp:50)
    plan.morphed.resize(n);
    for (size_t i = 0; i < n; ++i) {
        plan.morphed[i].x = (float)((1.0 - s) * a[i].x + s * b[i].x);
        plan.morphed[i].y = (float)((1.0 - s) * a[i].y + s * b[i].y);
    }
    clip_points_ref(plan.morphed, w, h);

    Triangulator tri(w, h);
    std::set<P2f, P2fLess> seen;
    for (size_t i = 0; i < n; ++i)
        if (seen.insert(plan.morphed[i]).second && !tri.add(plan.morphed[i].x, plan.morphed[i].y, (int)i)) return -3;

    plan.idx3.clear();
    tri.triangles(plan.idx3);
    const int T = (int)plan.idx3.size() / 3;
    plan.n_tris = T;
    plan.tri_xy.resize((size_t)T * 6);
    plan.M1.resize((size_t)T * 9); plan.M2.resize((size_t)T * 9);
    plan.inv1.resize((size_t)T * 9); plan.inv2.resize((size_t)T * 9);

    const double wI1 = 1.0 - (double)s, wH1 = (double)s;     // M1 = I*(1.0-r) + H*r
    const double wI2 = (double)s, wH2 = 1.0 - (double)s;     // M2 = I*r + inv(H)*(1.0-r)
    const float dI1 = (float)wI1, fH1 = (float)wH1, dI2 = (float)wI2, fH2 = (float)wH2;
    for (int t = 0; t < T; ++t) {
        int c1[6], c2[6];
        for (int k = 0; k < 3; ++k) {
            int id = plan.idx3[t * 3 + k];
            c1[2 * k] = (int)a[id].x; c1[2 * k + 1] = (int)a[id].y;
            c2[2 * k] = (int)b[id].x; c2[2 * k + 1] = (int)b[id].y;
            plan.tri_xy[t * 6 + 2 * k] = (int)plan.morphed[id].x;
            plan.tri_xy[t * 6 + 2 * k + 1] = (int)plan.morphed[id].y;
        }
        float P1[9], P2[9], P1i[9], Hm[9], Hi[9];
        homogeneous(c1, P1); homogeneous(c2, P2);
        invert3x3(P1, P1i);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                float acc = P2[r * 3] * P1i[c] + P2[r * 3 + 1] * P1i[3 + c] + P2[r * 3 + 2] * P1i[6 + c];
                Hm[r * 3 + c] = (float)(acc * 1.0 + 0.f * 0.0);
            }
        invert3x3(Hm, Hi);
        float* M1 = &plan.M1[(size_t)t * 9];
        float* M2 = &plan.M2[(size_t)t * 9];
        for (int i = 0; i < 9; ++i) {
            const bool diag = (i == 0 || i == 4 || i == 8);
            const float e1 = diag ? dI1 : 0.f, e2 = diag ? dI2 : 0.f;
            M1[i] = wH1 == 1.0 ? e1 + Hm[i] : Hm[i] * fH1 + e1;      // cv::add when the weight is 1, else scaleAdd
            M2[i] = wH2 == 1.0 ? e2 + Hi[i] : Hi[i] * fH2 + e2;
        }
        invert3x3(M1, &plan.inv1[(size_t)t * 9]);
        invert3x3(M2, &plan.inv2[(size_t)t * 9]);
    }
    build_raster(plan, w, h);
    build_outlines(plan, w, h);
    return 0;
}

namespace {
// What the tiled warp kernels assume of a matrix over the pixels that use it — the box [x0, x1] x [y0, y1]: every entry finite
// and moderate, the denominator z = m6 x + m7 y + m8 of one sign and within [2^-20, 2^20] (the hardware division sequence then
// needs no range fix-up).  z is affine in (x, y): its extremes over the box are at the corners.
bool warp_matrix_ok(const float* m, int x0, int x1, int y0, int y1, bool& zero) {
    zero = true;
    for (int i = 0; i < 9; ++i) {
        if (!std::isfinite(m[i]) || std::fabs(m[i]) > 1099511627776.f) return false;
        if (m[i] != 0.f) zero = false;
        // rows 0 and 1 reach the kernel scaled by 32 (pack_warp_records): exact as long as no product is subnormal before the scaling
        if (i < 6 && m[i] != 0.f && std::fabs(m[i]) < 0x1p-100f) return false;
    }
    if (zero) return true;
    double zlo = 0, zhi = 0;
    for (int k = 0; k < 4; ++k) {
        const double x = (k & 1) ? x1 : x0, y = (k & 2) ? y1 : y0;
        const double z = (double)m[6] * x + (double)m[7] * y + (double)m[8];
        zlo = k ? std::min(zlo, z) : z; zhi = k ? std::max(zhi, z) : z;
    }
    const double kMin = 1.0 / 1048576.0, kMax = 1048576.0;
    return (zlo >= kMin && zhi <= kMax) || (zhi <= -kMin && zlo >= -kMax);
}
}  // namespace

// tri_xy (optional): the triangles' integer corners, 6 per triangle — a record is only used by pixels of its triangle's raster,
// which lies inside the corners' bounding box; without it the whole image is assumed.
bool pack_warp_records(const float* inv1, const float* inv2, int n_tris, int w, int h, float* rec, const int* tri_xy) {
    static const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    bool ok = true;
    for (int t = -1; t < n_tris; ++t, rec += kWarpRecordFloats) {
        const float* a = t < 0 ? ident : inv1 + (size_t)t * 9;
        const float* b = t < 0 ? ident : inv2 + (size_t)t * 9;
        int x0 = 0, x1 = w - 1, y0 = 0, y1 = h - 1;
        if (t >= 0 && tri_xy) {
            const int* v = tri_xy + (size_t)t * 6;
            x0 = std::max(0, std::min(std::min(v[0], v[2]), v[4]) - 1); x1 = std::min(w - 1, std::max(std::max(v[0], v[2]), v[4]) + 1);
            y0 = std::max(0, std::min(std::min(v[1], v[3]), v[5]) - 1); y1 = std::min(h - 1, std::max(std::max(v[1], v[3]), v[5]) + 1);
            if (x0 > x1 || y0 > y1) { x0 = x1 = std::min(std::max(x0, 0), w - 1); y0 = y1 = std::min(std::max(y0, 0), h - 1); }   // wholly outside: paints nothing
        }
        bool za, zb;
        ok = warp_matrix_ok(a, x0, x1, y0, y1, za) && ok;
        ok = warp_matrix_ok(b, x0, x1, y0, y1, zb) && ok;
        // The numerator rows carry remap's sub-pixel scale: cvRound(32 (m0 x + m1 y + m2) / z) = cvRound((32 m0 x + 32 m1 y + 32 m2) / z) bit for bit — a
        // power of two commutes with every rounding of the expression (no product is subnormal: warp_matrix_ok) —, which saves the kernels the
        // multiplication by 32 per coordinate pair (round 4: the fused warp kernel is bound by its arithmetic, profiles/r04_notes.md section 2).
        rec[0] = 32.f * a[0]; rec[1] = 32.f * a[3]; rec[2] = 32.f * a[1]; rec[3] = 32.f * a[4]; rec[4] = 32.f * a[2]; rec[5] = 32.f * a[5];
        rec[6] = 32.f * b[0]; rec[7] = 32.f * b[3]; rec[8] = 32.f * b[1]; rec[9] = 32.f * b[4]; rec[10] = 32.f * b[2]; rec[11] = 32.f * b[5];
        rec[12] = a[6]; rec[13] = b[6]; rec[14] = a[7]; rec[15] = b[7];
        rec[16] = za ? 0.00001f : a[8]; rec[17] = zb ? 0.00001f : b[8];
        rec[18] = rec[19] = 0.f;
    }
    return ok;
}

}  // namespace poppy_hip
