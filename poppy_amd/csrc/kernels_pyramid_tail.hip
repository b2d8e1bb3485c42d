// kernels_pyramid_tail.hip — every small level of the Laplacian blend in ONE workgroup, driven by a per-geometry tap table.
//
//   src/blend.hpp:25-77 (mask pyramid, Laplacian build, per-level mix, collapse) on
//   OCV/imgproc/src/pyramids.cpp:745-900 (pyrDown), :903-1005 (pyrUp), :1260-1304, :1367-1406
//
// Levels first..levels (a few hundred pixels at most, then ~50 single-pixel levels with pyramid_levels = 64) live in LDS.
// The kernel is a single workgroup on the chained frames' critical path, so its time is a chain of dependent steps, and a
// step's time is the number of instructions ONE wave executes for one output: nothing here is throughput.
//
// Round 1/2 form: every output derived its taps from the level geometry (divisions by the row width, reflections,
// association split points): ~250-300 instructions per output, 1.4-2.6 us per level step, 27.5 us per launch
// (tools/micro/tail_probe.hip).  The geometry is fixed for the life of a pair, so now the HOST writes one 16-byte descriptor
// per output once per pair (build_pyr_tail_plan): the five source rows / columns of a pyrDown output and its two association
// flags; the three rows / columns, the edge form and the row parity of a pyrUp output.  The kernel stages the descriptors in
// LDS beside the levels; an output is then "unpack, load taps, the reference's expression tree, store".
//
// The single-pixel levels: pyrDown of a 1x1 image is v -> f(v) with f a fixed rounding pattern, and on the floats of an image
// it reaches a bitwise fixed point within a few applications (the ties of v*6 round to even, which clears low mantissa
// bits).  Once f(v) == v bitwise for all seven chains (L and R per channel, mask) every deeper level is identical, so the
// walk stops there: exact by construction, data-dependent only in how early it stops.  Going up, the Laplacian residual is
// the same value r for all levels past that point, and  cur -> pyrUp(cur) + r  is iterated until IT is a bitwise fixed point
// (or the range ends); the remaining few levels run as before.  53 + 53 dependent level steps become ~4 + ~4.
#include "kernels.h"
#include "pyramid_device.h"
#include <mutex>

namespace poppy_hip {

namespace {

constexpr int kTailThreads = 1024;
constexpr int kResMax = 3 * 258;

__device__ __forceinline__ float down_1x1(float v) {                    // pyrDown of a single pixel: every tap is the pixel
    const float h = v * 6.f + (v + v) * 4.f + v + v;
    return (h * 6.f + (h + h) * 4.f + h + h) * (1.f / 256);
}
__device__ __forceinline__ float up_1x1(float v) {                      // pyrUp of a single pixel to a single pixel
    const float h = v * 8.f;
    return (h + h * 6.f + h) * (1.f / 64);
}

// pyrDown output from its descriptor: w0 = y0 | y1 << 10 | y2 << 20 | hBody << 30 | vBody << 31, w1 = y3 | y4 << 10 | c2 << 20,
// w2 = four signed bytes: the other columns relative to c2.  Rows are row numbers of the source level, columns element offsets.
__device__ __forceinline__ float down_from_desc(const float* __restrict__ lvl, int stride, uint4 d) {
    const int y0 = d.x & 1023, y1 = (d.x >> 10) & 1023, y2 = (d.x >> 20) & 1023, y3 = d.y & 1023, y4 = (d.y >> 10) & 1023;
    const bool hBody = (d.x >> 30) & 1u, vBody = d.x >> 31;
    const int c2 = (int)(d.y >> 20);
    const int col[5] = {c2 + ((int)(d.z << 24) >> 24), c2 + ((int)(d.z << 16) >> 24), c2, c2 + ((int)(d.z << 8) >> 24), c2 + ((int)d.z >> 24)};
    const float* row[5] = {lvl + y0 * stride, lvl + y1 * stride, lvl + y2 * stride, lvl + y3 * stride, lvl + y4 * stride};
    float t[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int m = 0; m < 5; ++m) t[k][m] = row[k][col[m]];
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; ++k)
        r[k] = hBody ? t[k][2] * 6.f + ((t[k][1] + t[k][3]) * 4.f + (t[k][0] + t[k][4]))
                     : t[k][2] * 6.f + (t[k][1] + t[k][3]) * 4.f + t[k][0] + t[k][4];
    const float s = 1.f / 256;
    return vBody ? ((r[1] + r[3] + r[2]) * 4.f + (r[0] + r[4] + (r[2] + r[2]))) * s
                 : (r[2] * 6.f + (r[1] + r[3]) * 4.f + r[0] + r[4]) * s;
}

// horizontal pyrUp value (pyramids.cpp:945-977) from the three neighbours and the output's edge form:
//   0  sm + s0*6 + sp     1  (s0 + sp)*4     2  s0*6 + sp*2     3  sm + s0*7     4  s0*8
// written with shared operations: forms 0, 2, 3 are (x + s0*k) [+ sp], forms 1, 4 are (s0 + y)*4 — sp*2 and (s0 + s0)*4 are
// exact, and x + s0*6 is s0*6 + x.
struct UpForm { bool f0, f2, f3, odd, f4; };
__device__ __forceinline__ float pyrup_h_form(float sm, float s0, float sp, const UpForm& f) {
    const float k = f.f3 ? 7.f : 6.f;
    const float x = f.f2 ? sp + sp : sm;
    float u = s0 * k;
    u = f.f2 ? u + x : x + u;                          // the same sum either way; kept in the reference's operand order
    const float a = f.f0 ? u + sp : u;
    const float v = (s0 + (f.f4 ? s0 : sp)) * 4.f;
    return f.odd ? v : a;
}

// collapse output from its descriptor: w0 = y0 | y1 << 10 | y2 << 20 | oddy << 30, w1 = cm | c0 << 11 | form << 22,
// w2 = cp | mask pixel << 11.  Rows are row numbers of the coarser level, columns element offsets in a row.
__device__ __forceinline__ float up3_from_desc(const float* __restrict__ nl, int stride, uint4 d, const UpForm& f, bool oddy) {
    const int y0 = d.x & 1023, y1 = (d.x >> 10) & 1023, y2 = (d.x >> 20) & 1023;
    const int cm = d.y & 2047, c0 = (d.y >> 11) & 2047, cp = d.z & 2047;
    const float* r0 = nl + y0 * stride; const float* r1 = nl + y1 * stride; const float* r2 = nl + y2 * stride;
    const float h0 = pyrup_h_form(r0[cm], r0[c0], r0[cp], f), h1 = pyrup_h_form(r1[cm], r1[c0], r1[cp], f);
    const float h2 = pyrup_h_form(r2[cm], r2[c0], r2[cp], f);
    const float s = 1.f / 64;
    return oddy ? ((h1 + h2) * 4.f) * s : (h0 + h1 * 6.f + h2) * s;
}

#ifdef POPPY_TAIL_STAMPS                    // tools/micro/tail_probe.hip: where a launch spends its time (100 MHz stamps by thread 0)
__device__ long long g_tail_stamps[64];
#define TAIL_STAMP(n) do { if (threadIdx.x == 0) g_tail_stamps[n] = (long long)wall_clock64(); } while (0)
#else
#define TAIL_STAMP(n) do { } while (0)
#endif

__global__ void __launch_bounds__(kTailThreads) k_pyr_tail(const float* __restrict__ gL, const float* __restrict__ gR,
                                                           const float* __restrict__ gM, float* __restrict__ gB,
                                                           const uint4* __restrict__ g_desc, const PyrTailArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* s_desc = (uint4*)lds;
    float* sL = (float*)(s_desc + a.n_desc);     // n3 floats each, indexed by (off3 - off3 of level `first`)
    float* sR = sL + a.n3;
    float* sB = sR + a.n3;
    float* sM = sB + a.n3;                       // n1 floats
    float* sRes = sM + a.n1;                     // residuals of the single-pixel levels that are walked
    int* sMeta = (int*)(sRes + kResMax);
    const int tid = threadIdx.x, nth = kTailThreads;
    TAIL_STAMP(0);
    // stage: the descriptors and level `first` (produced by the previous pyrDown launch), one round trip
    for (int e = tid; e < a.n_desc; e += nth) s_desc[e] = g_desc[e];
    for (int e = tid; e < a.first_c3; e += nth) { sL[e] = gL[a.g_off3 + e]; sR[e] = gR[a.g_off3 + e]; }
    for (int e = tid; e < a.first_c1; e += nth) sM[e] = gM[a.g_off1 + e];
    __syncthreads();
    TAIL_STAMP(1);

    // A step's first output per thread has its descriptor (and the step's table entry) fetched BEFORE the barrier that ends the
    // previous step: the kernel is a chain of short dependent steps, and those two reads were at the head of each.
    auto down_desc = [&](const PyrTailDown& st, int e) { return s_desc[st.desc + (e >= st.c3 ? e - st.c3 : e)]; };       // L and R share the 3-channel descriptors
    PyrTailDown nst = a.down[0];
    uint4 nd = a.n_wide > 0 && tid < 2 * nst.c3 + nst.c1 ? down_desc(nst, tid) : uint4{0, 0, 0, 0};
    for (int k = 0; k < a.n_wide; ++k) {
        const PyrTailDown st = nst;
        uint4 d = nd;
        if (k + 1 < a.n_wide) nst = a.down[k + 1];
        for (int e = tid; e < 2 * st.c3 + st.c1; e += nth) {
            if (e != tid) d = down_desc(st, e);
            const bool mask = e >= 2 * st.c3, right = e >= st.c3;
            float* plane = mask ? sM : right ? sR : sL;
            const float v = down_from_desc(plane + (mask ? st.so1 : st.so3), mask ? st.stride1 : st.stride3, d);
            plane[mask ? st.do1 + (e - 2 * st.c3) : st.do3 + (right ? e - st.c3 : e)] = v;
        }
        if (k + 1 < a.n_wide && tid < 2 * nst.c3 + nst.c1) nd = down_desc(nst, tid);
        __syncthreads();
        TAIL_STAMP(2 + k);
    }
    PyrTailUp nup = a.n_wide > 0 ? a.up[a.n_wide - 1] : PyrTailUp{0, 0, 0, 0, 0, 0};
    uint4 nud = tid < nup.cnt ? s_desc[nup.desc + tid] : uint4{0, 0, 0, 0};

    if (a.nl >= 0) {
        // single-pixel levels k1 .. k1 + nl; chain values at (o3 + 3 j + channel), (o1 + j)
        const int nl = a.nl, o3 = a.one_o3, o1 = a.one_o1;
        if (tid < 64) {
            int jz = nl;                                   // first level from which all seven chains are constant (nl: none found)
            if (tid < 7) {
                float* chain = tid < 3 ? sL + o3 + tid : tid < 6 ? sR + o3 + (tid - 3) : sM + o1;
                const int step = tid < 6 ? 3 : 1;
                float v = chain[0];
                int j = 0;
                while (j < nl) {
                    const float nv = down_1x1(v);
                    ++j;
                    chain[j * step] = nv;
                    const bool moved = __float_as_uint(nv) != __float_as_uint(v);
                    v = nv;
                    if (__builtin_amdgcn_ballot_w64(moved) == 0) { jz = j - 1; break; }          // f(v) == v on every chain
                }
            }
            jz = __builtin_amdgcn_readfirstlane(jz);       // lanes 0..6 agree; lane 0 is one of them
            if (tid == 0) sMeta[0] = jz;
        }
        __syncthreads();
        TAIL_STAMP(14);
        const int jz = sMeta[0];
        // residuals of levels k1 + j, j <= min(jz, nl - 1); levels past jz have the residual of jz
        const int nres = (jz < nl ? jz + 1 : nl) * 3;
        for (int e = tid; e < nres; e += nth) {
            const int j = e / 3;
            const float lapL = sL[o3 + e] - up_1x1(sL[o3 + e + 3]), lapR = sR[o3 + e] - up_1x1(sR[o3 + e + 3]);
            sRes[e] = mix_lr(lapL, lapR, sM[o1 + j]);
        }
        __syncthreads();
        TAIL_STAMP(15);
        if (tid < 3) {
            const int jt = jz < nl ? jz : nl;              // index of the deepest level's values (constant from jz on)
            float cur = mix_lr(sL[o3 + jt * 3 + tid], sR[o3 + jt * 3 + tid], sM[o1 + jt]);
            if (jz < nl) {                                 // levels nl-1 .. jz: one residual, iterate to the bitwise fixed point
                const float r = sRes[jz * 3 + tid];
                for (int i = nl - jz; i > 0; --i) {
                    const float nc = up_1x1(cur) + r;
                    const bool same = __float_as_uint(nc) == __float_as_uint(cur);
                    cur = nc;
                    if (same) break;
                }
            }
            for (int j = jt - 1; j >= 0; --j) cur = up_1x1(cur) + sRes[j * 3 + tid];
            sB[o3 + tid] = cur;
        }
    } else {
        for (int e = tid; e < a.top_c3; e += nth) sB[a.top_o3 + e] = mix_lr(sL[a.top_o3 + e], sR[a.top_o3 + e], sM[a.top_o1 + e / 3]);
    }
    __syncthreads();
    TAIL_STAMP(16);

    for (int k = a.n_wide - 1; k >= 0; --k) {
        const PyrTailUp st = nup;
        uint4 d = nud;
        if (k > 0) nup = a.up[k - 1];
        for (int e = tid; e < st.cnt; e += nth) {
            if (e != tid) d = s_desc[st.desc + e];
            const unsigned form = d.y >> 22;
            UpForm f;
            f.f0 = form == 0; f.f2 = form == 2; f.f3 = form == 3; f.f4 = form == 4; f.odd = form == 1 || form == 4;
            const bool oddy = (d.x >> 30) & 1u;
            const float m = sM[st.co1 + (int)(d.z >> 11)];
            const float gl = sL[st.co3 + e], gr = sR[st.co3 + e];
            const float uL = up3_from_desc(sL + st.no3, st.nstride, d, f, oddy), uR = up3_from_desc(sR + st.no3, st.nstride, d, f, oddy);
            const float uB = up3_from_desc(sB + st.no3, st.nstride, d, f, oddy);
            sB[st.co3 + e] = uB + mix_lr(gl - uL, gr - uR, m);
        }
        if (k > 0 && tid < nup.cnt) nud = s_desc[nup.desc + tid];
        __syncthreads();
        TAIL_STAMP(17 + k);
    }
    for (int e = tid; e < a.first_c3; e += nth) gB[a.g_off3 + e] = sB[e];
    TAIL_STAMP(30);
}

int reflect101_host(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
    return p;
}

}  // namespace

// Descriptors and step table of the tail for a pyramid whose levels are `lv[0..levels]`; the tail takes levels first..levels.
// ok == false when a field would not fit (levels wider than 2047 elements or higher than 1023 rows, more than kPyrTailMaxWide
// steps, more LDS than a workgroup has): the caller then runs the per-level kernels all the way down.
PyrTailPlan build_pyr_tail_plan(const PyrLevel* lv, int first, int levels) {
    PyrTailPlan p;
    PyrTailArgs& a = p.args;
    memset(&a, 0, sizeof(a));
    p.ok = false;
    if (first > levels) return p;
    int k1 = levels;                                           // first single-pixel level in [first, levels], else `levels`
    for (int i = first; i <= levels; ++i) if (lv[i].w == 1 && lv[i].h == 1) { k1 = i; break; }
    const bool one = lv[k1].w == 1 && lv[k1].h == 1;
    const int wide_end = k1;                                   // levels [first, wide_end) are reduced by the whole workgroup
    a.n_wide = wide_end - first;
    if (a.n_wide > kPyrTailMaxWide) return p;
    const size_t b3 = lv[first].off3, b1 = lv[first].off1;
    size_t n3 = 0, n1 = 0;
    for (int i = first; i <= levels; ++i) { n3 += (size_t)lv[i].w * lv[i].h * 3; n1 += (size_t)lv[i].w * lv[i].h; }
    if (n3 >= (1u << 20)) return p;
    a.n3 = (int)n3; a.n1 = (int)n1;
    a.first_c3 = lv[first].w * lv[first].h * 3; a.first_c1 = lv[first].w * lv[first].h;
    a.g_off3 = b3; a.g_off1 = b1;
    if (one && levels - k1 > 256) return p;
    a.nl = one ? levels - k1 : -1;
    a.one_o3 = (int)(lv[k1].off3 - b3); a.one_o1 = (int)(lv[k1].off1 - b1);
    a.top_o3 = (int)(lv[levels].off3 - b3); a.top_o1 = (int)(lv[levels].off1 - b1); a.top_c3 = lv[levels].w * lv[levels].h * 3;
    std::vector<uint32_t>& blob = p.desc;
    auto push = [&](uint32_t x, uint32_t y, uint32_t z) { blob.push_back(x); blob.push_back(y); blob.push_back(z); blob.push_back(0u); };
    for (int k = 0; k < a.n_wide; ++k) {
        const PyrLevel &s = lv[first + k], &d = lv[first + k + 1];
        if (s.w * 3 > 2047 || s.h > 1023) return p;
        PyrTailDown& st = a.down[k];
        st.c3 = d.w * d.h * 3; st.c1 = d.w * d.h; st.desc = (int)(blob.size() / 4);
        st.so3 = (int)(s.off3 - b3); st.so1 = (int)(s.off1 - b1); st.do3 = (int)(d.off3 - b3); st.do1 = (int)(d.off1 - b1);
        st.stride3 = s.w * 3; st.stride1 = s.w;
        for (int cn = 3; cn >= 1; cn -= 2) {                   // the 3-channel descriptors (L and R), then the mask's
            const DownGeom g = make_down_geom(s.w, s.h, cn);
            for (int y = 0; y < d.h; ++y)
                for (int xe = 0; xe < d.w * cn; ++xe) {
                    const int px = xe / cn, c = xe - px * cn;
                    const bool hBody = xe >= cn && xe < g.hBodyEnd, vBody = xe < g.vBodyEnd;
                    int col[5], row[5];
                    for (int t = 0; t < 5; ++t) { col[t] = reflect101_host(2 * px + t - 2, s.w) * cn + c; row[t] = reflect101_host(2 * y + t - 2, s.h); }
                    auto rel = [&](int t) { return (uint32_t)((col[t] - col[2]) & 255); };
                    push((uint32_t)row[0] | (uint32_t)row[1] << 10 | (uint32_t)row[2] << 20 | (hBody ? 1u << 30 : 0u) | (vBody ? 1u << 31 : 0u),
                         (uint32_t)row[3] | (uint32_t)row[4] << 10 | (uint32_t)col[2] << 20,
                         rel(0) | rel(1) << 8 | rel(3) << 16 | rel(4) << 24);
                }
        }
    }
    for (int k = 0; k < a.n_wide; ++k) {
        const PyrLevel &c = lv[first + k], &n = lv[first + k + 1];
        if (n.w * 3 > 2047 || n.h > 1023 || c.w * c.h > (1 << 21)) return p;
        PyrTailUp& st = a.up[k];
        st.cnt = c.w * c.h * 3; st.desc = (int)(blob.size() / 4);
        st.co3 = (int)(c.off3 - b3); st.co1 = (int)(c.off1 - b1); st.no3 = (int)(n.off3 - b3); st.nstride = n.w * 3;
        for (int y = 0; y < c.h; ++y)
            for (int xe = 0; xe < c.w * 3; ++xe) {
                const int dpx = xe / 3, ch = xe - dpx * 3, spx = dpx >> 1, sy = y >> 1;
                const bool oddx = dpx & 1, oddy = y & 1;
                // pyrup_h (pyramid_device.h): single column, left edge, right edge, interior
                int form, cm = spx, cp = spx;
                if (n.w == 1) form = 4;
                else if (spx == 0) { form = oddx ? 1 : 2; cp = 1; }
                else if (spx >= n.w - 1) { form = oddx ? 4 : 3; cm = n.w - 2; }
                else { form = oddx ? 1 : 0; cm = spx - 1; cp = spx + 1; }
                const int c0 = (n.w == 1 ? 0 : spx >= n.w - 1 ? n.w - 1 : spx);
                const int y1 = sy, y2 = reflect101_host((sy + 1) * 2, n.h * 2) >> 1;
                const int y0 = oddy ? y1 : reflect101_host((sy - 1) * 2, n.h * 2) >> 1;
                push((uint32_t)y0 | (uint32_t)y1 << 10 | (uint32_t)y2 << 20 | (oddy ? 1u << 30 : 0u),
                     (uint32_t)(cm * 3 + ch) | (uint32_t)(c0 * 3 + ch) << 11 | (uint32_t)form << 22,
                     (uint32_t)(cp * 3 + ch) | (uint32_t)(y * c.w + dpx) << 11);
            }
    }
    a.n_desc = (int)(blob.size() / 4);
    p.lds_bytes = (size_t)a.n_desc * 16 + ((size_t)3 * a.n3 + a.n1 + kResMax + 4) * sizeof(float);
    p.ok = p.lds_bytes <= 160 * 1024;
    return p;
}

// The dynamic-LDS limit of a kernel is process state per device, not per context: it is only ever RAISED here, so a context
// with a smaller geometry cannot lower it under a live one.  Call outside any stream capture.
bool prepare_pyr_tail(size_t lds_bytes) {
    static std::mutex mu;
    static size_t granted[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (lds_bytes <= granted[dev]) return true;
    if (hipFuncSetAttribute((const void*)k_pyr_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return false;
    granted[dev] = lds_bytes;
    return true;
}

void launch_pyr_tail(const float* pyrL, const float* pyrR, const float* pyrM, float* pyrB, const void* d_desc, const PyrTailArgs& args,
                     size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL(k_pyr_tail, dim3(1), dim3(kTailThreads), lds_bytes, s, pyrL, pyrR, pyrM, pyrB, (const uint4*)d_desc, args);
}

}  // namespace poppy_hip
