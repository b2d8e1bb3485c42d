// orb_detect.cpp — host side of ORB::detect on the GPU: level geometry, kernel sequencing and the
// order-sensitive selection steps that the reference performs with libstdc++ algorithms.
//
//   OCV/features2d/src/orb.cpp:970-1127  detectAndCompute (atlas layout, resize chain)
//   OCV/features2d/src/orb.cpp:784-959   computeKeyPoints (per-level quota, FAST -> border filter -> retainBest(2n)
//                                        -> Harris -> retainBest(n) -> IC angle -> pt *= scale)
//   OCV/features2d/src/keypoint.cpp:69-90 retainBest: std::nth_element + std::partition.  The resulting ORDER
//                                        is part of the contract (Poppy truncates and pairs keypoints by position,
//                                        src/extractor.cpp:96-99), so these two calls stay on the host and use
//                                        the very same library routines.
#include "orb_detect.h"
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace poppy_hip {

namespace {
inline int round_half_even(float v) { return (int)std::nearbyintf(v); }

struct Cand { int x, y, level; float response; float angle = 0.f; };
struct Key { float response; int i; };            // what retainBest compares, plus where the candidate sits in the raster-ordered list
struct ByResponse { template <typename T> bool operator()(const T& a, const T& b) const { return a.response > b.response; } };

template <typename T>
void retain_best(std::vector<T>& k, int n) {
    if (n >= 0 && k.size() > (size_t)n) {
        if (n == 0) { k.clear(); return; }
        std::nth_element(k.begin(), k.begin() + n - 1, k.end(), ByResponse());
        const float ambiguous = k[n - 1].response;
        auto new_end = std::partition(k.begin() + n, k.end(), [ambiguous](const T& c) { return c.response >= ambiguous; });
        k.resize(new_end - k.begin());
    }
}
}  // namespace

void OrbDetector::release() {
    void* bufs[] = {d_img, d_atlas, d_blur, d_scores, d_counters, d_kp, d_val, d_desc, d_nms};        // (d_cand lies inside d_counters' allocation)
    for (void* b : bufs) if (b) (void)hipFree(b);
    d_img = d_atlas = d_blur = d_scores = nullptr; d_counters = d_cand = d_kp = nullptr; d_val = nullptr; d_desc = nullptr; d_nms = nullptr;
    if (h_cand) (void)hipHostFree(h_cand);
    if (h_kp) (void)hipHostFree(h_kp);
    if (h_val) (void)hipHostFree(h_val);
    h_cand = nullptr; h_kp = nullptr; h_val = nullptr;
    W = H = 0;
}

hipError_t OrbDetector::prepare(int w, int h) {
    if (w == W && h == H) return hipSuccess;
    release();
    const double scaleFactor = (double)1.2f;                       // ORB::create takes a float, stores a double
    size_t off = 0, soff = 0;
    S.n = kOrbLevels;
    for (int l = 0; l < kOrbLevels; ++l) {
        OrbLevel& L = S.lv[l];
        L.scale = (float)std::pow(scaleFactor, (double)l);
        const float inv = 1.0f / L.scale;
        L.w = round_half_even(w * inv); L.h = round_half_even(h * inv);
        L.stride = (size_t)L.w + 2 * kOrbBorder;
        L.offset = off; L.score_offset = soff;
        off += L.stride * ((size_t)L.h + 2 * kOrbBorder);
        soff += (size_t)L.w * L.h;
        off = (off + 255) & ~(size_t)255; soff = (soff + 255) & ~(size_t)255;
    }
    atlas_bytes = off;
    static const int forced_cap = getenv("POPPY_ORB_CAP") ? std::max(16, atoi(getenv("POPPY_ORB_CAP"))) : 0;   // tests: make the first lists too short
    cap = forced_cap ? forced_cap : std::max(4096, w * h / 16);
    static const int forced_kp = getenv("POPPY_ORB_KPCAP") ? std::max(16, atoi(getenv("POPPY_ORB_KPCAP"))) : 0;    // tests: make the keypoint buffers too small
    kp_cap = forced_kp ? forced_kp : 1 << 16;
    hipError_t e;
    if ((e = hipMalloc((void**)&d_img, (size_t)w * h)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_atlas, off)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_blur, off)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_scores, soff)) != hipSuccess) return e;
    // counts (16 ints, 8 used) and, right behind them, the candidate lists of all levels one after the other: ONE copy of a guessed length fetches both
    if ((e = hipMalloc((void**)&d_counters, (kCandHeader + (size_t)kOrbLevels * cap * 2) * sizeof(int))) != hipSuccess) return e;
    d_cand = d_counters + kCandHeader;
    last_total = 0;
    if ((e = hipMalloc(&d_nms, fast_nms_scratch_bytes(S))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_kp, (size_t)kp_cap * 3 * sizeof(int))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_val, (size_t)kp_cap * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&d_desc, (size_t)kp_cap * 32)) != hipSuccess) return e;
    if ((e = hipHostMalloc((void**)&h_cand, (kCandHeader + (size_t)kOrbLevels * cap * 2) * sizeof(int))) != hipSuccess) return e;
    if ((e = hipHostMalloc((void**)&h_kp, (size_t)kp_cap * 3 * sizeof(int))) != hipSuccess) return e;
    if ((e = hipHostMalloc((void**)&h_val, (size_t)kp_cap * sizeof(float))) != hipSuccess) return e;
    W = w; H = h;
    return hipSuccess;
}

// room for n keypoints (positions, responses / angles, descriptors; device buffers and their pinned mirrors)
hipError_t OrbDetector::grow_keypoints(int n) {
    if (n <= kp_cap) return hipSuccess;
    // the new buffers first, the swap only when all of them exist: a failed allocation leaves the detector as it was (smaller, but consistent)
    int *nd_kp = nullptr, *nh_kp = nullptr; float *nd_val = nullptr, *nh_val = nullptr; uint8_t* nd_desc = nullptr;
    hipError_t e = hipMalloc((void**)&nd_kp, (size_t)n * 3 * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&nd_val, (size_t)n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&nd_desc, (size_t)n * 32);
    if (e == hipSuccess) e = hipHostMalloc((void**)&nh_kp, (size_t)n * 3 * sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc((void**)&nh_val, (size_t)n * sizeof(float));
    if (e != hipSuccess) {
        if (nd_kp) (void)hipFree(nd_kp);
        if (nd_val) (void)hipFree(nd_val);
        if (nd_desc) (void)hipFree(nd_desc);
        if (nh_kp) (void)hipHostFree(nh_kp);
        if (nh_val) (void)hipHostFree(nh_val);
        return e;
    }
    if (d_kp) (void)hipFree(d_kp);
    if (d_val) (void)hipFree(d_val);
    if (d_desc) (void)hipFree(d_desc);
    if (h_kp) (void)hipHostFree(h_kp);
    if (h_val) (void)hipHostFree(h_val);
    d_kp = nd_kp; d_val = nd_val; d_desc = nd_desc; h_kp = nh_kp; h_val = nh_val;
    kp_cap = n;
    return hipSuccess;
}

// room for new_cap candidates per level (device lists + their pinned mirror); the level geometry stays
hipError_t OrbDetector::grow_candidates(int new_cap) {
    if (new_cap <= cap) return hipSuccess;
    int *nd = nullptr, *nh = nullptr;
    const size_t bytes = (kCandHeader + (size_t)kOrbLevels * new_cap * 2) * sizeof(int);
    hipError_t e = hipMalloc((void**)&nd, bytes);
    if (e == hipSuccess) e = hipHostMalloc((void**)&nh, bytes);
    if (e != hipSuccess) { if (nd) (void)hipFree(nd); return e; }      // (the old lists stay: the detector is unchanged)
    if (d_counters) (void)hipFree(d_counters);
    if (h_cand) (void)hipHostFree(h_cand);
    d_counters = nd; d_cand = d_counters + kCandHeader; h_cand = nh;
    cap = new_cap;
    last_total = 0;
    return hipSuccess;
}

#define ORB_CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } } while (0)

int OrbDetector::detect(const uint8_t* gray, size_t stride, int w, int h, int nfeatures, hipStream_t s, std::vector<OrbKeyPoint>& out,
                        bool gray_on_device) {
    out.clear();
    const int rc = detect_begin(gray, stride, w, h, s, gray_on_device);
    return rc < 0 ? rc : detect_finish(nfeatures, s, out);
}

static const bool g_stage_times = getenv("POPPY_SETUP_TIMING") != nullptr;              // host wall time of the detection's stages on stderr

int OrbDetector::detect_begin(const uint8_t* gray, size_t stride, int w, int h, hipStream_t s, bool gray_on_device) {
    ORB_CHK(prepare(w, h));
    const int edge = 31, fastThreshold = 20;
    t_begin_ = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin_).count(); };
    ORB_CHK(hipMemcpy2DAsync(d_img, w, gray, stride, w, h, gray_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    launch_orb_pyramid(d_img, w, h, w, d_atlas, S, s);
    static const long forced_guess = getenv("POPPY_ORB_GUESS") ? atol(getenv("POPPY_ORB_GUESS")) : -1;       // tests: make the first copy short
    size_t guess = 0, total = 0;
    const int* h_counts = nullptr;
    // The candidate lists have room for one pixel in sixteen per level (flat shapes: 1 in 80; a photograph: 1 in 300).  Noise-like content after
    // the pre-filter's equalizeHist can exceed that (strict 3 x 3 maxima: at most 1 in 4): the counters hold the true counts, so the lists are
    // then re-allocated for what was counted and FAST runs once more — a slow first image of such content instead of an error.
    for (int attempt = 0; ; ++attempt) {
        launch_fast(d_atlas, S, d_scores, fastThreshold, edge, d_counters, d_cand, cap, d_nms, s);
        // The counts and the candidates come back in one copy: its length is a guess — a quarter more than the last image's candidates (pairs
        // follow each other with similar images), 1/32 of the pixels the first time — and a second copy fetches the rest when the guess was short.
        const size_t room = (size_t)kOrbLevels * cap;
        guess = std::min(room, forced_guess >= 0 ? (size_t)forced_guess : last_total ? last_total + last_total / 4 + 1024 : (size_t)w * h / 32 + 1024);
        ORB_CHK(hipMemcpyAsync(h_cand, d_counters, (kCandHeader + guess * 2) * sizeof(int), hipMemcpyDeviceToHost, s));
        ORB_CHK(hipStreamSynchronize(s));
        h_counts = h_cand;
        int most = 0;
        for (int l = 0; l < kOrbLevels; ++l) most = std::max(most, h_counts[l]);
        if (most <= cap) break;
        if (attempt) { err = "FAST candidate buffer overflow"; return -1; }
        ORB_CHK(grow_candidates(most + most / 8 + 1024));
    }
    ms_fast_ = since();
    for (int l = 0; l < kOrbLevels; ++l) {
        level_base_[l] = total;
        total += (size_t)h_counts[l];
    }
    if (total > guess) {
        ORB_CHK(hipMemcpyAsync(h_cand + kCandHeader + guess * 2, d_cand + guess * 2, (total - guess) * 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        ORB_CHK(hipStreamSynchronize(s));
    }
    last_total = total;
    ms_cand_ = since();
    return 0;
}

int OrbDetector::detect_finish(int nfeatures, hipStream_t s, std::vector<OrbKeyPoint>& out) {
    out.clear();
    if (!W) { err = "detect_finish without detect_begin"; return -1; }
    const int w = W, h = H, patch = 31;
    const bool stage_times = g_stage_times;
    const auto t_finish = std::chrono::steady_clock::now();
    auto since = [&]() { return ms_cand_ + std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_finish).count(); };
    const double ms_fast = ms_fast_, ms_cand = ms_cand_;
    double ms_sort = 0, ms_harris = 0;
    const int* h_counts = h_cand;

    // per-level quota (orb.cpp:803-813)
    int quota[kOrbLevels];
    {
        const float factor = (float)(1.0 / (double)1.2f);
        float desired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)kOrbLevels));
        int sum = 0;
        for (int l = 0; l < kOrbLevels - 1; ++l) { quota[l] = round_half_even(desired); sum += quota[l]; desired *= factor; }
        quota[kOrbLevels - 1] = std::max(nfeatures - sum, 0);
    }

    std::vector<Cand> all, k;
    int counts[kOrbLevels];
    // The levels are independent up to here: each one's counting sort + retainBest runs on its own host thread (level 0 holds
    // ~45 % of the ~10^5 candidates of a 1080p image: 0.9 -> 0.5 ms of host time per image).
    std::vector<Cand> per_level[kOrbLevels];
    auto level_job = [&](int l) {
        const int n = h_counts[l], lw = S.lv[l].w;
        const int* c = h_cand + kCandHeader + level_base_[l] * 2;
        // the candidates arrive in raster order = FAST's emission order (fast.cpp:271-290): the kernels compact them that way
        // retainBest on 8-byte keys (response, index): std::nth_element / std::partition make the same comparisons in the same order
        // whatever else an element carries, so the keys end up permuted exactly as the reference's KeyPoints would
        std::vector<Key> keys(n);
        for (int i = 0; i < n; ++i) keys[i] = Key{(float)c[2 * i + 1], i};
        retain_best(keys, 2 * quota[l]);
        std::vector<Cand>& kl = per_level[l];
        kl.resize(keys.size());
        for (size_t j = 0; j < keys.size(); ++j) { const int p = c[2 * keys[j].i]; kl[j] = Cand{p & 0xffff, p >> 16, l, keys[j].response}; }
        (void)lw;
    };
    {   // level 0 holds ~45 % of the candidates, level 1 ~25 %: the caller takes level 0, one helper level 1, another levels 2-7 (a thread per
        // level cost more in thread creation, ~50 us each, than the levels' work — these two are persistent)
        const bool split = h_counts[0] > 2000;
        if (split) {
            helper_.run([&]() { level_job(1); });
            helper2_.run([&]() { for (int l = 2; l < kOrbLevels; ++l) level_job(l); });
        }
        level_job(0);
        if (split) {
            const bool ok1 = helper_.wait(), ok2 = helper2_.wait();
            if (!ok1 || !ok2) { err = "keypoint selection helper thread: " + (ok1 ? helper2_.error() : helper_.error()); return -1; }
        }
        else for (int l = 1; l < kOrbLevels; ++l) level_job(l);
    }
    for (int l = 0; l < kOrbLevels; ++l) {
        counts[l] = (int)per_level[l].size();
        all.insert(all.end(), per_level[l].begin(), per_level[l].end());
    }
    ms_sort = since();
    if (all.empty()) return 0;
    // retainBest keeps every tie with the n-th response (keypoint.cpp:69-90) and FAST scores are small integers: on noise-like content the survivors
    // of the first selection can outnumber any fixed buffer — the keypoint buffers grow to what was selected
    // (room for two floats per selected candidate: Harris response and angle come back in one copy)
    if (2 * (int)all.size() > kp_cap) ORB_CHK(grow_keypoints(2 * (int)all.size() + (int)all.size() / 2));

    auto upload = [&](const std::vector<Cand>& v) -> hipError_t {
        for (size_t i = 0; i < v.size(); ++i) { h_kp[3 * i] = v[i].level; h_kp[3 * i + 1] = v[i].x; h_kp[3 * i + 2] = v[i].y; }
        return hipMemcpyAsync(d_kp, h_kp, v.size() * 3 * sizeof(int), hipMemcpyHostToDevice, s);
    };
    ORB_CHK(upload(all));
    // Harris responses AND the intensity-centroid angles of all survivors of the first selection in one submission: the angle of a keypoint does not
    // depend on the second selection, so computing it for the ~2 x nfeatures candidates (a few us more) saves the second round trip of upload,
    // launch, copy and wait that followed the host's retainBest until round 4
    launch_harris(d_atlas, S, d_kp, (int)all.size(), d_val, s);
    launch_ic_angle(d_atlas, S, d_kp, (int)all.size(), d_val + all.size(), s);
    ORB_CHK(hipMemcpyAsync(h_val, d_val, 2 * all.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    ORB_CHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < all.size(); ++i) { all[i].response = h_val[i]; all[i].angle = h_val[all.size() + i]; }
    ms_harris = since();

    std::vector<Cand> fin;
    size_t off = 0;
    for (int l = 0; l < kOrbLevels; ++l) {
        k.assign(all.begin() + off, all.begin() + off + counts[l]);
        off += counts[l];
        retain_best(k, quota[l]);
        fin.insert(fin.end(), k.begin(), k.end());
    }
    if (fin.empty()) return 0;
    out.resize(fin.size());
    last_levels.resize(fin.size() * 3);
    for (size_t i = 0; i < fin.size(); ++i) {
        const float sc = S.lv[fin[i].level].scale;
        out[i] = OrbKeyPoint{(float)fin[i].x * sc, (float)fin[i].y * sc, patch * sc, fin[i].angle, fin[i].response, fin[i].level, -1};
        last_levels[3 * i] = fin[i].level; last_levels[3 * i + 1] = fin[i].x; last_levels[3 * i + 2] = fin[i].y;
    }
    if (stage_times) {
        long long ncand = 0;
        for (int l = 0; l < kOrbLevels; ++l) ncand += h_counts[l];
        fprintf(stderr, "  orb detect %dx%d: pyramid + FAST %.3f, %lld candidates on the host %.3f, order + retainBest %.3f, Harris %.3f, angle %.3f ms (cumulative)\n",
                w, h, ms_fast, ncand, ms_cand, ms_sort, ms_harris, since());
    }
    return (int)fin.size();
}

// ORB::compute(image, keypoints, descriptors), WTA_K = 2 (orb.cpp:219-285,1148-1216).  kps7 in cv::KeyPoint field order.
int OrbDetector::describe(const uint8_t* gray, size_t stride, int w, int h, const float* kps7, int n, hipStream_t s, uint8_t* desc_out) {
    if (n <= 0) return 0;
    ORB_CHK(prepare(w, h));
    if (n > kp_cap) ORB_CHK(grow_keypoints(n));
    int nlev = 0;
    for (int i = 0; i < n; ++i) {
        int oct = (int)kps7[(size_t)i * 7 + 5];
        if (oct < 0 || oct >= kOrbLevels) { err = "keypoint octave out of range"; return -1; }
        nlev = std::max(nlev, oct + 1);
    }
    OrbLevelSet Sd = S; Sd.n = nlev;
    ORB_CHK(hipMemcpy2DAsync(d_img, w, gray, stride, w, h, hipMemcpyHostToDevice, s));
    launch_orb_pyramid(d_img, w, h, w, d_atlas, Sd, s);
    launch_orb_blur(d_atlas, d_blur, Sd, s);
    float* cs = h_val;                                   // 2 floats per keypoint: room is kp_cap floats, so stage in halves
    std::vector<float> csv((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        const float* k = kps7 + (size_t)i * 7;
        const int oct = (int)k[5];
        const float scale = 1.f / S.lv[oct].scale;
        float angle = k[3];
        angle *= (float)(M_PI / 180.f);
        csv[2 * i] = cosf(angle); csv[2 * i + 1] = sinf(angle);     // float overloads, as in the reference TU
        h_kp[3 * i] = oct; h_kp[3 * i + 1] = round_half_even(k[0] * scale); h_kp[3 * i + 2] = round_half_even(k[1] * scale);
        const OrbLevel& L = S.lv[oct];
        if (h_kp[3 * i + 1] < 23 - kOrbBorder || h_kp[3 * i + 1] >= L.w + kOrbBorder - 23 || h_kp[3 * i + 2] < 23 - kOrbBorder || h_kp[3 * i + 2] >= L.h + kOrbBorder - 23) {
            err = "keypoint too close to the border for a 31x31 steered patch"; return -1;
        }
    }
    (void)cs;
    float* d_cs = nullptr;
    ORB_CHK(hipMalloc((void**)&d_cs, (size_t)n * 2 * sizeof(float)));
    hipError_t e1 = hipMemcpyAsync(d_kp, h_kp, (size_t)n * 3 * sizeof(int), hipMemcpyHostToDevice, s);
    hipError_t e2 = hipMemcpyAsync(d_cs, csv.data(), (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, s);
    launch_orb_describe(d_blur, S, d_kp, d_cs, n, d_desc, s);
    hipError_t e3 = hipMemcpyAsync(desc_out, d_desc, (size_t)n * 32, hipMemcpyDeviceToHost, s);
    hipError_t e4 = hipStreamSynchronize(s);
    (void)hipFree(d_cs);
    ORB_CHK(e1); ORB_CHK(e2); ORB_CHK(e3); ORB_CHK(e4);
    return n;
}

int OrbDetector::hamming(const uint8_t* q, int nq, const uint8_t* t, int nt, hipStream_t s, int* out3) {
    if (nq <= 0 || nt <= 0) return 0;
    uint8_t *dq = nullptr, *dt = nullptr; int* dout = nullptr;
    ORB_CHK(hipMalloc((void**)&dq, (size_t)nq * 32));
    hipError_t e = hipMalloc((void**)&dt, (size_t)nt * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, (size_t)nq * 2 * sizeof(int));
    std::vector<int> tmp((size_t)nq * 2);
    if (e == hipSuccess) e = hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { launch_hamming_match(dq, nq, dt, nt, dout, s); e = hipMemcpyAsync(tmp.data(), dout, tmp.size() * sizeof(int), hipMemcpyDeviceToHost, s); }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (dq) (void)hipFree(dq);
    if (dt) (void)hipFree(dt);
    if (dout) (void)hipFree(dout);
    ORB_CHK(e);
    for (int i = 0; i < nq; ++i) { out3[3 * i] = i; out3[3 * i + 1] = tmp[2 * i]; out3[3 * i + 2] = tmp[2 * i + 1]; }
    return nq;
}

int OrbDetector::hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, hipStream_t s, int* out4) {
    if (nq <= 0) return 0;
    if (nt <= 0) { for (int i = 0; i < nq * 4; ++i) out4[i] = -1; return nq; }
    uint8_t *dq = nullptr, *dt = nullptr; int* dout = nullptr;
    ORB_CHK(hipMalloc((void**)&dq, (size_t)nq * 32));
    hipError_t e = hipMalloc((void**)&dt, (size_t)nt * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, (size_t)nq * 4 * sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) { launch_hamming_knn2(dq, nq, dt, nt, dout, s); e = hipMemcpyAsync(out4, dout, (size_t)nq * 4 * sizeof(int), hipMemcpyDeviceToHost, s); }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (dq) (void)hipFree(dq);
    if (dt) (void)hipFree(dt);
    if (dout) (void)hipFree(dout);
    ORB_CHK(e);
    return nq;
}

}  // namespace poppy_hip
