// pair_setup.cpp — the once-per-pair half of the C ABI (include/poppy_hip.h): ORB / descriptor / matching entry points, the
// pre-ORB filter chain (Extractor::foreground, dft_detail2, unsharp + Gabor banks), pair set-up from raw images
// (poppy::morph up to its frame loop, src/poppy.hpp:46-160, incl. Matcher::find / prepare and the auto-align), blur_margin.
// The per-frame path and the context itself live in poppy_hip.cpp; both share context.h.
#include "context.h"
#include <chrono>
#include "dft_exact.h"

extern "C" {

int poppy_hip_orb_detect(poppy_hip_ctx* c, const uint8_t* gray, size_t stride, int W, int H, int nfeatures, float* kps7, int max_kps, int* n_kps) {
    if (!c || !gray || !n_kps || W <= 0 || H <= 0 || stride < (size_t)W || nfeatures < 0) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<OrbKeyPoint> kps;
    int n = c->orb.detect(gray, stride, W, H, nfeatures, c->stream, kps);
    if (n < 0) { c->err = "orb_detect: " + c->orb.err; return n == -2 ? POPPY_E_DEVICE : POPPY_E_ARG; }
    *n_kps = n;
    if (n > max_kps) return fail(c, POPPY_E_ARG, "max_kps too small");
    for (int i = 0; i < n && kps7; ++i) {
        float* o = kps7 + (size_t)i * 7;
        o[0] = kps[i].x; o[1] = kps[i].y; o[2] = kps[i].size; o[3] = kps[i].angle; o[4] = kps[i].response;
        o[5] = (float)kps[i].octave; o[6] = (float)kps[i].class_id;
    }
    return POPPY_OK;
}

int poppy_hip_orb_describe(poppy_hip_ctx* c, const uint8_t* gray, size_t stride, int W, int H, const float* kps7, int n, uint8_t* desc) {
    if (!c || !gray || W <= 0 || H <= 0 || stride < (size_t)W || n < 0 || (n && (!kps7 || !desc))) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = c->orb.describe(gray, stride, W, H, kps7, n, c->stream, desc);
    if (rc < 0) { c->err = "orb_describe: " + c->orb.err; return rc == -2 ? POPPY_E_DEVICE : POPPY_E_ARG; }
    return POPPY_OK;
}

int poppy_hip_hamming_match(poppy_hip_ctx* c, const uint8_t* query, int nq, const uint8_t* train, int nt, int* out3, int* n_matches) {
    if (!c || nq < 0 || nt < 0 || !n_matches || (nq && !query) || (nt && !train) || (nq && nt && !out3)) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = c->orb.hamming(query, nq, train, nt, c->stream, out3);
    if (rc < 0) { c->err = "hamming_match: " + c->orb.err; return POPPY_E_DEVICE; }
    *n_matches = rc;
    return POPPY_OK;
}

int poppy_hip_hamming_knn2(poppy_hip_ctx* c, const uint8_t* query, int nq, const uint8_t* train, int nt, int* out4) {
    if (!c || nq < 0 || nt < 0 || (nq && !query) || (nt && !train) || (nq && !out4)) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->orb.hamming_knn2(query, nq, train, nt, c->stream, out4) < 0) { c->err = "hamming_knn2: " + c->orb.err; return POPPY_E_DEVICE; }
    return POPPY_OK;
}

int poppy_ratio_symmetry(const int* knn12, int n1, const int* knn21, int n2, float ratio, int* out3, int* n_out) {
    if (n1 < 0 || n2 < 0 || !n_out || (n1 && !knn12) || (n2 && !knn21)) return POPPY_E_ARG;
    std::vector<int> o;
    ratio_symmetry(knn12, n1, knn21, n2, ratio, o);
    *n_out = (int)o.size() / 3;
    if (out3 && !o.empty()) memcpy(out3, o.data(), o.size() * sizeof(int));
    return POPPY_OK;
}

// ---- auto-align ------------------------------------------------------------------------------------------------------
static int align_stage(poppy_hip_ctx* c, const uint8_t* img, size_t stride, int W, int H) {
    const size_t bytes = (size_t)W * H * 3;
    if (c->d_align_bytes < bytes) {
        if (c->d_align) (void)hipFree(c->d_align);
        c->d_align = nullptr; c->d_align_bytes = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_align, bytes));
        c->d_align_bytes = bytes;
    }
    HIPCHK(c, copy_rows_async(c->d_align, (size_t)W * 3, img, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
    return POPPY_OK;
}

int poppy_hip_warp_affine(poppy_hip_ctx* c, const uint8_t* src, size_t ss, int W, int H, const double* M, uint8_t* dst, size_t ds) {
    if (!c) return POPPY_E_ARG;
    if (!src || !dst || !M || W <= 0 || H <= 0 || ss < (size_t)W * 3 || ds < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad warp_affine arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = align_stage(c, src, ss, W, H); if (rc) return rc;
    uint8_t* d_out = nullptr; int* d_tab = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_out, (size_t)W * H * 3));
    hipError_t e = hipMalloc((void**)&d_tab, (size_t)2 * (W + H) * sizeof(int));
    bool ok = e == hipSuccess && warp_affine_device(c->d_align, d_out, W, H, M, d_tab, c->stream);
    if (ok) ok = copy_rows_async(dst, ds, d_out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipStreamSynchronize(c->stream) == hipSuccess;
    (void)hipFree(d_out); if (d_tab) (void)hipFree(d_tab);
    return ok ? POPPY_OK : fail(c, POPPY_E_DEVICE, "warp_affine failed");
}

static int align_host_entry(poppy_hip_ctx* c, int which, uint8_t* img, size_t stride, int W, int H, const float* p1, float* p2, int n, double* dist) {
    if (!c) return POPPY_E_ARG;
    if (!img || !p1 || !p2 || n < 4 || W <= 0 || H <= 0 || stride < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad align arguments (at least 4 point pairs)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = align_stage(c, img, stride, W, H); if (rc) return rc;
    std::vector<P2f> a(n), b(n);
    memcpy(a.data(), p1, (size_t)n * 8); memcpy(b.data(), p2, (size_t)n * 8);
    rc = which < 0 ? c->aligner.run(c->d_align, W, H, a, b, c->stream, dist) : c->aligner.step(which, c->d_align, W, H, a, b, c->stream, dist);
    if (rc) return fail(c, rc == -1 ? POPPY_E_ARG : POPPY_E_DEVICE, c->aligner.err.c_str());
    memcpy(p2, b.data(), (size_t)n * 8);
    HIPCHK(c, copy_rows_async(img, stride, c->d_align, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}
int poppy_hip_auto_align(poppy_hip_ctx* c, uint8_t* img, size_t stride, int W, int H, const float* p1, float* p2, int n, double* dist) {
    return align_host_entry(c, -1, img, stride, W, H, p1, p2, n, dist);
}
int poppy_hip_align_step(poppy_hip_ctx* c, int which, uint8_t* img, size_t stride, int W, int H, const float* p1, float* p2, int n, double* dist) {
    if (which < 0 || which > 2) return c ? fail(c, POPPY_E_ARG, "align_step: which must be 0, 1 or 2") : POPPY_E_ARG;
    return align_host_entry(c, which, img, stride, W, H, p1, p2, n, dist);
}
int poppy_procrustes(const float* x, const float* y, int n, float* rot4, float* se2, float* yprime) {
    if (!x || !y || n < 1) return POPPY_E_ARG;
    std::vector<P2f> a(n), b(n);
    memcpy(a.data(), x, (size_t)n * 8); memcpy(b.data(), y, (size_t)n * 8);
    ProcrustesFit f;
    procrustes_fit(a, b, f);
    if (rot4) memcpy(rot4, f.rotation, 16);
    if (se2) { se2[0] = f.scale; se2[1] = f.error; }
    if (yprime) memcpy(yprime, f.yprime.data(), (size_t)n * 8);
    return POPPY_OK;
}
int poppy_perspective_from4(const float* s4, const float* d4, double* m) {
    if (!s4 || !d4 || !m) return POPPY_E_ARG;
    perspective_from_4((const P2f*)s4, (const P2f*)d4, m);
    return POPPY_OK;
}

static int match_points_with(Worker* helper, const float* p1, const float* p2, int n, int W, int H, double tol, float* o1, float* o2, int* n_out, double* imd) {
    if (n < 0 || W <= 0 || H <= 0 || !n_out || (n && (!p1 || !p2))) return POPPY_E_ARG;
    std::vector<P2f> a(n), b(n);
    if (n) { memcpy(a.data(), p1, (size_t)n * 8); memcpy(b.data(), p2, (size_t)n * 8); }
    drop_out_of_image(a, b, W, H);
    if (a.empty()) { *n_out = 0; if (imd) *imd = 0; return POPPY_OK; }     // caller falls back to the dissolve (poppy.hpp:125)
    std::vector<PointPair> pairs;
    const double d = morph_distance_pairs(a, b, W, H, pairs, helper);
    if (imd) *imd = d;
    match_and_prepare_from(pairs, a, b, W, H, tol, d);
    *n_out = (int)a.size();
    if (o1) memcpy(o1, a.data(), a.size() * 8);
    if (o2) memcpy(o2, b.data(), b.size() * 8);
    return POPPY_OK;
}
int poppy_match_points(const float* p1, const float* p2, int n, int W, int H, double tol, float* o1, float* o2, int* n_out, double* imd) {
    return match_points_with(nullptr, p1, p2, n, W, H, tol, o1, o2, n_out, imd);
}

int poppy_hip_pair_begin_prefiltered(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2,
                                     const uint8_t* g1, const uint8_t* g2, const float* gabor2, int W, int H, int nfeatures) {
    if (!c || !bgr1 || !bgr2 || !g1 || !g2 || !gabor2) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<OrbKeyPoint> k1, k2;
    if (c->orb.detect(g1, W, W, H, nfeatures, c->stream, k1) < 0 || c->orb.detect(g2, W, W, H, nfeatures, c->stream, k2) < 0) {
        c->err = "orb_detect: " + c->orb.err;
        return POPPY_E_DEVICE;
    }
    const size_t n = std::min(k1.size(), k2.size());                    // Extractor::points (extractor.cpp:96-99)
    std::vector<float> p1(n * 2), p2(n * 2), o1((n + 4) * 2), o2((n + 4) * 2);
    for (size_t i = 0; i < n; ++i) { p1[2 * i] = k1[i].x; p1[2 * i + 1] = k1[i].y; p2[2 * i] = k2[i].x; p2[2 * i + 1] = k2[i].y; }
    int m = 0;
    int rc = poppy_match_points(p1.data(), p2.data(), (int)n, W, H, c->cfg.match_tolerance, o1.data(), o2.data(), &m, &c->initial_morph_dist);
    if (rc) return fail(c, rc, "poppy_match_points failed");
    return poppy_hip_pair_load(c, bgr1, s1, bgr2, s2, gabor2, W, H, o1.data(), o2.data(), m);
}

int poppy_hip_foreground(poppy_hip_ctx* c, const uint8_t* bgr, size_t stride, int W, int H, uint8_t* out, const poppy_foreground_debug* dbg) {
    if (!c) return POPPY_E_ARG;
    if (!bgr || !out || W <= 0 || H <= 0 || stride < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    ForegroundDebugOut d;
    if (dbg) { d.grey = dbg->grey; d.stages = dbg->stages; d.floats = dbg->floats; d.masked = dbg->masked; }
    const int rc = c->foreground.run(bgr, stride, W, H, c->stream, out, dbg ? &d : nullptr);
    if (rc) { c->err = "foreground: " + c->foreground.err; return rc == -1 ? POPPY_E_ARG : POPPY_E_DEVICE; }
    return POPPY_OK;
}

int poppy_hip_median_blur(poppy_hip_ctx* c, const uint8_t* src, int W, int H, int ksize, int form, uint8_t* dst) {
    if (!c) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = c->foreground.median(src, W, H, ksize, form, c->stream, dst);
    if (rc) { c->err = "median: " + c->foreground.err; return rc == -1 ? POPPY_E_ARG : POPPY_E_DEVICE; }
    return POPPY_OK;
}

// Pair set-up from the raw images: the pre-ORB filter chain on the GPU, then the same steps as pair_begin_prefiltered.
static int pair_begin_impl(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H, float ratio,
                           bool on_device = false) {
    if (!c) return POPPY_E_ARG;
    if (!bgr1 || !bgr2 || W <= 0 || H <= 0 || s1 < (size_t)W * 3 || s2 < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    struct SetupHook {                                            // a pool lets one context per device set a pair up at a time (comm.cpp: the set-up gate)
        poppy_hip_ctx* c;
        explicit SetupHook(poppy_hip_ctx* c_) : c(c_) { if (c->setup_hook) c->setup_hook(c->setup_hook_user, c, 1); }
        ~SetupHook() { if (c->setup_hook) c->setup_hook(c->setup_hook_user, c, 0); }
    } setup_hook(c);
    const auto t_enter = std::chrono::steady_clock::now();
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    c->pair_ready = false;
    c->c2_raw_valid = false;
    const size_t P = (size_t)W * H;
    static const bool gabor2_first_env = getenv("POPPY_GABOR2_FIRST") != nullptr;
    // One image's chain after the other on the GPU: a context of a pool of three or more (three set-ups side by side fill the GPU; two chains each would only put
    // six chains on the process's four hardware queues — which layout a pool of three got was a lottery with a 25 % slower outcome in one pool of four, profiles/r05_notes.md
    // section 6), a caller's choice (poppy_hip_set_setup_chains), or POPPY_SETUP_SERIAL
    static const bool serial_env = getenv("POPPY_SETUP_SERIAL") != nullptr;
    const bool serial_chains = serial_env || c->setup_serial;
    // gabor2 (the second raw image only) at the very start of the set-up, beside the first image's chain: with the chains one after the other the context has one chain in flight
    // and room beside it (a pool step of six pairs 56.7 -> 56.0 ms); with two chains side by side it tripled the first medians' time (round 3) and waits for them
    const bool gabor2_first = gabor2_first_env || (serial_chains && getenv("POPPY_GABOR2_LATE") == nullptr);
    // Host images: the second image is uploaded by its own chain's thread on that chain's stream, so the first image's chain — stream-ordered behind its
    // own upload — has the GPU to itself for the length of a copy instead of both waiting for both (POPPY_SETUP_UPLOAD_BOTH=1: the order before round 5)
    static const bool upload_both = getenv("POPPY_SETUP_UPLOAD_BOTH") != nullptr;
    const bool staged = !on_device && !upload_both && !gabor2_first && !serial_chains;
    if (on_device) {
        HIPCHK(c, hipMemcpyAsync(c->c1, bgr1, P * 3, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->c2, bgr2, P * 3, hipMemcpyDeviceToDevice, c->stream));
    } else {
        rc = upload_image(c, c->c1, bgr1, s1, W, H); if (rc) return rc;
        if (!staged) { rc = upload_image(c, c->c2, bgr2, s2, W, H); if (rc) return rc; }
    }
    // POPPY_SETUP_TIMING: host wall time of the set-up's stages on stderr (chains = foreground + detail + ORB input (+ gabor2) of both
    // images side by side; detect = the two ORB detections; match = the host matcher; finish = m2 + the pair state)
    static const bool stage_times = getenv("POPPY_SETUP_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double ms_upload = 0, ms_chains = 0, ms_detect = 0, ms_match = 0, ms_chain_end[2] = {0, 0}, ms_joined = 0;
    std::vector<uint8_t> g[2];
    if (ratio >= 0.f) { g[0].resize(P); g[1].resize(P); }
    const uint8_t* g_dev[2] = {nullptr, nullptr};
    double d[2] = {0, 0};
    if (!c->aux_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    if (!staged) HIPCHK(c, hipStreamSynchronize(c->stream));            // the uploads above
    ms_upload = since(t_begin);
    // The two images go through the chain independently (Extractor::foreground -> dft_detail2 -> the ORB input of
    // Extractor::keypoints; image 2 also through gabor_filter(corrected2 / 255), src/poppy.hpp:119-122): one host thread and
    // one stream each, so that the medians of one image run beside the Gabor bank of the other.
    std::string errs[2];
    std::atomic<int> rcs[2] = {{POPPY_OK}, {POPPY_OK}};          // (each written by its own thread, read by the other once)
    // with auto-align, gabor2 belongs to the ALIGNED second image (src/poppy.hpp:116-122 runs after Matcher::find): computed further down
    const bool align_first = c->cfg.enable_auto_align != 0 && ratio < 0.f;
    // gabor2 depends on the second image alone, not on its chain (and has its own buffers): it goes to the plan-upload stream — idle during a
    // set-up — and starts when the second image's medians are through, beside the strings of small dependent launches that follow them (at
    // the very start it ran beside the first medians, one wave per histogram set, and tripled their time; POPPY_GABOR2_FIRST: that order)
    // (errors go to the CALLING thread's string: with gabor2_late the first image's thread queues this while the second image's thread is inside foreground_b)
    auto gabor2_on_side_stream = [&](std::string& e, bool other_thread_in_fg_b) -> bool {
        std::string ge;
        const float* gab = c->foreground_b.gabor_field(c->c2, W, H, c->copy_stream, other_thread_in_fg_b ? &ge : nullptr);
        if (!gab) { e = "gabor_field: " + (other_thread_in_fg_b ? ge : c->foreground_b.err); return false; }
        if (hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, c->copy_stream) != hipSuccess) { e = "gabor2 copy failed"; return false; }
        return true;
    };
    if (!align_first && gabor2_first && !gabor2_on_side_stream(c->err, false)) return POPPY_E_DEVICE;
    if (!c->setup_ev) HIPCHK(c, hipEventCreateWithFlags(&c->setup_ev, hipEventDisableTiming));
    if (!c->c2_up_ev) HIPCHK(c, hipEventCreateWithFlags(&c->c2_up_ev, hipEventDisableTiming));
    // POPPY_GABOR2_AT: where gabor2 starts — 0 behind the second image's medians, 1 / 2 behind the FIRST image's ORB input / FAST kernels (queued by that
    // image's thread): the first image's chain is through earlier than the second's, gabor2 then fills the GPU beside the second chain's tail of small launches
    static const int gabor2_at = getenv("POPPY_GABOR2_AT") ? atoi(getenv("POPPY_GABOR2_AT")) : 2;      // (1080p 3.03 -> 2.96 ms, 4K 8.5 -> 8.1: tools/experiments/gabor2_at_ab.sh)
    const int gabor2_late = (!align_first && !gabor2_first && !serial_chains) ? gabor2_at : 0;
    if (gabor2_late && (c->foreground_b.prepare(W, H) || c->foreground_b.prepare2(W, H))) return fail(c, POPPY_E_DEVICE, "foreground buffers");
    c->foreground_b.medians_done = (!align_first && !gabor2_first && !gabor2_late) ? c->setup_ev : nullptr;
    // Each image's thread goes on to the detector's second half by itself as soon as BOTH details are known (nfeatures, src/extractor.cpp:40-45):
    // the other image's detail is ready long before its own candidates are, so nobody waits for a whole chain.  `details` counts the images
    // whose detail is published (or whose chain failed before it).
    struct Details {                                                      // a counter two threads wait on (no spinning: the wait can be a chain's length)
        std::mutex m; std::condition_variable cv; int n = 0;
        void add() { { std::lock_guard<std::mutex> g(m); ++n; } cv.notify_all(); }
        void wait_for(int k) { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return n >= k; }); }
    } details;
    // The staged upload of the second image happens on the second chain's thread and stream; gabor2 reads c2 on copy_stream, queued by the FIRST chain's
    // thread: that thread waits (host) until the upload has been QUEUED and its event recorded, then makes copy_stream wait for the event (device).
    // 0 = not yet, 1 = event recorded, -1 = the upload failed (round 5 had no such edge: gabor2 could read a half-written c2 when the helper thread was late)
    Details c2_uploaded;
    std::atomic<int> c2_upload_state{staged ? 0 : 1};
    std::vector<OrbKeyPoint> k1, k2;
    int nfeatures = 0;
    struct Publish {                                                      // counts once, at the detail or at whichever exit comes before it
        Details& d; bool done = false;
        void now() { if (!done) { done = true; d.add(); } }
        ~Publish() { now(); }
    };
    if (!on_device) {                                                    // which median kernel each image's chain takes: from a sample of the host pixels
        c->foreground.median_cols_hint = median_cols_hint_from_host(bgr1, s1, W, H);
        c->foreground_b.median_cols_hint = median_cols_hint_from_host(bgr2, s2, W, H);
    }
    auto chain_of = [&](int i) {
        Publish publish{details};
        struct UploadKnown {                                      // whichever way the second chain leaves, the first is not left waiting for its upload
            Details& d; std::atomic<int>& state; bool mine;
            ~UploadKnown() { if (mine && state.load() == 0) { state = -1; d.add(); } }
        } upload_known{c2_uploaded, c2_upload_state, i == 1 && staged};
        if (hipSetDevice(c->device) != hipSuccess) { errs[i] = "hipSetDevice failed"; rcs[i] = POPPY_E_DEVICE; return; }
        ForegroundFilter& fg = i ? c->foreground_b : c->foreground;
        hipStream_t st = i ? c->aux_stream : c->stream;
        if (i == 1 && staged) {
            const bool ok = copy_rows_async(c->c2, (size_t)W * 3, bgr2, s2, (size_t)W * 3, H, hipMemcpyHostToDevice, st) == hipSuccess &&
                            hipEventRecord(c->c2_up_ev, st) == hipSuccess;
            c2_upload_state = ok ? 1 : -1;
            c2_uploaded.add();
            if (!ok) { errs[i] = "pair_begin: upload of the second image failed"; rcs[i] = POPPY_E_DEVICE; return; }
        }
        const uint8_t* gf = fg.run_device(i ? c->c2 : c->c1, (size_t)W * 3, W, H, st, nullptr);
        if (!gf) { errs[i] = "foreground: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        if (i == 1 && fg.medians_done) {                          // gabor2 starts when the second image's medians are through (queued now, long before)
            if (hipStreamWaitEvent(c->copy_stream, fg.medians_done, 0) != hipSuccess) { errs[i] = "gabor2: stream wait failed"; rcs[i] = POPPY_E_DEVICE; return; }
            if (!gabor2_on_side_stream(errs[i], false)) { rcs[i] = POPPY_E_DEVICE; return; }
        }
        // dft_detail2 and the ORB input both read goodFeatures: the ORB input's kernels are queued behind dft_detail2's before the host waits for the detail value
        // (until round 4 the chain's stream ran dry twice in mid-chain, at the two read-backs of dft_detail2)
        if (fg.detail_begin(gf, W, H, st)) { errs[i] = "dft_detail2: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        const uint8_t* gi = fg.orb_input(gf, W, H, 0, st);
        if (!gi) { errs[i] = "orb_input: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        if (fg.detail_end(&d[i])) { errs[i] = "dft_detail2: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        publish.now();
        g_dev[i] = gi;                                            // the detector reads it where it lies; only ORB::compute wants a host copy
        auto gabor2_behind_this_chain = [&]() {
            if (staged) {                                         // c2 is written on the other chain's stream: order copy_stream behind that copy
                c2_uploaded.wait_for(1);
                if (c2_upload_state.load() < 0) return false;     // (the other chain reports the error)
                if (hipStreamWaitEvent(c->copy_stream, c->c2_up_ev, 0) != hipSuccess) { errs[i] = "gabor2: stream wait failed"; rcs[i] = POPPY_E_DEVICE; return false; }
            }
            if (hipEventRecord(c->setup_ev, st) != hipSuccess || hipStreamWaitEvent(c->copy_stream, c->setup_ev, 0) != hipSuccess) {
                errs[i] = "gabor2: stream wait failed"; rcs[i] = POPPY_E_DEVICE; return false;
            }
            if (!gabor2_on_side_stream(errs[i], true)) { rcs[i] = POPPY_E_DEVICE; return false; }
            return true;
        };
        if (i == 0 && gabor2_late == 1 && !gabor2_behind_this_chain()) return;
        hipError_t e = ratio >= 0.f ? hipMemcpyAsync(g[i].data(), gi, P, hipMemcpyDeviceToHost, st) : hipSuccess;
        if (e != hipSuccess) { errs[i] = std::string("pair_begin: ") + hipGetErrorString(e); rcs[i] = POPPY_E_DEVICE; return; }
        // the detector's first half needs no nfeatures (which takes BOTH images' detail, src/extractor.cpp:40-45): it follows the chain at once,
        // so the image that is through first does not wait for the other with the GPU half idle
        OrbDetector& orb = i ? c->orb_b : c->orb;
        if (orb.detect_begin(gi, W, W, H, st, true) < 0) { errs[i] = "orb_detect: " + orb.err; rcs[i] = POPPY_E_DEVICE; return; }
        if (i == 0 && gabor2_late == 2 && !gabor2_behind_this_chain()) return;
        if (serial_chains) return;                                // (one chain after the other: the second half follows below)
        details.wait_for(2);
        if (rcs[i ^ 1]) return;                                   // the other chain failed (its error is reported)
        const int nf = (int)(c->cfg.max_keypoints * (255.0 / std::max(d[0], d[1])));
        if (orb.detect_finish(nf, st, i ? k2 : k1) < 0) { errs[i] = "orb_detect: " + orb.err; rcs[i] = POPPY_E_DEVICE; }
        ms_chain_end[i] = since(t_begin);
    };
    if (serial_chains) {
        const double t0 = since(t_begin);
        chain_of(0);
        const double t1 = since(t_begin);
        chain_of(1);
        if (stage_times) fprintf(stderr, "  chains one after the other: image 1 %.3f ms, image 2 (+ gabor2) %.3f ms\n", t1 - t0, since(t_begin) - t1);
    } else {
        c->setup_worker.run([&]() { chain_of(1); });
        chain_of(0);
        if (!c->setup_worker.wait()) { c->err = "pair set-up helper thread: " + c->setup_worker.error(); return POPPY_E_DEVICE; }
    }
    c->foreground_b.medians_done = nullptr;
    ms_joined = since(t_begin);
    if (!align_first) HIPCHK(c, hipStreamSynchronize(c->copy_stream));                // gabor2 is in place
    for (int i = 0; i < 2; ++i) if (rcs[i].load()) { c->err = errs[i]; return rcs[i].load(); }
    ms_chains = since(t_begin);
    const double detail = 255.0 / std::max(d[0], d[1]);                 // src/extractor.cpp:40-45
    c->last_detail[0] = d[0]; c->last_detail[1] = d[1];
    nfeatures = (int)(c->cfg.max_keypoints * detail);
    c->last_nfeatures = nfeatures;
    if (serial_chains) {
        int r1 = 0, r2 = 0;
        c->setup_worker.run([&]() { r2 = hipSetDevice(c->device) == hipSuccess ? c->orb_b.detect_finish(nfeatures, c->aux_stream, k2) : -2; });
        r1 = c->orb.detect_finish(nfeatures, c->stream, k1);
        if (!c->setup_worker.wait()) { c->err = "pair set-up helper thread: " + c->setup_worker.error(); return POPPY_E_DEVICE; }
        if (r1 < 0 || r2 < 0) { c->err = "orb_detect: " + (r1 < 0 ? c->orb.err : c->orb_b.err); return POPPY_E_DEVICE; }
    }
    ms_detect = since(t_begin);
    if (ratio >= 0.f) {
        // Opt-in descriptor mode (the reference only sketched it, src/experiments.hpp:14-144): ORB::compute on both
        // keypoint sets, 2-NN Hamming both ways, ratio test, symmetry test; the surviving pairs, in query order, become the
        // point sets (out-of-image pairs dropped, the four corners appended).  No positional re-pairing, no threshold.
        std::vector<uint8_t> d1(k1.size() * 32), d2(k2.size() * 32);
        auto rows7 = [](const std::vector<OrbKeyPoint>& k) {            // cv::KeyPoint field order, all as float
            std::vector<float> r(k.size() * 7);
            for (size_t i = 0; i < k.size(); ++i) {
                float* o = &r[i * 7];
                o[0] = k[i].x; o[1] = k[i].y; o[2] = k[i].size; o[3] = k[i].angle; o[4] = k[i].response; o[5] = (float)k[i].octave; o[6] = (float)k[i].class_id;
            }
            return r;
        };
        const std::vector<float> r1v = rows7(k1), r2v = rows7(k2);
        {
            int r1 = 0, r2 = 0;
            std::thread other([&]() { r2 = hipSetDevice(c->device) == hipSuccess ? c->orb_b.describe(g[1].data(), W, W, H, r2v.data(), (int)k2.size(), c->aux_stream, d2.data()) : -2; });
            r1 = c->orb.describe(g[0].data(), W, W, H, r1v.data(), (int)k1.size(), c->stream, d1.data());
            other.join();
            if (r1 < 0 || r2 < 0) { c->err = "orb_describe: " + (r1 < 0 ? c->orb.err : c->orb_b.err); return POPPY_E_DEVICE; }
        }
        std::vector<int> k12(k1.size() * 4), k21(k2.size() * 4), sym;
        if (c->orb.hamming_knn2(d1.data(), (int)k1.size(), d2.data(), (int)k2.size(), c->stream, k12.data()) < 0 ||
            c->orb.hamming_knn2(d2.data(), (int)k2.size(), d1.data(), (int)k1.size(), c->stream, k21.data()) < 0) {
            c->err = "hamming_knn2: " + c->orb.err;
            return POPPY_E_DEVICE;
        }
        ratio_symmetry(k12.data(), (int)k1.size(), k21.data(), (int)k2.size(), ratio, sym);
        std::vector<P2f> a, b;
        for (size_t i = 0; i + 3 <= sym.size(); i += 3) {
            a.push_back(P2f{k1[sym[i]].x, k1[sym[i]].y});
            b.push_back(P2f{k2[sym[i + 1]].x, k2[sym[i + 1]].y});
        }
        drop_out_of_image(a, b, W, H);
        c->last_descriptor_matches = (int)a.size();
        if (a.empty()) return fail(c, POPPY_E_NOMATCH, "no symmetric descriptor matches");
        c->initial_morph_dist = morph_distance_ref(a, b, W, H);
        add_image_corners(a, b, W, H);
        rc = set_points(c, (const float*)a.data(), (const float*)b.data(), (int)a.size()); if (rc) return rc;
    } else {
        const size_t n = std::min(k1.size(), k2.size());                    // Extractor::points (extractor.cpp:96-99)
        std::vector<float> p1(n * 2), p2(n * 2), o1((n + 4) * 2), o2((n + 4) * 2);
        for (size_t i = 0; i < n; ++i) { p1[2 * i] = k1[i].x; p1[2 * i + 1] = k1[i].y; p2[2 * i] = k2[i].x; p2[2 * i + 1] = k2[i].y; }
        if (align_first) {                                                  // Matcher::find, src/matcher.cpp:29-32
            if (n < 4) return fail(c, POPPY_E_UNSUPPORTED, "auto-align needs at least 4 keypoint pairs (the reference reads 4 unconditionally)");
            std::vector<P2f> a(n), b(n);
            memcpy(a.data(), p1.data(), n * 8); memcpy(b.data(), p2.data(), n * 8);
            if (!c->c2_raw) HIPCHK(c, hipMalloc((void**)&c->c2_raw, P * 3 + 16));       // phase == 1 writes the image as it came in
            HIPCHK(c, hipMemcpyAsync(c->c2_raw, c->c2, P * 3, hipMemcpyDeviceToDevice, c->stream));
            c->c2_raw_valid = true;
            if (c->aligner.run(c->c2, W, H, a, b, c->stream, nullptr)) return fail(c, POPPY_E_DEVICE, c->aligner.err.c_str());
            memcpy(p2.data(), b.data(), n * 8);
            const float* gab = c->foreground_b.gabor_field(c->c2, W, H, c->stream);
            if (!gab) { c->err = "gabor_field: " + c->foreground_b.err; return POPPY_E_DEVICE; }
            HIPCHK(c, hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, c->stream));
        }
        int m = 0;
        const double ms_m0 = since(t_begin);
        rc = match_points_with(&c->setup_worker, p1.data(), p2.data(), (int)n, W, H, c->cfg.match_tolerance, o1.data(), o2.data(), &m, &c->initial_morph_dist);
        if (rc) return fail(c, rc, "poppy_match_points failed");
        const double ms_m1 = since(t_begin);
        rc = set_points(c, o1.data(), o2.data(), m); if (rc) return rc;
        if (stage_times) fprintf(stderr, "  match stage: points out of the keypoints %.3f, matcher %.3f, set_points %.3f ms (cumulative)\n", ms_m0, ms_m1, since(t_begin));
    }
    ms_match = since(t_begin);
    rc = finish_pair_load(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (stage_times) fprintf(stderr, "  chains: image 1 through %.3f, image 2 through %.3f, both joined %.3f, gabor2 in place %.3f ms\n", ms_chain_end[0], ms_chain_end[1], ms_joined, ms_chains);
    if (stage_times)
        fprintf(stderr, "pair set-up %dx%d: upload %.3f, chains %.3f, detect %.3f, match %.3f, finish %.3f ms (cumulative); before them (drain + queueing the raw pair's copies) %.3f ms\n", W, H, ms_upload, ms_chains,
                ms_detect, ms_match, since(t_begin), std::chrono::duration<double, std::milli>(t_begin - t_enter).count());
    return POPPY_OK;
}

int poppy_hip_pair_begin(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H) {
    return pair_begin_impl(c, bgr1, s1, bgr2, s2, W, H, -1.f);
}
int poppy_hip_pair_begin_device(poppy_hip_ctx* c, const void* d1, const void* d2, int W, int H) {
    return pair_begin_impl(c, (const uint8_t*)d1, (size_t)W * 3, (const uint8_t*)d2, (size_t)W * 3, W, H, -1.f, true);
}
void poppy_count_frames_cb(void* user, const uint8_t*, int, int, size_t) { if (user) ++*(long long*)user; }
int poppy_hip_pair_begin_descriptors(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H, float ratio) {
    if (!(ratio >= 0.f)) return c ? fail(c, POPPY_E_ARG, "ratio must be >= 0") : POPPY_E_ARG;
    return pair_begin_impl(c, bgr1, s1, bgr2, s2, W, H, ratio);
}

int poppy_hip_pair_corrected2(poppy_hip_ctx* c, uint8_t* dst, size_t ds) {
    if (!c || !dst) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no resident pair");
    if (ds < (size_t)c->W * 3) return fail(c, POPPY_E_ARG, "stride too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, copy_rows_async(dst, ds, c->c2, (size_t)c->W * 3, (size_t)c->W * 3, c->H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

// intermediates of the last poppy_hip_pair_begin, for the tolerance tests: nfeatures, the two dft_detail2 values
int poppy_hip_pair_begin_info(poppy_hip_ctx* c, int* nfeatures, double* detail2) {
    if (!c) return POPPY_E_ARG;
    if (nfeatures) *nfeatures = c->last_nfeatures;
    if (detail2) { detail2[0] = c->last_detail[0]; detail2[1] = c->last_detail[1]; }
    return POPPY_OK;
}

// Extractor::keypoints' image chain for one goodFeatures image (host in / out): us = grey(unsharp), gb = Gabor mean, g = ORB input
int poppy_hip_orb_input(poppy_hip_ctx* c, const uint8_t* good_features, int W, int H, uint8_t* g, float* us, float* gb, double* detail) {
    if (!c || !good_features || W <= 0 || H <= 0) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    ForegroundFilter& fg = c->foreground;
    if (fg.prepare(W, H)) { c->err = "foreground: " + fg.err; return POPPY_E_DEVICE; }
    uint8_t* d_gf = fg.bgr_staging();                                   // any w*h device bytes will do as the staging area
    HIPCHK(c, hipMemcpyAsync(d_gf, good_features, (size_t)W * H, hipMemcpyHostToDevice, c->stream));
    if (detail && fg.detail(d_gf, W, H, c->stream, detail)) { c->err = "dft_detail2: " + fg.err; return POPPY_E_DEVICE; }
    const uint8_t* gi = fg.orb_input(d_gf, W, H, 0, c->stream, us, gb);
    if (!gi) { c->err = "orb_input: " + fg.err; return POPPY_E_DEVICE; }
    if (g) HIPCHK(c, hipMemcpyAsync(g, gi, (size_t)W * H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

// 1: the Gabor banks as direct double-precision sums (kernels_prefilter2.hip), 0 (default): by tiled FFTs (kernels_gabor_fft.hip)
int poppy_hip_set_gabor_direct(poppy_hip_ctx* c, int on) {
    if (!c) return POPPY_E_ARG;
    c->foreground.gabor_direct = c->foreground_b.gabor_direct = on != 0;
    return POPPY_OK;
}

int poppy_hip_set_setup_chains(poppy_hip_ctx* c, int serial) {
    if (!c) return POPPY_E_ARG;
    c->setup_serial = serial != 0;
    return POPPY_OK;
}

int poppy_hip_gabor_doubt(unsigned long long out[3]) { return out && gabor_fft_doubt(out) ? POPPY_OK : POPPY_E_DEVICE; }

// gabor_filter(bgr / 255) with the default arguments (host in / out, f32x3)
int poppy_hip_gabor_field(poppy_hip_ctx* c, const uint8_t* bgr, size_t stride, int W, int H, float* out) {
    if (!c || !bgr || !out || W <= 0 || H <= 0 || stride < (size_t)W * 3) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    ForegroundFilter& fg = c->foreground;
    if (fg.prepare(W, H)) { c->err = "foreground: " + fg.err; return POPPY_E_DEVICE; }
    HIPCHK(c, copy_rows_async(fg.bgr_staging(), (size_t)W * 3, bgr, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
    const float* gab = fg.gabor_field(fg.bgr_staging(), W, H, c->stream);
    if (!gab) { c->err = "gabor_field: " + fg.err; return POPPY_E_DEVICE; }
    HIPCHK(c, hipMemcpyAsync(out, gab, (size_t)W * H * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

// blur_margin (src/util.cpp:574-602): what the reference's CLI does to every image before poppy::morph when the phase is not 0 / 1
// (src/poppy.cpp:233-240,293-308): centre it in the union canvas and blur the four margin strips (127x127, sigma 6, fixed point).
int poppy_hip_blur_margin(poppy_hip_ctx* c, const uint8_t* src, size_t stride, int W, int H, int UW, int UH, uint8_t* dst, size_t dst_stride) {
    if (!c) return POPPY_E_ARG;
    if (!src || !dst || W <= 0 || H <= 0 || UW < W || UH < H || stride < (size_t)W * 3 || dst_stride < (size_t)UW * 3) return fail(c, POPPY_E_ARG, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t UB = (size_t)UW * UH * 3;
    uint8_t *canvas = nullptr, *out = nullptr; uint32_t* tmp = nullptr; int* d_taps = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)canvas, (void*)out, (void*)tmp, (void*)d_taps}) if (p) (void)hipFree(p); };
    // taps: exp(-x^2 / 2 sigma^2) / sum in double, to 8 fractional bits with error diffusion, centre = 256 - rest (smooth.dispatch.cpp:224-258)
    const int n = 127; const double sigma = 6;
    std::vector<double> v(n); double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; v[i] = std::exp(-(x * x) / (2 * sigma * sigma)); sum += v[i]; }
    std::vector<int> taps(n, 0);
    { double err = 0; int tot = 0;
      for (int i = 0; i < n / 2; ++i) { const double adj = v[i] / sum * 256 + err; const int v0 = (int)std::nearbyint(adj); err = adj - v0; taps[i] = taps[n - 1 - i] = v0; tot += v0; }
      taps[n / 2] = 256 - 2 * tot; }
    hipError_t e = hipMalloc((void**)&canvas, UB);
    if (e == hipSuccess) e = hipMalloc((void**)&out, UB);
    if (e == hipSuccess) e = hipMalloc((void**)&tmp, UB * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_taps, n * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_taps, taps.data(), n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(canvas, 0, UB, c->stream);
    const double margin = (W + H) / 100.0;
    double dx = std::fabs((double)(W - UW)) / 2.0, dy = std::fabs((double)(H - UH)) / 2.0;
    const int rx = (int)dx, ry = (int)dy;
    if (e == hipSuccess) e = copy_rows_async(canvas + ((size_t)ry * UW + rx) * 3, (size_t)UW * 3, src, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, canvas, UB, hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) { cleanup(); c->err = std::string("blur_margin: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    dx = (dx == 0 ? 1.3 : dx + margin);
    dy = (dy == 0 ? 1.3 : dy + margin);
    const int rects[4][4] = {{0, 0, (int)dx, UH}, {(int)(UW - dx), 0, (int)dx, UH}, {0, 0, UW, (int)dy}, {0, (int)(UH - dy), UW, (int)dy}};
    for (const auto& r : rects) launch_strip_blur(canvas, out, UW, tmp, d_taps, n, r[0], r[1], r[2], r[3], c->stream);   // left, right, top, bottom: later strips win
    e = copy_rows_async(dst, dst_stride, out, (size_t)UW * 3, (size_t)UW * 3, UH, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) { c->err = std::string("blur_margin: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}

int poppy_dft_plan(int n, int* factors, int* n_factors, int* itab, float* wave) {
    if (n < 1 || !factors || !n_factors || !itab || !wave) return POPPY_E_ARG;
    DftPlanHost p;
    try { dft_make_plan(n, p); } catch (...) { return POPPY_E_UNSUPPORTED; }
    *n_factors = p.nf;
    memcpy(factors, p.factors, sizeof(int) * 34);
    memcpy(itab, p.itab.data(), sizeof(int) * (size_t)n);
    memcpy(wave, p.wave.data(), sizeof(float) * 2 * (size_t)n);
    return POPPY_OK;
}

// Host-side tables of the device code, for tests that run without a GPU.
// which = 31: Extractor::keypoints' bank (sigma 5, lambda 2), 13: gabor_filter's default bank (sigma 5, lambda 10).  bank: 16 * ks * ks floats
// (the kernels as getGaborKernel returns them); spectra: 8 * 4096 * 2 doubles = what kernels_gabor_fft.hip multiplies the patch spectrum with.
int poppy_gabor_tables(int which, float* bank, double* spectra) {
    if (which != 31 && which != 13) return POPPY_E_ARG;
    std::vector<float> b;
    gabor_bank(which, 5, which == 31 ? 2 : 10, 0.04, M_PI / 4, b);
    if (bank) memcpy(bank, b.data(), b.size() * 4);
    if (spectra) { const std::vector<double> t = gabor_fft_tables(b, which); memcpy(spectra, t.data(), t.size() * 8); }
    return POPPY_OK;
}
// The tap table of the pyramid tail (kernels.h: PyrTailPlan) of a width x height frame: info[0..5] = first level of the tail, multi-pixel
// level steps, single-pixel reductions (-1: none), descriptors, LDS bytes, usable (0 / 1); desc (optional, room for info[3] * 4 words).
int poppy_pyr_tail_plan(int width, int height, int pyramid_levels, int tail_px, int* info, unsigned* desc) {
    if (width < 1 || height < 1 || pyramid_levels < 1 || pyramid_levels > 256 || !info) return POPPY_E_ARG;
    std::vector<PyrLevel> lv(pyramid_levels + 1);
    size_t o3 = 0, o1 = 0;
    int w = width, h = height;
    for (int i = 0; i <= pyramid_levels; ++i) { lv[i] = PyrLevel{w, h, o3, o1}; o3 += (size_t)w * h * 3; o1 += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
    int first = pyramid_levels;
    for (int i = 1; i <= pyramid_levels; ++i) if ((size_t)lv[i].w * lv[i].h <= (size_t)tail_px) { first = i; break; }
    const PyrTailPlan p = build_pyr_tail_plan(lv.data(), first, pyramid_levels);
    info[0] = first; info[1] = p.args.n_wide; info[2] = p.args.nl; info[3] = p.args.n_desc; info[4] = (int)p.lds_bytes; info[5] = p.ok ? 1 : 0;
    if (desc && !p.desc.empty()) memcpy(desc, p.desc.data(), p.desc.size() * 4);
    return POPPY_OK;
}

int poppy_radial_gradient(int W, int H, float* out) {
    if (W <= 0 || H <= 0 || !out) return POPPY_E_ARG;
    std::vector<float> r;
    radial_gradient(W, H, r);
    memcpy(out, r.data(), r.size() * 4);
    return POPPY_OK;
}

int poppy_radial_mask(int W, int H, float* out) {
    if (W <= 0 || H <= 0 || !out) return POPPY_E_ARG;
    std::vector<float> r;
    radial_mask(W, H, r);
    memcpy(out, r.data(), r.size() * 4);
    return POPPY_OK;
}

}  // extern "C"
