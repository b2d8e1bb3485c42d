// foreground.cpp — Extractor::foreground (src/extractor.cpp:136-229) for one image, as a sequence of HIP launches:
//   grey -> MOG2 -> 12 x { median(8i+1) -> MOG2 -> accumulate -> Gaussian 23x23 } -> log mask -> equalizeHist.
// Everything stays in HBM between the steps; the only host work is the schedule of learning rates.
#include "foreground.h"
#include "kernels_prefilter.h"
#include "kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace poppy_hip {

#define FG_CHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return -2; } \
    } while (0)

ForegroundFilter::~ForegroundFilter() {
    release();
    release2();
    if (logtab) (void)hipFree(logtab);
    if (h_easy) (void)hipHostFree(h_easy);
    if (easy_ev) (void)hipEventDestroy(easy_ev);
    if (detail_ev) (void)hipEventDestroy(detail_ev);
}

void ForegroundFilter::release() {
    void* bufs[] = {padded, d_bgr, grey, meds, acc[0], acc[1], flows, masked, out, lut, tmp16, dbgf, hist, radial12, med_pres};
    for (void* b : bufs) if (b) (void)hipFree(b);
    d_bgr = grey = meds = acc[0] = acc[1] = flows = masked = out = lut = nullptr;
    tmp16 = nullptr; padded = nullptr; dbgf = nullptr; hist = nullptr; radial12 = nullptr; med_pres = nullptr;
    W = H = 0;
}

int ForegroundFilter::ensure(int w, int h) {
    if (!prepared) {
        if (!prepare_median_u8()) { err = "hipFuncSetAttribute(k_median_u8) failed"; return -2; }
        // cv::log's table (OCV/core/src/mathfuncs.cpp:2151-2408): (ln(1 + i/256), 1/(1 + i/256)) as doubles, used as floats;
        // the last pair is (ln 2, 1/2)
        std::vector<float> tab(512);
        for (int i = 0; i < 255; ++i) {
            const long double t = 1.0L + (long double)i / 256.0L;
            tab[2 * i] = (float)(double)logl(t);
            tab[2 * i + 1] = (float)(double)(1.0L / t);
        }
        tab[510] = (float)0.69314718055994530941723212145818; tab[511] = 0.5f;
        FG_CHK(hipMalloc((void**)&logtab, 512 * sizeof(float)));
        FG_CHK(hipMemcpy(logtab, tab.data(), 512 * sizeof(float), hipMemcpyHostToDevice));
        prepared = true;
    }
    if (w == W && h == H) return 0;
    float* keep = logtab;
    release();
    logtab = keep;
    const size_t P = (size_t)w * h;
    FG_CHK(hipMalloc((void**)&d_bgr, P * 3)); FG_CHK(hipMalloc((void**)&grey, P));
    FG_CHK(hipMalloc((void**)&meds, P * 12));                       // the twelve median planes
    FG_CHK(hipMalloc((void**)&acc[0], P)); FG_CHK(hipMalloc((void**)&acc[1], P));
    FG_CHK(hipMalloc((void**)&flows, P * kMog2Steps));             // one mask plane per MOG2 step
    FG_CHK(hipMalloc((void**)&masked, P)); FG_CHK(hipMalloc((void**)&out, P)); FG_CHK(hipMalloc((void**)&lut, 256));
    FG_CHK(hipMalloc((void**)&tmp16, P * 2));
    FG_CHK(hipMalloc((void**)&padded, 2 * median_padded_bytes(w, h)));        // two: a median reads one and writes the next one's
    FG_CHK(hipMalloc((void**)&hist, 256 * sizeof(unsigned)));
    FG_CHK(hipMalloc((void**)&med_pres, (2 * median_presence_words(w, h) + 4) * sizeof(uint32_t)));   // presence maps of a median's source and result tiles + a counter
    if (!h_easy) FG_CHK(hipHostMalloc((void**)&h_easy, 16));
    W = w; H = h;
    return 0;
}

const uint8_t* ForegroundFilter::run_device(const uint8_t* bgr, size_t stride, int w, int h, hipStream_t s, const ForegroundDebugOut* dbg) {
    if (ensure(w, h)) return nullptr;
    const size_t P = (size_t)w * h;
    const int n = (int)P;
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e); };
    auto stage_out = [&](int k, const uint8_t* d) {           // debug: copy one stage plane to the host (stream ordered)
        if (dbg && dbg->stages) chk(hipMemcpyAsync(dbg->stages + (size_t)k * P, d, P, hipMemcpyDeviceToHost, s), "stage copy");
    };
    err.clear();
    launch_bgr2gray(bgr, stride, grey, w, h, s);
    // 1. the chain of input images: the grey image and its progressive medians (ksize 1, 9, ..., 89; each of the previous one).
    //    It does not depend on the masks, so it runs first and every plane is kept.
    const uint8_t* inputs[kMog2Steps];
    inputs[0] = grey;
    uint8_t* pad[2] = {padded, padded + median_padded_bytes(w, h)};
    int pc = 0, pp = 0;
    bool pres_valid = false;
    uint32_t* pres[2] = {med_pres, med_pres + median_presence_words(w, h)};
    // which kernel the large windows take (see foreground.h): the caller's hint, or the device's own count of tiles with few values
    static const int forced_cols = getenv("POPPY_MED_COLS_FORCE") ? atoi(getenv("POPPY_MED_COLS_FORCE")) : -1;
    int use_cols = forced_cols >= 0 ? forced_cols : median_cols_hint;
    median_cols_hint = -1;
    if (median_cols_min_ksize() > 89) use_cols = 0;
    // images with many values per tile (photographs): the lane-per-column kernel's time grows with the window, the column histograms' does not — they take over later
    const int cols_from_hard = forced_cols == 0 || median_cols_min_ksize() > 89 ? 999 : median_cols_min_ksize_hard();
    bool ask_device = false;
    if (use_cols < 0) {
        // the device counts the grey image's easy tiles now; the answer is read when the first large window comes up — behind the small windows' launches,
        // so that the host does not stall the queue for it
        uint32_t* d_easy = med_pres + 2 * median_presence_words(w, h);
        if (!easy_ev) chk(hipEventCreateWithFlags(&easy_ev, hipEventDisableTiming), "event create");
        chk(hipMemsetAsync(d_easy, 0, 4, s), "memset");
        launch_median_presence(grey, pres[pp], w, h, d_easy, s);
        chk(hipMemcpyAsync(h_easy, d_easy, 4, hipMemcpyDeviceToHost, s), "copy");
        if (easy_ev) chk(hipEventRecord(easy_ev, s), "event record");
        ask_device = true;
        pres_valid = median_cols_min_ksize() <= 9;            // (of the grey image: the first median's source)
    }
    launch_pad_cols(grey, pad[pc], w, h, s);                  // medianBlur(ksize 1) is a copy: the first real median reads the grey image
    for (int i = 0; i < 12; ++i) {
        uint8_t* med = meds + (size_t)i * P;
        const int ksize = i * 8 + 1;
        // medianBlur(ksize 1) is a copy: MOG2 reads the grey image itself unless the caller wants every stage plane
        if (ksize <= 1) { if (dbg && dbg->stages) chk(hipMemcpyAsync(med, grey, P, hipMemcpyDeviceToDevice, s), "copy"); else med = grey; }
        // (The round-5 advisor's note: since the column kernel takes every window this read comes before a single median is queued — a device-to-host round trip with the chain's
        // stream dry.  Measured in round 6 (profiles/r06_notes.md section 7): letting the first two windows take k_median_u8 while the answer travels costs more than the round trip —
        // set-up from device images 2.61 against 2.56 ms, the pool the same either way — so the read stays where it is.)
        if (ask_device && ksize >= median_cols_min_ksize()) {
            chk(easy_ev ? hipEventSynchronize(easy_ev) : hipStreamSynchronize(s), "sync");
            const MedianColsGeom g = median_cols_geom(w, h);
            use_cols = (unsigned long long)h_easy[0] * 10 >= (unsigned long long)g.tiles_x * g.tiles_y * 9;
            ask_device = false;
        }
        if (ksize <= 1) {}
        else if (ksize >= (use_cols > 0 ? median_cols_min_ksize() : cols_from_hard)) {   // by column histograms: every launch also leaves the presence map of its result's tiles
            if (!pres_valid) { launch_median_presence(inputs[i], pres[pp], w, h, nullptr, s); pres_valid = true; }
            launch_median_cols(pad[pc], med, i < 11 ? pad[pc ^ 1] : nullptr, w, h, ksize, pres[pp], pres[pp ^ 1], 0, s);
            pc ^= 1; pp ^= 1;
        } else {                                              // every median writes its result twice: tight (MOG2's input) and padded (the next median's)
            launch_median_padded(pad[pc], med, i < 11 ? pad[pc ^ 1] : nullptr, w, h, ksize, s);
            pc ^= 1; pres_valid = false;
        }
        inputs[i + 1] = med;
    }
    if (medians_done) chk(hipEventRecord(medians_done, s), "event record");
    // 2. all thirteen MOG2 applies in one launch (mixture in registers), one mask plane per step
    float alphaT[kMog2Steps], prune[kMog2Steps];
    for (int i = 0; i < kMog2Steps; ++i) {
        const double lr = 1. / std::min(2 * (i + 1), 500);   // learningRate < 0 -> 1 / min(2 nframes, history)
        alphaT[i] = (float)lr; prune[i] = (float)(-lr * 0.05f);
    }
    launch_mog2_all(inputs, alphaT, prune, kMog2Steps, flows, n, s);
    // 3. fgMask: accumulate the masks, blurring the accumulator after each of the twelve median steps
    const float acc_scale = (float)(1.0 / (12 / 2.0));       // flow * (1.0 / (iterations / 2.0)), iterations = 12
    int cur = 0;                                              // acc[cur] is fgMask (Mat::zeros: the first step writes it)
    launch_acc_flow(acc[cur], flows, n, acc_scale, s, true);
    stage_out(0, flows); stage_out(1, acc[cur]);
    // accumulate + blur: POPPY_ACC_STEPS of the twelve dependent steps per launch (1, 2, 3, 4, 6; 0: the byte-per-thread kernel, one step each)
    static const int acc_steps = getenv("POPPY_ACC_STEPS") ? atoi(getenv("POPPY_ACC_STEPS")) : 3;
    const bool fused = !(dbg && dbg->stages) && acc_steps > 0 && 12 % acc_steps == 0 && acc_steps != 5 && acc_steps <= 6 && acc_gauss23_fused_takes(w, h) && P % 4 == 0;
    for (int i = 0; fused && i < 12; i += acc_steps) {
        launch_acc_gauss23_fused(acc[cur], flows + (size_t)(i + 1) * P, P, acc_steps, acc[cur ^ 1], w, h, acc_scale, s);
        cur ^= 1;
    }
    for (int i = 0; !fused && i < 12; ++i) {
        if (!(dbg && dbg->stages)) {                          // accumulate + blur in one launch; the staged form below also shows the sum before the blur
            launch_acc_gauss23(acc[cur], flows + (size_t)(i + 1) * P, acc[cur ^ 1], w, h, acc_scale, s);
            cur ^= 1;
            continue;
        }
        launch_acc_flow(acc[cur], flows + (size_t)(i + 1) * P, n, acc_scale, s);
        stage_out(2 + 4 * i, inputs[i + 1]); stage_out(3 + 4 * i, flows + (size_t)(i + 1) * P); stage_out(4 + 4 * i, acc[cur]);
        launch_gauss23_u8(acc[cur], tmp16, acc[cur ^ 1], w, h, s);
        cur ^= 1;
        stage_out(5 + 4 * i, acc[cur]);
    }
    float* dbg_floats = nullptr;
    if (dbg && dbg->floats) {
        if (!dbgf) chk(hipMalloc((void**)&dbgf, P * 12), "hipMalloc");
        dbg_floats = dbgf;
    }
    if (radial_mask_on && !radial12) {                          // once per geometry: ~0.1 us per pixel of host libm, on threads
        std::vector<float> rad;
        radial_mask(w, h, rad);
        chk(hipMalloc((void**)&radial12, P * 4), "hipMalloc");
        if (radial12 && hipMemcpy(radial12, rad.data(), P * 4, hipMemcpyHostToDevice) != hipSuccess) {
            // a table that did not arrive must not survive: the next pair would multiply its mask with uninitialised memory unnoticed
            (void)hipFree(radial12); radial12 = nullptr;
            chk(hipErrorUnknown, "radial mask upload");
        }
        if (!radial12) return nullptr;
    }
    launch_fg_tail(grey, acc[cur], logtab, radial_mask_on ? radial12 : nullptr, masked, hist, lut, out, dbg_floats, n, s);
    if (dbg) {
        if (dbg->grey) chk(hipMemcpyAsync(dbg->grey, grey, P, hipMemcpyDeviceToHost, s), "copy");
        if (dbg->masked) chk(hipMemcpyAsync(dbg->masked, masked, P, hipMemcpyDeviceToHost, s), "copy");
        if (dbg->floats && dbgf) chk(hipMemcpyAsync(dbg->floats, dbgf, P * 12, hipMemcpyDeviceToHost, s), "copy");
    }
    chk(hipGetLastError(), "launch");
    return err.empty() ? out : nullptr;
}

int ForegroundFilter::median(const uint8_t* src, int w, int h, int ksize, int form, hipStream_t s, uint8_t* dst) {
    if (!src || !dst || w <= 0 || h <= 0 || ksize < 3 || ksize > 89 || !(ksize & 1) || form < 0 || form > 7) { err = "bad arguments"; return -1; }
    if (ensure(w, h)) return -2;
    const size_t P = (size_t)w * h;
    FG_CHK(hipMemcpyAsync(grey, src, P, hipMemcpyHostToDevice, s));
    launch_pad_cols(grey, padded, w, h, s);
    if (form == 0) form = ksize >= median_cols_min_ksize() ? 2 : 1;
    if (form == 1) launch_median_padded(padded, meds, nullptr, w, h, ksize, s);
    else {
        if (form != 3) launch_median_presence(grey, med_pres, w, h, nullptr, s);
        launch_median_cols(padded, meds, nullptr, w, h, ksize, form != 3 ? med_pres : nullptr, med_pres + median_presence_words(w, h), form == 3 || form == 5 ? 1 : form == 4 ? 2 : form == 6 ? 5 : form == 7 ? 9 : 0, s);
    }
    FG_CHK(hipGetLastError());
    FG_CHK(hipMemcpyAsync(dst, meds, P, hipMemcpyDeviceToHost, s));
    FG_CHK(hipStreamSynchronize(s));
    return 0;
}

int ForegroundFilter::run(const uint8_t* bgr, size_t stride, int w, int h, hipStream_t s, uint8_t* dst, const ForegroundDebugOut* dbg) {
    if (!bgr || !dst || w <= 0 || h <= 0 || stride < (size_t)w * 3) { err = "bad arguments"; return -1; }
    if (ensure(w, h)) return -2;
    FG_CHK(copy_rows_async(d_bgr, (size_t)w * 3, bgr, stride, (size_t)w * 3, h, hipMemcpyHostToDevice, s));
    median_cols_hint = median_cols_hint_from_host(bgr, stride, w, h);
    const uint8_t* r = run_device(d_bgr, (size_t)w * 3, w, h, s, dbg);
    if (!r) return -2;
    FG_CHK(hipMemcpyAsync(dst, r, (size_t)w * h, hipMemcpyDeviceToHost, s));
    FG_CHK(hipStreamSynchronize(s));
    return 0;
}

}  // namespace poppy_hip
