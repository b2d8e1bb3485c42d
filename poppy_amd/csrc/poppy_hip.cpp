// poppy_hip.cpp — context, HBM layout and orchestration behind the C ABI of include/poppy_hip.h.
//
// Host C++ (as the reference's own morph driver is, src/poppy.hpp:46-248) calling the hand-written
// gfx950 kernels of kernels_*.hip.  Per pair everything stays resident in HBM; per frame the host only
// plans the mesh (frame_plan.cpp, ~1k triangles) and uploads ~100 KB through a pinned ring.
//
// HBM layout for a W x H pair (P = W*H):
//   c1, c2            u8x3   3P each     sources (c1 is replaced by the previous frame in chained mode)
//   m2                f32    4P          1 - gray(gabor2), loop invariant (algo.cpp:250-252)
// and per frame SLOT (kSlots of them, each with its own HIP stream, so that the parts of frame j+1 that do not
// depend on frame j — id-map clear, raster, mask — or, in phase mode, whole frames run beside frame j's long tail
// of small launch-latency-bound pyramid kernels):
//   triMap            i32    4P          triangle id per pixel
//   tr1, tr2          u8x3   3P each     warped sources (never widened to float in memory)
//   pyrL, pyrR        f32x3  ~4P each    Gaussian levels 1..levels of the warped sources
//   pyrM              f32    ~5.3P       mask levels 0..levels
//   pyrB              f32x3  ~16P        blended levels; level 0 is lapBlend
//   out               u8x3   3P          the frame (chained mode feeds it to the next slot as c1)
// No CPU fallback exists in this library: every entry point either runs the kernels or fails.
#include "../../include/poppy_hip.h"
#include "foreground.h"
#include "frame_plan.h"
#include "kernels.h"
#include "kernels_prefilter.h"
#include "orb_detect.h"
#include "point_match.h"
#include "auto_align.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace poppy_hip;

static std::string g_create_error;

struct FrameSlot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;           // end of the frame last rendered here
    hipEvent_t prepared = nullptr;       // id map and mask of the frame being rendered here are written
    uint8_t *tr1 = nullptr, *tr2 = nullptr, *out = nullptr;
    float *pyrL = nullptr, *pyrR = nullptr, *pyrM = nullptr, *pyrB = nullptr;
    float *tmp = nullptr, *diff = nullptr;      // only for 1-pixel-wide / -high images (separate unsharp passes)
    float* unsharpF = nullptr;                  // debug copy of the float unsharp result, allocated on demand
    int32_t* triMap = nullptr;
    bool map_clean = false;                     // triMap is all zero (the warp kernel clears it behind itself)
    uint8_t *h_blob = nullptr, *d_blob = nullptr;   // this slot's frame plan (pinned host copy, device copy)
    void* h_blob_dev = nullptr;                     // device-side address of the pinned copy
    hipEvent_t uploaded = nullptr;                  // the device copy is complete
    hipGraphExec_t body = nullptr;                  // pyrdown .. unsharp of this slot, captured once per pair geometry
};
constexpr size_t kBlobHeader = 64;

struct poppy_hip_ctx {
    int device = 0;
    poppy_settings cfg;
    hipStream_t stream = nullptr;
    std::string err;

    int W = 0, H = 0;
    bool pair_ready = false;
    // resident buffers
    uint8_t *c1 = nullptr, *c2 = nullptr;
    float *gabor2 = nullptr, *m2 = nullptr;
    std::vector<FrameSlot> slots;        // per-frame working sets, used round-robin
    hipEvent_t inputs_ready = nullptr;   // c1 / c2 / m2 written (recorded on `stream` by the pair loaders)
    const uint8_t* cur1 = nullptr;       // what the next frame warps as "corrected1"
    hipEvent_t cur1_ready = nullptr;     // producer of cur1 when it is a slot's output, else null
    hipStream_t cur1_stream = nullptr;   // ... and the stream that producer ran on
    // Frames that feed on a previous frame all run on `stream`: a cross-stream event on the critical path costs more
    // than the kernels it would overlap.  Frames that read the loaded image run entirely on their slot's stream
    // (created on first use: every extra stream competes for the few hardware queues), so in phase mode several
    // frames are in flight at once.
    int next_slot = 0, last_slot = -1;
    std::vector<PyrLevel> levels;        // 0..pyramid_levels
    PyrLevel* d_levels = nullptr;
    int first_tail = 1;
    // points
    std::vector<P2f> pts1_0, pts1, pts2;
    // per-frame plan blobs (pinned host + device) live in the frame slots, so the host can plan ahead of the GPU
    int max_tris = 0;
    // blob layout: [header 64 B: f32 unsharp amount][warp records (T+1)*20 f32][tri_xy T*6 i32][inv1 T*9 f32][inv2 T*9 f32]
    //              [RasterTri T][work 2*n i32]
    size_t blob_bytes = 0;
    int tail_n3 = 0, tail_n1 = 0, tail_k1 = 0;    // tail_k1: first single-pixel level (or the last level)
    hipStream_t copy_stream = nullptr;
    FramePlan plan;
    OrbDetector orb, orb_b;
    ForegroundFilter foreground, foreground_b;      // two instances: the images of a pair are filtered side by side
    hipStream_t aux_stream = nullptr;
    double initial_morph_dist = 0;
    int last_nfeatures = 0;
    AutoAligner aligner;
    uint8_t* d_align = nullptr; size_t d_align_bytes = 0;      // staging image of the host-facing align entry points
    int last_descriptor_matches = 0;               // symmetric matches kept by the last pair_begin_descriptors
    bool last_warp_fast = false;                   // which warp kernel the last submitted frame used
    double last_detail[2] = {0, 0};
    // diagnostics
    bool debug = false;
    int timing = 0;                      // 0 off, 1 every kernel group (direct launches), 2 the warp kernel only
    struct Mark { const char* name; hipEvent_t ev; };   // name == nullptr opens a frame
    std::vector<Mark> marks; size_t marks_used = 0;
    // staging for host-image entry points
    uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0;
    static const int kStageRing = 3;          // pinned frames in flight towards the writer
    hipStream_t dl_stream = nullptr;
    hipEvent_t dl_done[kStageRing] = {};
};

#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return POPPY_E_DEVICE;                                                                    \
        }                                                                                             \
    } while (0)

static int fail(poppy_hip_ctx* c, int code, const char* msg) { c->err = msg; return code; }

extern "C" {

void poppy_settings_default(poppy_settings* s) {
    s->number_of_frames = 60; s->match_tolerance = 1.0; s->max_keypoints = 300; s->pyramid_levels = 64; s->enable_radial_mask = 0; s->enable_auto_align = 0;
}
const char* poppy_hip_create_error(void) { return g_create_error.c_str(); }
const char* poppy_hip_last_error(const poppy_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

poppy_hip_ctx* poppy_hip_create(int device, const poppy_settings* settings) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_create_error = "no HIP device (libpoppy_hip has no CPU fallback)"; return nullptr; }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return nullptr; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only";
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return nullptr; }
    poppy_hip_ctx* c = new poppy_hip_ctx();
    c->device = device;
    if (settings) c->cfg = *settings; else poppy_settings_default(&c->cfg);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
    (void)hipEventCreateWithFlags(&c->inputs_ready, hipEventDisableTiming);
    int k = 4;                                             // frames in flight
    if (const char* e = getenv("POPPY_HIP_SLOTS")) k = atoi(e);
    c->slots.resize(std::max(2, std::min(k, 8)));
    for (FrameSlot& f : c->slots) {
        if (hipEventCreateWithFlags(&f.prepared, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.uploaded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "hipStreamCreate failed"; delete c; return nullptr;
        }
    }
    return c;
}

static void free_pair(poppy_hip_ctx* c) {
    void* bufs[] = {c->c1, c->c2, c->gabor2, c->m2, c->d_levels};
    for (void* b : bufs) if (b) (void)hipFree(b);
    c->c1 = c->c2 = nullptr; c->gabor2 = c->m2 = nullptr; c->d_levels = nullptr;
    for (FrameSlot& f : c->slots) {
        void* fb[] = {f.tr1, f.tr2, f.out, f.pyrL, f.pyrR, f.pyrM, f.pyrB, f.tmp, f.diff, f.unsharpF, f.triMap};
        for (void* b : fb) if (b) (void)hipFree(b);
        f.tr1 = f.tr2 = f.out = nullptr; f.pyrL = f.pyrR = f.pyrM = f.pyrB = f.tmp = f.diff = f.unsharpF = nullptr; f.triMap = nullptr;
    }
    for (FrameSlot& f : c->slots) {
        if (f.body) { (void)hipGraphExecDestroy(f.body); f.body = nullptr; }
        if (f.h_blob) (void)hipHostFree(f.h_blob);
        if (f.d_blob) (void)hipFree(f.d_blob);
        f.h_blob = nullptr; f.d_blob = nullptr;
    }
    c->max_tris = 0; c->W = c->H = 0; c->pair_ready = false;
}

void poppy_hip_destroy(poppy_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (FrameSlot& f : c->slots) if (f.stream) (void)hipStreamSynchronize(f.stream);
    free_pair(c);
    for (FrameSlot& f : c->slots) {
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.prepared) (void)hipEventDestroy(f.prepared);
        if (f.uploaded) (void)hipEventDestroy(f.uploaded);
        if (f.stream) (void)hipStreamDestroy(f.stream);
    }
    if (c->inputs_ready) (void)hipEventDestroy(c->inputs_ready);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->dl_stream) { (void)hipStreamSynchronize(c->dl_stream); (void)hipStreamDestroy(c->dl_stream); }
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    for (hipEvent_t e : c->dl_done) if (e) (void)hipEventDestroy(e);
    for (auto& m : c->marks) (void)hipEventDestroy(m.ev);
    (void)hipStreamSynchronize(c->copy_stream);
    if (c->d_align) (void)hipFree(c->d_align);
    c->aligner.release();
    (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int poppy_warp_records(const float* inv1, const float* inv2, int n_tris, int width, int height, float* records) {
    if (n_tris < 0 || width < 1 || height < 1 || !records || (n_tris > 0 && (!inv1 || !inv2))) return POPPY_E_ARG;
    return pack_warp_records(inv1, inv2, n_tris, width, height, records) ? 1 : 0;
}
int poppy_hip_last_warp_kind(poppy_hip_ctx* c) { return c ? (c->last_warp_fast ? 1 : 0) : POPPY_E_ARG; }
int poppy_hip_set_debug(poppy_hip_ctx* c, int on) { if (!c) return POPPY_E_ARG; c->debug = on != 0; return POPPY_OK; }
int poppy_hip_set_timing(poppy_hip_ctx* c, int on) { if (!c) return POPPY_E_ARG; c->timing = on < 0 ? 0 : on; c->marks_used = 0; return POPPY_OK; }
void* poppy_hip_stream(poppy_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }
// `stream` waits on every frame as it is queued (submit_frame), so draining it drains the slots' streams too
int poppy_hip_sync(poppy_hip_ctx* c) { if (!c) return POPPY_E_ARG; HIPCHK(c, hipStreamSynchronize(c->stream)); return POPPY_OK; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
static int ensure_ring(poppy_hip_ctx* c, int n_points) {
    int need = 2 * n_points + 16;            // a planar triangulation of n points has < 2n triangles
    if (need <= c->max_tris) return POPPY_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    // worst case every triangle spans the whole image height
    const size_t items = (size_t)need * ((size_t)c->H / kRasterChunkRows + 3);
    const size_t bytes = ((kBlobHeader + (size_t)(need + 1) * kWarpRecordFloats * 4 +
                          (size_t)need * (6 * 4 + 18 * 4 + sizeof(RasterTri)) + items * 8 + 15) / 16) * 16;
    for (FrameSlot& f : c->slots) {
        if (f.body) { (void)hipGraphExecDestroy(f.body); f.body = nullptr; }      // it holds a pointer into the blob
        if (f.h_blob) (void)hipHostFree(f.h_blob);
        if (f.d_blob) (void)hipFree(f.d_blob);
        f.h_blob = f.d_blob = nullptr;
        HIPCHK(c, hipHostMalloc((void**)&f.h_blob, bytes, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer(&f.h_blob_dev, f.h_blob, 0));
        HIPCHK(c, hipMalloc((void**)&f.d_blob, bytes));
    }
    c->blob_bytes = bytes;
    c->max_tris = need;
    return POPPY_OK;
}

static int alloc_pair(poppy_hip_ctx* c, int W, int H) {
    if (c->W == W && c->H == H && c->c1) return POPPY_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (FrameSlot& f : c->slots) if (f.stream) HIPCHK(c, hipStreamSynchronize(f.stream));
    free_pair(c);
    if (c->cfg.pyramid_levels < 1 || c->cfg.pyramid_levels > 256) return fail(c, POPPY_E_UNSUPPORTED, "pyramid_levels must be in [1,256]");
    const size_t P = (size_t)W * H;
    const int L = c->cfg.pyramid_levels;
    c->levels.resize(L + 1);
    size_t off3 = 0, off1 = 0;
    int w = W, h = H;
    for (int i = 0; i <= L; ++i) {
        c->levels[i] = PyrLevel{w, h, off3, off1};
        off3 += (size_t)w * h * 3; off1 += (size_t)w * h;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    c->first_tail = L;
    static const size_t tail_px = getenv("POPPY_TAIL_PX") ? (size_t)atoi(getenv("POPPY_TAIL_PX")) : 600;
    for (int i = 1; i <= L; ++i)
        if ((size_t)c->levels[i].w * c->levels[i].h <= tail_px) { c->first_tail = i; break; }   // everything below runs in ONE workgroup: keep it small
    c->tail_n3 = c->tail_n1 = 0;
    c->tail_k1 = L;
    for (int i = std::min(c->first_tail, L); i <= L; ++i)
        if (c->levels[i].w == 1 && c->levels[i].h == 1) { c->tail_k1 = i; break; }
    for (int i = c->first_tail; i <= L; ++i) { c->tail_n3 += c->levels[i].w * c->levels[i].h * 3; c->tail_n1 += c->levels[i].w * c->levels[i].h; }
    const size_t tail_lds = pyr_tail_lds_bytes(L, c->tail_n3, c->tail_n1);
    if (tail_lds > 160 * 1024 || !prepare_pyr_tail(tail_lds))
        return fail(c, POPPY_E_UNSUPPORTED, "pyramid_levels too small for this image size: the coarsest level must fit the LDS-resident tail kernel");
    // +16: k_warp4 fetches footprints with 8-byte loads (6 bytes used), the last one may run 2 bytes past the image
    HIPCHK(c, hipMalloc((void**)&c->c1, P * 3 + 16)); HIPCHK(c, hipMalloc((void**)&c->c2, P * 3 + 16));
    HIPCHK(c, hipMalloc((void**)&c->gabor2, P * 12)); HIPCHK(c, hipMalloc((void**)&c->m2, P * 4));
    for (FrameSlot& f : c->slots) {
        HIPCHK(c, hipMalloc((void**)&f.tr1, P * 3 + 16)); HIPCHK(c, hipMalloc((void**)&f.tr2, P * 3 + 16));
        HIPCHK(c, hipMalloc((void**)&f.out, P * 3 + 16));
        HIPCHK(c, hipMalloc((void**)&f.triMap, P * 4));
        f.map_clean = false;
        HIPCHK(c, hipMalloc((void**)&f.pyrL, off3 * 4)); HIPCHK(c, hipMalloc((void**)&f.pyrR, off3 * 4));
        HIPCHK(c, hipMalloc((void**)&f.pyrB, off3 * 4)); HIPCHK(c, hipMalloc((void**)&f.pyrM, off1 * 4));
        if (W < 2 || H < 2) { HIPCHK(c, hipMalloc((void**)&f.tmp, P * 12)); HIPCHK(c, hipMalloc((void**)&f.diff, P * 12)); }
    }
    HIPCHK(c, hipMalloc((void**)&c->d_levels, sizeof(PyrLevel) * (L + 1)));
    HIPCHK(c, hipMemcpy(c->d_levels, c->levels.data(), sizeof(PyrLevel) * (L + 1), hipMemcpyHostToDevice));
    c->W = W; c->H = H;
    return POPPY_OK;
}

static int set_points(poppy_hip_ctx* c, const float* p1, const float* p2, int n) {
    if (n < 0 || (n > 0 && (!p1 || !p2))) return fail(c, POPPY_E_ARG, "bad point sets");
    c->pts1_0.resize(n); c->pts2.resize(n);
    if (n) { memcpy(c->pts1_0.data(), p1, (size_t)n * 8); memcpy(c->pts2.data(), p2, (size_t)n * 8); }
    c->pts1 = c->pts1_0;
    return ensure_ring(c, n);
}

static int finish_pair_load(poppy_hip_ctx* c) {
    launch_gray_inv(c->gabor2, c->m2, c->W * c->H, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->inputs_ready, c->stream));
    c->cur1 = c->c1; c->cur1_ready = nullptr; c->last_slot = -1; c->pair_ready = true;
    return POPPY_OK;
}

static int stage_host(poppy_hip_ctx* c, size_t bytes) {
    if (bytes <= c->h_stage_bytes) return POPPY_OK;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, bytes));
    c->h_stage_bytes = bytes;
    return POPPY_OK;
}

// Per-kernel timing: events are only RECORDED while frames are queued (no host sync); they are resolved
// in poppy_hip_timing_summary() after the caller has drained the stream.
struct Timer {
    poppy_hip_ctx* c;
    Timer(poppy_hip_ctx* c_, hipStream_t s_) : c(c_), s(s_) {}
    hipStream_t s = nullptr;
    hipEvent_t take(const char* name) {           // next event of the pool, labelled, not recorded
        if (c->marks_used >= c->marks.size()) { hipEvent_t e; (void)hipEventCreate(&e); c->marks.push_back({nullptr, e}); }
        c->marks[c->marks_used].name = name;
        return c->marks[c->marks_used++].ev;
    }
    void mark(const char* name) {
        if (!c->timing) return;
        (void)hipEventRecord(take(name), s);
    }
};

static int submit_frame(poppy_hip_ctx* c, double mask, bool chain);

// one frame on the resident pair; result in frame[slot]
static int render_frame(poppy_hip_ctx* c, double shape, double mask, bool chain) {
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    if (c->pts1.empty()) return fail(c, POPPY_E_NOMATCH, "no point pairs (use poppy_hip_dissolve)");
    int rc = plan_frame(c->W, c->H, c->pts1, c->pts2, shape, c->plan);
    if (rc) return fail(c, POPPY_E_RANGE, "point outside the image rectangle (Subdiv2D::insert would throw)");
    return submit_frame(c, mask, chain);
}

// Multi-frame calls plan on a small pool of host threads: only the POINT chain is sequential in chained mode
// (src/poppy.hpp:178-179,218: srcPoints1 <- morphedPoints), and that is a few hundred multiply-adds per frame; the
// triangulation and matrix work of the frames is independent once each frame's input points are known.
static int render_sequence(poppy_hip_ctx* c, const double* shape, const double* mask, int n, bool chain, poppy_write_cb write, void* user) {
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    if (c->pts1.empty()) return fail(c, POPPY_E_NOMATCH, "no point pairs (use poppy_hip_dissolve)");
    if (n <= 0) return POPPY_OK;
    const int W = c->W, H = c->H;
    std::vector<std::vector<P2f>> src1(n);
    src1[0] = c->pts1;
    if (chain)
        for (int j = 0; j + 1 < n; ++j) {                            // morph_points + clip_points of frame j
            const float s = (float)shape[j];
            std::vector<P2f> a = src1[j], b = c->pts2;
            clip_points_ref(a, W, H); clip_points_ref(b, W, H);
            std::vector<P2f>& m = src1[j + 1];
            m.resize(a.size());
            for (size_t i = 0; i < a.size(); ++i) {
                m[i].x = (float)((1.0 - s) * a[i].x + s * b[i].x);
                m[i].y = (float)((1.0 - s) * a[i].y + s * b[i].y);
            }
            clip_points_ref(m, W, H);
        }
    std::vector<FramePlan> plans(n);
    std::vector<int> rcs(n, 0);
    std::vector<std::atomic<int>> ready(n);
    for (auto& r : ready) r.store(0);
    std::atomic<int> next{0};
    const int nthreads = std::max(1, std::min({n, 16, (int)std::thread::hardware_concurrency() - 1}));
    auto worker = [&]() {
        for (;;) {
            const int j = next.fetch_add(1);
            if (j >= n) return;
            rcs[j] = plan_frame(W, H, chain ? src1[j] : src1[0], c->pts2, shape[j], plans[j]);
            ready[j].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
    int rc = POPPY_OK;
    const size_t row = (size_t)W * 3, frame_bytes = row * H;
    const int R = std::min(poppy_hip_ctx::kStageRing, (int)c->slots.size());
    int written = 0;
    if (write) {
        rc = stage_host(c, frame_bytes * R);
        if (rc == POPPY_OK && !c->dl_stream) {
            if (hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(c, POPPY_E_DEVICE, "hipStreamCreate failed");
            for (int k = 0; k < poppy_hip_ctx::kStageRing && rc == POPPY_OK; ++k)
                if (hipEventCreateWithFlags(&c->dl_done[k], hipEventDisableTiming) != hipSuccess) rc = fail(c, POPPY_E_DEVICE, "hipEventCreate failed");
        }
    }
    for (int j = 0; j < n && rc == POPPY_OK; ++j) {
        while (!ready[j].load(std::memory_order_acquire)) std::this_thread::yield();
        if (rcs[j]) { rc = fail(c, POPPY_E_RANGE, "point outside the image rectangle (Subdiv2D::insert would throw)"); break; }
        c->plan = std::move(plans[j]);
        if (chain) c->pts1 = src1[j];
        rc = submit_frame(c, mask[j], chain);
        if (rc == POPPY_OK && write) {
            // Frame hand-off: the download of frame j runs on its own stream into a ring of pinned buffers while the GPU
            // renders frame j+1; the writer gets frame j-(R-1).  R <= number of slots, so by the time a slot is rendered
            // into again its previous frame has been handed over (the host waited for that download).
            const int r = j % R;
            FrameSlot& f = c->slots[c->last_slot];
            hipError_t e = hipStreamWaitEvent(c->dl_stream, f.done, 0);
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_stage + (size_t)r * frame_bytes, f.out, frame_bytes, hipMemcpyDeviceToHost, c->dl_stream);
            if (e == hipSuccess) e = hipEventRecord(c->dl_done[r], c->dl_stream);
            if (e != hipSuccess) { c->err = std::string("frame download: ") + hipGetErrorString(e); rc = POPPY_E_DEVICE; break; }
            if (j >= R - 1) {
                const int jr = j - (R - 1), rr = jr % R;
                if (hipEventSynchronize(c->dl_done[rr]) != hipSuccess) { c->err = "frame download failed"; rc = POPPY_E_DEVICE; break; }
                write(user, c->h_stage + (size_t)rr * frame_bytes, W, H, row);
                ++written;
            }
        }
    }
    if (write && rc == POPPY_OK)
        for (int jr = written; jr < n; ++jr) {            // drain the ring
            const int rr = jr % R;
            if (hipEventSynchronize(c->dl_done[rr]) != hipSuccess) { c->err = "frame download failed"; rc = POPPY_E_DEVICE; break; }
            write(user, c->h_stage + (size_t)rr * frame_bytes, W, H, row);
        }
    next.store(n);                         // on an error: let the workers drain
    for (auto& t : pool) t.join();
    return rc;
}

static_assert(kPlanRasterRows == kRasterChunkRows, "the plan's work list and k_raster must agree on the chunk height");

// pyrdown .. unsharp of one slot.  Every argument is fixed for the life of the pair (the per-frame unsharp amount is
// read from the slot's plan blob), which is what lets the whole sequence be captured into one graph launch.
static void enqueue_body(poppy_hip_ctx* c, FrameSlot& f, hipStream_t s, Timer* tm, float amount, bool debug) {
    const int W = c->W, H = c->H, L = c->cfg.pyramid_levels;
    const int ft = c->first_tail < L ? c->first_tail : L;
    static const bool fuse = getenv("POPPY_HIP_NOFUSE") == nullptr;
    for (int i = 0; i < ft;) {
        const PyrLevel &a = c->levels[i], &b = c->levels[i + 1];
        if (fuse && i >= 1 && i + 2 <= ft && pyrdown2_eligible(a.w, a.h)) {           // two small levels in one launch
            const PyrLevel& d = c->levels[i + 2];
            launch_pyrdown2(f.pyrL + a.off3, f.pyrR + a.off3, f.pyrM + a.off1, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrM + b.off1,
                            f.pyrL + d.off3, f.pyrR + d.off3, f.pyrM + d.off1, a.w, a.h, s);
            i += 2;
            continue;
        }
        const void* sl = i == 0 ? (const void*)f.tr1 : (const void*)(f.pyrL + a.off3);
        const void* sr = i == 0 ? (const void*)f.tr2 : (const void*)(f.pyrR + a.off3);
        launch_pyrdown(sl, sr, f.pyrM + a.off1, i == 0, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrM + b.off1, a.w, a.h, s);
        ++i;
    }
    if (tm) tm->mark("pyrdown");
    launch_pyr_tail(f.pyrL, f.pyrR, f.pyrM, f.pyrB, c->d_levels, ft, L, c->tail_k1, c->tail_n3, c->tail_n1, s);
    if (tm) tm->mark("pyr_tail");
    for (int j = ft; j > 0;) {                     // blended level j is known; produce level j-2 or j-1
        if (fuse && j - 2 >= 1) {
            const PyrLevel &a = c->levels[j - 2], &m = c->levels[j - 1], &n = c->levels[j];
            if (collapse2_eligible(a.w, a.h, m.w, m.h, n.w, n.h)) {
                launch_collapse2(f.pyrL + a.off3, f.pyrR + a.off3, f.pyrM + a.off1, f.pyrL + m.off3, f.pyrR + m.off3, f.pyrM + m.off1,
                                 f.pyrL + n.off3, f.pyrR + n.off3, f.pyrB + n.off3, f.pyrB + a.off3, a.w, a.h, m.w, m.h, n.w, n.h, s);
                j -= 2;
                continue;
            }
        }
        const int i = j - 1;
        const PyrLevel &a = c->levels[i], &b = c->levels[i + 1];
        const void* gl = i == 0 ? (const void*)f.tr1 : (const void*)(f.pyrL + a.off3);
        const void* gr = i == 0 ? (const void*)f.tr2 : (const void*)(f.pyrR + a.off3);
        launch_collapse(gl, gr, i == 0, f.pyrM + a.off1, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrB + b.off3, f.pyrB + a.off3,
                        a.w, a.h, b.w, b.h, s);
        --j;
    }
    if (tm) tm->mark("collapse");
    launch_unsharp(f.pyrB, f.tmp, f.diff, f.out, debug ? f.unsharpF : nullptr, W, H, amount, (const float*)f.d_blob, (float)0.3, s);
    if (tm) tm->mark("unsharp");
}

static int capture_body(poppy_hip_ctx* c, FrameSlot& f) {
    hipGraph_t g = nullptr;
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    enqueue_body(c, f, c->stream, nullptr, 0.f, false);
    HIPCHK(c, hipStreamEndCapture(c->stream, &g));
    hipError_t e = hipGraphInstantiate(&f.body, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { f.body = nullptr; c->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}

static int submit_frame(poppy_hip_ctx* c, double mask, bool chain) {
    const int W = c->W, H = c->H;
    const int T = c->plan.n_tris;
    if (T > c->max_tris) return fail(c, POPPY_E_ARG, "triangle budget exceeded");

    // frame slot: the next one in the ring that does not hold the image this frame reads as corrected1
    int fi = c->next_slot;
    if (c->slots[fi].out == c->cur1) fi = (fi + 1) % (int)c->slots.size();
    c->next_slot = (fi + 1) % (int)c->slots.size();
    FrameSlot& f = c->slots[fi];

    HIPCHK(c, hipEventSynchronize(f.uploaded));                    // the pinned copy is free again
    const double amount = std::sin(mask * M_PI);
    *(float*)f.h_blob = (float)(1.0 - amount);                     // unsharp_mask(.., 1, 1.0 - amount, 0.3)
    const size_t rec_bytes = (size_t)(T + 1) * kWarpRecordFloats * sizeof(float);
    int* h_tri = (int*)(f.h_blob + kBlobHeader + rec_bytes);
    float* h_inv = (float*)(h_tri + (size_t)T * 6);
    RasterTri* h_edges = (RasterTri*)(h_inv + (size_t)T * 18);      // byte offset 144 + 176*T: 8-byte aligned
    int* h_work = (int*)(h_edges + T);
    const int n_work = (int)(c->plan.work.size() / 2);
    const size_t used = kBlobHeader + rec_bytes + (size_t)T * (6 + 18) * 4 + (size_t)T * sizeof(RasterTri) + (size_t)n_work * 8;
    if (used > c->blob_bytes) return fail(c, POPPY_E_ARG, "plan blob overflow");
    if (T) {
        memcpy(h_tri, c->plan.tri_xy.data(), (size_t)T * 6 * sizeof(int));
        memcpy(h_inv, c->plan.inv1.data(), (size_t)T * 9 * sizeof(float));
        memcpy(h_inv + (size_t)T * 9, c->plan.inv2.data(), (size_t)T * 9 * sizeof(float));
        memcpy(h_edges, c->plan.raster.data(), (size_t)T * sizeof(RasterTri));
        memcpy(h_work, c->plan.work.data(), (size_t)n_work * 8);
    }
    // the fast warp kernel takes the frame when every matrix passes the host's range check (always, short of degenerate input)
    static const bool exact_warp_only = getenv("POPPY_HIP_GENERALWARP") != nullptr;
    const bool fast_warp = pack_warp_records(c->plan.inv1.data(), c->plan.inv2.data(), T, W, H, (float*)(f.h_blob + kBlobHeader)) &&
                           warp_fast_geometry(W, H) && !exact_warp_only;
    c->last_warp_fast = fast_warp;
    const float* d_rec = (const float*)(f.d_blob + kBlobHeader);
    const int* d_tri = (const int*)(f.d_blob + kBlobHeader + rec_bytes);
    const float* d_inv = (const float*)(d_tri + (size_t)T * 6);
    const RasterTri* d_edges = (const RasterTri*)(d_inv + (size_t)T * 18);
    const int* d_work = (const int*)(d_edges + T);

    // POPPY_HIP_HEADMODE=1 moves a chained frame's clear/raster/mask to the slot's stream so that it overlaps the previous
    // frame.  Measured on MI355X (profiles/r01_e_streams.md) the cross-queue hand-over costs 12-20 us, more than the
    // ~45 us head saves once queue sharing is counted, so chained frames stay on one stream by default.
    static const int head_mode = getenv("POPPY_HIP_HEADMODE") ? atoi(getenv("POPPY_HIP_HEADMODE")) : 0;
    static const bool no_graph = getenv("POPPY_HIP_NOGRAPH") != nullptr;
    const bool chained = chain || c->cur1_ready;
    if (!f.stream && (!chained || head_mode != 0)) HIPCHK(c, hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
    hipStream_t s = chained ? c->stream : f.stream;                           // warp .. unsharp
    hipStream_t head = (chained && head_mode == 0) ? s : f.stream;            // clear, raster, mask
    if (c->debug && !f.unsharpF) HIPCHK(c, hipMalloc((void**)&f.unsharpF, (size_t)W * H * 12));
    const bool all_marks = c->timing == 1;
    // The captured body is for frames in flight beside each other (phase mode), where the submitting host thread is the
    // bottleneck.  On the chained critical path a graph launch leaves the GPU idle ~8 us longer than the same kernels
    // launched one by one (4600 vs 4785 frames/s, profiles/r01_e_streams.md), and the host keeps up easily.
    const bool use_graph = !no_graph && !chained && !c->debug && !all_marks && W > 1 && H > 1;
    if (use_graph && !f.body) { int rc = capture_body(c, f); if (rc) return rc; }

    Timer th(c, head), tm(c, s);
    // the plan goes up on its own stream, after the frame that last used this slot has let go of the device copy
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, f.done, 0));
    launch_upload(f.h_blob_dev, f.d_blob, used, c->copy_stream);
    HIPCHK(c, hipEventRecord(f.uploaded, c->copy_stream));
    if (head != c->stream) HIPCHK(c, hipStreamWaitEvent(head, c->inputs_ready, 0));
    if (head != c->stream) HIPCHK(c, hipStreamWaitEvent(head, f.done, 0));    // the frame that last used this slot's buffers
    if (all_marks) th.mark(nullptr);
    // -- independent of the previous frame ---------------------------------------------------------------------
    // the id map is left cleared by the warp kernel of the frame before (its only reader); only a slot's first frame
    // and frames after a debug frame (which keeps the map for poppy_hip_debug_fetch) need the memset
    if (!f.map_clean) HIPCHK(c, hipMemsetAsync(f.triMap, 0, (size_t)W * H * 4, head));
    // Chained frames: the host waits for the plan upload itself (it is ~100 us ahead of the GPU, the copy takes ~10) instead of
    // putting a cross-stream wait in front of the raster, which costs the critical path ~4 us per frame.
    static const bool host_wait = getenv("POPPY_HIP_STREAMWAIT") == nullptr;
    if (chained && host_wait) HIPCHK(c, hipEventSynchronize(f.uploaded));
    else HIPCHK(c, hipStreamWaitEvent(head, f.uploaded, 0));
    if (all_marks) th.mark("upload+clear");
    launch_raster(d_tri, d_edges, d_work, n_work, f.triMap, W, H, head);
    if (all_marks) th.mark("raster");
    if (s != head) {
        HIPCHK(c, hipEventRecord(f.prepared, head));
        HIPCHK(c, hipStreamWaitEvent(s, f.prepared, 0));            // (implies inputs_ready)
    }
    // -- chained mode: corrected1 is the previous frame (src/poppy.hpp:217) -------------------------------------
    if (c->cur1_ready && c->cur1_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->cur1_ready, 0));
    WarpExtras ex;
    ex.clear_ids = c->debug ? 0 : 1;
    ex.m2 = c->m2; ex.mask = f.pyrM; ex.alpha = 1.0 - mask; ex.beta = -mask;       // lbmask rides along (level 0 of pyrM)
    f.map_clean = ex.clear_ids != 0;
    if (c->timing == 2) {       // the dispatch's own begin / end timestamps: no marker packets in the stream
        hipEvent_t t0 = tm.take(nullptr), t1 = tm.take("warp");
        if (fast_warp) launch_warp_fast(f.triMap, d_rec, T + 1, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s, t0, t1);
        else launch_warp(f.triMap, d_inv, d_inv + (size_t)T * 9, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s, t0, t1);
    } else {
        tm.mark(nullptr);
        if (fast_warp) launch_warp_fast(f.triMap, d_rec, T + 1, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s);
        else launch_warp(f.triMap, d_inv, d_inv + (size_t)T * 9, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s);
        tm.mark("warp");
    }
    if (use_graph) HIPCHK(c, hipGraphLaunch(f.body, s));
    else enqueue_body(c, f, s, all_marks ? &tm : nullptr, (float)(1.0 - amount), c->debug);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(f.done, s));
    if (s != c->stream) HIPCHK(c, hipStreamWaitEvent(c->stream, f.done, 0));   // anything queued on `stream` later sees this frame
    c->last_slot = fi;
    if (chain) {                                   // src/poppy.hpp:217-218
        c->cur1 = f.out;
        c->cur1_ready = f.done;
        c->cur1_stream = s;
        c->pts1 = c->plan.morphed;
    }
    return POPPY_OK;
}

static int upload_image(poppy_hip_ctx* c, uint8_t* dst, const uint8_t* src, size_t stride, int W, int H) {
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)W * 3, src, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
    return POPPY_OK;
}

extern "C" {

int poppy_hip_pair_load(poppy_hip_ctx* c, const uint8_t* c1, size_t s1, const uint8_t* c2, size_t s2, const float* gabor2,
                        int W, int H, const float* p1, const float* p2, int n) {
    if (!c) return POPPY_E_ARG;
    if (!c1 || !c2 || !gabor2 || W <= 0 || H <= 0 || s1 < (size_t)W * 3 || s2 < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    rc = set_points(c, p1, p2, n); if (rc) return rc;
    rc = upload_image(c, c->c1, c1, s1, W, H); if (rc) return rc;
    rc = upload_image(c, c->c2, c2, s2, W, H); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->gabor2, gabor2, (size_t)W * H * 12, hipMemcpyHostToDevice, c->stream));
    rc = finish_pair_load(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));      // host buffers may be reused by the caller
    return POPPY_OK;
}

int poppy_hip_pair_load_device(poppy_hip_ctx* c, const void* d1, const void* d2, const void* dg, int W, int H,
                               const float* p1, const float* p2, int n) {
    if (!c) return POPPY_E_ARG;
    if (!d1 || !d2 || !dg || W <= 0 || H <= 0) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    rc = set_points(c, p1, p2, n); if (rc) return rc;
    const size_t P = (size_t)W * H;
    HIPCHK(c, hipMemcpyAsync(c->c1, d1, P * 3, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->c2, d2, P * 3, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gabor2, dg, P * 12, hipMemcpyDeviceToDevice, c->stream));
    rc = finish_pair_load(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

int poppy_hip_pair_reset(poppy_hip_ctx* c) {
    if (!c) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    c->cur1 = c->c1; c->cur1_ready = nullptr; c->pts1 = c->pts1_0; c->last_slot = -1;
    return POPPY_OK;
}

int poppy_hip_render(poppy_hip_ctx* c, double shape, double mask, int chain, uint8_t* dst, size_t dst_stride) {
    if (!c) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = render_frame(c, shape, mask, chain != 0); if (rc) return rc;
    if (dst) {
        if (dst_stride < (size_t)c->W * 3) return fail(c, POPPY_E_ARG, "dst_stride too small");
        HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, c->slots[c->last_slot].out, (size_t)c->W * 3, (size_t)c->W * 3, c->H, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return POPPY_OK;
}

const void* poppy_hip_frame_device(poppy_hip_ctx* c) { return (c && c->last_slot >= 0) ? c->slots[c->last_slot].out : nullptr; }

int poppy_hip_morph_images(poppy_hip_ctx* c, const uint8_t* c1, size_t s1, const uint8_t* c2, size_t s2, const float* gabor2, int W, int H,
                           const float* p1, const float* p2, int n, double shape, double mask, uint8_t* dst, size_t dst_stride, float* morphed) {
    if (!c) return POPPY_E_ARG;
    if (!dst) return fail(c, POPPY_E_ARG, "dst is null");
    int rc = poppy_hip_pair_load(c, c1, s1, c2, s2, gabor2, W, H, p1, p2, n); if (rc) return rc;
    rc = poppy_hip_render(c, shape, mask, 0, dst, dst_stride); if (rc) return rc;
    if (morphed && n) memcpy(morphed, c->plan.morphed.data(), (size_t)n * 8);
    return POPPY_OK;
}

double poppy_frame_ratio(int j, int number_of_frames, double phase) {
    const double N = (double)number_of_frames;
    const double linear = j / N;
    double progress = 0;
    if (phase >= 1.0) progress = 1;
    if (phase < 1.0 && phase >= 0) progress = 1.0 / N;
    else if (linear == 0) progress = 0;
    else if (linear == 1) progress = 1;
    else progress = (1.0 / (1.0 - linear)) / N;
    double shape = (phase < 1.0 && phase >= 0) ? progress * phase : progress;
    if (shape > 1) shape = 1;
    return shape;
}

int poppy_hip_morph_frames(poppy_hip_ctx* c, double phase, poppy_write_cb write, void* user) {
    if (!c) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const int N = c->cfg.number_of_frames;
    if (write) { int rc = stage_host(c, (size_t)c->W * 3 * c->H); if (rc) return rc; }
    const int n = phase >= 0 ? 1 : N;                          // phase mode: exactly one frame (src/poppy.hpp:234-235)
    std::vector<double> ratio(n);
    for (int j = 0; j < n; ++j) ratio[j] = poppy_frame_ratio(j, N, phase);
    return render_sequence(c, ratio.data(), ratio.data(), n, true, write, user);
}

int poppy_hip_dissolve(poppy_hip_ctx* c, const uint8_t* img1, size_t s1, const uint8_t* img2, size_t s2, int W, int H, double phase,
                       uint8_t* dst, size_t dst_stride) {
    if (!c) return POPPY_E_ARG;
    if (!img1 || !img2 || !dst || W <= 0 || H <= 0) return fail(c, POPPY_E_ARG, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    rc = upload_image(c, c->c1, img1, s1, W, H); if (rc) return rc;
    rc = upload_image(c, c->c2, img2, s2, W, H); if (rc) return rc;
    // Mat blend = img2*phase + img1*(1.0-phase)  ->  addWeighted(img2, phase, img1, 1-phase, 0)
    launch_dissolve(c->c2, c->c1, c->slots[0].out, (size_t)W * H * 3, (float)phase, (float)(1.0 - phase), c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, c->slots[0].out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pair_ready = false;
    return POPPY_OK;
}

int poppy_hip_debug_fetch(poppy_hip_ctx* c, const char* name, void* host, size_t bytes) {
    if (!c || !name || !host) return POPPY_E_ARG;
    if (!c->W) return fail(c, POPPY_E_STATE, "no pair loaded");
    const size_t P = (size_t)c->W * c->H;
    const void* src = nullptr; size_t need = 0;
    std::string n(name);
    if (n != "m2" && c->last_slot < 0) return fail(c, POPPY_E_STATE, "no frame rendered yet");
    const FrameSlot& f = c->slots[c->last_slot < 0 ? 0 : c->last_slot];          // intermediates of the last frame
    if (n == "triMap") { src = f.triMap; need = P * 4; }
    else if (n == "trImg1") { src = f.tr1; need = P * 3; }
    else if (n == "trImg2") { src = f.tr2; need = P * 3; }
    else if (n == "lbmask") { src = f.pyrM; need = P * 4; }
    else if (n == "lapBlend") { src = f.pyrB; need = P * 12; }
    else if (n == "unsharp") {
        if (!c->debug || !f.unsharpF) return fail(c, POPPY_E_STATE, "enable debug before rendering the frame");
        src = f.unsharpF; need = P * 12;
    }
    else if (n == "m2") { src = c->m2; need = P * 4; }
    else return fail(c, POPPY_E_ARG, "unknown debug buffer");
    if (bytes != need) return fail(c, POPPY_E_ARG, "debug buffer size mismatch");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(host, src, need, hipMemcpyDeviceToHost));
    return POPPY_OK;
}

int poppy_hip_debug_triangles(poppy_hip_ctx* c, int* n_tris, int* idx3, float* M1, float* M2, int max_tris) {
    if (!c || !n_tris) return POPPY_E_ARG;
    const int T = c->plan.n_tris;
    *n_tris = T;
    if (T > max_tris) return fail(c, POPPY_E_ARG, "max_tris too small");
    if (idx3 && T) memcpy(idx3, c->plan.idx3.data(), (size_t)T * 3 * sizeof(int));
    if (M1 && T) memcpy(M1, c->plan.M1.data(), (size_t)T * 9 * sizeof(float));
    if (M2 && T) memcpy(M2, c->plan.M2.data(), (size_t)T * 9 * sizeof(float));
    return POPPY_OK;
}

int poppy_plan_frame(int W, int H, const float* p1, const float* p2, int n, double shape, int max_tris,
                     int* n_tris, int* idx3, int* tri_xy, float* M1, float* M2, float* inv1, float* inv2, float* morphed) {
    if (W <= 0 || H <= 0 || n < 0 || !n_tris) return POPPY_E_ARG;
    std::vector<P2f> a(n), b(n);
    if (n) { memcpy(a.data(), p1, (size_t)n * 8); memcpy(b.data(), p2, (size_t)n * 8); }
    FramePlan plan;
    if (plan_frame(W, H, a, b, shape, plan)) return POPPY_E_RANGE;
    const int T = plan.n_tris;
    *n_tris = T;
    if (T > max_tris) return POPPY_E_ARG;
    if (idx3 && T) memcpy(idx3, plan.idx3.data(), (size_t)T * 12);
    if (tri_xy && T) memcpy(tri_xy, plan.tri_xy.data(), (size_t)T * 24);
    if (M1 && T) memcpy(M1, plan.M1.data(), (size_t)T * 36);
    if (M2 && T) memcpy(M2, plan.M2.data(), (size_t)T * 36);
    if (inv1 && T) memcpy(inv1, plan.inv1.data(), (size_t)T * 36);
    if (inv2 && T) memcpy(inv2, plan.inv2.data(), (size_t)T * 36);
    if (morphed && n) memcpy(morphed, plan.morphed.data(), (size_t)n * 8);
    return POPPY_OK;
}

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
    HIPCHK(c, hipMemcpy2DAsync(c->d_align, (size_t)W * 3, img, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
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
    if (ok) ok = hipMemcpy2DAsync(dst, ds, d_out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
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
    HIPCHK(c, hipMemcpy2DAsync(img, stride, c->d_align, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream));
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

int poppy_match_points(const float* p1, const float* p2, int n, int W, int H, double tol, float* o1, float* o2, int* n_out, double* imd) {
    if (n < 0 || W <= 0 || H <= 0 || !n_out || (n && (!p1 || !p2))) return POPPY_E_ARG;
    std::vector<P2f> a(n), b(n);
    if (n) { memcpy(a.data(), p1, (size_t)n * 8); memcpy(b.data(), p2, (size_t)n * 8); }
    drop_out_of_image(a, b, W, H);
    if (a.empty()) { *n_out = 0; if (imd) *imd = 0; return POPPY_OK; }     // caller falls back to the dissolve (poppy.hpp:125)
    const double d = morph_distance_ref(a, b, W, H);
    if (imd) *imd = d;
    match_and_prepare(a, b, W, H, tol, d);
    *n_out = (int)a.size();
    if (o1) memcpy(o1, a.data(), a.size() * 8);
    if (o2) memcpy(o2, b.data(), b.size() * 8);
    return POPPY_OK;
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

// Pair set-up from the raw images: the pre-ORB filter chain on the GPU, then the same steps as pair_begin_prefiltered.
static int pair_begin_impl(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H, float ratio) {
    if (!c) return POPPY_E_ARG;
    if (!bgr1 || !bgr2 || W <= 0 || H <= 0 || s1 < (size_t)W * 3 || s2 < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    c->pair_ready = false;
    rc = upload_image(c, c->c1, bgr1, s1, W, H); if (rc) return rc;
    rc = upload_image(c, c->c2, bgr2, s2, W, H); if (rc) return rc;
    const size_t P = (size_t)W * H;
    std::vector<uint8_t> g[2] = {std::vector<uint8_t>(P), std::vector<uint8_t>(P)};
    double d[2] = {0, 0};
    if (!c->aux_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamSynchronize(c->stream));                         // the uploads above
    // The two images go through the chain independently (Extractor::foreground -> dft_detail2 -> the ORB input of
    // Extractor::keypoints; image 2 also through gabor_filter(corrected2 / 255), src/poppy.hpp:119-122): one host thread and
    // one stream each, so that the medians of one image run beside the Gabor bank of the other.
    std::string errs[2];
    int rcs[2] = {POPPY_OK, POPPY_OK};
    // with auto-align, gabor2 belongs to the ALIGNED second image (src/poppy.hpp:116-122 runs after Matcher::find): computed further down
    const bool align_first = c->cfg.enable_auto_align != 0 && ratio < 0.f;
    auto chain_of = [&](int i) {
        if (hipSetDevice(c->device) != hipSuccess) { errs[i] = "hipSetDevice failed"; rcs[i] = POPPY_E_DEVICE; return; }
        ForegroundFilter& fg = i ? c->foreground_b : c->foreground;
        hipStream_t st = i ? c->aux_stream : c->stream;
        const uint8_t* gf = fg.run_device(i ? c->c2 : c->c1, (size_t)W * 3, W, H, st, nullptr);
        if (!gf) { errs[i] = "foreground: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        if (fg.detail(gf, W, H, st, &d[i])) { errs[i] = "dft_detail2: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        const uint8_t* gi = fg.orb_input(gf, W, H, 0, st);
        if (!gi) { errs[i] = "orb_input: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
        hipError_t e = hipMemcpyAsync(g[i].data(), gi, P, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && i == 1 && !align_first) {
            const float* gab = fg.gabor_field(c->c2, W, H, st);
            if (!gab) { errs[i] = "gabor_field: " + fg.err; rcs[i] = POPPY_E_DEVICE; return; }
            e = hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { errs[i] = std::string("pair_begin: ") + hipGetErrorString(e); rcs[i] = POPPY_E_DEVICE; }
    };
    {
        std::thread other(chain_of, 1);
        chain_of(0);
        other.join();
    }
    for (int i = 0; i < 2; ++i) if (rcs[i]) { c->err = errs[i]; return rcs[i]; }
    const double detail = 255.0 / std::max(d[0], d[1]);                 // src/extractor.cpp:40-45
    c->last_detail[0] = d[0]; c->last_detail[1] = d[1];
    const int nfeatures = (int)(c->cfg.max_keypoints * detail);
    c->last_nfeatures = nfeatures;
    std::vector<OrbKeyPoint> k1, k2;
    {   // the two detections are independent too
        int r1 = 0, r2 = 0;
        std::thread other([&]() { r2 = hipSetDevice(c->device) == hipSuccess ? c->orb_b.detect(g[1].data(), W, W, H, nfeatures, c->aux_stream, k2) : -2; });
        r1 = c->orb.detect(g[0].data(), W, W, H, nfeatures, c->stream, k1);
        other.join();
        if (r1 < 0 || r2 < 0) { c->err = "orb_detect: " + (r1 < 0 ? c->orb.err : c->orb_b.err); return POPPY_E_DEVICE; }
    }
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
            if (c->aligner.run(c->c2, W, H, a, b, c->stream, nullptr)) return fail(c, POPPY_E_DEVICE, c->aligner.err.c_str());
            memcpy(p2.data(), b.data(), n * 8);
            const float* gab = c->foreground_b.gabor_field(c->c2, W, H, c->stream);
            if (!gab) { c->err = "gabor_field: " + c->foreground_b.err; return POPPY_E_DEVICE; }
            HIPCHK(c, hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, c->stream));
        }
        int m = 0;
        rc = poppy_match_points(p1.data(), p2.data(), (int)n, W, H, c->cfg.match_tolerance, o1.data(), o2.data(), &m, &c->initial_morph_dist);
        if (rc) return fail(c, rc, "poppy_match_points failed");
        rc = set_points(c, o1.data(), o2.data(), m); if (rc) return rc;
    }
    rc = finish_pair_load(c); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

int poppy_hip_pair_begin(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H) {
    return pair_begin_impl(c, bgr1, s1, bgr2, s2, W, H, -1.f);
}
int poppy_hip_pair_begin_descriptors(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H, float ratio) {
    if (!(ratio >= 0.f)) return c ? fail(c, POPPY_E_ARG, "ratio must be >= 0") : POPPY_E_ARG;
    return pair_begin_impl(c, bgr1, s1, bgr2, s2, W, H, ratio);
}

int poppy_hip_pair_corrected2(poppy_hip_ctx* c, uint8_t* dst, size_t ds) {
    if (!c || !dst) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no resident pair");
    if (ds < (size_t)c->W * 3) return fail(c, POPPY_E_ARG, "stride too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(dst, ds, c->c2, (size_t)c->W * 3, (size_t)c->W * 3, c->H, hipMemcpyDeviceToHost, c->stream));
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

// gabor_filter(bgr / 255) with the default arguments (host in / out, f32x3)
int poppy_hip_gabor_field(poppy_hip_ctx* c, const uint8_t* bgr, size_t stride, int W, int H, float* out) {
    if (!c || !bgr || !out || W <= 0 || H <= 0 || stride < (size_t)W * 3) return POPPY_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    ForegroundFilter& fg = c->foreground;
    if (fg.prepare(W, H)) { c->err = "foreground: " + fg.err; return POPPY_E_DEVICE; }
    HIPCHK(c, hipMemcpy2DAsync(fg.bgr_staging(), (size_t)W * 3, bgr, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
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
    if (e == hipSuccess) e = hipMemcpy2DAsync(canvas + ((size_t)ry * UW + rx) * 3, (size_t)UW * 3, src, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, canvas, UB, hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) { cleanup(); c->err = std::string("blur_margin: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    dx = (dx == 0 ? 1.3 : dx + margin);
    dy = (dy == 0 ? 1.3 : dy + margin);
    const int rects[4][4] = {{0, 0, (int)dx, UH}, {(int)(UW - dx), 0, (int)dx, UH}, {0, 0, UW, (int)dy}, {0, (int)(UH - dy), UW, (int)dy}};
    for (const auto& r : rects) launch_strip_blur(canvas, out, UW, tmp, d_taps, n, r[0], r[1], r[2], r[3], c->stream);   // left, right, top, bottom: later strips win
    e = hipMemcpy2DAsync(dst, dst_stride, out, (size_t)UW * 3, (size_t)UW * 3, UH, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (e != hipSuccess) { c->err = std::string("blur_margin: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}

int poppy_radial_gradient(int W, int H, float* out) {
    if (W <= 0 || H <= 0 || !out) return POPPY_E_ARG;
    std::vector<float> r;
    radial_gradient(W, H, r);
    memcpy(out, r.data(), r.size() * 4);
    return POPPY_OK;
}

int poppy_hip_pair_points(poppy_hip_ctx* c, float* p1, float* p2, int max_points, int* n_points) {
    if (!c || !n_points) return POPPY_E_ARG;
    const int n = (int)c->pts1_0.size();
    *n_points = n;
    if (n > max_points) return fail(c, POPPY_E_ARG, "max_points too small");
    if (p1 && n) memcpy(p1, c->pts1_0.data(), (size_t)n * 8);
    if (p2 && n) memcpy(p2, c->pts2.data(), (size_t)n * 8);
    return POPPY_OK;
}

int poppy_hip_timing_summary(poppy_hip_ctx* c, const char** names, float* total_ms, int* launches, int max) {
    if (!c) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    int n = 0;
    for (size_t i = 1; i < c->marks_used; ++i) {
        const char* nm = c->marks[i].name;
        if (!nm) continue;                                  // frame boundary
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->marks[i - 1].ev, c->marks[i].ev) != hipSuccess) continue;
        int k = 0;
        while (k < n && strcmp(names[k], nm) != 0) ++k;
        if (k == n) { if (n >= max) continue; names[n] = nm; total_ms[n] = 0.f; launches[n] = 0; ++n; }
        total_ms[k] += ms; launches[k] += 1;
    }
    c->marks_used = 0;
    return n;
}

int poppy_hip_render_many(poppy_hip_ctx* c, const double* shape, const double* mask, int n, int chain, poppy_write_cb write, void* user) {
    if (!c || !shape || !mask || n < 0) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    HIPCHK(c, hipSetDevice(c->device));
    if (write) { int rc = stage_host(c, (size_t)c->W * 3 * c->H); if (rc) return rc; }
    return render_sequence(c, shape, mask, n, chain != 0, write, user);
}

}  // extern "C"
