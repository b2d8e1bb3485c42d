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
#include "context.h"

static std::string g_create_error;
static void free_slot_preps(poppy_hip_ctx* c);
static void stop_seq_plans(poppy_hip_ctx* c);
void start_default_seq_plans(poppy_hip_ctx* c);
static void drop_slot_preps(poppy_hip_ctx* c);

// Streams and hardware queues.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable
// says otherwise); streams that share a queue run in order, and queues are handed out — and spread over the command processor's
// pipes — in the order the streams are created.  Which queue the frame-download stream gets decides what the writer hand-off costs:
// created lazily as a context's fifth stream it got, with 5 or more queues allowed, a queue of its own that delayed EVERY dispatch of
// the rendering stream by ~40 us while a copy was pending (chained 1080p loop with writer: 3.9k frames/s against 5.9k; kernels of 8 us
// show up as 45 us in the trace), and with fewer than 4 it shared the rendering stream's queue (4.7k).  So a context creates its three
// hot streams first and in this order — rendering, plan upload, frame download — and the library leaves the queue count alone:
// 5.7-6.0k frames/s with 4, 5, 8 or 16 queues, with the image's runtime (downloads on the SDMA engines) and with the one bundled in
// the torch wheel (blit kernels).  Measurements: profiles/r02_notes.md section 7, tools/experiments/hwq_matrix.sh, hwq_sweep.sh, hwq_sweep4.sh, writer_gap.py.


extern "C" {

void poppy_settings_default(poppy_settings* s) {
    s->number_of_frames = 60; s->match_tolerance = 1.0; s->max_keypoints = 300; s->pyramid_levels = 64; s->enable_radial_mask = 0; s->enable_auto_align = 0;
}
static std::atomic<int> g_live_contexts{0};        // contexts alive in this process: they share the host's threads for their frame planners
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
    c->foreground.radial_mask_on = c->foreground_b.radial_mask_on = c->cfg.enable_radial_mask != 0;      // src/extractor.cpp:178-197
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "hipStreamCreate failed"; delete c; return nullptr; }
    for (hipEvent_t& e : c->dl_done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { g_create_error = "hipEventCreate failed"; delete c; return nullptr; }
    (void)hipEventCreateWithFlags(&c->inputs_ready, hipEventDisableTiming);
    int k = 4;                                             // frames in flight
    if (const char* e = getenv("POPPY_HIP_SLOTS")) k = atoi(e);
    c->slots.resize(std::max(2, std::min(k, 8)));
    for (FrameSlot& f : c->slots) {
        // (hipEventDisableSystemFence on `done` was measured: +0.6-1 % frames/s; not used, because frame downloads to the
        // host are ordered by this event)
        if (hipEventCreateWithFlags(&f.downloaded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.prepared, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.uploaded, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "hipStreamCreate failed"; delete c; return nullptr;
        }
    }
    g_live_contexts.fetch_add(1);
    return c;
}

static void free_pair(poppy_hip_ctx* c) {
    c->last_warp.valid = false;                     // its pointers go with the pair's buffers
    void* bufs[] = {c->arena, c->c2_raw, c->gabor2, c->d_levels};
    for (void* b : bufs) if (b) (void)hipFree(b);
    c->arena = nullptr; c->arena_bytes = 0;
    c->c1 = c->c2 = c->c2_raw = nullptr; c->gabor2 = c->m2 = nullptr; c->d_levels = nullptr;
    for (FrameSlot& f : c->slots) {
        void* fb[] = {f.tr1, f.tr2, f.out, f.pyrL, f.pyrR, f.pyrM, f.pyrB, f.tmp, f.diff, f.unsharpF, f.triMap};
        for (void* b : fb) if (b) (void)hipFree(b);
        f.tr1 = f.tr2 = f.out = nullptr; f.pyrL = f.pyrR = f.pyrM = f.pyrB = f.tmp = f.diff = f.unsharpF = nullptr; f.triMap = nullptr;
    }
    for (FrameSlot& f : c->slots) {
        if (f.body) { (void)hipGraphExecDestroy(f.body); f.body = nullptr; }
        if (f.h_blob) (void)hipHostFree(f.h_blob);
        if (f.d_blob) (void)hipFree(f.d_blob);
        if (f.tile_data) (void)hipFree(f.tile_data);
        f.h_blob = nullptr; f.d_blob = nullptr; f.tile_data = nullptr;
    }
    c->max_tris = 0; c->W = c->H = 0; c->pair_ready = false;
}

void poppy_hip_destroy(poppy_hip_ctx* c) {
    if (!c) return;
    g_live_contexts.fetch_sub(1);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)poppy_hip_comm_free(c);
    for (FrameSlot& f : c->slots) { if (f.stream) (void)hipStreamSynchronize(f.stream); if (f.own_stream) (void)hipStreamSynchronize(f.own_stream); }
    free_pair(c);
    for (FrameSlot& f : c->slots) {
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.prepared) (void)hipEventDestroy(f.prepared);
        if (f.downloaded) (void)hipEventDestroy(f.downloaded);
        if (f.uploaded) (void)hipEventDestroy(f.uploaded);
        if (f.own_stream) (void)hipStreamDestroy(f.own_stream);
    }
    if (c->inputs_ready) (void)hipEventDestroy(c->inputs_ready);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->dl_stream) { (void)hipStreamSynchronize(c->dl_stream); (void)hipStreamDestroy(c->dl_stream); c->dl_stream = nullptr; }
    for (hipStream_t& st : c->dl_ring) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); st = nullptr; }
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    stop_seq_plans(c);
    free_slot_preps(c);
    if (c->setup_ev) (void)hipEventDestroy(c->setup_ev);
    if (c->c2_up_ev) (void)hipEventDestroy(c->c2_up_ev);
    for (hipEvent_t e : c->dl_done) if (e) (void)hipEventDestroy(e);
    for (auto& m : c->marks) (void)hipEventDestroy(m.ev);
    (void)hipStreamSynchronize(c->copy_stream);
    if (c->d_align) (void)hipFree(c->d_align);
    if (c->d_comm_scratch) (void)hipFree(c->d_comm_scratch);
    c->aligner.release();
    (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int poppy_warp_records(const float* inv1, const float* inv2, int n_tris, int width, int height, float* records) {
    if (n_tris < 0 || width < 1 || height < 1 || !records || (n_tris > 0 && (!inv1 || !inv2))) return POPPY_E_ARG;
    return pack_warp_records(inv1, inv2, n_tris, width, height, records) ? 1 : 0;
}
int poppy_hip_last_warp_kind(poppy_hip_ctx* c) { return c ? (c->last_warp_bin ? 2 : c->last_warp_fast ? 1 : 0) : POPPY_E_ARG; }
int poppy_hip_warp_counts(poppy_hip_ctx* c, unsigned long long* fused, unsigned long long* tiled, unsigned long long* general) {
    if (!c) return POPPY_E_ARG;
    if (fused) *fused = c->n_warp_bin;
    if (tiled) *tiled = c->n_warp_fast;
    if (general) *general = c->n_warp_general;
    return POPPY_OK;
}
// Relaunches the last frame's fused raster + warp kernel `reps` times back to back on the context's stream, nothing else running, and
// returns the average time per launch between two events around the batch (milliseconds) — the kernel's duration as the kernel traces
// show it, without the per-dispatch stamps' overhead.  The relaunches write the same warped images again.
int poppy_hip_time_last_warp(poppy_hip_ctx* c, int reps, float* ms_per_launch) {
    if (!c || reps < 1 || !ms_per_launch) return POPPY_E_ARG;
    if (!c->last_warp.valid) return fail(c, POPPY_E_STATE, "no fused raster + warp launch to repeat");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = drain_frames(c); if (rc) return rc; }
    const auto& w = c->last_warp;
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    launch_warp_bin(w.rec, w.tile_data, w.tile_bytes, w.toff, w.tile_w, w.c1, w.c2, w.tr1, w.tr2, c->W, c->H, w.ex, c->stream);      // warm
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) launch_warp_bin(w.rec, w.tile_data, w.tile_bytes, w.toff, w.tile_w, w.c1, w.c2, w.tr1, w.tr2, c->W, c->H, w.ex, c->stream);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = ms / (float)reps;
    return POPPY_OK;
}

int poppy_hip_mask_rider(poppy_hip_ctx* c) { return c ? (c->lazy_mask ? 0 : 1) : POPPY_E_ARG; }
int poppy_hip_set_debug(poppy_hip_ctx* c, int on) { if (!c) return POPPY_E_ARG; c->debug = on != 0; return POPPY_OK; }
int poppy_hip_set_timing(poppy_hip_ctx* c, int on) { if (!c) return POPPY_E_ARG; c->timing = on < 0 ? 0 : on; c->marks_used = 0; return POPPY_OK; }
void* poppy_hip_stream(poppy_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }
int poppy_hip_sync(poppy_hip_ctx* c) { if (!c) return POPPY_E_ARG; HIPCHK(c, hipSetDevice(c->device)); return drain_frames(c); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
static int ensure_ring(poppy_hip_ctx* c, int n_points) {
    int need = 2 * n_points + 16;            // a planar triangulation of n points has < 2n triangles
    if (need <= c->max_tris) return POPPY_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    c->last_warp.valid = false;                     // the plan blobs and tile entries it points into are reallocated below
    // worst case every triangle spans the whole image height
    const size_t items = (size_t)need * ((size_t)c->H / kRasterChunkRows + 3);
    const int tw = warp_bin_tile_width(c->W, c->H), th = 1024 / tw;
    const size_t ntiles = (size_t)((c->W + tw - 1) / tw) * ((c->H + th - 1) / th);
    c->bins_cap = 64 * (size_t)need + 16 * ntiles;
    c->tile_bytes = warp_bin_data_bytes(ntiles, c->bins_cap);
    const size_t bytes = ((kBlobHeader + (size_t)(need + 1) * kWarpRecordFloats * 4 +
                          (size_t)need * (6 * 4 + 18 * 4 + sizeof(RasterTri)) + items * 8 +
                          (size_t)need * 3 * sizeof(OutlineSeg) + (ntiles + 1) * 4 + c->bins_cap * 2 + 64 + 6 * 16 + 15) / 16) * 16;
    for (FrameSlot& f : c->slots) {
        if (f.body) { (void)hipGraphExecDestroy(f.body); f.body = nullptr; }      // it holds a pointer into the blob
        if (f.h_blob) (void)hipHostFree(f.h_blob);
        if (f.d_blob) (void)hipFree(f.d_blob);
        if (f.tile_data) (void)hipFree(f.tile_data);
        f.h_blob = f.d_blob = f.tile_data = nullptr;
        HIPCHK(c, hipMalloc((void**)&f.tile_data, c->tile_bytes));
        HIPCHK(c, hipHostMalloc((void**)&f.h_blob, bytes, hipHostMallocMapped));
        HIPCHK(c, hipHostGetDevicePointer(&f.h_blob_dev, f.h_blob, 0));
        HIPCHK(c, hipMalloc((void**)&f.d_blob, bytes));
    }
    c->blob_bytes = bytes;
    c->max_tris = need;
    return POPPY_OK;
}

int alloc_pair(poppy_hip_ctx* c, int W, int H) {
    { int rc = drain_frames(c); if (rc) return rc; }              // every pair loader comes through here: no frame still reads the old pair
    if (c->W == W && c->H == H && c->c1) return POPPY_OK;
    free_pair(c);
    if (c->cfg.pyramid_levels < 1 || c->cfg.pyramid_levels > 256) return fail(c, POPPY_E_UNSUPPORTED, "pyramid_levels must be in [1,256]");
    const size_t P = (size_t)W * H;
    const int L = c->cfg.pyramid_levels;
    c->levels.resize(L + 1);
    size_t off3 = 0, off1 = 0;
    int w = W, h = H;
    for (int i = 0; i <= L; ++i) {
        const int pitch = i == 1 ? level1_pitch(w, h, c->levels[0].pitch, W) : level_pitch(w, h);      // rows of the large levels begin on 16-byte boundaries (kernels.h)
        c->levels[i] = PyrLevel{w, h, off3, off1, pitch};
        off3 += (size_t)pitch * h * 3; off1 += (size_t)pitch * h;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    const size_t P0 = (size_t)c->levels[0].pitch * H;             // pixels of a padded level-0 image (= P for widths that are multiples of 4)
    c->first_tail = L;
    static const size_t tail_px = getenv("POPPY_TAIL_PX") ? (size_t)atoi(getenv("POPPY_TAIL_PX")) : 600;
    for (int i = 1; i <= L; ++i)
        if ((size_t)c->levels[i].w * c->levels[i].h <= tail_px) { c->first_tail = i; break; }   // everything below runs in ONE workgroup: keep it small
    c->tail = build_pyr_tail_plan(c->levels.data(), std::min(c->first_tail, L), L);
    c->use_tail = c->tail.ok;
    static const bool rider = getenv("POPPY_HIP_LBMASK_RIDER") != nullptr;
    c->lazy_mask = !rider && c->cfg.pyramid_levels >= 1 && c->first_tail >= 1 && pyr_level0_vec_ok(W, H);
    if (c->use_tail) {
        if (!prepare_pyr_tail(c->tail.lds_bytes)) return fail(c, POPPY_E_DEVICE, "could not raise the tail kernel's LDS limit");
    } else {
        // A shallow pyramid (--pyramid 4 at 1080p ends at 120 x 68): every level goes through the per-level kernels and the
        // coarsest-level mix runs from global memory (blend.hpp simply loops `levels` times, any depth is legal).
        c->first_tail = L;
    }
    // +16: k_warp4 fetches footprints with 8-byte loads (6 bytes used), the last one may run 2 bytes past the image
    c->arena_bytes = pair_state_bytes(W, H);
    HIPCHK(c, hipMalloc((void**)&c->arena, c->arena_bytes));
    c->c1 = c->arena + kPairHeadBytes + 2 * (size_t)kPairMaxPoints * 8;
    c->c2 = c->c1 + pair_align(P * 3 + 16);
    c->m2 = (float*)(c->c2 + pair_align(P * 3 + 16));
    HIPCHK(c, hipMalloc((void**)&c->gabor2, P * 12));
    for (FrameSlot& f : c->slots) {
        HIPCHK(c, hipMalloc((void**)&f.tr1, P0 * 3 + 16)); HIPCHK(c, hipMalloc((void**)&f.tr2, P0 * 3 + 16));
        HIPCHK(c, hipMalloc((void**)&f.out, P * 3 + 16));
        HIPCHK(c, hipMalloc((void**)&f.triMap, P * 4));
        f.map_tag = 0;
        HIPCHK(c, hipMalloc((void**)&f.pyrL, off3 * 4)); HIPCHK(c, hipMalloc((void**)&f.pyrR, off3 * 4));
        HIPCHK(c, hipMalloc((void**)&f.pyrB, off3 * 4)); HIPCHK(c, hipMalloc((void**)&f.pyrM, off1 * 4));
        if (W < 2 || H < 2) { HIPCHK(c, hipMalloc((void**)&f.tmp, P * 12)); HIPCHK(c, hipMalloc((void**)&f.diff, P * 12)); }
    }
    if (c->use_tail) {
        HIPCHK(c, hipMalloc(&c->d_levels, c->tail.desc.size() * 4 + 16));
        if (!c->tail.desc.empty()) HIPCHK(c, hipMemcpy(c->d_levels, c->tail.desc.data(), c->tail.desc.size() * 4, hipMemcpyHostToDevice));
    }
    c->W = W; c->H = H;
    return POPPY_OK;
}

int set_points(poppy_hip_ctx* c, const float* p1, const float* p2, int n) {
    if (n < 0 || (n > 0 && (!p1 || !p2))) return fail(c, POPPY_E_ARG, "bad point sets");
    c->pts1_0.resize(n); c->pts2.resize(n);
    if (n) { memcpy(c->pts1_0.data(), p1, (size_t)n * 8); memcpy(c->pts2.data(), p2, (size_t)n * 8); }
    c->pts1 = c->pts1_0;
    const int rc = ensure_ring(c, n);
    if (rc == POPPY_OK) start_default_seq_plans(c);                // (the previous pair's plans, if a call never took them, are dropped in there)
    return rc;
}

int finish_pair_load(poppy_hip_ctx* c) {
    launch_gray_inv(c->gabor2, c->m2, c->W * c->H, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->inputs_ready, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));       // frames on other streams do not wait for the pair on the device: it is complete here
    c->cur1 = c->c1; c->cur1_ready = nullptr; c->last_slot = -1; c->pair_ready = true;
    return POPPY_OK;
}

// every frame queued on this context has finished (independent frames run on their slots' streams)
int drain_frames(poppy_hip_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (FrameSlot& f : c->slots) {
        if (f.stream) HIPCHK(c, hipStreamSynchronize(f.stream));
        if (f.own_stream && f.own_stream != f.stream) HIPCHK(c, hipStreamSynchronize(f.own_stream));
    }
    return POPPY_OK;
}

int stage_pair_state(poppy_hip_ctx* c) {
    const int n = (int)c->pts1_0.size();
    if (n > kPairMaxPoints) return fail(c, POPPY_E_UNSUPPORTED, "more point pairs than the packed pair state has room for");
    std::vector<uint8_t> head(kPairHeadBytes + 2 * (size_t)kPairMaxPoints * 8, 0);
    PairStateHeader h{};
    h.magic = kPairMagic; h.version = 1; h.W = c->W; h.H = c->H; h.n_points = n; h.nfeatures = c->last_nfeatures;
    h.initial_morph_dist = c->initial_morph_dist; h.detail[0] = c->last_detail[0]; h.detail[1] = c->last_detail[1];
    memcpy(head.data(), &h, sizeof h);
    if (n) {
        memcpy(head.data() + kPairHeadBytes, c->pts1_0.data(), (size_t)n * 8);
        memcpy(head.data() + kPairHeadBytes + (size_t)kPairMaxPoints * 8, c->pts2.data(), (size_t)n * 8);
    }
    HIPCHK(c, hipMemcpyAsync(c->arena, head.data(), head.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));           // `head` goes out of scope
    return POPPY_OK;
}

int adopt_pair_state(poppy_hip_ctx* c) {
    std::vector<uint8_t> head(kPairHeadBytes + 2 * (size_t)kPairMaxPoints * 8);
    HIPCHK(c, hipMemcpyAsync(head.data(), c->arena, head.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    PairStateHeader h;
    memcpy(&h, head.data(), sizeof h);
    if (h.magic != kPairMagic || h.version != 1) return fail(c, POPPY_E_ARG, "not a packed pair state");
    if (h.W != c->W || h.H != c->H) return fail(c, POPPY_E_ARG, "packed pair state has another geometry");
    if (h.n_points < 0 || h.n_points > kPairMaxPoints) return fail(c, POPPY_E_ARG, "packed pair state is corrupt");
    c->last_nfeatures = h.nfeatures; c->initial_morph_dist = h.initial_morph_dist;
    c->last_detail[0] = h.detail[0]; c->last_detail[1] = h.detail[1];
    int rc = set_points(c, (const float*)(head.data() + kPairHeadBytes), (const float*)(head.data() + kPairHeadBytes + (size_t)kPairMaxPoints * 8), h.n_points);
    if (rc) return rc;
    c->c2_raw_valid = false;
    HIPCHK(c, hipEventRecord(c->inputs_ready, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));       // frames on other streams do not wait for the pair on the device: it is complete here
    c->cur1 = c->c1; c->cur1_ready = nullptr; c->last_slot = -1; c->pair_ready = true;
    return POPPY_OK;
}

static int stage_host(poppy_hip_ctx* c, size_t bytes) {
    if (bytes <= c->h_stage_bytes) return POPPY_OK;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_stage, bytes, hipHostMallocMapped));
    HIPCHK(c, hipHostGetDevicePointer(&c->h_stage_dev, c->h_stage, 0));
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
static int prepare_ahead(poppy_hip_ctx* c, const FramePlan& plan, double mask);

// one frame on the resident pair; result in frame[slot]
static int render_frame(poppy_hip_ctx* c, double shape, double mask, bool chain) {
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    if (c->pts1.empty()) return fail(c, POPPY_E_NOMATCH, "no point pairs (use poppy_hip_dissolve)");
    int rc = plan_frame(c->W, c->H, c->pts1, c->pts2, shape, c->plan);
    if (rc) return fail(c, POPPY_E_RANGE, "point outside the image rectangle (Subdiv2D::insert would throw)");
    if (warp_bin_geometry(c->W, c->H)) { const int tw = warp_bin_tile_width(c->W, c->H); build_tile_bins(c->plan, c->W, c->H, tw, 1024 / tw, c->bins_cap); }
    return submit_frame(c, mask, chain);
}

// Multi-frame calls plan on a small pool of host threads: only the POINT chain is sequential in chained mode
// (src/poppy.hpp:178-179,218: srcPoints1 <- morphedPoints), and that is a few hundred multiply-adds per frame; the
// triangulation and matrix work of the frames is independent once each frame's input points are known.
constexpr int kPlanThrew = -1000;          // rcs[] marker: the planner of that frame threw

// The plans of one multi-frame call, made by the context's planner team.  They live on the heap because the team may be started BEFORE the call that consumes them
// (round 6): a pair loader starts the plans of the reference's default sequence — number_of_frames chained frames, src/poppy.hpp:177-210 — the moment the point pairs are
// known, while the set-up's last kernels and copies still run; poppy_hip_morph_frames then finds its first plans ready instead of idling the GPU for the 0.3-0.5 ms the
// first plan takes (one context, pair after pair: 4 % of a pair).  A call with other frames drops them (the planners stop at their next frame) and makes its own.
struct SeqPlans {
    int n = 0, W = 0, H = 0;
    bool chain = false, abandoned = false;                     // abandoned: told to stop before every frame was planned (never adopted)
    std::vector<double> shape;
    std::vector<P2f> pts1_at_start, pts2;                      // the point sets the plans were made from (a call adopts them only for the same ones)
    std::vector<std::vector<P2f>> src1;
    std::vector<FramePlan> plans;
    std::vector<int> rcs;
    std::vector<std::atomic<int>> ready;
    std::atomic<int> next{0};
    explicit SeqPlans(int n_) : n(n_), src1(n_), plans(n_), rcs(n_, 0), ready(n_) { for (auto& r : ready) r.store(0); }
};
static bool same_points(const std::vector<P2f>& a, const std::vector<P2f>& b) {           // bit for bit
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(P2f)) == 0);
}
static void stop_seq_plans(poppy_hip_ctx* c) {
    SeqPlans* sp = static_cast<SeqPlans*>(c->seq_plans);
    if (!sp) return;
    sp->next.store(sp->n);                                         // the planners stop at their next frame
    (void)c->planners.wait();
    delete sp;
    c->seq_plans = nullptr;
}
static SeqPlans* start_seq_plans(poppy_hip_ctx* c, const double* shape, int n, bool chain) {
    const int W = c->W, H = c->H;
    SeqPlans* sp = new SeqPlans(n);
    sp->W = W; sp->H = H; sp->chain = chain; sp->shape.assign(shape, shape + n); sp->pts1_at_start = c->pts1; sp->pts2 = c->pts2;
    std::vector<std::vector<P2f>>& src1 = sp->src1;
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
    // planner threads of this call: at most 16, and the contexts alive in this process share the host's threads between them (a pool of 3 contexts
    // x 8 devices would otherwise park ~400 planner threads; the planners of one context keep up with its GPU from ~4 threads: 0.3 ms per plan)
    const int alive = std::max(1, g_live_contexts.load());
    const int share = std::max(4, ((int)std::thread::hardware_concurrency() - 1) / alive);
    const int nthreads = std::max(1, std::min({n, 16, share, (int)std::thread::hardware_concurrency() - 1}));
    const int bin_tw = warp_bin_geometry(W, H) ? warp_bin_tile_width(W, H) : 0;
    const size_t bins_cap = c->bins_cap;
    auto worker = [sp, W, H, bin_tw, bins_cap, chain]() {
        for (;;) {
            const int j = sp->next.fetch_add(1);
            if (j >= sp->n) return;
            // a planner that throws (std::bad_alloc is the case that can happen) must still publish its frame: the calling thread spins on ready[j]
            try {
                sp->rcs[j] = plan_frame(W, H, chain ? sp->src1[j] : sp->src1[0], sp->pts2, sp->shape[j], sp->plans[j]);
                if (!sp->rcs[j] && bin_tw) build_tile_bins(sp->plans[j], W, H, bin_tw, 1024 / bin_tw, bins_cap);
            } catch (...) { sp->rcs[j] = kPlanThrew; }
            sp->ready[j].store(1, std::memory_order_release);
        }
    };
    c->planners.run(nthreads, worker);                            // persistent threads (worker.h): parked between calls
    c->seq_plans = sp;
    return sp;
}
// a pair loader's speculative start (set_points): the reference's default sequence on the new pair
void start_default_seq_plans(poppy_hip_ctx* c) {
    static const bool off = getenv("POPPY_HIP_NO_PLAN_AHEAD") != nullptr;
    if (SeqPlans* old = static_cast<SeqPlans*>(c->seq_plans)) {
        // The previous pair's plans were never taken: this caller loads pairs without rendering the default sequence in between (a set-up timing loop, a caller
        // of single frames).  Planning ahead for it only burns host threads beside its next set-up (0.3 ms per set-up in such a loop), and waiting for planners in
        // mid-frame cost another 0.2 ms: they are told to stop and left to finish, and no plans are started for a pair until a multi-frame call has been seen again.
        c->plan_ahead_credit = false;
        old->abandoned = true;
        old->next.store(old->n);
        if (!c->planners.idle()) return;
    }
    stop_seq_plans(c);
    if (!c->plan_ahead_credit) return;
    const int N = c->cfg.number_of_frames;
    if (off || N < 2 || c->pts1.empty() || c->debug) return;
    std::vector<double> ratio(N);
    for (int j = 0; j < N; ++j) ratio[j] = poppy_frame_ratio(j, N, -1.0);
    start_seq_plans(c, ratio.data(), N, true);
}

static int render_sequence(poppy_hip_ctx* c, const double* shape, const double* mask, int n, bool chain, poppy_write_cb write, void* user) {
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    if (c->pts1.empty()) return fail(c, POPPY_E_NOMATCH, "no point pairs (use poppy_hip_dissolve)");
    if (n <= 0) return POPPY_OK;
    const int W = c->W, H = c->H;
    if (n >= 2) c->plan_ahead_credit = true;                       // a caller of sequences: the next pair loader plans ahead again (start_default_seq_plans)
    // the plans a pair loader started for exactly these frames on exactly these points, or new ones
    SeqPlans* sp = static_cast<SeqPlans*>(c->seq_plans);
    if (!(sp && !sp->abandoned && sp->n == n && sp->chain == chain && sp->W == W && sp->H == H && same_points(sp->pts1_at_start, c->pts1) && same_points(sp->pts2, c->pts2) &&
          std::equal(shape, shape + n, sp->shape.begin()))) {
        stop_seq_plans(c);
        sp = start_seq_plans(c, shape, n, chain);
    }
    std::vector<std::vector<P2f>>& src1 = sp->src1;
    std::vector<FramePlan>& plans = sp->plans;
    std::vector<int>& rcs = sp->rcs;
    std::vector<std::atomic<int>>& ready = sp->ready;
    std::atomic<int>& next = sp->next;
    int rc = POPPY_OK;
    const size_t row = (size_t)W * 3, frame_bytes = row * H;
    static const int ring_pref = getenv("POPPY_HIP_RING") ? std::max(1, atoi(getenv("POPPY_HIP_RING"))) : 3;
    const int R = std::min({poppy_hip_ctx::kStageRing, ring_pref, (int)c->slots.size()});
    const size_t slot_bytes = (frame_bytes + 255) & ~(size_t)255;     // ring slots start on 256-byte boundaries
    int written = 0;
    if (write) rc = stage_host(c, slot_bytes * R);
    c->writer_attached = write != nullptr;                        // (phase-mode frames pick their streams by it: submit_frame)
    // Frame hand-off.  The download of a frame runs on its own stream into a ring of R pinned buffers while the GPU renders the
    // frames behind it, and the writer gets frames in order, R - 1 downloads behind.  A copy whose start depends on an event of
    // ANOTHER stream is launched by the runtime's asynchronous-event thread when that event fires; with two contexts rendering and
    // downloading at once those launches crawled (22 GB/s together against 52 GB/s for copies without a dependency:
    // tools/experiments/d2h_raw.py, overlap_probe.py).  So the host waits for frame j-1 itself — frame j is already queued, the GPU
    // never idles for it — and then issues a copy that depends on nothing.  (POPPY_HIP_DL_DEVWAIT=1: the dependent form.)  The copy is
    // the runtime's: a kernel of ours storing the frame into the mapped ring costs the frame kernels beside it far more (3.7k frames/s
    // against 5.8k, whatever its geometry: shader stores over PCIe hold up the other kernels' stores, profiles/r02_notes.md section 7).
    static const bool dev_wait = getenv("POPPY_HIP_DL_DEVWAIT") != nullptr;
    static const bool seq_times = getenv("POPPY_SEQ_TIMING") != nullptr;      // where the calling thread's time goes, on stderr
    using clk = std::chrono::steady_clock;
    double ms_plan = 0, ms_done = 0, ms_deliver = 0, ms_submit = 0;
    const auto t_seq = clk::now();
    const double w0[4] = {c->wait_ms[0], c->wait_ms[1], c->wait_ms[2], c->wait_ms[3]};
    auto lap = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    std::vector<int> slot_of(n, -1);
    int issued = 0;                                               // downloads queued so far (frames 0 .. issued-1)
    // Round 6: a frame copy goes to the stream of its pinned ring buffer, which carries nothing else, and NO event is recorded behind it — whoever needs the copy
    // finished synchronises that stream.  An event record behind a copy is a marker packet that waits, in one of the process's four hardware queues, for the copy's
    // signal, and every kernel of every stream mapped to that queue waits with it for the length of a frame copy (~120 us): a pool of six with the writer 6.2 -> 6.6-6.9k
    // frames/s (7.4-7.6k on the runtime bundled with torch), one context 5.2 -> 5.4k, the 480-frame phase-mode job 6.0-6.2 -> 7.2-7.5k (profiles/r06_dl_streams.txt).
    // POPPY_HIP_DL_EVENTS=1: one download stream + an event per copy, as until round 5.
    static const bool dl_streams = getenv("POPPY_HIP_DL_EVENTS") == nullptr && !dev_wait;
    auto issue_download = [&](int k) -> bool {
        FrameSlot& f = c->slots[slot_of[k]];
        const int r = k % R;
        const auto t0 = clk::now();
        hipError_t e = dev_wait ? hipStreamWaitEvent(c->dl_stream, f.done, 0) : hipEventSynchronize(f.done);
        ms_done += lap(t0);
        if (dl_streams) {
            if (e == hipSuccess && !c->dl_ring[r]) e = hipStreamCreateWithFlags(&c->dl_ring[r], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_stage + (size_t)r * slot_bytes, f.out, frame_bytes, hipMemcpyDeviceToHost, c->dl_ring[r]);
            f.dl_pending = true; f.dl_ring_idx = r;
            if (e != hipSuccess) { c->err = std::string("frame download: ") + hipGetErrorString(e); rc = POPPY_E_DEVICE; return false; }
            return true;
        }
        f.dl_ring_idx = -1;
#ifdef POPPY_EXPERIMENTS
        static const bool skip_copy = getenv("POPPY_DL_SKIP_COPY") != nullptr;      // timing experiment: every wait and event of the writer path, no bytes moved (wrong frames)
        if (!skip_copy)
#endif
        if (e == hipSuccess) e = hipMemcpyAsync(c->h_stage + (size_t)r * slot_bytes, f.out, frame_bytes, hipMemcpyDeviceToHost, c->dl_stream);
        if (e == hipSuccess) e = hipEventRecord(c->dl_done[r], c->dl_stream);
        if (e == hipSuccess) e = hipEventRecord(f.downloaded, c->dl_stream);          // the slot's own: ring events are re-recorded every R frames
        f.dl_pending = true;
        if (e != hipSuccess) { c->err = std::string("frame download: ") + hipGetErrorString(e); rc = POPPY_E_DEVICE; return false; }
        return true;
    };
    auto deliver = [&](int k) -> bool {
        const int rr = k % R;
        const auto t0 = clk::now();
        if ((dl_streams ? hipStreamSynchronize(c->dl_ring[rr]) : hipEventSynchronize(c->dl_done[rr])) != hipSuccess) { c->err = "frame download failed"; rc = POPPY_E_DEVICE; return false; }
        ms_deliver += lap(t0);
        write(user, c->h_stage + (size_t)rr * slot_bytes, W, H, row);
        ++written;
        return true;
    };
    for (int j = 0; j < n && rc == POPPY_OK; ++j) {
        const auto t_plan = clk::now();
        while (!ready[j].load(std::memory_order_acquire)) std::this_thread::yield();
        ms_plan += lap(t_plan);
        if (rcs[j] == kPlanThrew) { rc = fail(c, POPPY_E_DEVICE, "frame planner failed (out of memory?)"); break; }
        if (rcs[j]) { rc = fail(c, POPPY_E_RANGE, "point outside the image rectangle (Subdiv2D::insert would throw)"); break; }
        c->plan = std::move(plans[j]);
        if (chain) c->pts1 = src1[j];
        if (write) {
            // The slot frame j renders into may still hold a frame whose download has not been issued (few slots, or every frame
            // landing in the one slot that does not hold corrected1): that copy goes out first; submit_frame then waits for it.
            int ps = c->next_slot;
            if (c->slots[ps].out == c->cur1) ps = (ps + 1) % (int)c->slots.size();
            int last_user = -1;
            for (int k = issued; k < j; ++k) if (slot_of[k] == ps) last_user = k;
            while (issued <= last_user && rc == POPPY_OK) {
                while (issued - written >= R && rc == POPPY_OK) deliver(written);
                if (rc == POPPY_OK && issue_download(issued)) ++issued;
            }
            if (rc != POPPY_OK) break;
        }
        const auto t_sub = clk::now();
        rc = submit_frame(c, mask[j], chain);
        // chained frames: the NEXT frame's plan goes up and is expanded now, behind this frame's launches (the wait for it at the head of the next
        // submit_frame then finds it done); only when its plan is ready — the planners are normally far ahead
        if (rc == POPPY_OK && chain && j + 1 < n && ready[j + 1].load(std::memory_order_acquire) && rcs[j + 1] == 0) rc = prepare_ahead(c, plans[j + 1], mask[j + 1]);
        ms_submit += lap(t_sub);
        if (rc != POPPY_OK) break;
        slot_of[j] = c->last_slot;
        if (write) {
            const int upto = dev_wait ? j + 1 : j;                // frames whose download can be issued now
            while (issued < upto && rc == POPPY_OK) {
                while (issued - written >= R && rc == POPPY_OK) deliver(written);
                if (rc == POPPY_OK && issue_download(issued)) ++issued;
            }
        }
    }
    if (write && rc == POPPY_OK) {
        while (issued < n && rc == POPPY_OK) {                    // the last frame(s), then drain the ring
            while (issued - written >= R && rc == POPPY_OK) deliver(written);
            if (rc == POPPY_OK && issue_download(issued)) ++issued;
        }
        while (written < n && rc == POPPY_OK) deliver(written);
    }
    c->writer_attached = false;
    drop_slot_preps(c);                    // (a frame prepared ahead and never rendered — an error exit — must not meet a later call)
    next.store(n);                         // on an error: let the workers drain
    if (!c->planners.wait() && rc == POPPY_OK) rc = fail(c, POPPY_E_DEVICE, ("frame planner thread: " + c->planners.error()).c_str());
    delete sp;
    c->seq_plans = nullptr;
    if (seq_times)
        fprintf(stderr, "sequence of %d frames: %.2f ms; waiting for plans %.2f, submit_frame %.2f (of which waiting for: the slot's download %.2f, its pinned plan %.2f, "
                "its last frame %.2f, upload + expansion %.2f), waiting for frames to finish %.2f, waiting for downloads %.2f ms\n",
                n, lap(t_seq), ms_plan, ms_submit, c->wait_ms[0] - w0[0], c->wait_ms[1] - w0[1], c->wait_ms[2] - w0[2], c->wait_ms[3] - w0[3], ms_done, ms_deliver);
    return rc;
}

static_assert(kPlanRasterRows == kRasterChunkRows, "the plan's work list and k_raster must agree on the chunk height");

// pyrdown .. unsharp of one slot.  Every argument is fixed for the life of the pair (the per-frame unsharp amount is
// read from the slot's plan blob), which is what lets the whole sequence be captured into one graph launch.
static void enqueue_body(poppy_hip_ctx* c, FrameSlot& f, hipStream_t s, Timer* tm, float amount, bool debug, hipEvent_t done = nullptr) {
    const int W = c->W, H = c->H, L = c->cfg.pyramid_levels;
    const int ft = c->first_tail < L ? c->first_tail : L;
    static const bool fuse = getenv("POPPY_HIP_NOFUSE") == nullptr;
    for (int i = 0; i < ft;) {
        const PyrLevel &a = c->levels[i], &b = c->levels[i + 1];
        if (fuse && i >= 1 && i + 2 <= ft && b.pitch == b.w && c->levels[i + 2].pitch == c->levels[i + 2].w && pyrdown2_eligible(a.w, a.h)) {     // two small levels in one launch (the two it writes are tight)
            const PyrLevel& d = c->levels[i + 2];
            launch_pyrdown2(f.pyrL + a.off3, f.pyrR + a.off3, f.pyrM + a.off1, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrM + b.off1,
                            f.pyrL + d.off3, f.pyrR + d.off3, f.pyrM + d.off1, a.w, a.h, s, a.pitch);
            i += 2;
            continue;
        }
        const void* sl = i == 0 ? (const void*)f.tr1 : (const void*)(f.pyrL + a.off3);
        const void* sr = i == 0 ? (const void*)f.tr2 : (const void*)(f.pyrR + a.off3);
        const bool lazy = i == 0 && c->lazy_mask;      // level 0 reads the mask through m2 (kernels.h: launch_pyrdown)
        launch_pyrdown(sl, sr, lazy ? c->m2 : f.pyrM + a.off1, i == 0, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrM + b.off1, a.w, a.h, s,
                       lazy ? (const double*)(f.d_blob + kBlobMaskAB) : nullptr, a.pitch, lazy ? a.w : a.pitch, b.pitch);
        ++i;
    }
    if (tm) tm->mark("pyrdown");
    if (c->use_tail) launch_pyr_tail(f.pyrL, f.pyrR, f.pyrM, f.pyrB, c->d_levels, c->tail.args, c->tail.lds_bytes, s);
    else launch_mix_top(f.pyrL + c->levels[L].off3, f.pyrR + c->levels[L].off3, f.pyrM + c->levels[L].off1, f.pyrB + c->levels[L].off3,
                        c->levels[L].w * c->levels[L].h, s);
    if (tm) tm->mark("pyr_tail");
    // The way up: the small levels in ONE launch (round 6, kernels_pyramid_cone.hip): from the tail's level to the largest level of at most kConeMaxPixels
    // (level 1 at 1080p, level 2 at 4K).  POPPY_HIP_NOCONE: the launches of round 5 (k_collapse2 pairs + one k_collapse_level per remaining level).
    static const bool cone = getenv("POPPY_HIP_NOCONE") == nullptr;
    int j_top = ft;
    if (fuse && cone) {
        int k = 1;
        while (k < ft && (size_t)c->levels[k].w * c->levels[k].h > kConeMaxPixels) ++k;
        if (ft - k > kConeMaxLevels) k = ft - kConeMaxLevels;
        if (ft - k >= 2 && collapse_cone_eligible(&c->levels[k], ft - k)) {
            launch_collapse_cone(f.pyrL, f.pyrR, f.pyrM, f.pyrB, &c->levels[k], ft - k, s);
            j_top = k;
        }
    }
    for (int j = j_top; j > 0;) {                  // blended level j is known; produce level j-2 or j-1
        if (fuse && j - 2 >= 1) {
            const PyrLevel &a = c->levels[j - 2], &m = c->levels[j - 1], &n = c->levels[j];
            if (a.pitch == a.w && m.pitch == m.w && collapse2_eligible(a.w, a.h, m.w, m.h, n.w, n.h)) {
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
        const bool lazy = i == 0 && c->lazy_mask;
        launch_collapse(gl, gr, i == 0, lazy ? c->m2 : f.pyrM + a.off1, f.pyrL + b.off3, f.pyrR + b.off3, f.pyrB + b.off3, f.pyrB + a.off3,
                        a.w, a.h, b.w, b.h, s, lazy ? (const double*)(f.d_blob + kBlobMaskAB) : nullptr, a.pitch, lazy ? a.w : a.pitch, b.pitch);
        --j;
    }
    if (tm) tm->mark("collapse");
    launch_unsharp(f.pyrB, f.tmp, f.diff, f.out, debug ? f.unsharpF : nullptr, W, H, amount, (const float*)f.d_blob, (float)0.3, s, done, c->levels[0].pitch);
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

// A frame in two halves (round 6).  PREPARE: the slot's plan blob is filled, uploaded and expanded into id bytes + record slots (k_upload, k_tile_expand) — that depends
// on the plan only.  RENDER: everything that reads images.  For chained frames the first half runs on the copy stream and the HOST waits for it before it launches
// the warp kernel (no device-side wait across hardware queues, see below); until round 6 that wait sat between the two halves of the SAME frame — 38 us of every frame's
// ~100 us of host time alone, ~500 us per frame in a pool, where the copy stream's packets queue behind other contexts' kernels.  render_sequence now prepares frame j + 1
// right after it has launched frame j: the wait at the head of frame j + 1 finds the event complete.
struct SlotPrep {
    bool valid = false;
    int T = 0, n_work = 0, tile_w = 0;
    bool bin_warp = false, fast_warp = false, chained = false, use_graph = false;
    unsigned long long seq = 0;            // the submit_frame call this was prepared for (0: prepared by that call itself)
    size_t rec_bytes = 0, o_edges = 0, o_outl = 0, o_toff = 0, o_ttri = 0, o_tri = 0, o_inv = 0, o_work = 0, used = 0;
    double mask = 0;
    hipStream_t s = nullptr;
    std::vector<P2f> morphed;
};
static std::vector<SlotPrep>& preps_of(poppy_hip_ctx* c) {             // one record per slot (kept behind a pointer: context.h stays free of frame_plan.h types)
    if (!c->slot_prep_store) c->slot_prep_store = new std::vector<SlotPrep>();
    auto* v = static_cast<std::vector<SlotPrep>*>(c->slot_prep_store);
    if (v->size() != c->slots.size()) v->assign(c->slots.size(), SlotPrep());
    return *v;
}
static void free_slot_preps(poppy_hip_ctx* c) { delete static_cast<std::vector<SlotPrep>*>(c->slot_prep_store); c->slot_prep_store = nullptr; }
static void drop_slot_preps(poppy_hip_ctx* c) { if (c->slot_prep_store) for (SlotPrep& p : *static_cast<std::vector<SlotPrep>*>(c->slot_prep_store)) p.valid = false; }

// the slot the next frame renders into: the next one in the ring that does not hold the image that frame reads as corrected1
static int pick_slot(const poppy_hip_ctx* c) {
    int fi = c->next_slot;
    if (c->slots[fi].out == c->cur1) fi = (fi + 1) % (int)c->slots.size();
    return fi;
}

static auto waited_on(poppy_hip_ctx* c) {
    return [c](double& acc, hipEvent_t ev) -> hipError_t {             // host wait for an event, accounted (POPPY_SEQ_TIMING)
        (void)c;
        const auto t0 = std::chrono::steady_clock::now();
        const hipError_t e = hipEventSynchronize(ev);
        acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return e;
    };
}

// first half: `plan` into slot fi (blob, upload, expansion).  chained: on the copy stream, nobody waits here.
static int prepare_slot(poppy_hip_ctx* c, const FramePlan& plan, double mask, bool chain, int fi) {
    const int W = c->W, H = c->H;
    const int T = plan.n_tris;
    if (T > c->max_tris) return fail(c, POPPY_E_ARG, "triangle budget exceeded");
    static_assert(((long long)kIdTagMax << kIdTagShift) + (1ll << kIdTagShift) - 1 <= 0x7fffffffll, "tagged ids must stay positive int32 values");
    if (T + 1 >= (1 << kIdTagShift)) return fail(c, POPPY_E_UNSUPPORTED, "more triangles than the id map's tag scheme can number (2^20 - 2)");
    FrameSlot& f = c->slots[fi];
    SlotPrep& pr = preps_of(c)[fi];
    pr.valid = false;
    auto waited = waited_on(c);
    HIPCHK(c, waited(c->wait_ms[1], f.uploaded));                  // the pinned copy is free again
    const double amount = std::sin(mask * M_PI);
    *(float*)f.h_blob = (float)(1.0 - amount);                     // unsharp_mask(.., 1, 1.0 - amount, 0.3)
    ((double*)(f.h_blob + kBlobMaskAB))[0] = 1.0 - mask;           // lbmask = clamp(alpha + m2 * beta), read by the level-0 blend kernels
    ((double*)(f.h_blob + kBlobMaskAB))[1] = -mask;
    // The slot's plan blob: header | warp records | fill-edge tables | (fused path) outline segments, per-tile offsets, per-tile triangle lists |
    // (id-map path only) integer triangles, inverse matrices, k_raster's work list.  A frame uploads what ITS kernels read: the fused path's two kernels
    // never look at the last group (round 6: ~255 KB instead of ~400 KB per 1080p frame over PCIe, k_upload 12 -> 8 us).
    const size_t rec_bytes = (size_t)(T + 1) * kWarpRecordFloats * sizeof(float);
    const int n_work = (int)(plan.work.size() / 2);
    const size_t n_toff = plan.tile_off.size(), n_ttri = plan.tile_tris.size();
    static const bool idmap_only = getenv("POPPY_HIP_IDMAP") != nullptr;
    const bool bins = plan.bins_ok && !idmap_only && !c->debug && n_ttri <= c->bins_cap && plan.max_tile_entries <= warp_bin_max_tile_entries() &&
                      plan.tile_w == warp_bin_tile_width(W, H);
    // the fast warp kernels take the frame when every matrix passes the host's range check (always, short of degenerate input)
    static const bool exact_warp_only = getenv("POPPY_HIP_GENERALWARP") != nullptr;
    if (kBlobHeader + rec_bytes > c->blob_bytes) return fail(c, POPPY_E_ARG, "plan blob overflow");
    const bool records_ok = pack_warp_records(plan.inv1.data(), plan.inv2.data(), T, W, H, (float*)(f.h_blob + kBlobHeader), plan.tri_xy.data()) && !exact_warp_only;
    // raster fused into the warp kernel: no id map at all.  Any width whose level-0 rows begin on 16-byte boundaries: multiples of 4, and every width from
    // 150 001 pixels up (level_pitch); small images of other widths keep the id-map path
    const bool bin_warp = records_ok && bins && warp_bin_geometry(W, H) && (c->levels[0].pitch & 3) == 0;
    const bool fast_warp = bin_warp || (records_ok && warp_fast_geometry(W, H));
    auto pad16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    size_t off = kBlobHeader + rec_bytes;
    const size_t o_edges = off; off += (size_t)T * sizeof(RasterTri);                       // 96-byte entries: stays 16-byte aligned
    size_t o_outl = 0, o_toff = 0, o_ttri = 0, o_tri = 0, o_inv = 0, o_work = 0;
    if (bin_warp) {
        o_outl = off; off += (size_t)T * 3 * sizeof(OutlineSeg);
        o_toff = off; off += pad16(n_toff * 4);
        o_ttri = off; off += pad16(n_ttri * 2);
    } else {
        o_tri = off; off += pad16((size_t)T * 6 * sizeof(int));
        o_inv = off; off += pad16((size_t)T * 18 * sizeof(float));
        o_work = off; off += pad16((size_t)n_work * 8);
    }
    const size_t used = off;
    if (used > c->blob_bytes) return fail(c, POPPY_E_ARG, "plan blob overflow");
    if (T) memcpy(f.h_blob + o_edges, plan.raster.data(), (size_t)T * sizeof(RasterTri));
    if (bin_warp) {
        if (T) memcpy(f.h_blob + o_outl, plan.outline.data(), (size_t)T * 3 * sizeof(OutlineSeg));
        memcpy(f.h_blob + o_toff, plan.tile_off.data(), n_toff * 4);
        if (n_ttri) memcpy(f.h_blob + o_ttri, plan.tile_tris.data(), n_ttri * 2);
    } else if (T) {
        memcpy(f.h_blob + o_tri, plan.tri_xy.data(), (size_t)T * 6 * sizeof(int));
        memcpy(f.h_blob + o_inv, plan.inv1.data(), (size_t)T * 9 * sizeof(float));
        memcpy(f.h_blob + o_inv + (size_t)T * 9 * sizeof(float), plan.inv2.data(), (size_t)T * 9 * sizeof(float));
        memcpy(f.h_blob + o_work, plan.work.data(), (size_t)n_work * 8);
    }
    const float* d_rec = (const float*)(f.d_blob + kBlobHeader);
    const RasterTri* d_edges = (const RasterTri*)(f.d_blob + o_edges);
    const OutlineSeg* d_outl = (const OutlineSeg*)(f.d_blob + o_outl);
    const int* d_toff = (const int*)(f.d_blob + o_toff);
    const uint16_t* d_ttri = (const uint16_t*)(f.d_blob + o_ttri);

    // Streams.  Device-side waits between streams that sit on different hardware queues cost 12-20 us each on this part
    // (profiles/r01_e_streams.md), and which streams share a queue is the runtime's choice (GPU_MAX_HW_QUEUES); phase-mode frames
    // ran at 8.3k or at 5.2k frames/s depending on it (tools/experiments/frames_only.py).  So no frame waits on another stream's
    // event on the device:
    //   chained frames      every kernel on the context's stream; the plan upload and the raster expansion run on the copy stream
    //                       beside the previous frame and the HOST waits for them (it is a frame ahead of the GPU anyway);
    //   independent frames  everything — upload, expansion, kernels — in order on the slot's own stream; the other slots' frames
    //                       hide the upload.  Whoever needs all frames finished drains the slot streams (drain_frames).
    static const bool no_graph = getenv("POPPY_HIP_NOGRAPH") != nullptr;
    const bool chained = chain || c->cur1_ready;
    if (!chained) {
        // Independent frames: a stream per slot (created on first use) when the frames stay in HBM — four frames in flight, 10.7k frames/s at 1080p.
        // With a writer attached the slots take the context's OWN three compute streams in turn — rendering, plan upload, the set-up's second —:
        // a hardware queue is in order, the runtime spreads streams over four of them as they are created, and a fifth stream lands on the queue
        // of the download stream, whose packets wait for every frame copy in front of that slot's kernels (one slot in four behind the copies:
        // a 480-frame sequence with a writer took 80 ms; 74 with the slots on the three queues that carry no copies; without a writer three
        // queues are slower than four, 8.7k frames/s).  POPPY_PHASE_OWN_STREAMS: a stream per slot in both cases, as before round 3.
        static const bool own_streams = getenv("POPPY_PHASE_OWN_STREAMS") != nullptr;
        if (c->writer_attached && !own_streams) {
            if (!c->aux_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
            hipStream_t pick[3] = {c->stream, c->copy_stream, c->aux_stream};
            f.stream = pick[fi % 3];
        } else {
            if (!f.own_stream) HIPCHK(c, hipStreamCreateWithFlags(&f.own_stream, hipStreamNonBlocking));
            f.stream = f.own_stream;
        }
    }
    hipStream_t s = chained ? c->stream : f.stream;
    if (f.last_stream && f.last_stream != s) HIPCHK(c, hipEventSynchronize(f.done));     // the mode changed: settle the slot's last frame once, on the host
    f.last_stream = s;
    if (c->debug && !f.unsharpF) HIPCHK(c, hipMalloc((void**)&f.unsharpF, (size_t)W * H * 12));
    const bool all_marks = c->timing == 1;
    // The captured body is for frames in flight beside each other (phase mode), where the submitting host thread is the
    // bottleneck.  On the chained critical path a graph launch leaves the GPU idle ~8 us longer than the same kernels
    // launched one by one (4600 vs 4785 frames/s, profiles/r01_e_streams.md), and the host keeps up easily.
    const bool use_graph = !no_graph && !chained && !c->debug && !all_marks && W > 1 && H > 1;
    if (use_graph && !f.body) { int rc = capture_body(c, f); if (rc) return rc; }

    pr.T = T; pr.n_work = n_work; pr.tile_w = plan.tile_w; pr.bin_warp = bin_warp; pr.fast_warp = fast_warp; pr.chained = chained; pr.use_graph = use_graph;
    pr.rec_bytes = rec_bytes; pr.o_edges = o_edges; pr.o_outl = o_outl; pr.o_toff = o_toff; pr.o_ttri = o_ttri; pr.o_tri = o_tri; pr.o_inv = o_inv; pr.o_work = o_work; pr.used = used;
    pr.mask = mask; pr.s = s; pr.morphed = plan.morphed; pr.seq = c->frame_seq;
    hipStream_t up = chained ? c->copy_stream : s;
    if (chained) HIPCHK(c, waited(c->wait_ms[2], f.done));        // the frame that last read this slot's device copy of the plan (2+ frames back)
    launch_upload(f.h_blob_dev, f.d_blob, used, up);
    // the raster of the frame, as one id byte per pixel + the tiles' record slots: needs the plan only
    if (bin_warp) launch_tile_expand(d_rec, d_edges, d_outl, d_toff, d_ttri, f.tile_data, plan.tile_w, W, H, up);
    // (The blend mask also depends on the plan only.  Taking it out of the warp kernel — a kernel of its own on this stream —
    // made that kernel faster (20.5 -> 18.1 us at 1080p, 53.6 -> 46.5 us at 4K) and the chained FRAME slower (183.8 -> 188.3 us,
    // 403 -> 411 us): the extra traffic beside the chain costs the chain's bandwidth-bound kernels more than the rider did.
    // profiles/r02_notes.md.)
    HIPCHK(c, hipEventRecord(f.uploaded, up));
    pr.valid = true;
    return POPPY_OK;
}

// second half: the frame prepared in slot fi
static int render_slot(poppy_hip_ctx* c, int fi, bool chain) {
    const int W = c->W, H = c->H;
    FrameSlot& f = c->slots[fi];
    SlotPrep& pr = preps_of(c)[fi];
    if (!pr.valid) return fail(c, POPPY_E_STATE, "frame slot not prepared");
    pr.valid = false;
    auto waited = waited_on(c);
    const int T = pr.T, n_work = pr.n_work;
    const bool bin_warp = pr.bin_warp, fast_warp = pr.fast_warp, chained = pr.chained, use_graph = pr.use_graph;
    const double mask = pr.mask, amount = std::sin(mask * M_PI);
    hipStream_t s = pr.s;
    const float* d_rec = (const float*)(f.d_blob + kBlobHeader);
    const RasterTri* d_edges = (const RasterTri*)(f.d_blob + pr.o_edges);
    const int* d_toff = (const int*)(f.d_blob + pr.o_toff);
    const int* d_tri = (const int*)(f.d_blob + pr.o_tri);
    const float* d_inv = (const float*)(f.d_blob + pr.o_inv);
    const int* d_work = (const int*)(f.d_blob + pr.o_work);
    c->last_warp_fast = fast_warp; c->last_warp_bin = bin_warp;
    ++(bin_warp ? c->n_warp_bin : fast_warp ? c->n_warp_fast : c->n_warp_general);
    // a frame of this slot may still be on its way to the writer (the ring only orders the HOST side): nothing may render into
    // `out` before that copy has read it
    if (f.dl_pending) {
        if (f.dl_ring_idx >= 0) {                                  // (POPPY_HIP_DL_STREAMS: the copy's own stream instead of an event)
            const auto t0 = std::chrono::steady_clock::now();
            HIPCHK(c, hipStreamSynchronize(c->dl_ring[f.dl_ring_idx]));
            c->wait_ms[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        } else HIPCHK(c, waited(c->wait_ms[0], f.downloaded));
        f.dl_pending = false;
    }
    const bool all_marks = c->timing == 1;
    Timer tm(c, s);
    if (all_marks) tm.mark(nullptr);
    if (chained) HIPCHK(c, waited(c->wait_ms[3], f.uploaded));
    // -- independent of the previous frame ---------------------------------------------------------------------
    // Id-map path only (debug mode, POPPY_HIP_IDMAP, oversized tile lists).  The id map is not cleared between frames: every
    // frame writes its ids above a tag that grows from frame to frame, and its warp kernel reads everything else as "no
    // triangle" (kernels.h: launch_raster).  A memset is needed for a slot's first frame, when the tags run out (every 2047
    // frames), and around debug frames, which keep a plain map for poppy_hip_debug_fetch.
    if (!bin_warp) {
        if (c->debug || f.map_tag == 0 || f.map_tag >= kIdTagMax) {
            HIPCHK(c, hipMemsetAsync(f.triMap, 0, (size_t)W * H * 4, s));
            f.map_tag = 0;
        }
        if (!c->debug) ++f.map_tag;
    }
    const uint32_t id_base = (uint32_t)f.map_tag << kIdTagShift;
    if (all_marks) tm.mark("upload+clear");
    if (!bin_warp) launch_raster(d_tri, d_edges, d_work, n_work, f.triMap, W, H, id_base, s);
    if (all_marks) tm.mark("raster");
    // -- chained mode: corrected1 is the previous frame (src/poppy.hpp:217) -------------------------------------
    if (c->cur1_ready && c->cur1_stream != s) HIPCHK(c, hipStreamWaitEvent(s, c->cur1_ready, 0));
    WarpExtras ex;
    ex.id_base = id_base;
    // lbmask (level 0 of pyrM) rides along with the warp only where the blend kernels cannot read it through m2
    ex.m2 = c->lazy_mask ? nullptr : c->m2; ex.mask = f.pyrM; ex.alpha = 1.0 - mask; ex.beta = -mask;
    ex.out_pitch = c->levels[0].pitch;
    if (c->timing == 2) {       // the dispatch's own begin / end timestamps: no marker packets in the stream
        // A stamped dispatch still costs the frame loop ~5 us (it completes through a signal the host can read: 2.6 % of a
        // chained 1080p frame when every launch is stamped), so one launch in kWarpStampStride carries the stamps; the
        // stride is coprime with the usual sequence lengths, so over a few sequences every frame position is sampled.
        static const int stride = getenv("POPPY_HIP_WARP_STAMP_STRIDE") ? std::max(1, atoi(getenv("POPPY_HIP_WARP_STAMP_STRIDE"))) : kWarpStampStride;
        const bool stamp = (c->warp_seq++ % (unsigned)stride) == 0;
        hipEvent_t t0 = stamp ? tm.take(nullptr) : nullptr, t1 = stamp ? tm.take("warp") : nullptr;
        if (bin_warp) launch_warp_bin(d_rec, f.tile_data, c->tile_bytes, d_toff, pr.tile_w, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s, t0, t1);
        else if (fast_warp) launch_warp_fast(f.triMap, d_rec, T + 1, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s, t0, t1);
        else launch_warp(f.triMap, d_inv, d_inv + (size_t)T * 9, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s, t0, t1);
    } else {
        if (!all_marks) tm.mark(nullptr);
        if (bin_warp) launch_warp_bin(d_rec, f.tile_data, c->tile_bytes, d_toff, pr.tile_w, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s);
        else if (fast_warp) launch_warp_fast(f.triMap, d_rec, T + 1, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s);
        else launch_warp(f.triMap, d_inv, d_inv + (size_t)T * 9, c->cur1, c->c2, f.tr1, f.tr2, W, H, ex, s);
        tm.mark("warp");
    }
    if (bin_warp) {
        c->last_warp.rec = d_rec; c->last_warp.tile_data = f.tile_data; c->last_warp.tile_bytes = c->tile_bytes;
        c->last_warp.toff = d_toff; c->last_warp.tile_w = pr.tile_w; c->last_warp.c1 = c->cur1; c->last_warp.c2 = c->c2;
        c->last_warp.tr1 = f.tr1; c->last_warp.tr2 = f.tr2; c->last_warp.ex = ex; c->last_warp.valid = true;
    } else c->last_warp.valid = false;
    // The frame's completion event rides on its last dispatch when the kernels are launched one by one: an event record of
    // its own behind the last kernel leaves the stream idle for ~6 us before the next frame's first kernel.
    static const bool done_packet = getenv("POPPY_HIP_DONE_PACKET") != nullptr;
    const bool done_rides = !use_graph && !all_marks && !done_packet;
    if (use_graph) HIPCHK(c, hipGraphLaunch(f.body, s));
    else enqueue_body(c, f, s, all_marks ? &tm : nullptr, (float)(1.0 - amount), c->debug, done_rides ? f.done : nullptr);
    HIPCHK(c, hipGetLastError());
    if (!done_rides) HIPCHK(c, hipEventRecord(f.done, s));
    c->last_slot = fi;
    if (chain) {                                   // src/poppy.hpp:217-218
        c->cur1 = f.out;
        c->cur1_ready = f.done;
        c->cur1_stream = s;
        c->pts1 = pr.morphed;
    }
    return POPPY_OK;
}


static int submit_frame(poppy_hip_ctx* c, double mask, bool chain) {
    const int fi = pick_slot(c);
    c->next_slot = (fi + 1) % (int)c->slots.size();
    ++c->frame_seq;
    SlotPrep& pr = preps_of(c)[fi];
    if (!(pr.valid && pr.seq == c->frame_seq && pr.chained && chain)) {             // (not prepared ahead for THIS call: both halves now)
        drop_slot_preps(c);
        int rc = prepare_slot(c, c->plan, mask, chain, fi); if (rc) return rc;
    }
    return render_slot(c, fi, chain);
}

// Chained frames only: the first half of the NEXT frame, launched behind the frame just submitted.  `plan` is that frame's; its slot is the one submit_frame will pick.
static int prepare_ahead(poppy_hip_ctx* c, const FramePlan& plan, double mask) {
    static const bool off = getenv("POPPY_HIP_NO_PREPARE_AHEAD") != nullptr;
    if (off || c->debug || c->timing != 0 || c->slots.size() < 3) return POPPY_OK;
    const int fi = pick_slot(c);
    const int rc = prepare_slot(c, plan, mask, true, fi);
    if (rc == POPPY_OK) preps_of(c)[fi].seq = c->frame_seq + 1;
    return rc;
}

int upload_image(poppy_hip_ctx* c, uint8_t* dst, const uint8_t* src, size_t stride, int W, int H) {
    // (tight rows go as one linear copy: kernels.h copy_rows_async; only images that really have padded rows — ROIs — pay the 2-D copy's slow path at odd widths)
    HIPCHK(c, copy_rows_async(dst, (size_t)W * 3, src, stride, (size_t)W * 3, H, hipMemcpyHostToDevice, c->stream));
    return POPPY_OK;
}

extern "C" {

int poppy_hip_pair_load(poppy_hip_ctx* c, const uint8_t* c1, size_t s1, const uint8_t* c2, size_t s2, const float* gabor2,
                        int W, int H, const float* p1, const float* p2, int n) {
    if (!c) return POPPY_E_ARG;
    if (!c1 || !c2 || !gabor2 || W <= 0 || H <= 0 || s1 < (size_t)W * 3 || s2 < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    c->c2_raw_valid = false;
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
    c->c2_raw_valid = false;
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
        FrameSlot& f = c->slots[c->last_slot];
        hipStream_t fs = f.last_stream ? f.last_stream : c->stream;               // in order behind the frame itself
        HIPCHK(c, copy_rows_async(dst, dst_stride, f.out, (size_t)c->W * 3, (size_t)c->W * 3, c->H, hipMemcpyDeviceToHost, fs));
        HIPCHK(c, hipStreamSynchronize(fs));
    }
    return POPPY_OK;
}

const void* poppy_hip_frame_device(poppy_hip_ctx* c) { return (c && c->last_slot >= 0) ? c->slots[c->last_slot].out : nullptr; }
void* poppy_hip_frame_stream(poppy_hip_ctx* c) {
    if (!c || c->last_slot < 0) return nullptr;
    const FrameSlot& f = c->slots[c->last_slot];
    return (void*)(f.last_stream ? f.last_stream : c->stream);
}
int poppy_hip_frame_wait(poppy_hip_ctx* c, void* hip_stream) {
    if (!c) return POPPY_E_ARG;
    if (c->last_slot < 0) return fail(c, POPPY_E_STATE, "no frame rendered");
    HIPCHK(c, hipSetDevice(c->device));
    FrameSlot& f = c->slots[c->last_slot];
    if (hip_stream) HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, f.done, 0));
    else HIPCHK(c, hipEventSynchronize(f.done));
    return POPPY_OK;
}

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
    if (phase == 0 || phase == 1) {                            // src/poppy.hpp:54-70: N copies of image 1 / image 2, nothing rendered
        if (!write) return POPPY_OK;
        const uint8_t* img = phase == 0 ? c->c1 : (c->c2_raw_valid ? c->c2_raw : c->c2);
        const size_t row = (size_t)c->W * 3;
        HIPCHK(c, hipMemcpyAsync(c->h_stage, img, row * c->H, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int j = 0; j < N; ++j) write(user, c->h_stage, c->W, c->H, row);
        return POPPY_OK;
    }
    const int n = phase >= 0 ? 1 : N;                          // phase mode: exactly one frame (src/poppy.hpp:234-235)
    std::vector<double> ratio(n);
    for (int j = 0; j < n; ++j) ratio[j] = poppy_frame_ratio(j, N, phase);
    return render_sequence(c, ratio.data(), ratio.data(), n, true, write, user);
}

int poppy_hip_render_phases(poppy_hip_ctx* c, const double* t, int n, poppy_write_cb write, void* user) {
    if (!c || (n > 0 && !t) || n < 0) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t row = (size_t)c->W * 3;
    for (int i = 0; i < n;) {
        if (t[i] == 0 || t[i] == 1) {                             // a plain copy of image 1 / image 2
            if (write) {
                int rc = stage_host(c, row * c->H); if (rc) return rc;
                const uint8_t* img = t[i] == 0 ? c->c1 : (c->c2_raw_valid ? c->c2_raw : c->c2);
                HIPCHK(c, hipMemcpyAsync(c->h_stage, img, row * c->H, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                write(user, c->h_stage, c->W, c->H, row);
            }
            ++i;
            continue;
        }
        int j = i;
        while (j < n && t[j] != 0 && t[j] != 1) ++j;
        { int rc = poppy_hip_pair_reset(c); if (rc) return rc; }
        int rc = render_sequence(c, t + i, t + i, j - i, false, write, user); if (rc) return rc;
        i = j;
    }
    return POPPY_OK;
}

int poppy_printed_morph_distance(const float* p1, const float* p2, int n, int W, int H, double* out) {
    if (n < 1 || !p1 || !p2 || !out || W <= 0 || H <= 0) return POPPY_E_ARG;
    std::vector<P2f> a(n), b(n), u1, u2;
    memcpy(a.data(), p1, (size_t)n * 8); memcpy(b.data(), p2, (size_t)n * 8);
    clip_points_ref(a, W, H); unique_points_ref(a, u1);           // src/poppy.hpp:142-157
    clip_points_ref(b, W, H); unique_points_ref(b, u2);
    if (u1.size() > u2.size()) u1.resize(u2.size()); else u2.resize(u1.size());
    *out = morph_distance_ref(u1, u2, W, H);
    return POPPY_OK;
}

int poppy_hip_pair_distance(poppy_hip_ctx* c, double* out) {
    if (!c || !out) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no pair loaded");
    if (c->pts1_0.empty()) return fail(c, POPPY_E_NOMATCH, "no point pairs");
    return poppy_printed_morph_distance((const float*)c->pts1_0.data(), (const float*)c->pts2.data(), (int)c->pts1_0.size(), c->W, c->H, out);
}

long poppy_hypotf_selfcheck(long n, uint64_t seed) { return hypotf_selfcheck(n, seed); }

int poppy_hip_morph(poppy_hip_ctx* c, const uint8_t* bgr1, size_t s1, const uint8_t* bgr2, size_t s2, int W, int H, double phase,
                    int distance, poppy_write_cb write, void* user, double* morph_distance) {
    if (!c) return POPPY_E_ARG;
    if (!bgr1 || !bgr2 || W <= 0 || H <= 0 || s1 < (size_t)W * 3 || s2 < (size_t)W * 3) return fail(c, POPPY_E_ARG, "bad image arguments");
    const int N = c->cfg.number_of_frames;
    if (phase == 0 || phase == 1) {                            // src/poppy.hpp:54-70, before any feature work
        for (int j = 0; j < N && write; ++j) write(user, phase == 0 ? bgr1 : bgr2, W, H, phase == 0 ? s1 : s2);
        return POPPY_OK;
    }
    int rc = poppy_hip_pair_begin(c, bgr1, s1, bgr2, s2, W, H); if (rc) return rc;
    if (c->pts1_0.empty()) {                                   // :125-134 (see the header: the reference throws before reaching it)
        if (!distance && write) {
            std::vector<uint8_t> blend((size_t)W * 3 * H);
            rc = poppy_hip_dissolve(c, bgr1, s1, bgr2, s2, W, H, phase, blend.data(), (size_t)W * 3); if (rc) return rc;
            for (int j = 0; j < N; ++j) write(user, blend.data(), W, H, (size_t)W * 3);
        }
        return fail(c, POPPY_E_NOMATCH, "no point pairs: linear-blend fallback frames written (src/poppy.hpp:125-134)");
    }
    if (morph_distance || distance) {
        double d = 0;
        rc = poppy_hip_pair_distance(c, &d); if (rc) return rc;
        if (morph_distance) *morph_distance = d;
    }
    if (distance) return POPPY_OK;                             // :159-163 (the reference exits here)
    return poppy_hip_morph_frames(c, phase, write, user);
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
    HIPCHK(c, copy_rows_async(dst, dst_stride, c->slots[0].out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pair_ready = false;
    return POPPY_OK;
}

int poppy_hip_debug_fetch(poppy_hip_ctx* c, const char* name, void* host, size_t bytes) {
    if (!c || !name || !host) return POPPY_E_ARG;
    if (!c->W) return fail(c, POPPY_E_STATE, "no pair loaded");
    const size_t P = (size_t)c->W * c->H;
    const void* src = nullptr; size_t need = 0;
    size_t px_bytes = 0;                          // set for the buffers whose rows may be padded (level 0 of the slot's own images): bytes per pixel
    std::string n(name);
    if (n != "m2" && n != "gabor2" && c->last_slot < 0) return fail(c, POPPY_E_STATE, "no frame rendered yet");
    const FrameSlot& f = c->slots[c->last_slot < 0 ? 0 : c->last_slot];          // intermediates of the last frame
    if (n == "triMap") { src = f.triMap; need = P * 4; }
    else if (n == "trImg1") { src = f.tr1; need = P * 3; px_bytes = 3; }
    else if (n == "trImg2") { src = f.tr2; need = P * 3; px_bytes = 3; }
    else if (n == "lbmask") {
        if (c->lazy_mask) {                    // never materialised by the frame: made here from m2 and the frame's (alpha, beta)
            if (hipStreamSynchronize(f.last_stream ? f.last_stream : c->stream) != hipSuccess) return fail(c, POPPY_E_DEVICE, "sync failed");
            launch_lbmask(c->m2, (const double*)(f.d_blob + kBlobMaskAB), f.pyrM, P, c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(c, POPPY_E_DEVICE, "lbmask kernel failed");
        }
        src = f.pyrM; need = P * 4; px_bytes = 4;
    }
    else if (n == "lapBlend") { src = f.pyrB; need = P * 12; px_bytes = 12; }
    else if (n == "unsharp") {
        if (!c->debug || !f.unsharpF) return fail(c, POPPY_E_STATE, "enable debug before rendering the frame");
        src = f.unsharpF; need = P * 12;
    }
    else if (n == "m2") { src = c->m2; need = P * 4; }
    else if (n == "gabor2") { src = c->gabor2; need = P * 12; }
    else return fail(c, POPPY_E_ARG, "unknown debug buffer");
    if (bytes != need) return fail(c, POPPY_E_ARG, "debug buffer size mismatch");
    { int rc = drain_frames(c); if (rc) return rc; }
    if (px_bytes && c->levels[0].pitch != c->W)
        HIPCHK(c, hipMemcpy2D(host, (size_t)c->W * px_bytes, src, (size_t)c->levels[0].pitch * px_bytes, (size_t)c->W * px_bytes, c->H, hipMemcpyDeviceToHost));
    else HIPCHK(c, hipMemcpy(host, src, need, hipMemcpyDeviceToHost));
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
    if (drain_frames(c) != POPPY_OK) return 0;
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
