// comm.cpp — the multi-GPU half of the C ABI (include/poppy_hip.h, "multi-GPU"): SURVEY.md 8e.
//
// The path shards two ways, and neither needs a collective on the data path:
//   frames of ONE pair   phase-mode frames are independent (src/poppy.hpp:186-200,234-235): every GPU renders a contiguous
//                        sub-range of t_j = j / total.  The only exchange is the pair state — both images, the mask field's
//                        grey complement and the point sets, ONE contiguous allocation (context.h: arena) — which goes from
//                        the GPU that ran the pair set-up to all others in a single ncclBroadcast over RCCL / xGMI.
//   pairs                the pairs loop of the CLI (src/poppy.cpp:266-328) has no cross-pair state: every GPU takes whole
//                        pairs off a shared counter, each a chained sequence on its own context.  No communication at all.
// Two deployment shapes are served by the same primitives:
//   one process per GPU  (bench.py under torch.distributed.run): poppy_hip_comm_id on one rank, the 128 bytes reach the others
//                        by any out-of-band channel, poppy_hip_comm_init everywhere, poppy_hip_pair_broadcast per pair;
//   one process, N GPUs  (a drop-in behind the reference's single-process CLI): poppy_hip_morph_sharded / poppy_hip_morph_pairs
//                        run one host thread and one context per device and create their communicators with ncclCommInitAll.
// librccl is loaded on first use (dlopen): single-GPU callers never touch it, and all entry points come from ONE handle, so a
// second copy of RCCL in the process (PyTorch ships its own) cannot be mixed in by symbol interposition.
#include "context.h"
#include <dlfcn.h>
#include <atomic>
#include <mutex>

namespace {

struct Id128 { char b[128]; };                  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
constexpr int kNcclUint8 = 1, kNcclFloat64 = 8, kNcclMax = 2;       // rccl.h: ncclUint8, ncclFloat64 (ncclDouble), ncclMax

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {getenv("POPPY_HIP_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + s; return p; };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, Id128, int))sym("ncclCommInitRank");
        r.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
        r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclAllReduce");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    });
    return &r;
}

int rccl_fail(poppy_hip_ctx* c, const char* what, int code) {
    Rccl* r = rccl();
    c->err = std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "RCCL error");
    return POPPY_E_DEVICE;
}

}  // namespace

extern "C" {

int poppy_hip_comm_id(uint8_t* id128) {
    if (!id128) return POPPY_E_ARG;
    Rccl* r = rccl();
    if (!r->err.empty()) return POPPY_E_UNSUPPORTED;
    return r->GetUniqueId(id128) == 0 ? POPPY_OK : POPPY_E_DEVICE;
}

int poppy_hip_comm_init(poppy_hip_ctx* c, int rank, int world, const uint8_t* id128) {
    if (!c) return POPPY_E_ARG;
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(c, POPPY_E_ARG, "bad rank / world / id");
    Rccl* r = rccl();
    if (!r->err.empty()) return fail(c, POPPY_E_UNSUPPORTED, r->err.c_str());
    if (c->comm) return fail(c, POPPY_E_STATE, "this context already has a communicator");
    HIPCHK(c, hipSetDevice(c->device));
    Id128 id;
    memcpy(id.b, id128, 128);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, id, rank);
    if (rc != 0) return rccl_fail(c, "ncclCommInitRank", rc);
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return POPPY_OK;
}

int poppy_hip_comm_free(poppy_hip_ctx* c) {
    if (!c) return POPPY_E_ARG;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy(c->comm);
        c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    }
    return POPPY_OK;
}

int poppy_hip_pair_state_bytes(int width, int height, size_t* bytes) {
    if (width <= 0 || height <= 0 || !bytes) return POPPY_E_ARG;
    *bytes = pair_state_bytes(width, height);
    return POPPY_OK;
}

// The resident pair of rank `root` becomes the resident pair of every rank: one ncclBroadcast of the packed pair state.
// Everything that can fail on ONE rank (the root's pair missing or holding more points than the state has room for, an allocation on
// a receiver) happens first, and the ranks then agree on the outcome through a one-value reduction: a rank never returns with an
// error while the others are already blocked inside the broadcast.
int poppy_hip_pair_broadcast(poppy_hip_ctx* c, int root, int W, int H) {
    if (!c) return POPPY_E_ARG;
    if (!c->comm) return fail(c, POPPY_E_STATE, "no communicator (poppy_hip_comm_init)");
    if (root < 0 || root >= c->comm_world || W <= 0 || H <= 0) return fail(c, POPPY_E_ARG, "bad root / geometry");   // the same on every rank
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (c->comm_rank == root) {
        if (!c->pair_ready || c->W != W || c->H != H) rc = fail(c, POPPY_E_STATE, "the root has no resident pair of this geometry");
        else rc = stage_pair_state(c);
    } else {
        rc = alloc_pair(c, W, H);
        if (rc == POPPY_OK) c->pair_ready = false;
    }
    if (c->comm_world > 1) {
        double worst = rc == POPPY_OK ? 0.0 : 1.0;
        const int ra = poppy_hip_comm_max(c, &worst);              // a collective: entered by every rank whatever its own outcome
        if (ra != POPPY_OK) return ra;
        if (rc != POPPY_OK) return rc;
        if (worst != 0.0) return fail(c, POPPY_E_STATE, "another rank could not take part in the broadcast (its own error says why)");
    } else if (rc != POPPY_OK) return rc;
    const int nr = rccl()->Broadcast(c->arena, c->arena, c->arena_bytes, kNcclUint8, root, c->comm, c->stream);
    if (nr != 0) return rccl_fail(c, "ncclBroadcast", nr);
    if (c->comm_rank != root) return adopt_pair_state(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

// max over the ranks of a host double (step timing), through the same communicator
int poppy_hip_comm_max(poppy_hip_ctx* c, double* value) {
    if (!c || !value) return POPPY_E_ARG;
    if (!c->comm) return fail(c, POPPY_E_STATE, "no communicator (poppy_hip_comm_init)");
    HIPCHK(c, hipSetDevice(c->device));
    double* d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d, 8));
    hipError_t e = hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, c->stream);
    int nr = 0;
    if (e == hipSuccess) nr = rccl()->AllReduce(d, d, 1, kNcclFloat64, kNcclMax, c->comm, c->stream);
    if (e == hipSuccess && nr == 0) e = hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (nr != 0) return rccl_fail(c, "ncclAllReduce", nr);
    if (e != hipSuccess) { c->err = std::string("comm_max: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}

// The packed pair state to / from a caller's device buffer (one device copy): for callers that move it with their own
// transport (bench.py falls back to torch.distributed when librccl cannot be initialised a second time in the process).
int poppy_hip_pair_export_device(poppy_hip_ctx* c, void* d_dst, size_t bytes) {
    if (!c || !d_dst) return POPPY_E_ARG;
    if (!c->pair_ready) return fail(c, POPPY_E_STATE, "no resident pair");
    if (bytes < c->arena_bytes) return fail(c, POPPY_E_ARG, "buffer smaller than poppy_hip_pair_state_bytes");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = stage_pair_state(c); if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(d_dst, c->arena, c->arena_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}
int poppy_hip_pair_import_device(poppy_hip_ctx* c, const void* d_src, size_t bytes, int W, int H) {
    if (!c || !d_src || W <= 0 || H <= 0) return POPPY_E_ARG;
    if (bytes < pair_state_bytes(W, H)) return fail(c, POPPY_E_ARG, "buffer smaller than poppy_hip_pair_state_bytes");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = alloc_pair(c, W, H); if (rc) return rc;
    c->pair_ready = false;
    HIPCHK(c, hipMemcpyAsync(c->arena, d_src, c->arena_bytes, hipMemcpyDeviceToDevice, c->stream));
    return adopt_pair_state(c);
}

// ---- one process, several GPUs ---------------------------------------------------------------------------------------------
static void set_err(char* err, size_t n, const std::string& s) { if (err && n) { snprintf(err, n, "%s", s.c_str()); } }

// ONE total_frames-frame phase-mode morph (frame j = morph(img1, img2, ..., phase = j / total_frames) with number_of_frames = 1;
// frame 0 is the phase == 0 copy of image 1) rendered by n_devices GPUs, device k taking the k-th contiguous share.
int poppy_hip_morph_sharded(const int* devices, int n_devices, const poppy_settings* settings, const uint8_t* bgr1, size_t s1,
                            const uint8_t* bgr2, size_t s2, int W, int H, int total_frames, poppy_write_indexed_cb write, void* user,
                            char* err, size_t err_len) {
    if (!devices || n_devices < 1 || n_devices > 64 || !bgr1 || !bgr2 || W <= 0 || H <= 0 || total_frames < 1) { set_err(err, err_len, "bad arguments"); return POPPY_E_ARG; }
    poppy_settings cfg;
    if (settings) cfg = *settings; else poppy_settings_default(&cfg);
    cfg.number_of_frames = 1;
    std::vector<poppy_hip_ctx*> ctx(n_devices, nullptr);
    auto cleanup = [&]() { for (poppy_hip_ctx* c : ctx) if (c) { poppy_hip_comm_free(c); poppy_hip_destroy(c); } };
    for (int k = 0; k < n_devices; ++k) {
        ctx[k] = poppy_hip_create(devices[k], &cfg);
        if (!ctx[k]) { set_err(err, err_len, std::string("poppy_hip_create: ") + poppy_hip_create_error()); cleanup(); return POPPY_E_DEVICE; }
    }
    if (n_devices > 1) {
        Rccl* r = rccl();
        if (!r->err.empty()) { set_err(err, err_len, r->err); cleanup(); return POPPY_E_UNSUPPORTED; }
        std::vector<void*> comms(n_devices, nullptr);
        const int rc = r->CommInitAll(comms.data(), n_devices, devices);
        if (rc != 0) { set_err(err, err_len, std::string("ncclCommInitAll: ") + (r->GetErrorString ? r->GetErrorString(rc) : "")); cleanup(); return POPPY_E_DEVICE; }
        for (int k = 0; k < n_devices; ++k) { ctx[k]->comm = comms[k]; ctx[k]->comm_rank = k; ctx[k]->comm_world = n_devices; }
    }
    // the pair set-up runs before any other device's thread exists: nobody can be left waiting inside RCCL when it fails
    {
        int rc = poppy_hip_pair_begin(ctx[0], bgr1, s1, bgr2, s2, W, H);
        if (rc == POPPY_OK && ctx[0]->pts1_0.empty()) rc = fail(ctx[0], POPPY_E_NOMATCH, "no point pairs");
        if (rc != POPPY_OK) { set_err(err, err_len, std::string("pair set-up: ") + poppy_hip_last_error(ctx[0])); cleanup(); return rc; }
    }
    std::vector<int> rcs(n_devices, POPPY_OK);
    struct Relay { poppy_write_indexed_cb write; void* user; int base; };
    auto work = [&](int k) {
        poppy_hip_ctx* c = ctx[k];
        int rc = POPPY_OK;
        if (n_devices > 1) rc = poppy_hip_pair_broadcast(c, 0, W, H);
        const int lo = (int)((long long)total_frames * k / n_devices), hi = (int)((long long)total_frames * (k + 1) / n_devices);
        if (rc == POPPY_OK && hi > lo) {
            Relay relay{write, user, lo};
            poppy_write_cb cb = write ? +[](void* u, const uint8_t* bgr, int w, int h, size_t stride) {
                Relay* r = (Relay*)u;
                r->write(r->user, r->base++, bgr, w, h, stride);
            } : (poppy_write_cb) nullptr;
            std::vector<double> t(hi - lo);
            for (int j = lo; j < hi; ++j) t[j - lo] = (double)j / (double)total_frames;      // t_0 = 0: a copy of image 1 (src/poppy.hpp:54-62)
            rc = poppy_hip_render_phases(c, t.data(), hi - lo, cb, &relay);
        }
        rcs[k] = rc;
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_devices; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    int rc = POPPY_OK;
    for (int k = 0; k < n_devices; ++k)
        if (rcs[k] != POPPY_OK) { rc = rcs[k]; set_err(err, err_len, "device " + std::to_string(devices[k]) + ": " + poppy_hip_last_error(ctx[k])); break; }
    cleanup();
    return rc;
}

// The pairs loop of the reference's CLI (src/poppy.cpp:266-328) over several GPUs: independent pairs, each one whole
// poppy::morph (default chained mode unless phase says otherwise), handed out one at a time to contexts_per_device host threads
// per GPU — a chained sequence is a latency chain that leaves most of a GPU idle, and the pair set-up of one pair runs beside the
// frames of another.  The pool keeps its contexts (and their HBM) between batches.
struct poppy_hip_pool {
    std::vector<poppy_hip_ctx*> ctx;
    std::vector<int> device_of;
};

poppy_hip_pool* poppy_hip_pool_create(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings,
                                      char* err, size_t err_len) {
    if (!devices || n_devices < 1 || n_devices > 64 || contexts_per_device < 1 || contexts_per_device > 16) { set_err(err, err_len, "bad arguments"); return nullptr; }
    poppy_hip_pool* p = new poppy_hip_pool();
    for (int d = 0; d < n_devices; ++d)
        for (int k = 0; k < contexts_per_device; ++k) {
            poppy_hip_ctx* c = poppy_hip_create(devices[d], settings);
            if (!c) { set_err(err, err_len, std::string("poppy_hip_create: ") + poppy_hip_create_error()); poppy_hip_pool_destroy(p); return nullptr; }
            p->ctx.push_back(c); p->device_of.push_back(devices[d]);
        }
    return p;
}

void poppy_hip_pool_destroy(poppy_hip_pool* p) {
    if (!p) return;
    for (poppy_hip_ctx* c : p->ctx) poppy_hip_destroy(c);
    delete p;
}

int poppy_hip_pool_morph_pairs(poppy_hip_pool* p, int n_pairs, int W, int H, double phase, int inputs_on_device,
                               poppy_pair_source_cb source, poppy_write_pair_cb write, void* user, char* err, size_t err_len) {
    if (!p || n_pairs < 0 || !source || W <= 0 || H <= 0) { set_err(err, err_len, "bad arguments"); return POPPY_E_ARG; }
    // With auto-align the reference's pairs are NOT independent: the aligned second image of one pair is the first image of the next
    // (src/poppy.cpp:326: img1 = corrected2.clone()), which a caller's pair source cannot know in advance.  Such a sequence is a chain:
    // poppy_hip_morph pair after pair on one context, feeding poppy_hip_pair_corrected2 forward (include/poppy_hip_shim.hpp does).
    if (!p->ctx.empty() && p->ctx[0]->cfg.enable_auto_align && n_pairs > 1) {
        set_err(err, err_len, "enable_auto_align chains the pairs (src/poppy.cpp:326): render them in sequence with poppy_hip_morph, not through the pool");
        return POPPY_E_UNSUPPORTED;
    }
    std::atomic<int> next{0}, failed{POPPY_OK};
    std::mutex mu;
    std::string first_err;
    struct Relay { poppy_write_pair_cb write; void* user; int pair; int frame; };
    auto work = [&](int wk) {
        poppy_hip_ctx* c = p->ctx[wk];
        for (;;) {
            const int pi = next.fetch_add(1);
            if (pi >= n_pairs || failed.load() != POPPY_OK) break;
            const uint8_t *a = nullptr, *b = nullptr; size_t sa = 0, sb = 0;
            int rc = source(user, pi, p->device_of[wk], &a, &sa, &b, &sb) == 0 ? POPPY_OK : POPPY_E_ARG;
            if (rc != POPPY_OK) c->err = "the pair source failed";
            Relay relay{write, user, pi, 0};
            poppy_write_cb cb = write ? +[](void* u, const uint8_t* bgr, int w, int h, size_t stride) {
                Relay* r = (Relay*)u;
                r->write(r->user, r->pair, r->frame++, bgr, w, h, stride);
            } : (poppy_write_cb) nullptr;
            if (rc == POPPY_OK && !inputs_on_device) rc = poppy_hip_morph(c, a, sa, b, sb, W, H, phase, 0, cb, &relay, nullptr);
            else if (rc == POPPY_OK) {                              // the same call sequence on images that are already in this GPU's memory
                rc = poppy_hip_pair_begin_device(c, a, b, W, H);
                if (rc == POPPY_OK && c->pts1_0.empty()) rc = fail(c, POPPY_E_UNSUPPORTED, "no point pairs: the fallback needs the images on the host (poppy_hip_morph)");
                if (rc == POPPY_OK) rc = poppy_hip_morph_frames(c, phase, cb, &relay);
            }
            if (rc != POPPY_OK && rc != POPPY_E_NOMATCH) {          // a pair without matches got its fallback frames: not an error of the batch
                std::lock_guard<std::mutex> g(mu);
                if (first_err.empty()) first_err = "pair " + std::to_string(pi) + ": " + poppy_hip_last_error(c);
                failed = rc;
                break;
            }
        }
    };
    std::vector<std::thread> th;
    const int n_workers = std::min((int)p->ctx.size(), std::max(1, n_pairs));
    for (int k = 1; k < n_workers; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    if (failed.load() != POPPY_OK) set_err(err, err_len, first_err);
    return failed.load();
}

int poppy_hip_pool_set_timing(poppy_hip_pool* p, int on) {
    if (!p) return POPPY_E_ARG;
    for (poppy_hip_ctx* c : p->ctx) poppy_hip_set_timing(c, on);
    return POPPY_OK;
}
// per-kernel-group totals summed over the pool's contexts (same contract as poppy_hip_timing_summary)
int poppy_hip_pool_timing_summary(poppy_hip_pool* p, const char** names, float* total_ms, int* launches, int max) {
    if (!p) return 0;
    int n = 0;
    for (poppy_hip_ctx* c : p->ctx) {
        const char* nm[32]; float ms[32]; int cnt[32];
        const int k = poppy_hip_timing_summary(c, nm, ms, cnt, 32);
        for (int i = 0; i < k; ++i) {
            int j = 0;
            while (j < n && strcmp(names[j], nm[i]) != 0) ++j;
            if (j == n) { if (n >= max) continue; names[n] = nm[i]; total_ms[n] = 0.f; launches[n] = 0; ++n; }
            total_ms[j] += ms[i]; launches[j] += cnt[i];
        }
    }
    return n;
}
int poppy_hip_pool_warp_counts(poppy_hip_pool* p, unsigned long long* fused, unsigned long long* tiled, unsigned long long* general) {
    if (!p) return POPPY_E_ARG;
    unsigned long long a = 0, b = 0, f = 0;
    for (poppy_hip_ctx* c : p->ctx) { a += c->n_warp_fast; b += c->n_warp_general; f += c->n_warp_bin; }
    if (fused) *fused = f;
    if (tiled) *tiled = a;
    if (general) *general = b;
    return POPPY_OK;
}

int poppy_hip_pool_mask_rider(poppy_hip_pool* p) {
    if (!p || p->ctx.empty()) return POPPY_E_ARG;
    return poppy_hip_mask_rider(p->ctx[0]);
}

int poppy_hip_morph_pairs(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings, int n_pairs,
                          int W, int H, double phase, poppy_pair_source_cb source, poppy_write_pair_cb write, void* user,
                          char* err, size_t err_len) {
    poppy_hip_pool* p = poppy_hip_pool_create(devices, n_devices, contexts_per_device, settings, err, err_len);
    if (!p) return POPPY_E_DEVICE;
    const int rc = poppy_hip_pool_morph_pairs(p, n_pairs, W, H, phase, 0, source, write, user, err, err_len);
    poppy_hip_pool_destroy(p);
    return rc;
}

void poppy_count_pair_frames_cb(void* user, int, int, const uint8_t*, int, int, size_t) {
    if (user) __atomic_fetch_add((long long*)user, 1ll, __ATOMIC_RELAXED);
}

}  // extern "C"
