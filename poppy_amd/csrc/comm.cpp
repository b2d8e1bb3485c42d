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
#include <condition_variable>
#include <functional>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace {

struct Id128 { char b[128]; };                  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;      // optional: what the communicator itself says its size is (poppy_hip_comm_info)
    int (*CommUserRank)(void*, int*) = nullptr;
    int (*CommAbort)(void*) = nullptr;            // optional: unblocks the other device threads of poppy_hip_morph_sharded when one of them failed
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
        r.CommAbort = (int (*)(void*))dlsym(r.handle, "ncclCommAbort");
        r.CommCount = (int (*)(void*, int*))dlsym(r.handle, "ncclCommCount");
        r.CommUserRank = (int (*)(void*, int*))dlsym(r.handle, "ncclCommUserRank");
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
    if (c->comm.load() || c->comm_aborted.load()) return fail(c, POPPY_E_STATE, "this context already has a communicator (poppy_hip_comm_free first)");
    HIPCHK(c, hipSetDevice(c->device));
    Id128 id;
    memcpy(id.b, id128, 128);
    void* comm = nullptr;
    const int rc = r->CommInitRank(&comm, world, id, rank);
    if (rc != 0) return rccl_fail(c, "ncclCommInitRank", rc);
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    // the small reductions' scratch now, where a failure is this rank's alone: inside a collective sequence an allocation that fails
    // would leave the other ranks waiting in the reduction this rank never enters
    if (!c->d_comm_scratch && hipMalloc((void**)&c->d_comm_scratch, 8 * sizeof(double)) != hipSuccess) {
        (void)r->CommDestroy(comm); c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
        return fail(c, POPPY_E_DEVICE, "allocation of the reduction scratch");
    }
    return POPPY_OK;
}

// what the context believes (rank, world) and what its RCCL communicator reports (ncclCommUserRank, ncclCommCount; -1: no communicator / symbol missing)
int poppy_hip_comm_info(poppy_hip_ctx* c, int* rank, int* world, int* nccl_rank, int* nccl_count) {
    if (!c) return POPPY_E_ARG;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm_world;
    int nr = -1, nc = -1;
    Rccl* r = rccl();
    void* cm = c->comm.load();
    if (cm && r->handle) {
        if (r->CommUserRank && r->CommUserRank(cm, &nr) != 0) nr = -1;
        if (r->CommCount && r->CommCount(cm, &nc) != 0) nc = -1;
    }
    if (nccl_rank) *nccl_rank = nr;
    if (nccl_count) *nccl_count = nc;
    return POPPY_OK;
}

int poppy_hip_comm_free(poppy_hip_ctx* c) {
    if (!c) return POPPY_E_ARG;
    void* cm = c->comm.exchange(nullptr);                           // (an aborted communicator is gone already: its pointer was taken by the abort)
    if (cm) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)rccl()->CommDestroy(cm);
    }
    c->comm_aborted = false; c->comm_rank = 0; c->comm_world = 1;
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
    if (!c->comm.load()) return fail(c, POPPY_E_STATE, c->comm_aborted.load() ? "the communicator was aborted" : "no communicator (poppy_hip_comm_init)");
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
    void* cm = c->comm.load();
    if (!cm) return fail(c, POPPY_E_STATE, "the communicator was aborted");
    const int nr = rccl()->Broadcast(c->arena, c->arena, c->arena_bytes, kNcclUint8, root, cm, c->stream);
    if (nr != 0) return rccl_fail(c, "ncclBroadcast", nr);
    if (c->comm_rank != root) return adopt_pair_state(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return POPPY_OK;
}

// max over the ranks of n <= 8 host doubles (step timing, the set-up's detail values and status flags), through the same communicator;
// the device scratch is allocated once per context (hipMalloc / hipFree per call cost more than the reduction)
static int comm_max_n(poppy_hip_ctx* c, double* values, int n) {
    if (n < 1 || n > 8) return fail(c, POPPY_E_ARG, "comm_max_n: 1..8 values");
    // nothing below returns before the collective has been entered: a rank whose device selection or copy-in failed still takes part (and
    // returns its error afterwards), so that no rank is left alone inside ncclAllReduce
    if (!c->d_comm_scratch) return fail(c, POPPY_E_STATE, "communicator without its scratch (poppy_hip_comm_init allocates it)");
    double* d = c->d_comm_scratch;
    void* cm = c->comm.load();                                      // once: the abort path may take it away at any moment
    if (!cm) return fail(c, POPPY_E_STATE, "the communicator was aborted");   // (nobody is waiting for this rank in an aborted job)
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMemcpyAsync(d, values, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
    const int nr = rccl()->AllReduce(d, d, (size_t)n, kNcclFloat64, kNcclMax, cm, c->stream);   // entered whatever `e` says: see above
    if (e == hipSuccess && nr == 0) e = hipMemcpyAsync(values, d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (nr != 0) return rccl_fail(c, "ncclAllReduce", nr);
    if (e != hipSuccess) { c->err = std::string("comm_max: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}
int poppy_hip_comm_max(poppy_hip_ctx* c, double* value) {
    if (!c || !value) return POPPY_E_ARG;
    if (!c->comm.load()) return fail(c, POPPY_E_STATE, "no communicator (poppy_hip_comm_init)");
    return comm_max_n(c, value, 1);
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

}  // extern "C"

// ---- the pair set-up itself, spread over the ranks --------------------------------------------------------------------------
// The set-up is the serial part of a sharded morph (Amdahl: ~5.5 ms on one GPU against 60 frames x 90 us per rank at N = 8).  Its three
// heavy pieces are independent until the matcher (src/poppy.hpp:52,114-122, src/extractor.cpp:33-83):
//   A  image 1: Extractor::foreground -> dft_detail2 -> the ORB input -> ORB::detect        on rank `root`
//   B  image 2: the same                                                                     on rank root + 1
//   C  gabor_filter(corrected2 / 255) -> m2                                                  on rank root + 2 (with B when there are two ranks)
// The exchanges, all through the same communicator: the raw pair from `root` (one broadcast of the c1 | c2 region), the two detail
// values (one 3-double max-reduction, which also carries an error flag: nfeatures needs both, src/extractor.cpp:40-45), image 2's
// keypoint positions B -> A (one broadcast through the state's point area), then the matcher on A and two broadcasts that complete the
// pair state everywhere: header + points from A, m2 from C.  What can fail on ONE rank (an image, a detection, the matcher) never leaves the
// others inside a collective: it is reported through the reduction, through a count of -1 in the keypoint hand-off or through an invalid
// header that every rank refuses; from the keypoint hand-off on a TRANSPORT error is remembered while the remaining collectives are still
// entered (and nothing it left in the hand-off area is used).  The first two exchanges (the raw pair's broadcast, the reductions) return
// at once on a transport error: every rank sees the failed collective itself there — RCCL fails a collective on all its ranks, the local
// hub raises its abort flag — so none waits in a later one.  One limit the one-GPU set-up does not have: image 2's keypoints travel through the state's point
// area, so more than kPairMaxPoints - 1 (16 383) of them fail the set-up (the one-GPU path limits only the MATCHED pairs); max_keypoints
// x detail stays far below that for every setting the reference's CLI accepts.
// The transport is abstract so that the role logic can run — and be tested bit for bit — with several contexts of ONE process on one
// GPU (LocalHub: device-to-device copies between the contexts' buffers behind a thread barrier).
namespace {

struct Transport {
    int rank = 0, world = 1;
    std::function<int(poppy_hip_ctx*, void*, size_t, int)> bcast;     // in place, device memory, complete on return
    std::function<int(poppy_hip_ctx*, double*, int)> allmax;          // host doubles
};

int chain_image(poppy_hip_ctx* c, int i, bool with_gabor, double* detail, const uint8_t** g_dev, std::string* err) {
    const int W = c->W, H = c->H;
    const size_t P = (size_t)W * H;
    if (hipSetDevice(c->device) != hipSuccess) { *err = "hipSetDevice failed"; return POPPY_E_DEVICE; }
    ForegroundFilter& fg = i ? c->foreground_b : c->foreground;
    hipStream_t st = i ? c->aux_stream : c->stream;
    const uint8_t* gf = fg.run_device(i ? c->c2 : c->c1, (size_t)W * 3, W, H, st, nullptr);
    if (!gf) { *err = "foreground: " + fg.err; return POPPY_E_DEVICE; }
    if (fg.detail(gf, W, H, st, detail)) { *err = "dft_detail2: " + fg.err; return POPPY_E_DEVICE; }
    const uint8_t* gi = fg.orb_input(gf, W, H, 0, st);
    if (!gi) { *err = "orb_input: " + fg.err; return POPPY_E_DEVICE; }
    *g_dev = gi;
    hipError_t e = hipSuccess;
    if (with_gabor) {
        const float* gab = fg.gabor_field(c->c2, W, H, st);
        if (!gab) { *err = "gabor_field: " + fg.err; return POPPY_E_DEVICE; }
        e = hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { *err = std::string("pair set-up: ") + hipGetErrorString(e); return POPPY_E_DEVICE; }
    return POPPY_OK;
}

std::atomic<unsigned long long> g_sharded_setups{0};     // protocol runs in this process (poppy_hip_sharded_setups): lets a test see that the protocol, not a shortcut, ran

int setup_sharded(poppy_hip_ctx* c, Transport& T, const void* d1, const void* d2, int W, int H, int root) {
    g_sharded_setups.fetch_add(1);
    if (c->cfg.enable_auto_align) return fail(c, POPPY_E_UNSUPPORTED, "the sharded set-up does not take auto-align (image 2 changes after the match)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t P = (size_t)W * H;
    const int rA = root, rB = T.world >= 2 ? (root + 1) % T.world : root, rC = T.world >= 3 ? (root + 2) % T.world : rB;
    const bool isA = T.rank == rA, isB = T.rank == rB, isC = T.rank == rC;
    int rc = alloc_pair(c, W, H);
    if (rc == POPPY_OK) { c->pair_ready = false; c->c2_raw_valid = false; }
    if (rc == POPPY_OK && !c->aux_stream && hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(c, POPPY_E_DEVICE, "hipStreamCreate");
    if (rc == POPPY_OK && isA) {
        if (!d1 || !d2) rc = fail(c, POPPY_E_ARG, "the root has no images");
        else if (hipMemcpyAsync(c->c1, d1, P * 3, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                 hipMemcpyAsync(c->c2, d2, P * 3, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                 hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, POPPY_E_DEVICE, "copy of the raw pair");
    }
    double v[3] = {0, 0, rc == POPPY_OK ? 0.0 : 1.0};
    { const int ra = T.allmax(c, v, 3); if (ra) return ra; }
    if (v[2] != 0.0) return rc != POPPY_OK ? rc : fail(c, POPPY_E_STATE, "another rank could not start the set-up");
    // 1. the raw pair to every rank
    rc = T.bcast(c, c->c1, (size_t)((uint8_t*)c->m2 - c->c1), rA); if (rc) return rc;
    // 2. the independent pieces
    double d[2] = {0, 0};
    const uint8_t* g_dev[2] = {nullptr, nullptr};
    std::string errs[2];
    int rcs[2] = {POPPY_OK, POPPY_OK};
    {
        std::thread other;
        if (isB) {
            if (isA) other = std::thread([&]() { rcs[1] = chain_image(c, 1, isC, &d[1], &g_dev[1], &errs[1]); });
            else rcs[1] = chain_image(c, 1, isC, &d[1], &g_dev[1], &errs[1]);
        }
        if (isA) rcs[0] = chain_image(c, 0, false, &d[0], &g_dev[0], &errs[0]);
        if (other.joinable()) other.join();
        if (isC && !isB) {                                              // gabor2 alone
            const float* gab = c->foreground_b.gabor_field(c->c2, W, H, c->stream);
            if (!gab) { rcs[1] = POPPY_E_DEVICE; errs[1] = "gabor_field: " + c->foreground_b.err; }
            else if (hipMemcpyAsync(c->gabor2, gab, P * 12, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { rcs[1] = POPPY_E_DEVICE; errs[1] = "gabor2 copy"; }
        }
        if (isC && rcs[1] == POPPY_OK) {                                // m2 = 1 - gray(gabor2), where the frames read the mask field from
            launch_gray_inv(c->gabor2, c->m2, W * H, c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess) { rcs[1] = POPPY_E_DEVICE; errs[1] = "m2"; }
        }
    }
    rc = rcs[0] ? rcs[0] : rcs[1];
    if (rc) c->err = rcs[0] ? errs[0] : errs[1];
    v[0] = d[0]; v[1] = d[1]; v[2] = rc == POPPY_OK ? 0.0 : 1.0;
    { const int ra = T.allmax(c, v, 3); if (ra) return ra; }
    if (v[2] != 0.0) return rc != POPPY_OK ? rc : fail(c, POPPY_E_STATE, "another rank failed in its part of the set-up");
    const double detail = 255.0 / std::max(v[0], v[1]);                 // src/extractor.cpp:40-45
    c->last_detail[0] = v[0]; c->last_detail[1] = v[1];
    const int nfeatures = (int)(c->cfg.max_keypoints * detail);
    c->last_nfeatures = nfeatures;
    // 3. ORB::detect where the ORB inputs lie
    std::vector<OrbKeyPoint> k1, k2;
    int r1 = 0, r2 = 0;
    std::string root_err;                                  // this rank's own reason, kept apart from c->err (adopt_pair_state overwrites that)
    int bcast_rc = POPPY_OK;                               // a transport error: remembered, the remaining collectives are still entered
    {
        std::thread other;
        if (isB) {
            if (isA) other = std::thread([&]() { r2 = hipSetDevice(c->device) == hipSuccess ? c->orb_b.detect(g_dev[1], W, W, H, nfeatures, c->aux_stream, k2, true) : -2; });
            else r2 = c->orb_b.detect(g_dev[1], W, W, H, nfeatures, c->aux_stream, k2, true);
        }
        if (isA) r1 = c->orb.detect(g_dev[0], W, W, H, nfeatures, c->stream, k1, true);
        if (other.joinable()) other.join();
    }
    // 4. image 2's keypoint positions to A, through the state's point area: [count or -1][x, y pairs]
    uint8_t* const xarea = c->arena + kPairHeadBytes;
    const size_t xbytes = 2 * (size_t)kPairMaxPoints * 8;
    if (isB) {
        std::vector<float> buf(2 + 2 * (size_t)kPairMaxPoints, 0.f);
        int n2 = r2 < 0 || (int)k2.size() > kPairMaxPoints - 1 ? -1 : (int)k2.size();
        memcpy(&buf[0], &n2, 4);
        for (int i = 0; i < n2; ++i) { buf[2 + 2 * i] = k2[i].x; buf[3 + 2 * i] = k2[i].y; }
        if (n2 < 0) root_err = r2 < 0 ? "orb_detect (image 2): " + c->orb_b.err : "image 2 has more keypoints than the hand-off area holds (" + std::to_string(k2.size()) + " > " + std::to_string(kPairMaxPoints - 1) + ")";
        bool sent = hipMemcpyAsync(xarea, buf.data(), (2 + 2 * (size_t)std::max(n2, 0)) * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                    hipStreamSynchronize(c->stream) == hipSuccess;
        if (!sent) {                                       // not a return: A and the other ranks are about to enter the broadcast.  Say "failed" through
            n2 = -1; r2 = -1;                              // the protocol (count -1) if the device still takes a 4-byte write, and go on into the collective
            root_err = "keypoint hand-off to the matcher's rank";
            (void)hipMemcpy(xarea, &n2, 4, hipMemcpyHostToDevice);
        }
    }
    if (rA != rB) { rc = T.bcast(c, xarea, xbytes, rB); if (rc) bcast_rc = rc; }
    // 5. the matcher on A (host), then header + points for everybody
    bool valid = true;
    if (isA) {
        std::vector<float> p2v;
        int n2 = (int)k2.size();
        if (rA != rB) {
            std::vector<float> buf(2 + 2 * (size_t)kPairMaxPoints);
            if (hipMemcpyAsync(buf.data(), xarea, buf.size() * 4 > xbytes ? xbytes : buf.size() * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess) valid = false;
            memcpy(&n2, &buf[0], 4);
            // a hand-off broadcast that failed leaves stale floats in the area: nothing of it may be used (n2 would be an unbounded count)
            if (bcast_rc) { valid = false; root_err = "keypoint hand-off to the matcher's rank failed"; }
            if (n2 > kPairMaxPoints - 1) n2 = -1;
            if (valid && n2 >= 0) p2v.assign(buf.begin() + 2, buf.begin() + 2 + 2 * (size_t)n2);
        } else {
            if (r2 < 0) n2 = -1;
            for (int i = 0; i < n2; ++i) { p2v.push_back(k2[i].x); p2v.push_back(k2[i].y); }
        }
        if (r1 < 0 || n2 < 0) { valid = false; root_err = r1 < 0 ? "orb_detect (image 1): " + c->orb.err : std::string("image 2's rank reported a failed detection or too many keypoints"); }
        if (valid) {
            const size_t n = std::min(k1.size(), (size_t)n2);                   // Extractor::points (extractor.cpp:96-99)
            std::vector<float> p1(n * 2), p2(n * 2), o1((n + 4) * 2), o2((n + 4) * 2);
            for (size_t i = 0; i < n; ++i) { p1[2 * i] = k1[i].x; p1[2 * i + 1] = k1[i].y; p2[2 * i] = p2v[2 * i]; p2[2 * i + 1] = p2v[2 * i + 1]; }
            int m = 0;
            int mr = poppy_match_points(p1.data(), p2.data(), (int)n, W, H, c->cfg.match_tolerance, o1.data(), o2.data(), &m, &c->initial_morph_dist);
            if (mr == POPPY_OK) mr = set_points(c, o1.data(), o2.data(), m);
            if (mr == POPPY_OK) mr = stage_pair_state(c);
            if (mr != POPPY_OK) { valid = false; root_err = "matcher / pair state: " + c->err; }
        }
        if (!valid) {                                                           // a header every rank refuses (adopt_pair_state checks the magic)
            (void)hipMemsetAsync(c->arena, 0, kPairHeadBytes, c->stream);
            (void)hipStreamSynchronize(c->stream);
        }
    }
    // both final broadcasts are entered by every rank whatever happened before (a transport error on one rank must not leave the others inside the next one)
    rc = T.bcast(c, c->arena, kPairHeadBytes + xbytes, rA); if (rc && !bcast_rc) bcast_rc = rc;
    rc = T.bcast(c, c->m2, P * 4, rC); if (rc && !bcast_rc) bcast_rc = rc;
    if (bcast_rc) return bcast_rc;
    rc = adopt_pair_state(c);
    if (rc != POPPY_OK && !root_err.empty()) return fail(c, POPPY_E_DEVICE, ("sharded set-up: " + root_err).c_str());
    return rc;
}

// several contexts of one process (any devices): broadcasts are device copies from the root context's buffer behind a barrier
struct LocalHub {
    int n;
    std::mutex mu; std::condition_variable cv; int arrived = 0; unsigned phase = 0;
    const void* src = nullptr; int src_device = 0;
    std::vector<double> vals;
    bool aborted = false;
    explicit LocalHub(int n_) : n(n_) {}
    // false = a context left the protocol with an error (abort()): whoever waits here returns an error too instead of waiting for ever
    bool barrier() {
        std::unique_lock<std::mutex> g(mu);
        if (aborted) return false;
        const unsigned ph = phase;
        if (++arrived == n) { arrived = 0; ++phase; cv.notify_all(); }
        else cv.wait(g, [&] { return phase != ph || aborted; });
        return !aborted;
    }
    void abort() { { std::lock_guard<std::mutex> g(mu); aborted = true; } cv.notify_all(); }
};

}  // namespace

extern "C" {

unsigned long long poppy_hip_sharded_setups(void) { return g_sharded_setups.load(); }

int poppy_hip_pair_begin_sharded(poppy_hip_ctx* c, const void* d1, const void* d2, int W, int H, int root) {
    if (!c) return POPPY_E_ARG;
    if (!c->comm.load()) return fail(c, POPPY_E_STATE, c->comm_aborted.load() ? "the communicator was aborted" : "no communicator (poppy_hip_comm_init)");
    if (root < 0 || root >= c->comm_world || W <= 0 || H <= 0) return fail(c, POPPY_E_ARG, "bad root / geometry");
    // a world of one needs no exchange — unless POPPY_HIP_SHARD_WORLD1 asks for the protocol anyway: all three roles on this rank, every
    // broadcast and reduction a real RCCL call on the one-rank communicator (what a box with a single GPU can run of the multi-rank path)
    static const bool world1_protocol = getenv("POPPY_HIP_SHARD_WORLD1") != nullptr;
    if (c->comm_world == 1 && !world1_protocol) return poppy_hip_pair_begin_device(c, d1, d2, W, H);
    Transport T;
    T.rank = c->comm_rank; T.world = c->comm_world;
    T.bcast = [](poppy_hip_ctx* cc, void* buf, size_t bytes, int r) -> int {
        void* cm = cc->comm.load();
        if (!cm) return fail(cc, POPPY_E_STATE, "the communicator was aborted");
        const int nr = rccl()->Broadcast(buf, buf, bytes, kNcclUint8, r, cm, cc->stream);
        if (nr != 0) return rccl_fail(cc, "ncclBroadcast", nr);
        if (hipStreamSynchronize(cc->stream) != hipSuccess) return fail(cc, POPPY_E_DEVICE, "broadcast");
        return POPPY_OK;
    };
    T.allmax = [](poppy_hip_ctx* cc, double* v, int n) -> int { return comm_max_n(cc, v, n); };
    return setup_sharded(c, T, d1, d2, W, H, root);
}

// The same set-up by n contexts of THIS process (one host thread each; the contexts may share a GPU): context k plays rank k, the raw
// pair (device pointers valid for context `root`'s device) ends up resident in every context.  What the multi-rank path does, minus RCCL.
int poppy_hip_pair_begin_sharded_local(poppy_hip_ctx** ctxs, int n, const void* d1, const void* d2, int W, int H, int root) {
    if (!ctxs || n < 1 || n > 64 || root < 0 || root >= n || W <= 0 || H <= 0) return POPPY_E_ARG;
    for (int k = 0; k < n; ++k) if (!ctxs[k]) return POPPY_E_ARG;
    if (n == 1) return poppy_hip_pair_begin_device(ctxs[0], d1, d2, W, H);
    LocalHub hub(n);
    hub.vals.assign(64, 0.0);
    std::vector<int> rcs(n, POPPY_OK);
    auto work = [&](int k) {
        Transport T;
        T.rank = k; T.world = n;
        T.bcast = [&hub, k](poppy_hip_ctx* cc, void* buf, size_t bytes, int r) -> int {
            if (k == r) { hub.src = buf; hub.src_device = cc->device; }
            if (!hub.barrier()) return fail(cc, POPPY_E_STATE, "another context left the sharded set-up with an error");
            hipError_t e = hipSuccess;
            if (k != r) {
                e = hipMemcpyAsync(buf, hub.src, bytes, hipMemcpyDeviceToDevice, cc->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(cc->stream);
            }
            if (!hub.barrier()) return fail(cc, POPPY_E_STATE, "another context left the sharded set-up with an error");   // the root's buffer may change again only now
            return e == hipSuccess ? POPPY_OK : fail(cc, POPPY_E_DEVICE, "local broadcast");
        };
        T.allmax = [&hub, k](poppy_hip_ctx* cc, double* v, int m) -> int {
            bool ok = hub.barrier();
            if (ok && k == 0) std::fill(hub.vals.begin(), hub.vals.end(), -1e300);
            ok = ok && hub.barrier();
            if (ok) { std::lock_guard<std::mutex> g(hub.mu); for (int i = 0; i < m; ++i) hub.vals[i] = std::max(hub.vals[i], v[i]); }
            ok = ok && hub.barrier();
            if (!ok) return fail(cc, POPPY_E_STATE, "another context left the sharded set-up with an error");
            for (int i = 0; i < m; ++i) v[i] = hub.vals[i];
            return POPPY_OK;
        };
        rcs[k] = setup_sharded(ctxs[k], T, k == root ? d1 : nullptr, k == root ? d2 : nullptr, W, H, root);
        if (rcs[k] != POPPY_OK) hub.abort();                               // nobody waits for a context that has returned
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n; ++k) th.emplace_back(work, k);
    work(0);
    for (auto& t : th) t.join();
    for (int k = 0; k < n; ++k) if (rcs[k] != POPPY_OK && rcs[k] != POPPY_E_STATE) return rcs[k];      // the cause before its echoes
    for (int k = 0; k < n; ++k) if (rcs[k] != POPPY_OK) return rcs[k];
    return POPPY_OK;
}

}  // extern "C"

extern "C" {

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
        for (int k = 0; k < n_devices; ++k)               // the reductions' scratch before any thread can be inside a collective
            if (hipSetDevice(devices[k]) != hipSuccess || hipMalloc((void**)&ctx[k]->d_comm_scratch, 8 * sizeof(double)) != hipSuccess) {
                set_err(err, err_len, "allocation of the reduction scratch"); cleanup(); return POPPY_E_DEVICE;
            }
    }
    // Several devices: the set-up runs on device 0 before any other device's thread exists (nobody can be left waiting inside RCCL when it
    // fails) and the pair state is broadcast.  POPPY_HIP_SHARD_SETUP=1 spreads the set-up itself over the devices instead
    // (poppy_hip_pair_begin_sharded: image 1 on device 0, image 2 on device 1, the mask field on device 2; not under auto-align): opt-in
    // until its RCCL transport has run on a node with three or more GPUs (round-3 advisor finding; the role logic is tested through the
    // in-process transport, the transport through a world of one).
    static const bool shard_on = getenv("POPPY_HIP_SHARD_SETUP") && atoi(getenv("POPPY_HIP_SHARD_SETUP")) != 0;
    const bool shard_setup = n_devices > 1 && !cfg.enable_auto_align && shard_on;
    uint8_t* d_raw = nullptr;
    const size_t P3 = (size_t)W * H * 3;
    if (shard_setup) {
        hipError_t e = hipSetDevice(devices[0]);
        if (e == hipSuccess) e = hipMalloc((void**)&d_raw, 2 * P3);
        if (e == hipSuccess) e = hipMemcpy2D(d_raw, (size_t)W * 3, bgr1, s1, (size_t)W * 3, H, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy2D(d_raw + P3, (size_t)W * 3, bgr2, s2, (size_t)W * 3, H, hipMemcpyHostToDevice);
        if (e != hipSuccess) { set_err(err, err_len, std::string("upload of the raw pair: ") + hipGetErrorString(e)); if (d_raw) (void)hipFree(d_raw); cleanup(); return POPPY_E_DEVICE; }
    } else {
        int rc = poppy_hip_pair_begin(ctx[0], bgr1, s1, bgr2, s2, W, H);
        if (rc == POPPY_OK && ctx[0]->pts1_0.empty()) rc = fail(ctx[0], POPPY_E_NOMATCH, "no point pairs");
        if (rc != POPPY_OK) { set_err(err, err_len, std::string("pair set-up: ") + poppy_hip_last_error(ctx[0])); cleanup(); return rc; }
    }
    std::vector<int> rcs(n_devices, POPPY_OK);
    struct Relay { poppy_write_indexed_cb write; void* user; int base; };
    // a device thread that fails before or inside the collectives must not leave the others waiting in RCCL for ever: it aborts every
    // communicator of the job once (ncclCommAbort makes pending and later operations on it return an error)
    std::once_flag abort_once;
    auto abort_all = [&]() {
        std::call_once(abort_once, [&]() {
            Rccl* r = rccl();
            if (n_devices > 1 && r->CommAbort)
                // (ncclCommAbort FREES the communicator: the pointer is taken out of the context first, so a device thread that enters a collective
                // later finds null and fails with POPPY_E_STATE instead of handing RCCL a freed handle; one that is inside a collective already
                // is what ncclCommAbort exists to unblock; the flag keeps the context from being given a new communicator before comm_free)
                for (int k = 0; k < n_devices; ++k) {
                    ctx[k]->comm_aborted = true;
                    void* cm = ctx[k]->comm.exchange(nullptr);
                    if (cm) (void)r->CommAbort(cm);
                }
        });
    };
    std::atomic<int> past_setup{0};
    auto work = [&](int k) {
        poppy_hip_ctx* c = ctx[k];
        int rc = POPPY_OK;
        if (shard_setup) {
            rc = poppy_hip_pair_begin_sharded(c, k == 0 ? d_raw : nullptr, k == 0 ? d_raw + P3 : nullptr, W, H, 0);
            if (rc == POPPY_OK && c->pts1_0.empty()) rc = fail(c, POPPY_E_NOMATCH, "no point pairs");      // every rank sees the same (empty) point sets
        } else if (n_devices > 1) rc = poppy_hip_pair_broadcast(c, 0, W, H);
        // (POPPY_E_NOMATCH is every device's outcome at once, after its last collective: nothing to unblock)
        if (rc != POPPY_OK && rc != POPPY_E_NOMATCH && past_setup.load() < n_devices) abort_all();      // the exchanges are over once every device is past this point
        past_setup.fetch_add(1);
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
    if (d_raw) { (void)hipSetDevice(devices[0]); (void)hipFree(d_raw); }
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
    // The set-up gate (round 6): at most `setup_gate` contexts of a device run a pair set-up at a time (0: no gate).  A set-up takes the whole GPU whatever runs
    // beside it; contexts that all start one at once — a batch of as many pairs as contexts — then all render at once, and the copy link idles for the length of
    // the set-up round; set-ups side by side also slow each other (each takes the whole GPU).  One at a time: a pair's frames stream out while the next pair sets up.
    int setup_gate = 0, contexts_per_device = 1;
    std::mutex gate_mu;
    std::condition_variable gate_cv;
    std::map<int, int> setups_running;           // per device
    // Batches handed in without waiting for them (poppy_hip_pool_submit_pairs / poppy_hip_pool_wait): one persistent feeder thread per context takes pairs off the
    // queue's first batch, so a batch's last pairs render beside the next batch's first set-ups — the contexts never start a round of set-ups together.
    struct Batch {
        int n_pairs = 0, W = 0, H = 0, inputs_on_device = 0, taken = 0;
        double phase = -1.0;
        poppy_pair_source_cb source = nullptr; poppy_write_pair_cb write = nullptr; void* user = nullptr;
    };
    std::mutex q_mu;
    std::condition_variable q_cv, q_idle;
    std::deque<std::shared_ptr<Batch>> queue;    // batches with pairs nobody has taken yet
    long long outstanding = 0;                   // pairs submitted and not finished
    int async_rc = POPPY_OK;                     // the first failure since the last poppy_hip_pool_wait (the pairs behind it are dropped)
    std::string async_err;
    std::vector<std::thread> feeders;
    bool quit = false;
};

static void pool_setup_hook(void* user, poppy_hip_ctx* c, int begin);

poppy_hip_pool* poppy_hip_pool_create(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings,
                                      char* err, size_t err_len) {
    if (!devices || n_devices < 1 || n_devices > 64 || contexts_per_device < 1 || contexts_per_device > 16) { set_err(err, err_len, "bad arguments"); return nullptr; }
    poppy_hip_pool* p = new poppy_hip_pool();
    p->contexts_per_device = contexts_per_device;
    {   // POPPY_POOL_SETUPS: set-ups side by side per device (0 = as many as contexts).  Default from three contexts on: ONE — six contexts, 36 pairs in one call 7.0-7.5k -> 7.8k
        // frames/s, four contexts 7.0-7.2k -> 7.4k, the bench's batches of six pairs 6.3-6.6k -> 6.7k; two or three at a time: in between (profiles/r06_gate.txt)
        static const int forced = getenv("POPPY_POOL_SETUPS") ? atoi(getenv("POPPY_POOL_SETUPS")) : -1;
        p->setup_gate = forced >= 0 ? forced : (contexts_per_device >= 3 ? 1 : 0);
        if (p->setup_gate >= contexts_per_device) p->setup_gate = 0;
    }
    for (int d = 0; d < n_devices; ++d)
        for (int k = 0; k < contexts_per_device; ++k) {
            poppy_hip_ctx* c = poppy_hip_create(devices[d], settings);
            if (!c) { set_err(err, err_len, std::string("poppy_hip_create: ") + poppy_hip_create_error()); poppy_hip_pool_destroy(p); return nullptr; }
            // (pair_setup.cpp: from three contexts on the chains of a pair run one after the other — six chains on four hardware queues were a lottery, profiles/r05_notes.md
            // section 6; still the better form behind the set-up gate: profiles/r06_gate.txt; POPPY_POOL_CHAINS=0 / 1 forces side by side / serial)
            static const int chains_env = getenv("POPPY_POOL_CHAINS") ? atoi(getenv("POPPY_POOL_CHAINS")) : -1;
            c->setup_serial = chains_env >= 0 ? chains_env != 0 : contexts_per_device >= 3;
            if (p->setup_gate > 0) { c->setup_hook = pool_setup_hook; c->setup_hook_user = p; }
            p->ctx.push_back(c); p->device_of.push_back(devices[d]);
        }
    return p;
}

void poppy_hip_pool_destroy(poppy_hip_pool* p) {
    if (!p) return;
    if (!p->feeders.empty()) {                                     // batches still queued are rendered first (their writers expect every frame)
        {
            std::unique_lock<std::mutex> lk(p->q_mu);
            p->q_idle.wait(lk, [&] { return p->outstanding == 0; });
            p->quit = true;
        }
        p->q_cv.notify_all();
        for (std::thread& t : p->feeders) t.join();
    }
    for (poppy_hip_ctx* c : p->ctx) poppy_hip_destroy(c);
    delete p;
}

// The set-up gate: every pair set-up of a pool's contexts — from host images (poppy_hip_morph) or from resident ones (poppy_hip_pair_begin_device) — passes through here
// (context.h: setup_hook, called by pair_setup.cpp at the set-up's beginning and end)
static void pool_setup_hook(void* user, poppy_hip_ctx* c, int begin) {
    poppy_hip_pool* p = static_cast<poppy_hip_pool*>(user);
    const int dev = c->device;
    if (begin) {
        std::unique_lock<std::mutex> g(p->gate_mu);
        p->gate_cv.wait(g, [&] { return p->setups_running[dev] < p->setup_gate; });
        ++p->setups_running[dev];
    } else {
        { std::lock_guard<std::mutex> g(p->gate_mu); --p->setups_running[dev]; }
        p->gate_cv.notify_all();
    }
}

// one pair of a batch on context wk: the pair source, the set-up (behind the pool's gate), the frames
static int pool_render_pair(poppy_hip_pool* p, int wk, int pi, int W, int H, double phase, int inputs_on_device,
                            poppy_pair_source_cb source, poppy_write_pair_cb write, void* user) {
    poppy_hip_ctx* c = p->ctx[wk];
    struct Relay { poppy_write_pair_cb write; void* user; int pair; int frame; };
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
        rc = poppy_hip_pair_begin_device(c, a, b, W, H);           // (behind the pool's set-up gate: pool_setup_hook)
        if (rc == POPPY_OK && c->pts1_0.empty()) rc = fail(c, POPPY_E_UNSUPPORTED, "no point pairs: the fallback needs the images on the host (poppy_hip_morph)");
        if (rc == POPPY_OK) rc = poppy_hip_morph_frames(c, phase, cb, &relay);
    }
    return rc;
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
    auto work = [&](int wk) {
        poppy_hip_ctx* c = p->ctx[wk];
        for (;;) {
            const int pi = next.fetch_add(1);
            if (pi >= n_pairs || failed.load() != POPPY_OK) break;
            const int rc = pool_render_pair(p, wk, pi, W, H, phase, inputs_on_device, source, write, user);
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

// The asynchronous form of poppy_hip_pool_morph_pairs: the batch is queued and the call returns; poppy_hip_pool_wait returns when every pair submitted so far has
// been rendered and handed to its writer.  Batches are taken up in submission order, pair by pair, by whichever context is free — a batch's last pairs run beside the
// next batch's first set-ups.  `source`, `write` and `user` must stay valid until the wait; they are called from the pool's threads (as in the synchronous form).
// After a failure the pairs still queued are dropped and the wait reports the first error.  Not to be mixed with a poppy_hip_pool_morph_pairs call in flight.
static void pool_feeder(poppy_hip_pool* p, int wk) {
    for (;;) {
        std::shared_ptr<poppy_hip_pool::Batch> b;
        int pi = 0;
        bool drop = false;
        {
            std::unique_lock<std::mutex> lk(p->q_mu);
            p->q_cv.wait(lk, [&] { return p->quit || !p->queue.empty(); });
            if (p->queue.empty()) return;                           // (quit)
            b = p->queue.front();
            pi = b->taken++;
            if (b->taken >= b->n_pairs) p->queue.pop_front();
            drop = p->async_rc != POPPY_OK;
        }
        int rc = POPPY_OK;
        std::string thrown;
        if (!drop) {
            // (an exception on a pool thread — std::bad_alloc is the one that can happen — must reach the waiter as a status: nobody else would decrement `outstanding`)
            try { rc = pool_render_pair(p, wk, pi, b->W, b->H, b->phase, b->inputs_on_device, b->source, b->write, b->user); }
            catch (const std::exception& e) { rc = POPPY_E_DEVICE; thrown = e.what(); }
            catch (...) { rc = POPPY_E_DEVICE; thrown = "unknown exception"; }
        }
        {
            std::lock_guard<std::mutex> lk(p->q_mu);
            if (rc != POPPY_OK && rc != POPPY_E_NOMATCH && p->async_rc == POPPY_OK) {
                p->async_rc = rc;
                p->async_err = "pair " + std::to_string(pi) + ": " + (thrown.empty() ? std::string(poppy_hip_last_error(p->ctx[wk])) : "exception on a pool thread: " + thrown);
            }
            if (--p->outstanding == 0) p->q_idle.notify_all();
        }
    }
}

int poppy_hip_pool_submit_pairs(poppy_hip_pool* p, int n_pairs, int W, int H, double phase, int inputs_on_device,
                                poppy_pair_source_cb source, poppy_write_pair_cb write, void* user) {
    if (!p || n_pairs < 0 || !source || W <= 0 || H <= 0) return POPPY_E_ARG;
    if (!p->ctx.empty() && p->ctx[0]->cfg.enable_auto_align && n_pairs > 1) return POPPY_E_UNSUPPORTED;     // (see poppy_hip_pool_morph_pairs)
    if (n_pairs == 0) return POPPY_OK;
    auto b = std::make_shared<poppy_hip_pool::Batch>();
    b->n_pairs = n_pairs; b->W = W; b->H = H; b->phase = phase; b->inputs_on_device = inputs_on_device;
    b->source = source; b->write = write; b->user = user;
    {
        std::lock_guard<std::mutex> lk(p->q_mu);
        if (p->feeders.empty())
            for (int k = 0; k < (int)p->ctx.size(); ++k) p->feeders.emplace_back(pool_feeder, p, k);
        p->outstanding += n_pairs;
        p->queue.push_back(std::move(b));
    }
    p->q_cv.notify_all();
    return POPPY_OK;
}

int poppy_hip_pool_wait(poppy_hip_pool* p, char* err, size_t err_len) {
    if (!p) { set_err(err, err_len, "bad arguments"); return POPPY_E_ARG; }
    std::unique_lock<std::mutex> lk(p->q_mu);
    p->q_idle.wait(lk, [&] { return p->outstanding == 0; });
    const int rc = p->async_rc;
    if (rc != POPPY_OK) set_err(err, err_len, p->async_err);
    p->async_rc = POPPY_OK;
    p->async_err.clear();
    return rc;
}

// A pool whose contexts came out well.  The contexts of a pool get their streams, hardware queues and buffers from the runtime when they are made, and about
// one pool in ten then runs every batch 10 - 25 % slower for as long as it lives.  What a service does about that once, at start-up, is done
// here: up to max_candidates pools are made, each renders a small calibration batch (a built-in synthetic pair of the given geometry, two pairs per context, the
// frames handed to a counting writer) once untimed and twice timed; the fastest pool is returned, the others are destroyed.  Two pools that agree within 4 %
// end the search — that is the normal state.  candidates_ms (may be NULL, room for max_candidates) receives the timed batch time of every pool made, in the
// order they were made; *n_made their number; *kept the index of the one returned.
namespace {
void calibration_pair(int W, int H, std::vector<uint8_t>& a, std::vector<uint8_t>& b) {
    // flat rectangles on grey, the second image's shifted by a few pixels: corners for the detector, plateaus like the bench's own content
    a.assign((size_t)W * H * 3, 96); b = a;
    auto hash = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    for (uint32_t k = 0; k < 40; ++k) {
        const int w = 24 + (int)(hash(k * 7 + 1) % (uint32_t)std::max(8, W / 10)), h = 24 + (int)(hash(k * 7 + 2) % (uint32_t)std::max(8, H / 10));
        const int x0 = (int)(hash(k * 7 + 3) % (uint32_t)std::max(1, W - w - 16)), y0 = (int)(hash(k * 7 + 4) % (uint32_t)std::max(1, H - h - 16));
        const int dx = 2 + (int)(hash(k * 7 + 5) % 7), dy = 2 + (int)(hash(k * 7 + 6) % 7);
        const uint32_t col = hash(k * 7);
        for (int im = 0; im < 2; ++im) {
            std::vector<uint8_t>& img = im ? b : a;
            const int ox = x0 + (im ? dx : 0), oy = y0 + (im ? dy : 0);
            for (int y = oy; y < std::min(oy + h, H); ++y)
                for (int x = ox; x < std::min(ox + w, W); ++x) {
                    uint8_t* px = &img[((size_t)y * W + x) * 3];
                    px[0] = (uint8_t)col; px[1] = (uint8_t)(col >> 8); px[2] = (uint8_t)(col >> 16);
                }
        }
    }
}
}  // namespace

poppy_hip_pool* poppy_hip_pool_create_tuned(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings, int W, int H,
                                            int max_candidates, float* candidates_ms, int* n_made, int* kept, char* err, size_t err_len) {
    if (n_made) *n_made = 0;
    if (kept) *kept = -1;
    if (W <= 0 || H <= 0 || max_candidates < 1 || max_candidates > 8) { set_err(err, err_len, "bad arguments"); return nullptr; }
    std::vector<uint8_t> a, b;
    calibration_pair(W, H, a, b);
    struct Src { const uint8_t *a, *b; size_t stride; long long frames; } src{a.data(), b.data(), (size_t)W * 3, 0};
    auto source = +[](void* u, int, int, const uint8_t** p1, size_t* s1, const uint8_t** p2, size_t* s2) { Src* s = (Src*)u; *p1 = s->a; *p2 = s->b; *s1 = *s2 = s->stride; return 0; };
    auto count = +[](void* u, int, int, const uint8_t*, int, int, size_t) { __atomic_fetch_add(&((Src*)u)->frames, 1ll, __ATOMIC_RELAXED); };
    std::vector<std::pair<poppy_hip_pool*, double>> made;
    auto drop_all = [&]() { for (auto& m : made) poppy_hip_pool_destroy(m.first); };
    for (int k = 0; k < max_candidates; ++k) {
        poppy_hip_pool* p = poppy_hip_pool_create(devices, n_devices, contexts_per_device, settings, err, err_len);
        if (!p) { drop_all(); return nullptr; }
        const int batch = 2 * (int)p->ctx.size();
        int rc = poppy_hip_pool_morph_pairs(p, batch, W, H, -1.0, 0, source, count, &src, err, err_len);      // allocates
        const auto t0 = std::chrono::steady_clock::now();
        for (int q = 0; q < 2 && rc == POPPY_OK; ++q) rc = poppy_hip_pool_morph_pairs(p, batch, W, H, -1.0, 0, source, count, &src, err, err_len);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 2;
        if (rc != POPPY_OK) { poppy_hip_pool_destroy(p); drop_all(); return nullptr; }
        made.emplace_back(p, ms);
        if (candidates_ms) candidates_ms[k] = (float)ms;
        if (made.size() >= 2) {
            std::vector<double> t;
            for (auto& m : made) t.push_back(m.second);
            std::sort(t.begin(), t.end());
            if (t[0] * 1.04 >= t[1]) break;
        }
    }
    size_t best = 0;
    for (size_t k = 1; k < made.size(); ++k) if (made[k].second < made[best].second) best = k;
    for (size_t k = 0; k < made.size(); ++k) if (k != best) poppy_hip_pool_destroy(made[k].first);
    if (n_made) *n_made = (int)made.size();
    if (kept) *kept = (int)best;
    return made[best].first;
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
