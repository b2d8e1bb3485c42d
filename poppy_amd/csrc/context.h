// context.h — the library's context (one per GPU) and the helpers shared by its translation units:
//   poppy_hip.cpp   context life cycle, HBM layout, the per-frame path and the resident-pair API
//   pair_setup.cpp  everything that happens once per pair: pre-ORB chain, ORB, matching, auto-align, margins
#pragma once
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
#include <mutex>
#include <string>
#include <thread>
#include <vector>


using namespace poppy_hip;

constexpr int kWarpStampStride = 7;

struct FrameSlot {
    hipStream_t stream = nullptr;        // the stream the slot's independent (phase-mode) frames last ran on: its own, or one of the context's three
    hipStream_t own_stream = nullptr;    // the slot's own (created at its first frame that stays in HBM)
    hipStream_t last_stream = nullptr;   // the stream the frame last rendered here ran on (the context's for chained frames, `stream` otherwise)
    hipEvent_t done = nullptr;           // end of the frame last rendered here
    hipEvent_t prepared = nullptr;       // id map and mask of the frame being rendered here are written
    uint8_t *tr1 = nullptr, *tr2 = nullptr, *out = nullptr;
    float *pyrL = nullptr, *pyrR = nullptr, *pyrM = nullptr, *pyrB = nullptr;
    float *tmp = nullptr, *diff = nullptr;      // only for 1-pixel-wide / -high images (separate unsharp passes)
    float* unsharpF = nullptr;                  // debug copy of the float unsharp result, allocated on demand
    int32_t* triMap = nullptr;
    int map_tag = 0;                            // frame tag of the values last written to triMap (kernels.h: launch_raster); 0 = must be zeroed first
    uint8_t *h_blob = nullptr, *d_blob = nullptr;   // this slot's frame plan (pinned host copy, device copy)
    uint8_t* tile_data = nullptr;                   // the plan's raster expanded on the device: id bytes, record slots, overflow records (kernels_warp_bin.hip)
    void* h_blob_dev = nullptr;                     // device-side address of the pinned copy
    hipEvent_t uploaded = nullptr;                  // the device copy is complete
    hipGraphExec_t body = nullptr;                  // pyrdown .. unsharp of this slot, captured once per pair geometry
    hipEvent_t downloaded = nullptr;                // completes when the last download of this slot's `out` towards the writer has read it
    bool dl_pending = false;                        // ... and whether such a download was issued since the slot was last rendered into
    int dl_ring_idx = -1;                           // POPPY_HIP_DL_STREAMS: the ring stream that carries that download
};
constexpr size_t kBlobHeader = 64;            // [0] float: unsharp amount; [16], [24] double: the frame's mask (alpha, beta)
constexpr size_t kBlobMaskAB = 16;

struct poppy_hip_ctx {
    int device = 0;
    poppy_settings cfg;
    hipStream_t stream = nullptr;
    std::string err;

    int W = 0, H = 0;
    bool pair_ready = false;
    // resident buffers
    // c1, c2 and m2 live in ONE allocation behind a small header + point area (PairStateHeader): the "pair state" that a
    // frame needs.  Being contiguous, it travels to other GPUs as a single ncclBroadcast / a single device copy (comm.cpp).
    uint8_t* arena = nullptr; size_t arena_bytes = 0;
    uint8_t *c1 = nullptr, *c2 = nullptr;
    uint8_t* c2_raw = nullptr;           // image 2 before auto-align (only allocated when auto-align ran): what phase == 1 writes
    bool c2_raw_valid = false;           // ... and whether it belongs to the resident pair
    float *gabor2 = nullptr, *m2 = nullptr;
    std::vector<FrameSlot> slots;        // per-frame working sets, used round-robin
    void* slot_prep_store = nullptr;     // per slot: the frame prepared there (poppy_hip.cpp: SlotPrep)
    unsigned long long frame_seq = 0;    // submit_frame calls so far (a slot prepared ahead names the call it is for)
    void* seq_plans = nullptr;           // the plans of a multi-frame call in the making (poppy_hip.cpp: SeqPlans), possibly started ahead by a pair loader
    // called at the beginning (1) and at the end (0) of every pair set-up from raw images (pair_setup.cpp: pair_begin_impl): a pool's set-up gate (comm.cpp)
    void (*setup_hook)(void* user, poppy_hip_ctx* c, int begin) = nullptr;
    void* setup_hook_user = nullptr;
    bool plan_ahead_credit = true;       // pair loaders start the default sequence's plans (false after a pair whose plans nobody took, until a multi-frame call comes again)
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
    void* d_levels = nullptr;            // the tail kernel's tap descriptors (kernels.h: PyrTailPlan), written once per pair geometry
    PyrTailPlan tail;                    // ... and its step table (a kernel argument)
    int first_tail = 1;
    bool use_tail = true;                // false: the coarsest level is too large for one workgroup's LDS (shallow --pyramid): per-level kernels all the way
    // points
    std::vector<P2f> pts1_0, pts1, pts2;
    // per-frame plan blobs (pinned host + device) live in the frame slots, so the host can plan ahead of the GPU
    int max_tris = 0;
    // blob layout: [header 64 B: f32 unsharp amount][warp records (T+1)*20 f32][tri_xy T*6 i32][inv1 T*9 f32][inv2 T*9 f32]
    //              [RasterTri T][work 2*n i32]
    size_t blob_bytes = 0; size_t bins_cap = 0;       // bins_cap: most per-tile triangle-list entries a plan blob has room for
    size_t tile_bytes = 0;                            // size of every slot's tile_data (kernels.h: warp_bin_data_bytes)
    hipStream_t copy_stream = nullptr;
    FramePlan plan;
    OrbDetector orb, orb_b;
    Worker setup_worker;                            // the second image's half of a pair set-up (chain, detector)
    Team planners;                                  // the frame planners of multi-frame calls
    bool writer_attached = false;                   // a multi-frame call with a writer is in progress
    double wait_ms[4] = {0, 0, 0, 0};               // host waits inside submit_frame since the context was made (POPPY_SEQ_TIMING prints the per-sequence share)
    ForegroundFilter foreground, foreground_b;      // two instances: the images of a pair are filtered side by side
    hipStream_t aux_stream = nullptr;
    hipEvent_t setup_ev = nullptr;                  // "the second image's medians are through" (pair set-up: gabor2 starts there)
    hipEvent_t c2_up_ev = nullptr;                  // "the second host image is in c2" (recorded on aux_stream by the second chain's thread; gabor2 on copy_stream waits for it)
    double initial_morph_dist = 0;
    int last_nfeatures = 0;
    AutoAligner aligner;
    uint8_t* d_align = nullptr; size_t d_align_bytes = 0;      // staging image of the host-facing align entry points
    unsigned long long n_warp_fast = 0, n_warp_general = 0, n_warp_bin = 0;   // frames by warp kernel since create: tiled (id map), general, fused raster
    // the arguments of the last fused raster + warp launch (poppy_hip_time_last_warp relaunches it)
    struct LastWarp { const float* rec = nullptr; const void* tile_data = nullptr; size_t tile_bytes = 0; const int* toff = nullptr; int tile_w = 0;
                      const uint8_t* c1 = nullptr; const uint8_t* c2 = nullptr; uint8_t* tr1 = nullptr; uint8_t* tr2 = nullptr; WarpExtras ex; bool valid = false; } last_warp;
    int last_descriptor_matches = 0;               // symmetric matches kept by the last pair_begin_descriptors
    // RCCL communicator of this context (comm.cpp), or null.  Atomic: morph_sharded's abort path takes the pointer AWAY (exchange to null) before
    // ncclCommAbort frees the communicator, and every collective wrapper loads it once — a late entrant finds null (POPPY_E_STATE), never a freed handle
    std::atomic<void*> comm{nullptr}; int comm_rank = 0, comm_world = 1;
    std::atomic<bool> comm_aborted{false};                          // the communicator was aborted under this context: its collectives fail with POPPY_E_STATE until poppy_hip_comm_free
    double* d_comm_scratch = nullptr;                               // 8 doubles for the small reductions (comm.cpp), allocated on first use
    unsigned warp_seq = 0;                      // warp launches issued in timing mode 2 (every kWarpStampStride-th is stamped)
    bool last_warp_fast = false;                   // which warp kernel the last submitted frame used
    bool last_warp_bin = false;
    double last_detail[2] = {0, 0};
    // diagnostics
    bool debug = false;
    bool lazy_mask = false;                     // the level-0 blend kernels compute lbmask from m2; the warp kernel does not write it
    int timing = 0;                      // 0 off, 1 every kernel group (direct launches), 2 the warp kernel only
    struct Mark { const char* name; hipEvent_t ev; };   // name == nullptr opens a frame
    std::vector<Mark> marks; size_t marks_used = 0;
    // staging for host-image entry points
    uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0; void* h_stage_dev = nullptr;
    static const int kStageRing = 8;          // most pinned frames in flight towards the writer (POPPY_HIP_RING, default 3)
    hipStream_t dl_stream = nullptr;
    bool setup_serial = false;                  // pair set-up: the two images' chains one after the other (set by pools of >= 3 contexts per device and by poppy_hip_set_setup_chains)
    hipEvent_t dl_done[kStageRing] = {};
    hipStream_t dl_ring[kStageRing] = {};     // POPPY_HIP_DL_STREAMS: a stream per pinned ring buffer, carrying nothing but that buffer's copies (no event is recorded behind a copy)
};

// ---- packed pair state (see poppy_hip_ctx::arena) -------------------------------------------------------------------------
constexpr size_t kPairHeadBytes = 4096;
constexpr int kPairMaxPoints = 16384;              // point pairs the packed state has room for
struct PairStateHeader {
    uint32_t magic, version;
    int32_t W, H, n_points, nfeatures;
    double initial_morph_dist, detail[2];
};
static_assert(sizeof(PairStateHeader) <= kPairHeadBytes, "header area too small");
constexpr uint32_t kPairMagic = 0x50505931u;        // "PPY1"
inline size_t pair_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline size_t pair_state_bytes(int W, int H) {
    const size_t P = (size_t)W * H;
    return kPairHeadBytes + 2 * (size_t)kPairMaxPoints * 8 + 2 * pair_align(P * 3 + 16) + pair_align(P * 4);
}
int drain_frames(poppy_hip_ctx* c);                // waits for every frame queued on this context (all streams)
int stage_pair_state(poppy_hip_ctx* c);            // header + points -> arena head (queued on c->stream)
int adopt_pair_state(poppy_hip_ctx* c);            // arena head -> points, chain state; the pair becomes ready

#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return POPPY_E_DEVICE;                                                                    \
        }                                                                                             \
    } while (0)

inline int fail(poppy_hip_ctx* c, int code, const char* msg) { c->err = msg; return code; }

// shared between the translation units (defined in poppy_hip.cpp)
int alloc_pair(poppy_hip_ctx* c, int W, int H);                                   // resident buffers + frame slots for a W x H pair
int upload_image(poppy_hip_ctx* c, uint8_t* dst, const uint8_t* src, size_t stride, int W, int H);
int set_points(poppy_hip_ctx* c, const float* p1, const float* p2, int n);
int finish_pair_load(poppy_hip_ctx* c);                                           // m2 from gabor2, chain state reset, pair_ready
