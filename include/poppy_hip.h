/* poppy_hip.h — C ABI of the MI355X-native morph hot path (libpoppy_hip.so).
 *
 * Drop-in boundary for kallaballa/Poppy's feature-match -> dense-warp -> blend path.  Every entry
 * point names the reference interface it replaces (paths relative to the reference tree; OCV =
 * third/opencv-4.6.0/modules).  Plain pointers and sizes only; no C++ or torch types cross this line.
 *
 * Conventions
 *   - images are 8-bit BGR, row stride in BYTES given explicitly (cv::Mat::step); float images are
 *     tightly packed; point sets are float pairs (cv::Point2f).
 *   - every function returns POPPY_OK (0) or a negative poppy_status; poppy_hip_last_error(ctx)
 *     returns the message.  Nothing here calls exit() or throws across the boundary (the reference
 *     does: src/poppy.hpp:162,229).
 *   - one ctx per GPU, used from one host thread at a time; different ctx are independent.
 *   - there is NO CPU fallback: without a usable gfx950 device poppy_hip_create() fails.
 *
 * Environment variables (read once per process).  NONE of those the shipped library reads changes a result bit — they choose
 * between forms that the tests hold to the same bytes (tests/test_gpu_prefilter.py::test_setup_alternative_forms_stay_exact,
 * test_gpu_bstage.py::test_unsharp_kernel_forms_on_every_geometry, ::test_pyramid_launch_forms_stay_exact) or print timings:
 *   frames    POPPY_HIP_SLOTS (frames in flight, 4), POPPY_HIP_RING (pinned frames towards the writer, 3), POPPY_HIP_NOGRAPH,
 *             POPPY_HIP_NOFUSE (one launch per small pyramid level), POPPY_HIP_NOCONE (the way up in pairs of levels, as round 5),
 *             POPPY_TAIL_PX, POPPY_TILE_W (64 / 128), POPPY_HIP_IDMAP, POPPY_HIP_GENERALWARP, POPPY_HIP_LBMASK_RIDER,
 *             POPPY_HIP_DL_DEVWAIT, POPPY_HIP_DL_EVENTS (one download stream + an event per frame copy, as until round 5), POPPY_HIP_DONE_PACKET, POPPY_HIP_NO_PREPARE_AHEAD, POPPY_PHASE_OWN_STREAMS, POPPY_UNSHARP_STREAM / _TILE / _ROWS,
 *             POPPY_HIP_WARP_STAMP_STRIDE, POPPY_SEQ_TIMING (stderr)
 *   set-up    POPPY_SETUP_SERIAL, POPPY_SETUP_UPLOAD_BOTH, POPPY_GABOR2_FIRST / _LATE / _AT, POPPY_GABOR_DIRECT, POPPY_ACC_STEPS,
 *             POPPY_MED_SETS / _WAVES, POPPY_MED_COLS_MIN / _MIN_HARD / _FORCE / _ROWS, POPPY_ORB_GUESS / _CAP / _KPCAP (test
 *             forms: short lists that must grow), POPPY_SETUP_TIMING (stderr)
 *   several GPUs  POPPY_HIP_RCCL (path of librccl), POPPY_HIP_SHARD_SETUP, POPPY_HIP_SHARD_WORLD1, POPPY_POOL_SETUPS (pair set-ups side by side per device in a pool: 1 from three contexts on), POPPY_POOL_CHAINS
 * Measurement switches that DO change results (parts of a kernel left out, the Gabor transform without its exactness hand-over:
 * POPPY_MED_COLS_SKIP, POPPY_GABOR_NO_REDO, POPPY_GABOR_BAND, POPPY_DL_SKIP_COPY) and the launch-by-launch A/B of wave priorities
 * (POPPY_STAGGER_AB) exist only in a build made with -DPOPPY_EXPERIMENTS (`python -m poppy_amd.build --experiments` writes
 * libpoppy_hip_experiments.so beside the shipped library, never in its place); the shipped library does not read them.
 */
#ifndef POPPY_HIP_H_
#define POPPY_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct poppy_hip_ctx poppy_hip_ctx;

typedef enum {
    POPPY_OK = 0,
    POPPY_E_ARG = -1,        /* bad argument (size mismatch, null pointer, ...)                       */
    POPPY_E_DEVICE = -2,     /* HIP error / no gfx950 device                                           */
    POPPY_E_RANGE = -3,      /* a point fell outside the image where the reference throws StsOutOfRange
                                (OCV/imgproc/src/subdivision2d.cpp:287)                                */
    POPPY_E_STATE = -4,      /* call order violated (e.g. render before pair_begin)                    */
    POPPY_E_NOMATCH = -5,    /* no point pairs: caller should use poppy_hip_dissolve (src/poppy.hpp:125) */
    POPPY_E_UNSUPPORTED = -6 /* stage outside this round's scope (fails loudly, never falls back)      */
} poppy_status;

/* Mirror of poppy::Settings (src/settings.hpp:14-27), passed by value instead of a global singleton. */
typedef struct {
    int number_of_frames;    /* 60  */
    double match_tolerance;  /* 1.0 */
    int max_keypoints;       /* 300 */
    int pyramid_levels;      /* 64  */
    int enable_radial_mask;  /* 0: Extractor::foreground multiplies its mask with draw_radial_gradiant's (src/extractor.cpp:178-197; pinned by restatement only) */
    int enable_auto_align;   /* 0: Matcher::autoAlign on corrected2 and its points before matching (src/matcher.cpp:29-32) */
} poppy_settings;

void poppy_settings_default(poppy_settings* s);

/* ---- lifetime ------------------------------------------------------------------------------ */
poppy_hip_ctx* poppy_hip_create(int device, const poppy_settings* settings);     /* replaces poppy::init, src/poppy.hpp:30-44 */
void poppy_hip_destroy(poppy_hip_ctx* ctx);
const char* poppy_hip_last_error(const poppy_hip_ctx* ctx);
const char* poppy_hip_create_error(void);          /* message of the last failed poppy_hip_create() */

/* ---- per-frame operator (inner boundary) ----------------------------------------------------
 * poppy::morph_images(img1,img2,corrected1,corrected2,gabor2,gf1,gf2,dst,last,morphedPoints,
 *                     srcPoints1,srcPoints2,shapeRatio,maskRatio,linear)   src/algo.hpp:26, src/algo.cpp:178-273
 * Host buffers in, host buffers out, synchronous: this is the parity-test entry point.
 * morphed_pts (n x 2) may be NULL.                                                               */
int poppy_hip_morph_images(poppy_hip_ctx* ctx,
                           const uint8_t* corrected1, size_t stride1,
                           const uint8_t* corrected2, size_t stride2,
                           const float* gabor2_f32x3, int width, int height,
                           const float* src_points1, const float* src_points2, int n_points,
                           double shape_ratio, double mask_ratio,
                           uint8_t* dst, size_t dst_stride, float* morphed_pts);

/* ---- resident pair API (throughput path; frames stay in HBM) ---------------------------------
 * pair_load : uploads corrected1/2 + gabor2 and the matched point sets once per pair
 *             (state that poppy::morph keeps in locals, src/poppy.hpp:114-157).
 * render    : one morph_images step on the resident pair.  chain != 0 reproduces the default CLI
 *             loop body (src/poppy.hpp:177-219): srcPoints1 <- morphedPoints and corrected1 <- frame.
 *             dst may be NULL (frame stays on the device; poppy_hip_frame_device() exposes it).      */
int poppy_hip_pair_load(poppy_hip_ctx* ctx,
                        const uint8_t* corrected1, size_t stride1,
                        const uint8_t* corrected2, size_t stride2,
                        const float* gabor2_f32x3, int width, int height,
                        const float* src_points1, const float* src_points2, int n_points);
/* same, but the three images are already device pointers (tight rows) of this ctx's GPU */
int poppy_hip_pair_load_device(poppy_hip_ctx* ctx, const void* d_corrected1, const void* d_corrected2,
                               const void* d_gabor2_f32x3, int width, int height,
                               const float* src_points1, const float* src_points2, int n_points);
int poppy_hip_render(poppy_hip_ctx* ctx, double shape_ratio, double mask_ratio, int chain,
                     uint8_t* dst, size_t dst_stride);
int poppy_hip_pair_reset(poppy_hip_ctx* ctx);                 /* back to the state right after pair_load */
const void* poppy_hip_frame_device(poppy_hip_ctx* ctx);       /* device pointer of the last frame (u8x3, tight); see poppy_hip_frame_wait */
int poppy_hip_sync(poppy_hip_ctx* ctx);                       /* wait for all queued work of this ctx (every frame stream included) */
void* poppy_hip_stream(poppy_hip_ctx* ctx);                   /* the hipStream_t the pair set-up and CHAINED frames run on */
/* Independent frames (chain == 0, phase mode), several in flight, do not run in the order of poppy_hip_stream(): frames that STAY in HBM
 * (no writer) run on per-slot streams of their own — work queued on poppy_hip_stream() is NOT ordered behind them —; a sequence with a
 * WRITER attached (poppy_hip_render_many / _phases / _morph with a callback) takes the context's own three compute streams in turn, slot
 * index mod 3 — poppy_hip_stream(), the plan-upload stream and the set-up's second stream: the three hardware queues that carry no frame
 * copies —, so work a caller queues on poppy_hip_stream() DURING such a sequence is serialised behind every third frame
 * (POPPY_PHASE_OWN_STREAMS=1: a stream per slot in both cases, the form before round 3).  Either way a caller that renders with
 * dst == NULL and consumes poppy_hip_frame_device() on the device makes its own stream wait for the frame with poppy_hip_frame_wait (a
 * hipStreamWaitEvent on the frame's completion event; stream == NULL: the host waits), or calls poppy_hip_sync().
 * poppy_hip_frame_stream: the stream the last frame was rendered on.  Every pair loader drains all of them first. */
int poppy_hip_frame_wait(poppy_hip_ctx* ctx, void* hip_stream);
void* poppy_hip_frame_stream(poppy_hip_ctx* ctx);

/* Reference frame scheduler (src/poppy.hpp:181-210): shape (= color) ratio of frame j.
 * phase < 0 : default chained mode; 0 <= phase < 1 : phase mode (one frame).                          */
double poppy_frame_ratio(int j, int number_of_frames, double phase);

/* Whole frame loop of poppy::morph on the resident pair (src/poppy.hpp:177-235).  Each finished
 * frame is handed to `write` (the Twriter::write(cv::Mat&) of the reference); the pointer is valid
 * only during the call.  write may be NULL (frames stay on the device: benchmark mode).              */
typedef void (*poppy_write_cb)(void* user, const uint8_t* bgr, int width, int height, size_t stride);
int poppy_hip_morph_frames(poppy_hip_ctx* ctx, double phase, poppy_write_cb write, void* user);
/* phase == 0 / phase == 1 follow the reference's short-circuit (src/poppy.hpp:54-70): number_of_frames copies of image 1 /
 * image 2 as they were handed to the pair set-up (with auto-align: the UNALIGNED image 2), no frame is rendered.            */

/* The whole of poppy::morph(img1, img2, corrected1, corrected2, phase, distance, output) (src/poppy.hpp:46-248) in one call,
 * in the reference's order: phase == 0 / 1 short-circuits (:54-70, before any feature work) -> pair set-up from the raw
 * images (poppy_hip_pair_begin) -> the printed morph distance (:142-159) -> frame loop with the reference's scheduler
 * (:177-235; one frame when 0 < phase < 1).  distance != 0 reproduces --distance: the value is computed, no frame is
 * written and the call returns POPPY_OK (the reference prints it and calls exit(0), :159-163; the shim does that).
 * *morph_distance (may be NULL) receives the printed value whenever the set-up ran.
 * No point pairs: poppy::morph means to write number_of_frames frames of img2*phase + img1*(1-phase) (:125-134) — with the
 * phase ARGUMENT, i.e. -1 in the default mode — but the reference never gets there: with empty point lists Matcher::find ->
 * morph_distance -> cv::convexHull throws first (tried with the golden generator).  This call writes the intended
 * fallback frames and returns POPPY_E_NOMATCH so that a caller can tell; corrected2 comes from poppy_hip_pair_corrected2. */
int poppy_hip_morph(poppy_hip_ctx* ctx, const uint8_t* bgr1, size_t stride1, const uint8_t* bgr2, size_t stride2,
                    int width, int height, double phase, int distance, poppy_write_cb write, void* user,
                    double* morph_distance);
/* The value poppy::morph prints as "morph distance" for the resident pair (src/poppy.hpp:142-159: clip_points, make_uniq,
 * truncate to the shorter list, morph_distance) — what --distance reports.  Host arithmetic, computed on demand.          */
int poppy_hip_pair_distance(poppy_hip_ctx* ctx, double* morph_distance);
/* the same value from explicit point lists (host only, no ctx): points as Matcher::prepare leaves them */
int poppy_printed_morph_distance(const float* points1, const float* points2, int n_points, int width, int height, double* morph_distance);

/* ---- once-per-pair stage (outer boundary: poppy::morph, src/poppy.hpp:46-157) -------------------
 * ORB::create(nfeatures)->detect(gray, keypoints)  (src/extractor.cpp:45,77-78).  gray is an 8-bit single
 * channel host image.  kps7 receives max_kps x 7 floats per keypoint in cv::KeyPoint field order
 * (x, y, size, angle, response, octave, class_id); the ORDER equals the reference's.                   */
int poppy_hip_orb_detect(poppy_hip_ctx* ctx, const uint8_t* gray, size_t stride, int width, int height,
                         int nfeatures, float* kps7, int max_kps, int* n_kps);

/* North-star kernels WITHOUT a call site in Poppy (it never computes descriptors); pinned against
 * OpenCV directly:  ORB::compute (WTA_K 2, 32 bytes/keypoint; OCV/features2d/src/orb.cpp:219-285,1148-1216) and
 * BFMatcher(NORM_HAMMING).match (OCV/features2d/src/matchers.cpp:757, OCV/core/src/batch_distance.cpp:199-262;
 * rows of out3 = queryIdx, trainIdx, distance; lowest train index wins ties).                            */
int poppy_hip_orb_describe(poppy_hip_ctx* ctx, const uint8_t* gray, size_t stride, int width, int height,
                           const float* kps7, int n_kps, uint8_t* descriptors32);
int poppy_hip_hamming_match(poppy_hip_ctx* ctx, const uint8_t* query32, int n_query, const uint8_t* train32, int n_train,
                            int* out3, int* n_matches);

/* Descriptor matching as the reference sketched it (src/experiments.hpp:14-144, dead code there):
 *   hamming_knn2   : BFMatcher(NORM_HAMMING).knnMatch(k = 2) — rows of out4 = trainIdx0, distance0, trainIdx1, distance1,
 *                    ordered by (distance, train index) (OCV/core/src/batch_distance.cpp:225-248), -1 where there is none;
 *   ratio_symmetry : ratioTest (a query survives with two neighbours and distance0 / distance1 <= ratio) on both
 *                    directions, then symmetryTest; rows of out3 = queryIdx, trainIdx, distance.  Host only.
 * ransacTest (findFundamentalMat) of the same sketch is not provided.                                          */
int poppy_hip_hamming_knn2(poppy_hip_ctx* ctx, const uint8_t* query32, int n_query, const uint8_t* train32, int n_train, int* out4);
int poppy_ratio_symmetry(const int* knn12, int n1, const int* knn21, int n2, float ratio, int* out3, int* n_out);

/* ---- auto-align (Settings::enable_auto_align) ---------------------------------------------------
 * warp_affine : cv::warpAffine(src, dst, M, src.size()) for 8UC3, INTER_LINEAR, BORDER_CONSTANT 0, M = 2x3 forward map
 *               (OCV/imgproc/src/imgwarp.cpp:2155-2290,2582-2640) — what Transformer::translate / rotate and
 *               reprocrustes apply to corrected2 (src/transformer.cpp:21-30,268).
 * auto_align  : Matcher::autoAlign (src/matcher.cpp:133-244) on a host image and two point lists: image2 and points2 are
 *               replaced by their aligned versions; *distance = morph distance afterwards.
 * align_step  : one Transformer step on the same arguments — 0 retranslate, 1 reprocrustes, 2 rerotate
 *               (src/transformer.cpp:99-217,260-269); *distance = the value the step returns.
 * procrustes / perspective_from4 (host only): Procrustes(true,false)::procrustes (src/procrustes.cpp:52-114; rotation 2x2,
 *               scale_error[2], yprime n x 2) and cv::getPerspectiveTransform of four point pairs (3x3 doubles).
 * poppy_hip_pair_begin runs auto_align itself when settings.enable_auto_align is set.                              */
int poppy_hip_warp_affine(poppy_hip_ctx* ctx, const uint8_t* src, size_t src_stride, int width, int height, const double* m2x3,
                          uint8_t* dst, size_t dst_stride);
int poppy_hip_auto_align(poppy_hip_ctx* ctx, uint8_t* image2, size_t stride, int width, int height,
                         const float* points1, float* points2, int n_points, double* distance);
int poppy_hip_align_step(poppy_hip_ctx* ctx, int which, uint8_t* image2, size_t stride, int width, int height,
                         const float* points1, float* points2, int n_points, double* distance);
int poppy_procrustes(const float* x, const float* y, int n_points, float* rotation4, float* scale_error2, float* yprime);
int poppy_perspective_from4(const float* src4, const float* dst4, double* m3x3);

/* Matcher::find (general branch) + Matcher::prepare on raw point lists (src/matcher.cpp:118-131,246-332):
 * drop out-of-image pairs, morph distance, greedy nearest-neighbour pairing, threshold filter, 4 corners.
 * Host only.  out1/out2 need room for n_points + 4 pairs.                                              */
/* Self-check of the matcher's hypotf (point_match.cpp): compares its closed form with the libm hypotf the library is linked
 * against on n pseudo-random float pairs; returns the number of mismatches (0 expected).  Host only.                    */
long poppy_hypotf_selfcheck(long n, uint64_t seed);
int poppy_match_points(const float* points1, const float* points2, int n_points, int width, int height,
                       double match_tolerance, float* out1, float* out2, int* n_out, double* initial_morph_distance);

/* Extractor::foreground for ONE image (src/extractor.cpp:136-229; the reference runs it on both images of a pair,
 * three times per pair: poppy.hpp:52,116 and matcher.cpp:17): grey -> 13 x MOG2 on progressively median-blurred copies
 * (k = 1, 9, ..., 89), accumulated and Gaussian-smoothed -> log mask -> x grey -> equalizeHist.  Returns the 8-bit
 * `goodFeatures` image (width*height bytes).  Bit-exact with the reference.  First part of the pre-ORB filter chain
 * (SURVEY 8f-1); poppy_hip_orb_input / poppy_hip_gabor_field are the rest, poppy_hip_pair_begin runs all of it.
 * debug (optional, host pointers, each may be NULL): grey w*h; stages 50 planes of w*h in the order flow0, acc0, then
 * 12 x (med, flow, acc, blur); floats 3 planes of w*h: lin, logged, finalMask; masked w*h.                      */
typedef struct poppy_foreground_debug {
    uint8_t* grey;
    uint8_t* stages;
    float* floats;
    uint8_t* masked;
} poppy_foreground_debug;
int poppy_hip_foreground(poppy_hip_ctx* ctx, const uint8_t* bgr, size_t stride, int width, int height,
                         uint8_t* good_features, const poppy_foreground_debug* debug);
/* One link of that chain on its own: cv::medianBlur(src, dst, ksize) on a tight 8-bit single-channel host image, odd 3 <= ksize <= 89
 * (src/extractor.cpp:149; OCV/imgproc/src/median_blur.simd.hpp:84-346).  form selects the kernel (diagnostics, tests): 0 = what the chain
 * takes for this ksize, 1 = a lane per image column (k_median_u8), 2 = column histograms over the ranks of the values each tile's
 * footprint holds (k_median_cols: one count per lane up to 64 values, two up to 129, windows of 128 ranks beyond), 3 = k_median_cols
 * without the presence maps (every tile by windows over all 256 values), 4 = as 2 without the one-count form, 5 = as 2 with every tile
 * by windows, 6 / 7 = as 5 with the first window at the top / bottom of the ranks (the windows beside it do the work).  All forms return
 * the same bytes.                                                                                                                     */
int poppy_hip_median_blur(poppy_hip_ctx* ctx, const uint8_t* src, int width, int height, int ksize, int form, uint8_t* dst);

/* Pair set-up from the two ORB input images g1/g2 (what Extractor::keypoints feeds the detector,
 * src/extractor.cpp:50-78) and gabor2 (src/poppy.hpp:119-122): ORB x2 -> truncate to the shorter list
 * (extractor.cpp:96-99) -> poppy_match_points -> resident pair.  nfeatures = int(max_keypoints * detail).  */
int poppy_hip_pair_begin_prefiltered(poppy_hip_ctx* ctx,
                                     const uint8_t* bgr1, size_t stride1, const uint8_t* bgr2, size_t stride2,
                                     const uint8_t* orb_input1, const uint8_t* orb_input2, const float* gabor2_f32x3,
                                     int width, int height, int nfeatures);
/* Same from the raw BGR pair (the set-up half of poppy::morph, src/poppy.hpp:46-157 with face detection and auto-align
 * off): Extractor::foreground x2 -> dft_detail2 x2 -> nfeatures = int(max_keypoints * 255 / max(d1, d2)) -> unsharp(sigma 2),
 * grey, 31x31 Gabor bank, radial gradient, equalizeHist -> ORB x2 -> matcher -> gabor_filter(image2 / 255) -> resident pair.
 * Parity: bit-exact with the reference end to end (tests/test_gpu_prefilter2.py reproduces the point pairs and frames of
 * the real poppy::morph).  dft_detail2 restates cv::dft operation for operation; the Gabor banks produce the once-rounded
 * exact sums that OpenCV's double-precision DFT correlation yields (DESIGN.md section 7 has the fine print).          */
int poppy_hip_pair_begin(poppy_hip_ctx* ctx, const uint8_t* bgr1, size_t stride1, const uint8_t* bgr2, size_t stride2,
                         int width, int height);
/* poppy_hip_pair_begin with the two raw images already in this GPU's memory (tight rows, width*3 bytes each): the throughput
 * path of a caller whose decoder or previous stage left the images in HBM, and what bench.py times.                      */
int poppy_hip_pair_begin_device(poppy_hip_ctx* ctx, const void* d_bgr1, const void* d_bgr2, int width, int height);
/* A poppy_write_cb that only counts: ++*(long long*)user.  For callers (and the benchmark) that want the frame hand-off —
 * every frame downloaded into pinned host memory and presented to the writer — without a consumer of their own.           */
void poppy_count_frames_cb(void* user, const uint8_t* bgr, int width, int height, size_t stride);
/* nfeatures and the two dft_detail2 values of the last poppy_hip_pair_begin */
/* Opt-in quality mode with no counterpart in the reference's live code: pair set-up as poppy_hip_pair_begin, but the point
 * pairs come from ORB descriptors (ORB::compute on both keypoint sets, 2-NN both ways, ratio test, symmetry test) instead of
 * the positional greedy matcher; surviving pairs in query order, out-of-image pairs dropped, four corners appended.
 * POPPY_E_NOMATCH when nothing survives (use poppy_hip_dissolve).  ratio: 0.7 in the reference's sketch.        */
int poppy_hip_pair_begin_descriptors(poppy_hip_ctx* ctx, const uint8_t* bgr1, size_t stride1, const uint8_t* bgr2, size_t stride2,
                                     int width, int height, float ratio);

/* The second image as the resident pair holds it — after poppy_hip_pair_begin with enable_auto_align this is the ALIGNED
 * corrected2 that poppy::morph hands back to its caller (src/poppy.hpp:46-47; src/poppy.cpp:326 chains it into the next pair). */
int poppy_hip_pair_corrected2(poppy_hip_ctx* ctx, uint8_t* dst, size_t dst_stride);

int poppy_hip_pair_begin_info(poppy_hip_ctx* ctx, int* nfeatures, double* detail2);
/* Pieces of the chain, host in / host out, for tests and for callers that cache intermediates:
 * poppy_hip_orb_input: goodFeatures (w*h) -> g = the ORB input image; optional us (grey of the unsharp-masked image), gb (Gabor
 * mean), detail (dft_detail2).  poppy_hip_gabor_field: gabor_filter(bgr / 255) with the default arguments -> f32x3.
 * poppy_radial_gradient: draw_radial_gradiant2 (src/draw.cpp:40-59), host only.  poppy_radial_mask: draw_radial_gradiant + the
 * conversion to float of src/extractor.cpp:181-183 (src/draw.cpp:21-38): the mask enable_radial_mask multiplies in, host only.      */
int poppy_hip_orb_input(poppy_hip_ctx* ctx, const uint8_t* good_features, int width, int height, uint8_t* g, float* us, float* gb, double* detail);
int poppy_hip_gabor_field(poppy_hip_ctx* ctx, const uint8_t* bgr, size_t stride, int width, int height, float* gabor);
/* The two Gabor banks (src/util.cpp:31-61: filter2D per angle -> OCV/imgproc/src/templmatch.cpp:566-760, double-precision DFT
 * correlation) run as tiled double-precision FFTs by default; on != 0 selects the direct double sums instead.  The two forms give the
 * same planes in every bit: the transforms are ~1e-15 from the direct sums before the one rounding to float, and the FFT form hands
 * every pixel with a plane value that close to a float rounding boundary (or to zero) to the direct sums (the tests compare the two).
 * poppy_hip_gabor_doubt (diagnostic): since the last call on the current device, out[0] plane values near zero, out[1] near a
 * midpoint, out[2] pixels formed again because of them. */
int poppy_hip_set_gabor_direct(poppy_hip_ctx* ctx, int on);
int poppy_hip_gabor_doubt(unsigned long long out[3]);
/* How poppy_hip_pair_begin* runs the two images' filter chains (Extractor::foreground -> dft_detail2 -> ORB input -> detector, src/extractor.cpp:33-83, once per image):
 * serial = 0 (default of a context) side by side on two streams and two host threads — the shortest set-up when the context has the GPU to itself —, serial != 0 one
 * after the other.  The contexts of a pool with three or more contexts per device are created with serial = 1: their set-ups run side by side anyway, and six chains on
 * the process's four hardware queues made the pool's speed a lottery.  Both orders leave the same pair state in every bit.                                          */
int poppy_hip_set_setup_chains(poppy_hip_ctx* ctx, int serial);
int poppy_radial_gradient(int width, int height, float* out);
int poppy_radial_mask(int width, int height, float* out);
/* Host-side tables behind two device kernels, exposed for tests that run without a GPU (no reference counterpart):
 * poppy_gabor_tables: the 16 kernels of a Gabor bank as cv::getGaborKernel returns them (which = 31: src/extractor.cpp:63-64, 13:
 * src/util.hpp:95; 16 * which^2 floats) and their paired, conjugated 64 x 64 spectra (8 * 4096 complex doubles) for the FFT form;
 * poppy_pyr_tail_plan: the tap table of the pyramid tail kernel for a frame geometry (info: first tail level, multi-pixel level
 * steps, single-pixel reductions, descriptors, LDS bytes, usable; desc: 4 words per descriptor). */
int poppy_gabor_tables(int which, float* bank, double* spectra);
int poppy_pyr_tail_plan(int width, int height, int pyramid_levels, int tail_px, int* info, unsigned* desc);
/* Host only: the plan of the length-n transform dft_detail2's kernels run (pass order, load permutation, float twiddles; see
 * poppy_amd/csrc/dft_exact.cpp).  factors needs room for 34 ints, itab for n ints, wave for 2n floats.  For the test suite. */
int poppy_dft_plan(int n, int* factors, int* n_factors, int* itab, float* wave);

/* blur_margin (src/util.cpp:574-602), the CLI's padding step before poppy::morph (src/poppy.cpp:233-240,293-308): the image
 * centred in a union_width x union_height canvas, the four margin strips Gaussian-blurred (127x127, sigma 6).  Bit-exact.   */
int poppy_hip_blur_margin(poppy_hip_ctx* ctx, const uint8_t* bgr, size_t stride, int width, int height,
                          int union_width, int union_height, uint8_t* dst, size_t dst_stride);
/* copies of the resident point sets after pair_begin / pair_load (n x 2 floats each); n via *n_points */
int poppy_hip_pair_points(poppy_hip_ctx* ctx, float* points1, float* points2, int max_points, int* n_points);

/* n frames with explicit ratios on the resident pair (frame-range sharding: each GPU renders its own
 * sub-range of phase-mode frames; shape[j] = mask[j] = t_j reproduces morph(..., phase = t_j) with
 * number_of_frames = 1).  chain as in poppy_hip_render.  write may be NULL.                            */
int poppy_hip_render_many(poppy_hip_ctx* ctx, const double* shape_ratio, const double* mask_ratio, int n, int chain,
                          poppy_write_cb write, void* user);

/* ---- multi-GPU (implementation notes in poppy_amd/csrc/comm.cpp) ----------------------------------------------
 * The path shards by FRAMES of one pair in phase mode (each frame = morph(.., phase = t_j) with number_of_frames = 1,
 * src/poppy.hpp:186-200,234-235) and by PAIRS (the pairs loop of the CLI, src/poppy.cpp:266-328).  The only exchange is the pair
 * state — both images, the mask field's grey complement, the point sets: one contiguous allocation — from the GPU that ran the
 * pair set-up to the others, as ONE ncclBroadcast over RCCL / xGMI.  librccl is loaded on first use.
 *
 * One process per GPU:  comm_id on one rank -> the 128 bytes to the others by any out-of-band channel -> comm_init on every
 * rank's context -> per pair: pair_begin on the root, pair_broadcast everywhere, render_many on every rank's frame range.
 *   comm_id / comm_init / comm_free   ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy for this context's device
 *   pair_broadcast                    the resident pair of `root` becomes the resident pair of every rank (non-root contexts
 *                                     allocate for width x height); with auto-align, phase == 1 on a receiver writes the ALIGNED image
 *   comm_max                          max over the ranks of a host double (step timing)
 *   pair_state_bytes / pair_export_device / pair_import_device   the same packed state to / from a caller's device buffer, for
 *                                     callers with a transport of their own                                                 */
#define POPPY_COMM_ID_BYTES 128
int poppy_hip_comm_id(uint8_t* id128);
int poppy_hip_comm_init(poppy_hip_ctx* ctx, int rank, int world, const uint8_t* id128);
int poppy_hip_comm_free(poppy_hip_ctx* ctx);
/* rank / world as handed to poppy_hip_comm_init, and what the RCCL communicator itself reports (ncclCommUserRank / ncclCommCount; -1 without one):
 * a launcher's check that the job it started is the job the library communicates in (bench.py --gpus N asserts count == N).  Any pointer may be NULL. */
int poppy_hip_comm_info(poppy_hip_ctx* ctx, int* rank, int* world, int* nccl_rank, int* nccl_count);
int poppy_hip_pair_broadcast(poppy_hip_ctx* ctx, int root, int width, int height);
/* The pair set-up ITSELF spread over the communicator's ranks — a collective that replaces poppy_hip_pair_begin_device + poppy_hip_pair_broadcast
 * (the serial part of a sharded morph): rank `root` holds the raw pair (device pointers; the other ranks pass NULL) and filters / detects image 1,
 * rank root + 1 does image 2, rank root + 2 the mask field (src/poppy.hpp:52,114-122: independent until the matcher); the raw pair, the
 * two dft_detail2 values, image 2's keypoint positions, the matched point sets and the mask field's grey complement are exchanged through the
 * communicator.  Afterwards every rank holds the resident pair exactly as poppy_hip_pair_begin_device would have produced it on one GPU
 * (one limit of its own: image 2's keypoints travel through the state's point area, 16 383 at most — far above max_keypoints x detail
 * for every setting of the reference's CLI).  POPPY_E_UNSUPPORTED with enable_auto_align.  Opt-in in poppy_hip_morph_sharded
 * (POPPY_HIP_SHARD_SETUP=1) and bench.py (--shard-setup) until its RCCL transport has run on three or more GPUs.  _local: the same protocol between n contexts of THIS process (host threads; context k = rank k). */
int poppy_hip_pair_begin_sharded(poppy_hip_ctx* ctx, const void* d_bgr1, const void* d_bgr2, int width, int height, int root);
/* How often the sharded protocol itself ran in this process (a world of one takes poppy_hip_pair_begin_device instead unless
 * POPPY_HIP_SHARD_WORLD1 is set): lets a test on a one-GPU box see that the protocol, over RCCL, is what it exercised. */
unsigned long long poppy_hip_sharded_setups(void);
int poppy_hip_pair_begin_sharded_local(poppy_hip_ctx** ctxs, int n, const void* d_bgr1, const void* d_bgr2, int width, int height, int root);
int poppy_hip_comm_max(poppy_hip_ctx* ctx, double* value);
int poppy_hip_pair_state_bytes(int width, int height, size_t* bytes);
int poppy_hip_pair_export_device(poppy_hip_ctx* ctx, void* d_dst, size_t bytes);
int poppy_hip_pair_import_device(poppy_hip_ctx* ctx, const void* d_src, size_t bytes, int width, int height);
/* One process, several GPUs (a drop-in behind the reference's single-process CLI): one host thread + one context per device,
 * communicators from ncclCommInitAll.
 *   morph_sharded  ONE total_frames-frame phase-mode morph of the pair: frame j = morph(img1, img2, .., phase = j / total_frames)
 *                  with number_of_frames = 1 (frame 0 is the phase == 0 copy of image 1); device k renders the k-th contiguous
 *                  share.  `write` is called concurrently from n_devices threads, each with ascending frame indices.
 *   morph_pairs    n_pairs independent pairs, each one whole poppy_hip_morph(.., phase, ..) (default chained mode for phase < 0),
 *                  taken off a shared counter by contexts_per_device host threads per GPU (2-3 fill a GPU: a chained sequence is
 *                  a latency chain, and one pair's set-up runs beside another pair's frames).  `source` hands out pair p's two
 *                  images for the device that will render it (pointers must stay valid until the pair's last frame was written;
 *                  return 0); `write` gets (pair, frame) and is called concurrently.  No communication.
 *   pool_*         the same with contexts (and their HBM) kept between batches; inputs_on_device != 0: `source` returns device
 *                  pointers (tight rows) in the memory of the device it is asked for.
 * err (may be NULL) receives the message of the first failure.                                                              */
typedef struct poppy_hip_pool poppy_hip_pool;
typedef void (*poppy_write_indexed_cb)(void* user, int frame_index, const uint8_t* bgr, int width, int height, size_t stride);
typedef int (*poppy_pair_source_cb)(void* user, int pair_index, int device, const uint8_t** bgr1, size_t* stride1, const uint8_t** bgr2, size_t* stride2);
typedef void (*poppy_write_pair_cb)(void* user, int pair_index, int frame_index, const uint8_t* bgr, int width, int height, size_t stride);
poppy_hip_pool* poppy_hip_pool_create(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings,
                                      char* err, size_t err_len);
/* The same, with the start-up check a long-lived service makes once: the contexts of a pool get their streams, hardware queues and buffers from the
 * runtime, and about one pool in ten runs every batch 10-25 % slower for as long as it lives.  Up to max_candidates (1..8) pools are made, each renders a
 * built-in calibration batch of the given geometry (two pairs per context, once untimed, twice timed); two pools that agree within 4 % end the search;
 * the fastest is returned, the others are destroyed.  candidates_ms (may be NULL, room for max_candidates): the batch time of every pool made, in order;
 * *n_made, *kept (may be NULL): how many were made, which one was kept.                                                                             */
poppy_hip_pool* poppy_hip_pool_create_tuned(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings,
                                            int width, int height, int max_candidates, float* candidates_ms, int* n_made, int* kept,
                                            char* err, size_t err_len);
void poppy_hip_pool_destroy(poppy_hip_pool* pool);
int poppy_hip_pool_morph_pairs(poppy_hip_pool* pool, int n_pairs, int width, int height, double phase, int inputs_on_device,
                               poppy_pair_source_cb source, poppy_write_pair_cb write, void* user, char* err, size_t err_len);
/* The same batch without waiting for it: poppy_hip_pool_submit_pairs queues it and returns, poppy_hip_pool_wait returns when every pair submitted so far has been
 * rendered and written (the first failure's code and message; the pairs queued behind a failure are dropped).  Batches are taken up in order, pair by pair, by
 * whichever context is free, so the last pairs of one batch render beside the first set-ups of the next — the contexts of a pool fed this way never start a round of
 * set-ups together (the reference's CLI loop over pairs, src/poppy.cpp:266-328, fed by a service instead of a directory listing).  `source`, `write` and `user` must
 * stay valid until the wait.  Do not mix with a poppy_hip_pool_morph_pairs call in flight; poppy_hip_pool_destroy renders what is still queued first.        */
int poppy_hip_pool_submit_pairs(poppy_hip_pool* pool, int n_pairs, int width, int height, double phase, int inputs_on_device,
                                poppy_pair_source_cb source, poppy_write_pair_cb write, void* user);
int poppy_hip_pool_wait(poppy_hip_pool* pool, char* err, size_t err_len);
/* poppy_hip_set_timing / poppy_hip_timing_summary / poppy_hip_warp_counts over all contexts of a pool */
int poppy_hip_pool_set_timing(poppy_hip_pool* pool, int on);
int poppy_hip_pool_timing_summary(poppy_hip_pool* pool, const char** names, float* total_ms, int* launches, int max);
int poppy_hip_pool_warp_counts(poppy_hip_pool* pool, unsigned long long* fused, unsigned long long* tiled, unsigned long long* general);
/* poppy_hip_mask_rider of the pool's contexts (they share settings and geometry) */
int poppy_hip_pool_mask_rider(poppy_hip_pool* pool);
/* a poppy_write_pair_cb that only counts, atomically: ++*(long long*)user */
void poppy_count_pair_frames_cb(void* user, int pair_index, int frame_index, const uint8_t* bgr, int width, int height, size_t stride);
int poppy_hip_morph_sharded(const int* devices, int n_devices, const poppy_settings* settings,
                            const uint8_t* bgr1, size_t stride1, const uint8_t* bgr2, size_t stride2, int width, int height,
                            int total_frames, poppy_write_indexed_cb write, void* user, char* err, size_t err_len);
int poppy_hip_morph_pairs(const int* devices, int n_devices, int contexts_per_device, const poppy_settings* settings, int n_pairs,
                          int width, int height, double phase, poppy_pair_source_cb source, poppy_write_pair_cb write, void* user,
                          char* err, size_t err_len);

/* File sinks for the frame hand-off, host only, no codec library: poppy_sink_write has the poppy_write_cb signature
 * (user = the sink), so  poppy_hip_morph(ctx, .., poppy_sink_write, sink, ..)  writes the sequence to disk.  RAW: one file, BGR rows back
 * to back; PPM: one P6 file per frame, `path` holds exactly one %d, %<width>d or %0<width>d for the frame index (the library substitutes
 * it itself; any other conversion, or none, makes poppy_sink_open return NULL); Y4M: one YUV4MPEG2 file, C444, full-range BT.601.
 * poppy_sink_close returns the number of frames written, or a negative status if a write failed or a frame had another geometry.   */
typedef struct poppy_sink poppy_sink;
enum { POPPY_SINK_RAW = 0, POPPY_SINK_PPM = 1, POPPY_SINK_Y4M = 2 };
poppy_sink* poppy_sink_open(const char* path, int format, int width, int height, int fps_num, int fps_den);
void poppy_sink_write(void* sink, const uint8_t* bgr, int width, int height, size_t stride);
int poppy_sink_close(poppy_sink* sink);

/* n frames of the sharded job on the resident pair: frame k = morph(img1, img2, .., phase = t[k]) with number_of_frames = 1, i.e. a
 * copy of image 1 / image 2 for t == 0 / t == 1 (src/poppy.hpp:54-70) and an independent phase-mode frame otherwise.  What a rank
 * renders for its frame range t_j = j / total (poppy_hip_morph_sharded does the same internally).                              */
int poppy_hip_render_phases(poppy_hip_ctx* ctx, const double* t, int n, poppy_write_cb write, void* user);

/* No-match fallback  img2*phase + img1*(1-phase)  (src/poppy.hpp:125-134; u8 addWeighted,
 * OCV/core/src/arithm.simd.hpp:1705-1755).                                                          */
int poppy_hip_dissolve(poppy_hip_ctx* ctx, const uint8_t* img1, size_t stride1, const uint8_t* img2, size_t stride2,
                       int width, int height, double phase, uint8_t* dst, size_t dst_stride);

/* ---- diagnostics: copy an intermediate of the LAST frame to the host (parity tests) -----------
 * names: "triMap"(i32 HxW) "trImg1" "trImg2"(u8 HxWx3) "lbmask"(f32 HxW) "lapBlend" "unsharp"(f32 HxWx3);
 * of the resident pair: "gabor2"(f32 HxWx3, not on ranks that received the pair by broadcast) "m2"(f32 HxW)
 * "unsharp" is only available after poppy_hip_set_debug(ctx, 1).                                       */
int poppy_hip_set_debug(poppy_hip_ctx* ctx, int on);
/* Which warp kernel rendered the last frame: 2 = the fused raster + map + remap kernel (k_warp_bin: triangle ids rasterised per
 * tile in LDS, no id map; the normal case), 1 = the packed-arithmetic kernel on an id map from k_raster (k_warp_tile: debug mode,
 * POPPY_HIP_IDMAP, or a plan whose per-tile lists outgrew the blob), 0 = the general kernel (degenerate matrices or odd
 * geometry).  Same output bits in all three; exported so that the parity tests can tell which one they exercised.           */
int poppy_hip_last_warp_kind(poppy_hip_ctx* ctx);
/* frames rendered by each of the three since the context was created */
int poppy_hip_warp_counts(poppy_hip_ctx* ctx, unsigned long long* fused, unsigned long long* tiled, unsigned long long* general);
/* Measurement aid: relaunches the last frame's fused raster + warp kernel (create_map + remap, src/algo.cpp:146-176,230-238) `reps`
 * times back to back with nothing else running; *ms_per_launch = time between two events around the batch / reps.  POPPY_E_STATE when
 * the last frame did not take that kernel. */
int poppy_hip_time_last_warp(poppy_hip_ctx* ctx, int reps, float* ms_per_launch);
/* 1 when the warp kernel of this pair also writes the blend mask (lbmask = clamp((1 - mr) - m2 * mr), src/algo.cpp:262-263) beside the
 * two warped images, 0 when the level-0 blend kernels compute it from the pair's m2 field on the values they load (the normal case:
 * 8 B/px per frame less; POPPY_HIP_LBMASK_RIDER=1 or a geometry the wide blend kernels do not take select the former).           */
int poppy_hip_mask_rider(poppy_hip_ctx* ctx);
int poppy_hip_debug_fetch(poppy_hip_ctx* ctx, const char* name, void* host_dst, size_t bytes);
int poppy_hip_debug_triangles(poppy_hip_ctx* ctx, int* n_tris, int* idx3, float* M1, float* M2, int max_tris);

/* Host-only part of one frame (no GPU touched): mesh planning of morph_images, src/algo.cpp:184-228 +
 * the matrix inversions of create_map (:154-157).  Outputs may be NULL.  Used by the CPU test suite.  */
int poppy_plan_frame(int width, int height, const float* src_points1, const float* src_points2, int n_points,
                     double shape_ratio, int max_tris, int* n_tris, int* idx3, int* tri_xy,
                     float* M1, float* M2, float* inv1, float* inv2, float* morphed_pts);

/* Host-only: packs T pairs of inverse matrices (as poppy_plan_frame returns them) into the (T+1) x 20 float records the
 * tiled warp kernel reads (record 0 = identity; layout in poppy_amd/csrc/frame_plan.h) and returns 1 when every matrix is
 * inside the range for which that kernel's arithmetic is proven identical to the general one, 0 when the frame must take
 * the general kernel, < 0 on bad arguments.  Exported for the CPU test suite.                              */
int poppy_warp_records(const float* inv1, const float* inv2, int n_tris, int width, int height, float* records);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on.
 *   on = 0  off;
 *   on = 1  around every kernel group of a frame (the frame is then issued launch by launch instead of through its
 *           captured graph, so whole-frame throughput is a little lower while this is on);
 *   on = 2  around the fused map+remap kernel (k_warp_tile / k_warp4) only; the rest of the frame runs as usual.
 * timing_summary drains the streams and returns, per kernel group, the summed duration and the number of
 * launches since the last summary / set_timing call; returns the number of entries written.            */
int poppy_hip_set_timing(poppy_hip_ctx* ctx, int on);
int poppy_hip_timing_summary(poppy_hip_ctx* ctx, const char** names, float* total_ms, int* launches, int max);

#ifdef __cplusplus
}
#endif
#endif /* POPPY_HIP_H_ */
