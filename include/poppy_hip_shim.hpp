// poppy_hip_shim.hpp — the reference-side binding: drop this header (and poppy_hip.h) into Poppy's src/, link -lpoppy_hip, and switch
// the call at src/poppy.cpp:324 from poppy::morph(...) to poppy_hip::morph(...).  Nothing else in the CLI changes: Twriter still only
// needs write(cv::Mat&) (cv::VideoWriter, ChannelWriter src/poppy.cpp:50-55, SDLWriter :57-84).
//
// Same signature and observable behaviour as poppy::morph (src/poppy.hpp:46-248, general — non-face — branch); where the C ABI
// reports a status, this header turns it back into what the reference does at that point (exit, exception, stderr text).
// Syntax-checked against the vendored OpenCV 4.6.0 headers and Poppy's own settings.hpp by __graft_entry__.build() when the
// reference tree is present (tools/check_shim.sh).
#pragma once
#include <opencv2/core.hpp>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include "poppy_hip.h"
#include "settings.hpp"

namespace poppy_hip {

// One context per process and GPU, created on first use from the Settings singleton that poppy::init fills (src/poppy.hpp:30-44).
struct Ctx {
    poppy_hip_ctx* h;
    explicit Ctx(int device = 0) {
        poppy_settings s;
        poppy_settings_default(&s);
        auto& cfg = poppy::Settings::instance();                       // src/settings.hpp:14-27
        s.number_of_frames = (int)cfg.number_of_frames;
        s.match_tolerance = cfg.match_tolerance;
        s.max_keypoints = (int)cfg.max_keypoints;
        s.pyramid_levels = (int)cfg.pyramid_levels;
        s.enable_radial_mask = cfg.enable_radial_mask ? 1 : 0;
        s.enable_auto_align = cfg.enable_auto_align ? 1 : 0;
        h = poppy_hip_create(device, &s);
        if (!h) throw std::runtime_error(std::string("poppy_hip_create: ") + poppy_hip_create_error());   // no CPU fallback by design
    }
    ~Ctx() { poppy_hip_destroy(h); }
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
};

template <typename Twriter>
struct WriterThunk {                                                   // poppy_write_cb -> Twriter::write(cv::Mat&)
    static void call(void* user, const uint8_t* bgr, int w, int h, size_t stride) {
        cv::Mat frame(h, w, CV_8UC3, const_cast<uint8_t*>(bgr), stride);   // valid during the call, like the reference's reused `morphed`
        static_cast<Twriter*>(user)->write(frame);
    }
};

template <typename Twriter>
void morph(const cv::Mat& img1, const cv::Mat& img2, cv::Mat& corrected1, cv::Mat& corrected2, double phase, bool distance, Twriter& output) {
    static Ctx ctx;
    CV_Assert(img1.type() == CV_8UC3 && img2.type() == CV_8UC3 && img1.size() == img2.size());
    if (poppy::Settings::instance().enable_face_detection)
        throw std::runtime_error("poppy_hip::morph: face-landmark mode is not part of libpoppy_hip (SURVEY.md section 8: out of scope)");
    if (phase == 0) std::cerr << "zero phase. inserting image 1" << std::endl;          // src/poppy.hpp:55
    else if (phase == 1) std::cerr << "full phase. inserting image 2" << std::endl;     // :63
    double dist = 0;
    // phase == 0 / 1 short-circuits, pair set-up, the printed morph distance, the frame loop: all inside the one call
    const int rc = poppy_hip_morph(ctx.h, img1.data, img1.step, img2.data, img2.step, img1.cols, img1.rows, phase, distance ? 1 : 0,
                                   &WriterThunk<Twriter>::call, &output, &dist);
    if (phase == 0 || phase == 1) return;
    if (rc == POPPY_OK || rc == POPPY_E_NOMATCH) {
        corrected1 = img1.clone();                                     // Matcher::find hands back clones (src/matcher.cpp:24-25) ...
        corrected2 = img2.clone();
        if (rc == POPPY_OK && poppy::Settings::instance().enable_auto_align)        // ... and the ALIGNED second image, which run()
            poppy_hip_pair_corrected2(ctx.h, corrected2.data, corrected2.step);     // chains into the next pair (src/poppy.cpp:326)
    }
    if (rc == POPPY_E_NOMATCH) {
        // The reference means to write the linear blend here (src/poppy.hpp:125-134) but, with empty point lists, throws a
        // cv::Exception from Matcher::find -> morph_distance -> convexHull before it gets there.  The library wrote the blend.
        std::cerr << "No matches found. Inserting linear blend." << std::endl;
        return;
    }
    if (rc != POPPY_OK) throw std::runtime_error(std::string("poppy_hip_morph: ") + poppy_hip_last_error(ctx.h));
    std::cerr << "morph distance: " << dist << std::endl;              // src/poppy.hpp:159
    if (distance) exit(0);                                             // :161-163
}

// One pair across several GPUs of the node (what `--frames N --phase ...` loops would add up to: N phase-mode frames): see
// poppy_hip_morph_sharded in poppy_hip.h.  Frames arrive out of order, from one thread per GPU, tagged with their index.
template <typename Tindexed>
void morph_sharded(const int* devices, int n_devices, const cv::Mat& img1, const cv::Mat& img2, int total_frames, Tindexed& sink) {
    poppy_settings s;
    poppy_settings_default(&s);
    auto& cfg = poppy::Settings::instance();
    s.match_tolerance = cfg.match_tolerance; s.max_keypoints = (int)cfg.max_keypoints; s.pyramid_levels = (int)cfg.pyramid_levels;
    s.enable_radial_mask = cfg.enable_radial_mask ? 1 : 0; s.enable_auto_align = cfg.enable_auto_align ? 1 : 0;
    struct Thunk {
        static void call(void* user, int index, const uint8_t* bgr, int w, int h, size_t stride) {
            cv::Mat frame(h, w, CV_8UC3, const_cast<uint8_t*>(bgr), stride);
            static_cast<Tindexed*>(user)->write(index, frame);
        }
    };
    char err[512] = {0};
    const int rc = poppy_hip_morph_sharded(devices, n_devices, &s, img1.data, img1.step, img2.data, img2.step, img1.cols, img1.rows,
                                           total_frames, &Thunk::call, &sink, err, sizeof err);
    if (rc != POPPY_OK) throw std::runtime_error(std::string("poppy_hip_morph_sharded: ") + err);
}

}  // namespace poppy_hip
