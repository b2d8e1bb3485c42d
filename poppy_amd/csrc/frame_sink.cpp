// frame_sink.cpp — file sinks for the frame hand-off ("optional raw/PPM/Y4M sink"): the reference writes through
// cv::VideoWriter (FFV1, src/poppy.cpp:249), which needs a codec library; these need none.  Host only, no GPU involved:
// poppy_sink_write has the poppy_write_cb signature, so a sink plugs straight into poppy_hip_morph / poppy_hip_morph_frames.
//   POPPY_SINK_RAW   one file, frames back to back, width*3 bytes per row, BGR (what the writer callback receives)
//   POPPY_SINK_PPM   one binary PPM (P6, RGB) per frame; the path holds exactly one %d / %<width>d / %0<width>d (frame index from 0)
//   POPPY_SINK_Y4M   one YUV4MPEG2 file, C444, full-range BT.601 in 8-bit integer arithmetic (lossless containers downstream can
//                    re-encode it; the conversion is this file's, not the reference's)
#include "../../include/poppy_hip.h"
#include <cstdio>
#include <string>
#include <vector>

struct poppy_sink {
    int format = 0, w = 0, h = 0, frames = 0;
    bool failed = false;
    std::string path;                  // PPM: the text before the pattern's conversion
    std::string tail;                  // PPM: the text after it
    int pad = 0; bool zero = false;    // PPM: width and zero flag of the conversion (%d, %5d, %05d)
    FILE* f = nullptr;
    std::vector<uint8_t> row;
};

extern "C" {

poppy_sink* poppy_sink_open(const char* path, int format, int width, int height, int fps_num, int fps_den) {
    if (!path || width <= 0 || height <= 0 || format < POPPY_SINK_RAW || format > POPPY_SINK_Y4M) return nullptr;
    poppy_sink* s = new poppy_sink();
    s->format = format; s->w = width; s->h = height; s->path = path;
    if (format == POPPY_SINK_PPM) {
        // The pattern is parsed here, never handed to printf: exactly one conversion of the form %d / %<width>d / %0<width>d ("%%" is a
        // literal percent sign); anything else — %s, %n, a second conversion, no conversion at all (every frame would overwrite the
        // same file) — is refused.
        std::string head, tail;
        bool seen = false, bad = false;
        const std::string pat = path;
        for (size_t i = 0; i < pat.size() && !bad; ++i) {
            if (pat[i] != '%') { (seen ? tail : head) += pat[i]; continue; }
            if (i + 1 < pat.size() && pat[i + 1] == '%') { (seen ? tail : head) += '%'; ++i; continue; }
            if (seen) { bad = true; break; }
            size_t j = i + 1;
            if (j < pat.size() && pat[j] == '0') { s->zero = true; ++j; }
            int wdt = 0;
            while (j < pat.size() && pat[j] >= '0' && pat[j] <= '9' && wdt < 100) wdt = wdt * 10 + (pat[j++] - '0');
            if (j >= pat.size() || pat[j] != 'd' || wdt > 20) { bad = true; break; }
            s->pad = wdt; seen = true; i = j;
        }
        if (bad || !seen) { delete s; return nullptr; }
        s->path = head; s->tail = tail;
    }
    if (format != POPPY_SINK_PPM) {
        s->f = fopen(path, "wb");
        if (!s->f) { delete s; return nullptr; }
        if (format == POPPY_SINK_Y4M)
            fprintf(s->f, "YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C444 XCOLORRANGE=FULL\n", width, height, fps_num > 0 ? fps_num : 30, fps_den > 0 ? fps_den : 1);
    }
    s->row.resize((size_t)width * 3);
    return s;
}

void poppy_sink_write(void* user, const uint8_t* bgr, int width, int height, size_t stride) {
    poppy_sink* s = (poppy_sink*)user;
    if (!s || s->failed) return;
    if (!bgr || width != s->w || height != s->h || stride < (size_t)width * 3) { s->failed = true; return; }
    FILE* f = s->f;
    if (s->format == POPPY_SINK_PPM) {
        std::string num = std::to_string(s->frames);
        if ((int)num.size() < s->pad) num.insert(0, (size_t)s->pad - num.size(), s->zero ? '0' : ' ');
        const std::string name = s->path + num + s->tail;
        f = fopen(name.c_str(), "wb");
        if (!f) { s->failed = true; return; }
        fprintf(f, "P6\n%d %d\n255\n", width, height);
    } else if (s->format == POPPY_SINK_Y4M) {
        fputs("FRAME\n", f);
    }
    bool ok = true;
    if (s->format == POPPY_SINK_RAW) {
        for (int y = 0; y < height && ok; ++y) ok = fwrite(bgr + (size_t)y * stride, 1, (size_t)width * 3, f) == (size_t)width * 3;
    } else if (s->format == POPPY_SINK_PPM) {
        for (int y = 0; y < height && ok; ++y) {
            const uint8_t* p = bgr + (size_t)y * stride;
            for (int x = 0; x < width; ++x) { s->row[3 * x] = p[3 * x + 2]; s->row[3 * x + 1] = p[3 * x + 1]; s->row[3 * x + 2] = p[3 * x]; }
            ok = fwrite(s->row.data(), 1, (size_t)width * 3, f) == (size_t)width * 3;
        }
        fclose(f);
    } else {
        // planar Y, U, V; JFIF full-range BT.601 with 16 fractional bits: Y = 0.299 R + 0.587 G + 0.114 B, U = 128 - 0.168736 R - 0.331264 G + 0.5 B, ...
        for (int plane = 0; plane < 3 && ok; ++plane)
            for (int y = 0; y < height && ok; ++y) {
                const uint8_t* p = bgr + (size_t)y * stride;
                for (int x = 0; x < width; ++x) {
                    const int b = p[3 * x], g = p[3 * x + 1], r = p[3 * x + 2];
                    int v;
                    if (plane == 0) v = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
                    else if (plane == 1) v = ((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16) + 128;
                    else v = ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128;
                    s->row[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                }
                ok = fwrite(s->row.data(), 1, (size_t)width, f) == (size_t)width;
            }
    }
    if (!ok) s->failed = true; else ++s->frames;
}

int poppy_sink_close(poppy_sink* s) {
    if (!s) return POPPY_E_ARG;
    const int n = s->failed ? POPPY_E_DEVICE : s->frames;
    if (s->f) fclose(s->f);
    delete s;
    return n;
}

}  // extern "C"
