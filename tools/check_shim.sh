#!/bin/bash
# Syntax check of include/poppy_hip_shim.hpp against the headers it is meant to sit beside: the vendored OpenCV 4.6.0 and
# Poppy's own settings.hpp.  CONTAINER-ONLY (needs /root/reference and the survey-stage OpenCV build tree for cvconfig.h /
# opencv_modules.hpp); compiles nothing into the product.  Instantiates the templates with a writer like the reference's.
set -euo pipefail
REF=/root/reference
OCV=$REF/third/opencv-4.6.0
OCVB=${OCV_BUILD:-/tmp/ocv-build}
HERE=$(cd "$(dirname "$0")/.." && pwd)
[ -f "$OCVB/opencv2/opencv_modules.hpp" ] || { echo "no OpenCV build tree at $OCVB: skipped" >&2; exit 3; }
TMP=$(mktemp -d)
cat > "$TMP/t.cpp" <<'CPP'
#include "poppy_hip_shim.hpp"
struct W { void write(cv::Mat&) {} };
struct WI { void write(int, cv::Mat&) {} };
void use(const cv::Mat& a, const cv::Mat& b, cv::Mat& c1, cv::Mat& c2, W& w, WI& wi) {
    poppy_hip::morph(a, b, c1, c2, -1.0, false, w);
    const int dev[2] = {0, 1};
    poppy_hip::morph_sharded(dev, 2, a, b, 480, wi);
}
CPP
g++ -std=c++20 -fsyntax-only -Wall -I"$HERE/include" -I"$REF/src" -isystem "$OCVB" -isystem "$OCV/include" -isystem "$OCV/modules/core/include" "$TMP/t.cpp"
rm -rf "$TMP"
echo "poppy_hip_shim.hpp: syntax ok against OpenCV 4.6.0 + Poppy settings.hpp"
