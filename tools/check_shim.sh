#!/bin/bash
# include/poppy_hip_shim.hpp against what it is meant to sit beside — the vendored OpenCV 4.6.0 and Poppy's own settings: a translation unit that
# instantiates the shim's templates with writers like the reference's is COMPILED and LINKED against libopencv_core.a, Poppy's own settings.cpp and
# libpoppy_hip.so (every poppy_* symbol the shim calls resolves; nothing runs: there is no GPU here).  CONTAINER-ONLY (needs /root/reference and the
# OpenCV build tree of oracle/golden_gen/build_ocv.sh); compiles nothing into the product.
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
cat >> "$TMP/t.cpp" <<'CPP'
int main(int argc, char**) {
    if (argc > 1000) {                       // never true: the calls only have to link
        cv::Mat a, b, c1, c2; W w; WI wi;
        use(a, b, c1, c2, w, wi);
    }
    return 0;
}
CPP
INC="-I$HERE/include -I$REF/src -isystem $OCVB -isystem $OCV/include -isystem $OCV/modules/core/include"
g++ -std=c++20 -O1 -Wall $INC -c "$TMP/t.cpp" -o "$TMP/t.o"
g++ -std=c++20 -O1 -w $INC -c "$REF/src/settings.cpp" -o "$TMP/settings.o"
[ -f "$HERE/poppy_amd/libpoppy_hip.so" ] || { echo "no poppy_amd/libpoppy_hip.so: build it first (__graft_entry__.build)" >&2; exit 4; }
g++ -o "$TMP/shim_link" "$TMP/t.o" "$TMP/settings.o" -L"$OCVB/lib" -L"$OCVB/3rdparty/lib" -lopencv_core -lzlib -L"$HERE/poppy_amd" -lpoppy_hip \
    -Wl,-rpath,"$HERE/poppy_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lpthread -ldl
nm -u "$TMP/shim_link" | grep -c " U poppy_" | sed 's/^/poppy_* symbols bound from libpoppy_hip.so: /'
rm -rf "$TMP"
echo "poppy_hip_shim.hpp: compiles and links against OpenCV 4.6.0 (libopencv_core.a) + Poppy settings.cpp + libpoppy_hip.so"
