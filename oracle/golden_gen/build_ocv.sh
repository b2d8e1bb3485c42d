#!/bin/bash
# Recreates the OpenCV 4.6.0 tree the golden generator (build.sh) and tools/check_shim.sh link / include against:
# SURVEY.md section 8(c)'s recipe verbatim — the vendored source under /root/reference/third/opencv-4.6.0, static, Release,
# CPU_BASELINE = SSE3 (the default), CPU_DISPATCH empty (no AVX2 / FMA variants: v_fma stays mul-then-add), no IPP / OpenCL.
# Round 5: highgui is in the list too (no window back end: GTK / Qt off, videoio not built) — Poppy's util.cpp references cv::namedWindow /
# imshow / waitKey from show_image(), never reached with show_gui = false; the vendored module now provides those symbols instead of three
# stand-ins in gen_golden.cpp.
#
# CONTAINER-ONLY test infrastructure.  It runs the dependency's own build system (cmake + ninja), which is why the fixtures
# under tests/golden/ are listed as "captured reference outputs" and not as a reference build in DESIGN.md section 2: this
# script exists so that anybody can regenerate them and check tests/golden/manifest.json byte for byte (run_all.sh --verify).
# Outputs only under /tmp (OCV_BUILD, default /tmp/ocv-build); nothing of it travels to the GPU box or into the product.
# ~4-6 minutes on 8 vCPU.   usage: build_ocv.sh [-j N]
set -euo pipefail
REF=/root/reference
OCV=$REF/third/opencv-4.6.0
OCVB=${OCV_BUILD:-/tmp/ocv-build}
JOBS=${2:-$(nproc)}
[ -d "$OCV" ] || { echo "no vendored OpenCV at $OCV" >&2; exit 3; }
if [ -f "$OCVB/lib/libopencv_core.a" ] && [ -f "$OCVB/lib/libopencv_video.a" ] && [ -f "$OCVB/lib/libopencv_highgui.a" ] && [ -f "$OCVB/opencv2/opencv_modules.hpp" ]; then
  echo "OpenCV build tree already at $OCVB"; exit 0
fi
cmake -G Ninja -S "$OCV" -B "$OCVB" -DCMAKE_BUILD_TYPE=Release \
  -DBUILD_LIST=core,imgproc,features2d,flann,video,photo,imgcodecs,calib3d,highgui -DBUILD_SHARED_LIBS=OFF \
  -DBUILD_TESTS=OFF -DBUILD_PERF_TESTS=OFF -DBUILD_EXAMPLES=OFF -DBUILD_opencv_apps=OFF \
  -DWITH_IPP=OFF -DWITH_OPENCL=OFF -DWITH_ITT=OFF -DWITH_ADE=OFF -DWITH_PROTOBUF=OFF -DWITH_QUIRC=OFF -DWITH_EIGEN=OFF -DWITH_LAPACK=OFF \
  -DWITH_FFMPEG=OFF -DWITH_GSTREAMER=OFF -DWITH_V4L=OFF -DWITH_GTK=OFF -DWITH_OPENEXR=OFF -DWITH_JASPER=OFF -DWITH_OPENJPEG=OFF -DWITH_WEBP=OFF -DWITH_TIFF=OFF \
  -DBUILD_ZLIB=ON -DBUILD_PNG=ON -DBUILD_JPEG=ON -DCPU_DISPATCH= -DOPENCV_DOWNLOAD_PATH=/tmp/ocv-cache -DCMAKE_POLICY_VERSION_MINIMUM=3.5
ninja -C "$OCVB" -j "$JOBS"
ls "$OCVB"/lib/libopencv_{core,imgproc,features2d,flann,video,photo,imgcodecs,calib3d,highgui}.a
echo "built $OCVB"
