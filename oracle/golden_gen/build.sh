#!/bin/bash
# Builds the golden-vector generator.  CONTAINER-ONLY: needs /root/reference and the
# survey-stage OpenCV build tree (vendored OpenCV 4.6.0, static,
# Release, CPU_BASELINE=SSE3, CPU_DISPATCH empty, no IPP/OpenCL).  This script does NOT run
# cmake or any reference build system; if the prebuilt libraries are not there it stops.
# Outputs go to oracle/_ref/ (git-ignored).  Nothing here is used at test/bench time —
# tests read the committed fixtures in tests/golden/.
set -euo pipefail
REF=/root/reference
OCV=$REF/third/opencv-4.6.0
OCVB=${OCV_BUILD:-/tmp/ocv-build}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
mkdir -p "$OUT/obj"
[ -f "$OCVB/lib/libopencv_core.a" ] || { echo "no prebuilt OpenCV at $OCVB (survey-stage build missing); cannot regenerate goldens" >&2; exit 3; }
INC="-I$REF/src -I$OCVB -I$OCV/include"
for m in core imgproc features2d flann video photo imgcodecs calib3d highgui videoio objdetect ml dnn stitching; do
  INC="$INC -I$OCV/modules/$m/include"
done
CXXFLAGS="-std=c++20 -O3 -pthread -fno-strict-aliasing -D_NO_TIMETRACK -D_NO_FACE_DETECT -w"
for tu in algo util extractor matcher transformer procrustes draw settings terminal face; do
  [ "$OUT/obj/$tu.o" -nt "$REF/src/$tu.cpp" ] || g++ $CXXFLAGS $INC -c "$REF/src/$tu.cpp" -o "$OUT/obj/$tu.o" &
done
wait
g++ $CXXFLAGS $INC -c "$HERE/gen_golden.cpp" -o "$OUT/obj/gen_golden.o"
g++ -o "$OUT/gen_golden" "$OUT/obj/"*.o \
  -L"$OCVB/lib" -L"$OCVB/3rdparty/lib" \
  -lopencv_highgui -lopencv_video -lopencv_photo -lopencv_calib3d -lopencv_features2d -lopencv_flann -lopencv_imgcodecs -lopencv_imgproc -lopencv_core \
  -llibpng -llibjpeg-turbo -lzlib -lpthread -ldl
echo "built $OUT/gen_golden"
