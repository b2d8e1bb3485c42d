#!/usr/bin/env python3
"""Packs the raw reference dumps (oracle/_dumps/cases/*/out) into committed fixtures.

tests/golden/<case>.npz      arrays small enough to commit, stored in full (compressed)
tests/golden/manifest.json   for EVERY dumped array: dtype, shape, sha256 of the raw bytes,
                             and whether the full array is in the npz

A fixture is data only: expected outputs of the reference for inputs that the tests
regenerate from poppy_amd/synth.py.  Provenance is recorded in the manifest.
"""
import glob
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CASES = os.path.join(ROOT, "oracle", "_dumps", "cases")
GOLD = os.path.join(ROOT, "tests", "golden")

DT = {"u8": np.uint8, "f32": np.float32, "i32": np.int32, "f64": np.float64, "i16": np.int16, "u16": np.uint16}
FULL_LIMIT = 200 * 1024          # bytes: anything at or below is stored in full
FRAME_LIMIT = 1536 * 1024        # final frames / warped images / ORB inputs may be larger


def load_raw(path):
    name, dt, shp, _ = os.path.basename(path).rsplit(".", 3)
    shape = tuple(int(s) for s in shp.split("x")) if shp else ()
    a = np.fromfile(path, dtype=DT[dt])
    return name, a.reshape(shape)


# long sequences: frames are pinned by their sha256 only (first and last kept in full for debugging)
HASH_ONLY_FRAMES = {"a_512x512_chain30": ("frame0", "frame29"), "a_1920x1080_chain60": (), "a_3840x2160_phase": (), "a_1920x1080_photo60": (),
                    "a_639x480_numbers": ("frame0",), "a_749x480_cars": ("frame0",)}


def keep_full(name, arr, case=""):
    if case in HASH_ONLY_FRAMES and "frame" in name:
        return name in HASH_ONLY_FRAMES[case]
    if arr.nbytes <= FULL_LIMIT:
        return True
    big_ok = name.startswith("frame") or name.endswith("frame") or name in ("g1", "g2", "gabor2", "us1", "gb1", "radial", "padded") or "trImg" in name
    return big_ok and arr.nbytes <= FRAME_LIMIT


def main():
    """pack.py            repack every case under oracle/_dumps/cases
    pack.py CASE ...   (re)pack only these and merge them into the existing manifest (npz files are not
                       byte-reproducible, so untouched cases keep their committed files)"""
    only = sys.argv[1:]
    os.makedirs(GOLD, exist_ok=True)
    manifest = {
        "provenance": {
            "reference": "kallaballa/Poppy @ /root/reference (src/*.cpp compiled -std=c++20 -O3 -D_NO_FACE_DETECT)",
            "opencv": "vendored third/opencv-4.6.0, static Release, CPU_BASELINE=SSE3, CPU_DISPATCH empty, no IPP/OpenCL, "
                      "1 thread; build tree produced by the survey stage (SURVEY.md F7/8c), not by this repo",
            "generator": "oracle/golden_gen/{make_inputs.py,gen_golden.cpp,build.sh,pack.py}",
            "inputs": "poppy_amd/synth.py (integer-defined); case tables in oracle/golden_gen/make_inputs.py",
        },
        "cases": {},
    }
    mpath = os.path.join(GOLD, "manifest.json")
    if only and os.path.exists(mpath):
        manifest["cases"] = json.load(open(mpath))["cases"]
    for cdir in sorted(glob.glob(os.path.join(CASES, "*"))):
        case = os.path.basename(cdir)
        if only and case not in only:
            continue
        full = {}
        entries = {}
        for path in sorted(glob.glob(os.path.join(cdir, "out", "*.bin"))):
            name, arr = load_raw(path)
            e = {"dtype": str(arr.dtype), "shape": list(arr.shape), "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}
            if keep_full(name, arr, case):
                full[name] = arr
                e["full"] = True
            else:
                e["full"] = False
            entries[name] = e
        manifest["cases"][case] = entries
        np.savez_compressed(os.path.join(GOLD, case + ".npz"), **full)
        sz = os.path.getsize(os.path.join(GOLD, case + ".npz"))
        print(f"{case}: {len(entries)} arrays, {len(full)} full, npz {sz / 1024:.0f} KiB")
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
