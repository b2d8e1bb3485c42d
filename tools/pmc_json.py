#!/usr/bin/env python3
"""Writes profiles/<round>_warp_pmc.json (round = the tag up to its first underscore) (the per-launch HBM bytes bench.py quotes as roofline.traffic) and prints the per-kernel
tables of profiles/<tag>_pmc.md from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh:
    tools/pmc_json.py <tag>      reads gpurun_out/<tag>_{fetch,write}_{1920,3840}/*.db
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at half their bytes on this image);
the calibration row is k_gray_inv, which reads exactly 12 B/px and writes 4."""
import glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import per_kernel

tag = sys.argv[1]
out = {}
hsh = hashlib.sha256()
for f in ("kernels_warp_bin.hip", "warp_fast_device.h", "warp_device.h"):
    hsh.update(open(os.path.join(ROOT, "poppy_amd", "csrc", f), "rb").read())
out["kernel_src_sha16"] = hsh.hexdigest()[:16]
out["run_id"] = os.environ.get("POPPY_RUN_ID")      # tools/profile_round.sh: hostname + UTC time + tag of the one call every file of the tag comes from
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/experiments/frames_only.py W H 60 chain 1; bytes per launch; "
               "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads (calibration in the same run: k_gray_inv reads 12 B/px, writes 4)")
for w, h in ((1920, 1080), (3840, 2160)):
    fdb = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_fetch_{w}", "*.db"))
    wdb = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_write_{w}", "*.db"))
    if not fdb or not wdb:
        continue
    fetch, write = per_kernel(fdb[0], "FETCH_SIZE"), per_kernel(wdb[0], "WRITE_SIZE")
    px = w * h
    sz = {}
    best = {}
    for (name, grid), (n, f) in fetch.items():          # the largest launch geometry of each kernel = its level-0 / full-frame form
        base = name.split("<")[0]
        if base not in ("k_unsharp_tile", "k_unsharp_stream", "k_collapse_level", "k_pyrdown_level", "k_tile_expand", "k_warp_bin", "k_gray_inv"):
            continue
        if base not in best or grid > best[base][0]:
            best[base] = (grid, name)
    for base, (grid, name) in best.items():
        f = fetch[(name, grid)][1]; wv = write.get((name, grid), (0, 0.0))[1]
        sz[base] = {"fetch_bytes": int(round(2 * f)), "write_bytes": int(round(wv))}
        print(f"{w}x{h} {name} grid {grid}: read {2 * f / px:.2f} B/px, written {wv / px:.2f} B/px")
    out[f"{w}x{h}"] = sz
json.dump(out, open(os.path.join(ROOT, "profiles", tag.split("_")[0] + "_warp_pmc.json"), "w"), indent=1)
