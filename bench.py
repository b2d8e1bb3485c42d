#!/usr/bin/env python3
"""bench.py — morph frames/sec at 1080p for 60-frame sequences (BASELINE.json metric) on N MI355X GPUs of one node.

N = 1 (BASELINE.json configs[1]).  A step = PAIRS (6) independent synthetic 1080p pairs (seeds 1234 + k), each
taken through the WHOLE of poppy::morph (src/poppy.hpp:46-248) by the library:
    poppy_hip_pair_begin_device   raw BGR pair (resident in HBM before the timed region) -> foreground x2, dft_detail2 x2, ORB
                                  input x2, ORB x2, matcher, gabor2: the real point sets and mask field, nothing synthetic
    poppy_hip_morph_frames(-1)    the reference's default mode: 60 chained frames, every frame downloaded into pinned host
                                  memory and handed to a writer callback (poppy_count_frames_cb)
`value` = frames written / wall time of the timed region: set-up, rendering and the writer hand-off are all inside.  The
per-frame operator alone on a resident pair with the frames left in HBM (what round 1 reported as `value`) is the extra key
`resident_pair_fps`.

N > 1 (configs[3]): ONE 480-frame phase-mode morph of one pair for EVERY N (`--total-frames`, default 480: north_star's "480-frame 1080p
morph"): rank 0 runs the pair set-up, the pair state (both images, the mask field's grey complement, the point sets) goes to every rank
in one broadcast over RCCL/xGMI (`--shard-setup`: the set-up itself spread over ranks 0-2, used when it leaves the same point lists and is faster), rank r renders the r-th contiguous share of the frames t_j = j / 480 — each equal to
morph(img1, img2, ..., phase = t_j) with number_of_frames = 1 — and hands them to its writer.  No data-path collective afterwards; the
total work is fixed => "strong".  The N = 1 point of THAT job is the extra key `scaling_baseline_480` of the N = 1 line (the same
480-frame morph on one GPU, set-up and writer inside), so an N-sweep is read against it and not against the chained headline; every
N > 1 line carries its own measured serial part (`amdahl`: set-up, broadcast, frames).

One JSON line on rank 0 (driver contract) with
  roofline      (`isolated` inside it: the same kernel's launches with nothing else on the GPU, measured after the timed region)
                the dominant kernel (fused create_map + remap of both sources): algorithmic bytes per launch = 16 B/px (SURVEY.md
                8d, faithful path: id 4 + src 3 + 3 in, warped 3 + 3 out) x pixels, / the kernel's average launch duration
                measured live with HIP events attached to the dispatch on the library's stream (one launch in seven of the timed
                region).  The kernel moves about what the contract counts: no id map (the raster reaches it as one id byte per pixel
                + 2.5 KB of record slots per tile, ~3.6 B/px) and no blend mask (the level-0 blend kernels derive it from m2) — 12 B/px of images; `moved_*` states
                that.  Only when the mask rider is on (POPPY_HIP_LBMASK_RIDER) `frac_with_rider` adds its 8 B/px;
  cpu_baseline  oracle/ (CPU restatement, "port"): 20 chained 1080p frames of pair 0 from the GPU's own pair state on 1 thread (`value`),
                the same operator on every host thread at once (`all_cores`), and one whole morph incl. the oracle's pair set-up at
                512x512x30 (`end_to_end`: BASELINE.json configs[0], so that a CPU figure covers the same work as `value`);
  parity_check  the 20th frame of that oracle run against the 20th frame the GPU wrote (bit-exact expected); `cfg3_4k.parity_check`:
                the last frame of the 120-frame 4K sequence against the oracle;
  cfg3_4k       configs[2]: one 3840x2160 pair, 120 phase-mode frames, set-up and writer hand-off included, with its own roofline.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, FRAMES = 1920, 1080, 60
PAIRS = 6                      # pairs per step at N = 1 (two per context of the default pool)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
WARP_CONTRACT_B_PER_PX = 16.0  # the faithful fused map+remap kernel
WARP_RIDER_B_PER_PX = 8.0      # m2 in, lbmask out (the kernel computes the blend mask on the way)

# algorithmic HBM bytes per frame of each kernel group, per full-resolution pixel
ALGO_BYTES_PER_PX = {
    "upload+clear": 0.0, "raster": 4.0, "warp": WARP_CONTRACT_B_PER_PX,
    "pyrdown": (6 + 4) + (24 + 4) / 4 * (4 / 3), "pyr_tail": 0.0,
    "collapse": (6 + 4 + 12) + (36 / 4) * (4 / 3) + 12 * (1 / 3), "unsharp": 12 + 3,
}


def synth_pair(w, h, k):
    from poppy_amd import synth
    return synth.gen_pair(w, h, seed=1234 + k)


def measured_traffic(kernel, w, h):
    """HBM bytes per launch of `kernel` from the committed counter passes (profiles/r06_warp_pmc.json) — only when those passes were
    taken from the kernel sources that are being run (hash of the source files recorded with them); otherwise None."""
    import hashlib
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r06_warp_pmc.json")))
        hsh = hashlib.sha256()
        for f in ("kernels_warp_bin.hip", "warp_fast_device.h", "warp_device.h"):
            hsh.update(open(os.path.join(ROOT, "poppy_amd", "csrc", f), "rb").read())
        if hsh.hexdigest()[:16] != pm.get("kernel_src_sha16"):
            return None, "profiles/r06_warp_pmc.json was taken from other kernel sources: not quoted"
        e = pm.get(f"{w}x{h}", {}).get(kernel)
        if not e:
            return None, "no counter pass for this kernel / size in profiles/r06_warp_pmc.json"
        return e["fetch_bytes"] + e["write_bytes"], ("profiles/r06_warp_pmc.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, same kernel sources; "
                                                     "profiles/r06_z_pmc.md; run id %s)" % pm.get("run_id"))
    except (OSError, ValueError, KeyError):
        return None, "profiles/r06_warp_pmc.json missing"


def profile_facts(kernel, w, h):
    """What the committed profiles say about `kernel` at this size (profiles/r06_warp_facts.json, written by tools/warp_facts.py from the
    round's rocprofv3 kernel trace and from the kernel's ISA): the trace's average duration and the kernel's vector-issue bound."""
    try:
        f = json.load(open(os.path.join(ROOT, "profiles", "r06_warp_facts.json")))
        e = f.get(f"{w}x{h}", {}).get(kernel)
        if e is not None:
            e = dict(e, run_id=f.get("run_id"))          # tools/profile_round.sh: the one gpurun call trace, counters and (when it is that call) this line come from
        return e
    except (OSError, ValueError):
        return None


def roofline_of(ctx, warp_ms, warp_n, w, h):
    per_launch_ms = warp_ms / max(warp_n, 1)
    P = w * h
    contract = WARP_CONTRACT_B_PER_PX * P
    rate = lambda b: b / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
    ach = rate(contract)
    rider = bool(ctx.mask_rider())
    out = {"bound": "hbm", "kernel": ctx.warp_kernel_name() + " (fused create_map + remap of both sources" + ("; also emits lbmask)" if rider else ")"),
           "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
           "algo_bytes_per_launch": int(contract), "algo_bytes_note": "16 B/px x pixels (SURVEY.md 8d faithful path: id 4 + src 3 + 3 in, warped 3 + 3 out)"}
    if rider:
        ach_r = rate((WARP_CONTRACT_B_PER_PX + WARP_RIDER_B_PER_PX) * P)
        out.update({"achieved_with_rider": round(ach_r, 1), "frac_with_rider": round(ach_r / HBM_PEAK_GBS, 4)})
    else:
        out.update({"moved_image_bytes_per_launch": int(12 * P), "moved_GBps": round(rate(12.0 * P), 1),
                    "moved_note": "what the kernel itself reads and writes: c1 3 + c2 3 in, tr1 3 + tr2 3 out = 12 B/px (+ 1 B/px of id bytes and ~2.5 B/px of record slots: 15.6 B/px by the counters); no id map, no blend mask"})
    facts = profile_facts(ctx.warp_kernel_name(), w, h)
    if facts:
        if facts.get("trace_avg_us"):
            ach_p = contract / (facts["trace_avg_us"] * 1e-6) / 1e9
            out["from_profiles"] = {"avg_launch_us": facts["trace_avg_us"], "achieved": round(ach_p, 1), "frac": round(ach_p / HBM_PEAK_GBS, 4),
                                    "moved_frac": round(12.0 * P / (facts["trace_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                    "source": facts.get("trace_source"), "run_id": facts.get("run_id")}
        if facts.get("issue_cycles_per_wave"):
            waves = (w // 4) * h / 64.0
            clk = facts.get("clock_GHz", 2.4)
            bound_us = facts["issue_cycles_per_wave"] * waves / 1024.0 / (clk * 1e3)
            out["valu"] = {"issue_cycles_per_wave": facts["issue_cycles_per_wave"], "instructions_per_wave": facts.get("instructions_per_wave"), "waves": int(waves),
                           "bound_us": round(bound_us, 2), "frac_of_launch": round(bound_us / (per_launch_ms * 1e3), 3) if per_launch_ms > 0 else None,
                           "note": "vector issue cycles of the kernel's main path by the measured per-instruction costs of gfx950 (profiles/r03_notes.md section 1: "
                                   "2 cycles full rate, 4 half rate incl. every packed / perm / dot / cvt / min-max, 8 quarter) x waves / 1024 SIMDs / %.1f GHz: the time the "
                                   "arithmetic alone needs; the binding resource beside the HBM fraction" % clk,
                           # the kernel's arithmetic run ALONE on register data (tools/micro/warp_alu.hip, profiles/r04_warp_alu.txt) and what deleting pieces of
                           # TODAY'S kernel does to its time in the chained loop (tools/experiments/abl_build.sh + abl_run.sh, profiles/r05_warp_ablation.txt)
                           "alu_floor_us": {"1920x1080": 7.7, "3840x2160": 30.5}.get(f"{w}x{h}"),
                           "ablation_4k_us": {"as_shipped": 45.8, "no_division": 46.8, "constant_weights": 46.3, "no_blend": 43.9, "no_footprint_loads": 41.4, "all_four": 30.2},
                           # round 6 (tools/micro/warp_skeleton2.hip, profiles/r06_warp_skeleton.txt): the kernel's SKELETON alone — id bytes, record slots through LDS + barrier,
                           # five 16-byte record reads per pixel, two 12-byte stores per four pixels, nothing else — in today's geometry and in eleven others
                           "skeleton_alone_us": {"3840x2160": {"as_shipped_256x4": 16.6, "best_found_256x16": 13.0, "stores_only": 13.3, "three_dword_stores": 21.0, "scalar_records": 23.2},
                                                 "1920x1080": {"as_shipped_256x4": 4.7, "best_found": 4.7, "stores_only": 4.0},
                                                 "reading": "the skeleton is not the defect: alone it moves its 9.5 B/px at 4.75 TB/s (16.6 us at 4K, 4.7 at 1080p); what the all-four "
                                                            "ablation still held was arithmetic (rounding, clamps, tap addresses).  The kernel's time is its arithmetic (30.5 us alone at 4K) "
                                                            "plus the memory time the co-resident waves of a round, in lock step, do not hide; at 1080p — one round — a wave priority per "
                                                            "workgroup takes them out of step: 17.6 -> 14.6 us (pyramid_device.h: stagger_priority); at 4K, four rounds, nothing; DESIGN.md section 4"}}
    try:        # the frame's other full-resolution kernels, from the same committed trace: duration, algorithmic bytes (DESIGN.md's kernel table), fraction of 8 TB/s
        allf = json.load(open(os.path.join(ROOT, "profiles", "r06_warp_facts.json"))).get(f"{w}x{h}", {})
        others = {v["kernel"]: {k2: v[k2] for k2 in ("what", "algo_bytes_per_px", "trace_avg_us", "achieved_GBps", "frac_of_8_TBps")}
                  for k, v in allf.items() if k != "k_warp_bin" and "kernel" in v}
        if others:
            out["other_kernels_from_profiles"] = others
    except (OSError, ValueError, KeyError):
        pass
    out.update({"avg_launch_ms": round(per_launch_ms, 5), "launches_timed": warp_n,
                "frames_by_kernel": dict(zip(("k_warp_bin", "k_warp_tile", "k_warp4"), ctx.warp_counts())),
                **dict(zip(("traffic", "traffic_source"), measured_traffic(ctx.warp_kernel_name(), w, h)))})
    return out


def roofline_isolated(ctx, shapes_or_ts, w, h, reps=60):
    """The same kernel with nothing else on the GPU: one chained sequence of the resident pair on one context, then its last frame's
    launch repeated `reps` times back to back between two events (poppy_hip_time_last_warp).  Outside every timed region; this is the
    duration the kernel traces under profiles/ show (the per-dispatch stamps used inside the timed regions read 10-25 % above it)."""
    ctx.reset(); ctx.render_many(shapes_or_ts[:8], chain=True); ctx.sync()
    try:
        per = ctx.time_last_warp(reps)
    except Exception as e:                                   # the frame took another warp kernel
        return {"error": str(e)}
    ach = WARP_CONTRACT_B_PER_PX * w * h / (per * 1e-3) / 1e9
    return {"avg_launch_ms": round(per, 5), "launches_timed": reps, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
            "what": "the kernel relaunched back to back on the resident pair's 8th chained frame, nothing else running, outside the timed region; its inputs stay in the memory-side cache between relaunches, so this is the kernel's best case (in the chained frame loop the kernel trace under profiles/ shows it ~10 % slower, the per-dispatch stamps of the timed regions ~25 % slower)"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_end_to_end(w=512, h=512, n=30):
    """The WHOLE of poppy::morph on the CPU — the oracle's pair set-up from the raw images + n chained frames — at BASELINE.json configs[0]
    (512x512, 30 frames; the 1080p set-up of the oracle takes minutes): the CPU figure that covers the same work as the GPU `value`."""
    import oracle_lib as O
    from poppy_amd import synth
    a, b = synth.gen_pair(w, h, seed=1234)
    t0 = time.perf_counter()
    frames = O.morph(a, b, n)
    dt = time.perf_counter() - t0
    return {"value": round(len(frames) / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"one {w}x{h} pair, oracle pair set-up from the raw images + {len(frames)} chained frames, {dt:.1f} s on 1 thread (BASELINE.json configs[0]); "
                      "BASELINE.md: the real reference 7.0 frames/s on the build container for the same job"}


def cpu_baseline_and_parity(ctx, a, b, gpu_frames, frames=20):
    """oracle/ on `frames` chained frames of pair 0, fed with the pair state the GPU set-up produced (points, gabor2), one thread;
    its last frame is compared with the same frame of the GPU run."""
    import oracle_lib as O
    O.lib()
    p1, p2 = ctx.pair_points()
    g = ctx.fetch("gabor2")
    cur, pts = a, p1
    t0 = time.perf_counter()
    for j in range(frames):
        s = O.frame_ratio(j, FRAMES, -1.0)
        cur, pts = O.morph_images(cur, b, g, pts, p2, s, s, 64)
    dt = time.perf_counter() - t0
    diff = int((cur != gpu_frames[frames - 1]).sum())
    # every host thread at once: thread k renders its own short chained sequence of the same pair (the oracle library releases the GIL;
    # independent pairs are what the reference's CLI would run as separate processes)
    import threading
    ncores = max(1, min(os.cpu_count() or 1, 64))
    per_thread = 3
    def worker():
        c_, p_ = a, p1
        for j in range(per_thread):
            s_ = O.frame_ratio(j, FRAMES, -1.0)
            c_, p_ = O.morph_images(c_, b, g, p_, p2, s_, s_, 64)
    th = [threading.Thread(target=worker) for _ in range(ncores)]
    t1 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt_all = time.perf_counter() - t1
    base = {"value": round(frames / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
            "all_cores": {"value": round(ncores * per_thread / dt_all, 3), "unit": "frames/s", "cores": ncores,
                          "sample": f"{ncores} threads x {per_thread} chained frames of the same pair at once, {dt_all:.1f} s",
                          "cpu_model": cpu_model()},
            "sample": f"{frames} chained {a.shape[1]}x{a.shape[0]} frames (j = 0..{frames - 1} of the 60-frame sequence of pair 0, the per-frame operator "
                      f"only: the oracle's pair set-up at this size takes minutes) through oracle/liboracle.so, {dt:.1f} s on 1 of {os.cpu_count()} host threads; "
                      "BASELINE.md: the real reference ran 0.87 frames/s incl. set-up on the build container"}
    par = {"frame": frames - 1, "equal": diff == 0, "differing_bytes": diff,
           "what": "frame 19 of pair 0 as written by the GPU run vs the oracle's 20th chained frame from the same pair state"}
    return base, par



def measure_ceilings(torch, dev, w, h):
    """"also report a measured device-copy ceiling from the same run" — and the ceiling that bounds the END-TO-END metric, pinned
    device-to-host copies of whole frames (every frame of `value` leaves through one)."""
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {}
    for key, nbytes, reps in (("device_copy_hbm", 1 << 30, 6), ("device_copy_cache_resident", 64 << 20, 40)):
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
        src.fill_(3); dst.copy_(src); torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        out[key] = {"GBps": round(2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1), "bytes_per_copy": nbytes,
                    "what": "device-to-device copy, read + written bytes / time" + ("; source and destination (2 GB) exceed the 256 MB memory-side cache" if nbytes >= (1 << 30) else "; fits the memory-side cache")}
        del src, dst
    fb = w * h * 3
    d_frame = torch.empty(fb, dtype=torch.uint8, device=dev); d_frame.fill_(7)
    h_frame = torch.empty(fb, dtype=torch.uint8).pin_memory()
    h_frame.copy_(d_frame, non_blocking=True); torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    reps = 100
    e0.record()
    for _ in range(reps):
        h_frame.copy_(d_frame, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    rate = fb * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    out["pinned_d2h_frames"] = {"GBps": round(rate, 1), "frame_bytes": fb, "frames_per_s_cap": round(rate * 1e9 / fb, 1),
                                "what": f"{reps} copies of one {w}x{h} frame from HBM into a pinned host buffer, back to back on one stream: what a writer that takes every frame can be handed at most"}
    del d_frame, h_frame
    return out


def predicted_speedups(job_ms, setup_ms, state_bytes):
    """The N = 2 / 4 / 8 points of the sharded 480-frame job as the one-GPU terms predict them (Amdahl): T_N = serial + (job - set-up) / N."""
    frames_ms = max(job_ms - setup_ms, 0.0)
    bcast_ms = 0.05 + state_bytes / 100e9 * 1e3          # one ncclBroadcast of the pair state: ~100 GB/s per-link-bound ring over xGMI + latency (not measurable on one GPU)
    forms = {"rank0_then_broadcast": setup_ms + bcast_ms,
             # sharded set-up: one image's chain alone is 0.66 of the two side by side (POPPY_SETUP_SERIAL), + the raw-pair broadcast and five small collectives
             "sharded_over_ranks_0_2_estimate": 0.66 * setup_ms + 0.4 + bcast_ms}
    out = {"terms_ms": {"job_on_one_gpu": round(job_ms, 3), "pair_setup": round(setup_ms, 3), "frames_incl_writer": round(frames_ms, 3), "broadcast_estimate": round(bcast_ms, 3)},
           "note": "speed-up over scaling_baseline_480.fps; the one-GPU job is bound by the writer's PCIe path, each rank of an N-GPU job has its own"}
    for name, serial in forms.items():
        out[name] = {f"x{n}": round(job_ms / (serial + frames_ms / n), 2) for n in (2, 4, 8)}
        out[name]["serial_ms"] = round(serial, 3)
    return out


def content_sensitivity(capi, ctx, w, h):
    """Pair set-up and chained-frame time on content other than the 40 flat shapes of the synthetic pair: the medians' whole-wave skips and the ORB
    candidate guess depend on content.  `textured`: poppy_amd/synth.py textured_bgr (hash noise over gradients); `photo`: the reference's own
    sample pair images/amir1.jpg / amir2.jpg (pixels committed as tests/golden/photo_pair_720x405.npz), upscaled to the frame size in integers."""
    from poppy_amd import synth
    cases = {"synthetic_shapes": lambda: synth.gen_pair(w, h, seed=1234),
             "textured": lambda: (synth.textured_bgr(w, h, 7), synth.textured_bgr(w, h, 8)),
             "photo": lambda: synth.photo_pair(w, h)}
    shapes = np.array([capi.lib().poppy_frame_ratio(j, FRAMES, -1.0) for j in range(FRAMES)])
    out = {}
    for name, make in cases.items():
        try:
            a, b = make()
            ctx.pair_begin(a, b)
            t = []
            for _ in range(5):
                t0 = time.perf_counter(); ctx.pair_begin(a, b); t.append((time.perf_counter() - t0) * 1e3)
            p1, _ = ctx.pair_points()
            k0 = ctx.warp_counts()
            ctx.reset(); ctx.render_many(shapes, chain=True); ctx.sync()
            reps, blocks = 2, []                  # the fastest of four blocks of two sequences (a single block has shown 150 - 180 us on one box, run to run)
            for _ in range(4):
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.reset(); ctx.render_many(shapes, chain=True)
                ctx.sync()
                blocks.append((time.perf_counter() - t0) / (reps * FRAMES) * 1e6)
            us = min(blocks)
            k1 = ctx.warp_counts()
            out[name] = {"pair_setup_ms_from_host_images": round(sorted(t)[len(t) // 2], 3), "point_pairs": int(len(p1)), "chained_frame_us": round(us, 1),
                         "frames_by_kernel": dict(zip(("k_warp_bin", "k_warp_tile", "k_warp4"), (int(y - x) for x, y in zip(k0, k1))))}
        except Exception as e:
            out[name] = {"error": str(e)}
    return out


def odd_width(capi, dev_index):
    """Widths that are not multiples of 4 (the reference's CLI pads to the union size of its inputs, src/poppy.cpp:186-239: any width; 5 of its 26 sample images
    are 639 or 749 wide): pair set-up from host images and the chained frame, each beside the next multiple of 4 — 1918 / 1920 x 1080 on the synthetic pair,
    749 / 752 x 480 and 639 / 640 x 480 on the reference's own demo pairs (tests/golden/demo_pairs.npz; the wider canvas = the image padded with its edge).
    Rows of the frame slots' large levels are padded to 16 bytes (kernels.h: level_pitch), so every width takes the fused warp kernel and the wide
    pyramid kernels (`frames_by_kernel`)."""
    from poppy_amd import synth
    def widened(img, w):          # the same content on a canvas w pixels wide (edge replicated): the multiple-of-4 neighbour of an odd-width demo image
        return np.ascontiguousarray(np.pad(img, ((0, 0), (0, w - img.shape[1]), (0, 0)), mode="edge"))
    cars, numbers = synth.demo_pair("cars"), synth.demo_pair("numbers")
    cases = [("1918x1080_synthetic", lambda: synth.gen_pair(1918, 1080, seed=1234)), ("1920x1080_synthetic", lambda: synth.gen_pair(1920, 1080, seed=1234)),
             ("749x480_cars", lambda: cars), ("752x480_cars", lambda: (widened(cars[0], 752), widened(cars[1], 752))),
             ("639x480_numbers", lambda: numbers), ("640x480_numbers", lambda: (widened(numbers[0], 640), widened(numbers[1], 640)))]
    shapes = np.array([capi.lib().poppy_frame_ratio(j, FRAMES, -1.0) for j in range(FRAMES)])
    out = {}
    for name, make in cases:
        ctx = capi.Context(dev_index, number_of_frames=FRAMES)
        try:
            a, b = make()
            ctx.pair_begin(a, b)
            t = []
            for _ in range(5):
                t0 = time.perf_counter(); ctx.pair_begin(a, b); t.append((time.perf_counter() - t0) * 1e3)
            p1, _ = ctx.pair_points()
            k0 = ctx.warp_counts()
            ctx.reset(); ctx.render_many(shapes, chain=True); ctx.sync()
            reps, blocks = 2, []                  # the fastest of four blocks of two sequences (a single block has shown 150 - 180 us on one box, run to run)
            for _ in range(4):
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.reset(); ctx.render_many(shapes, chain=True)
                ctx.sync()
                blocks.append((time.perf_counter() - t0) / (reps * FRAMES) * 1e6)
            us = min(blocks)
            k1 = ctx.warp_counts()
            out[name] = {"pair_setup_ms_from_host_images": round(sorted(t)[len(t) // 2], 3), "point_pairs": int(len(p1)), "chained_frame_us": round(us, 1),
                         "frames_by_kernel": dict(zip(("k_warp_bin", "k_warp_tile", "k_warp4"), (int(y - x) for x, y in zip(k0, k1))))}
        except Exception as e:
            out[name] = {"error": str(e)}
        ctx.close()
    for odd, even in (("1918x1080_synthetic", "1920x1080_synthetic"), ("749x480_cars", "752x480_cars"), ("639x480_numbers", "640x480_numbers")):
        if "chained_frame_us" in out.get(odd, {}) and "chained_frame_us" in out.get(even, {}):
            out[odd]["chained_frame_vs_multiple_of_4"] = round(out[odd]["chained_frame_us"] / out[even]["chained_frame_us"], 3)
    return out


def run_cfg3_4k(capi, torch, dev, steps, check=True):
    """BASELINE.json configs[2]: 3840x2160 pair, 120 phase-mode frames (t_j = j / 120), set-up and writer hand-off inside."""
    w, h, n = 3840, 2160, 120
    a, b = synth_pair(w, h, 0)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ctx = capi.Context(dev.index or 0, number_of_frames=1)
    ts = np.arange(1, n + 1) / float(n + 1)            # 120 in-between frames: 0 < t < 1 (t = 0 and 1 are plain copies, src/poppy.hpp:54-70)

    def step():
        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), w, h)
        return ctx.render_many_counted(ts, chain=False)
    step()
    ctx.set_timing(2)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = 0
    for _ in range(steps):
        got += step()
    ctx.sync()
    dt = time.perf_counter() - t0
    warp_ms, warp_n = next(((ms, c) for nm, ms, c in ctx.timing_summary() if nm == "warp"), (0.0, 0))
    ctx.set_timing(0)
    # the same frames left in HBM, pair resident (frame loop only)
    ctx.render_many(ts, chain=False); ctx.sync()
    t1 = time.perf_counter()
    for _ in range(steps):
        ctx.render_many(ts, chain=False)
    ctx.sync()
    dt_res = time.perf_counter() - t1
    t2 = time.perf_counter()
    ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), w, h)
    setup_ms = (time.perf_counter() - t2) * 1e3
    roof = roofline_of(ctx, warp_ms, warp_n, w, h)
    chained = np.array([capi.lib().poppy_frame_ratio(j, 60, -1.0) for j in range(60)])
    roof["isolated"] = roofline_isolated(ctx, chained, w, h)
    parity = None
    if check:
        try:            # the LAST frame of the 120-frame sequence as the GPU hands it to a writer, against the oracle's frame from the same pair state
            import oracle_lib as O
            p1, p2 = ctx.pair_points()
            g = ctx.fetch("gabor2")
            last = []
            ctx.reset()                          # back to the pair's own first image (the isolated-kernel run above left a chained state)
            ctx.render_many(ts[-1:], chain=False, write=lambda f: last.append(f.copy()))
            want, _ = O.morph_images(a, b, g, p1, p2, float(ts[-1]), float(ts[-1]), 64)
            diff = int((want != last[0]).sum())
            parity = {"frame": n - 1, "equal": diff == 0, "differing_bytes": diff,
                      "what": f"frame {n - 1} (t = {n}/{n + 1}) of the {n}-frame {w}x{h} sequence as written by the GPU vs the oracle's frame from the same pair state"}
        except Exception as e:
            parity = {"error": str(e)}
    out = {"workload": f"{w}x{h} pair, {n} phase-mode frames per step, pair set-up from the raw images and the writer hand-off included",
           "parity_check": parity,
           "value": round(got / dt, 2), "unit": "frames/s", "mpix_per_s": round(got / dt * w * h / 1e6, 1), "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 3), "resident_pair_fps": round(steps * n / dt_res, 1), "pair_setup_ms": round(setup_ms, 2),
           "roofline": roof}
    ctx.close()
    # The same geometry through the library's POOL, as the 1080p headline is: two 3840x2160 pairs per step on two contexts, each pair the whole of poppy::morph in the
    # reference's default mode (set-up from the raw images + 120 CHAINED frames + the writer hand-off): one pair's set-up runs beside the other's frames, whose rate is
    # the D2H path's (a 4K frame renders in ~0.3 ms and leaves in ~0.47 ms).
    try:
        pool = capi.Pool([dev.index or 0], contexts_per_device=2, number_of_frames=n)
        a2, b2 = synth_pair(w, h, 1)
        tc, td = torch.from_numpy(a2).to(dev), torch.from_numpy(b2).to(dev)
        ptrs = [(ta.data_ptr(), tb.data_ptr()), (tc.data_ptr(), td.data_ptr())]
        pool.morph_pairs_device_counted(ptrs, w, h, -1.0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ps = max(2, steps // 2)
        for _ in range(ps):                                           # queued, as the 1080p headline's steps are: one wait at the end
            pool.submit_pairs_device_counted(ptrs, w, h, -1.0)
        k = pool.wait()
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t3
        t3 = time.perf_counter()
        ks = 0
        for _ in range(ps):
            ks += pool.morph_pairs_device_counted(ptrs, w, h, -1.0)   # every step waited for (the form of this figure up to round 5)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t3
        pool.close()
        out["pooled"] = {"value": round(k / dtp, 2), "unit": "frames/s", "workload": f"two {w}x{h} pairs per step on a pool of two contexts, each the whole poppy::morph: set-up from the raw images + {n} chained frames + writer; the steps queued (poppy_hip_pool_submit_pairs), one wait",
                         "steps": ps, "ms_per_step": round(dtp / ps * 1e3, 3), "value_step_synchronous": round(ks / dts, 2)}
        del tc, td
    except Exception as e:
        out["pooled"] = {"error": str(e)}
    del ta, tb
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=45)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the configs[2] (3840x2160 x 120) object")
    ap.add_argument("--headline-only", action="store_true", help="only the timed region (the command the rocprofv3 summaries under profiles/ are taken from)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None, help="frames per sequence (default 60)")
    ap.add_argument("--pairs", type=int, default=None, help="pairs per step at N = 1 (default 6)")
    ap.add_argument("--pairs-per-gpu", type=int, default=8, help="N > 1: pairs per GPU of the configs[4] object")
    ap.add_argument("--total-frames", type=int, default=480, help="N > 1: frames of the ONE phase-mode morph that is sharded by frame range (fixed for every N: strong scaling); also the size of the N = 1 line's scaling_baseline_480")
    ap.add_argument("--shard-setup", action="store_true", help="N > 1: ALSO time the pair set-up spread over ranks 0-2 (poppy_hip_pair_begin_sharded) before the warm-up, compare the point lists it leaves with the rank-0 "
                    "set-up's on every rank, and use it in the timed region when it is both identical and faster.  Opt-in: its RCCL transport has not run on more than one GPU yet (the default is the pair set-up on rank 0 + one broadcast of the pair state)")
    ap.add_argument("--no-shard-setup", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-cpu-end-to-end", action="store_true", help="skip the 512x512x30 whole-morph CPU figure (~40 s of oracle time)")
    ap.add_argument("--contexts", type=int, default=0, help="contexts (host threads) a rank's pairs are spread over; default 6 = one per pair of a step (round 5, ms per step of 6 pairs on one box, ten pools each: 3 contexts 59.2 - 60.4 with the chains of a set-up one after the other — side by side one pool in four ran at 75 -, 4: 57.2 - 59.3, 5: 57.1 - 60.5, 6: 55.9 - 59.4; profiles/r05_notes.md section 6)")
    args = ap.parse_args()
    global W, H, FRAMES, PAIRS
    if args.width and args.height:
        W, H = args.width, args.height
    if args.frames:
        FRAMES = args.frames
    if args.pairs:
        PAIRS = args.pairs

    import torch
    import torch.distributed as dist
    from poppy_amd import capi, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.contexts <= 0:
        args.contexts = 6
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    rehearsal = os.environ.get("POPPY_BENCH_ONE_GPU") == "1" and world > 1      # every rank on GPU 0 over gloo: numbers mean nothing
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # POPPY_BENCH_SHARDED_SELFTEST=1 (with --gpus 1): the N > 1 code path on a world of ONE rank over RCCL — the library's communicator, both forms of
    # the set-up (with --shard-setup the sharded protocol itself: POPPY_HIP_SHARD_WORLD1 makes the library run it on a world of one, all three roles
    # on rank 0, every broadcast and reduction a real RCCL call), the choice between them — for boxes with a single GPU.  Its numbers mean nothing.
    selftest = world == 1 and os.environ.get("POPPY_BENCH_SHARDED_SELFTEST") == "1"
    if selftest:
        os.environ.setdefault("POPPY_HIP_SHARD_WORLD1", "1")
        args.shard_setup = True
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    P = W * H
    if world == 1 and not selftest:
        out = bench_single(args, torch, capi, dev, local)
    else:
        out = bench_sharded(args, torch, dist, capi, sharding, dev, local, rank, world, rehearsal)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or selftest:
        dist.destroy_process_group()


def bench_single(args, torch, capi, dev, local):
    P = W * H
    pairs_host = [synth_pair(W, H, k) for k in range(PAIRS)]
    pairs_dev = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in pairs_host]
    torch.cuda.synchronize()
    ctx = capi.Context(local, number_of_frames=FRAMES)
    # The step's pairs are independent (the CLI's pairs loop, src/poppy.cpp:266-328, has no cross-pair state): the library's pool
    # renders them on CONTEXTS contexts of this GPU, one host thread each, so that one pair's set-up runs beside another's frames.
    ptrs = [(ta.data_ptr(), tb.data_ptr()) for ta, tb in pairs_dev]
    # `value` is measured on the pool the LIBRARY hands out through poppy_hip_pool_create_tuned: its start-up check (about one pool in ten comes out 10 - 25 % slower on
    # every step for as long as it lives: DESIGN.md section 5, `step_ms`; up to three pools, a calibration batch each, the fastest kept) is product behaviour, not a pick
    # made by this script.  `value_unselected` is the same timed region on a pool from plain poppy_hip_pool_create — the first pool made, no check.
    # A step = one batch of PAIRS pairs handed to the pool.  queued (the headline since round 6): the K steps are SUBMITTED one after the other
    # (poppy_hip_pool_submit_pairs returns at once, as a training step's launches do) and the timed region ends with poppy_hip_pool_wait + synchronize — every frame of
    # every step handed to the writer; the pool takes the batches up in order, so a step's last pairs render beside the next step's first set-ups.  not queued
    # (`value_step_synchronous`, the headline's definition up to round 5): every step waits for its own last frame before the next is handed in.
    def timed_region(pool_, timing, ptrs=ptrs, queued=True):
        for _ in range(args.warmup):
            pool_.morph_pairs_device_counted(ptrs, W, H, -1.0)
        if timing:
            pool_.set_timing(2)    # HIP events on the roofline kernel's own dispatch, one launch in seven, on the stream it is launched on
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        n_, per_step = 0, []
        for _ in range(args.steps):
            ts0 = time.perf_counter()
            if queued:
                pool_.submit_pairs_device_counted(ptrs, W, H, -1.0)
            else:
                n_ += pool_.morph_pairs_device_counted(ptrs, W, H, -1.0)  # returns after every frame of the batch was handed to the writer
                per_step.append((time.perf_counter() - ts0) * 1e3)
        if queued:
            n_ = pool_.wait()                                             # every frame of every step has been handed to the writer
        torch.cuda.synchronize()
        return n_, time.perf_counter() - t0_, per_step

    plain = capi.Pool([local], contexts_per_device=args.contexts, number_of_frames=FRAMES)
    plain.morph_pairs_device_counted(ptrs, W, H, -1.0)                    # allocates
    n_plain, dt_plain, _ = timed_region(plain, False)
    # the same pairs handed to the pool in ONE call (informational, not `value`): the contexts take pairs off a shared counter and run out of step — one pair's set-up
    # beside other pairs' frames — where the bench's steps start six set-ups together and then render together
    torch.cuda.synchronize()
    t1c = time.perf_counter()
    n_one = plain.morph_pairs_device_counted(ptrs * args.steps, W, H, -1.0)
    torch.cuda.synchronize()
    dt_one = time.perf_counter() - t1c
    plain.close()
    pool = capi.Pool([local], contexts_per_device=args.contexts, tuned_for=(W, H), number_of_frames=FRAMES)
    pool_selection = {"candidates_ms_per_batch": [round(x, 2) for x in pool.candidates_ms], "kept": pool.kept, "made_by": "poppy_hip_pool_create_tuned",
                      "what": "pools the library made at start-up and timed on its built-in calibration batch (two pairs per context, twice); the fastest is the pool"}
    written, dt, _ = timed_region(pool, True)
    warp_ms, warp_n = next(((ms, c) for nm, ms, c in pool.timing_summary() if nm == "warp"), (0.0, 0))
    pool.set_timing(0)
    n_sync, dt_sync, step_ms = timed_region(pool, False, queued=False)
    assert written == args.steps * PAIRS * FRAMES, (written, args.steps * PAIRS * FRAMES)
    fps = written / dt
    roof = roofline_of(pool, warp_ms, warp_n, W, H)
    pool.close()

    out = {
        "metric": "morph frames/sec at 1080p, 60-frame sequence; Mpix/s warped" if (W, H, FRAMES) == (1920, 1080, 60) else f"morph frames/sec at {W}x{H}, {FRAMES}-frame sequence; Mpix/s warped",
        "value": round(fps, 2), "unit": "frames/s", "mpix_per_s": round(fps * P / 1e6, 1),
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "value_unselected": round(n_plain / dt_plain, 2),
        "value_step_synchronous": {"fps": round(n_sync / dt_sync, 2), "ms_per_step": round(dt_sync / args.steps * 1e3, 3),
                                   "what": "the same pool, the same steps, but every step waits for its own last frame before the next batch is handed in "
                                           "(poppy_hip_pool_morph_pairs per step): `value`'s definition up to round 5 — the contexts then start every step with a round of "
                                           "set-ups together; `value` queues the steps (poppy_hip_pool_submit_pairs) and waits once, at the end of the timed region"},
        "pairs_in_one_call": {"fps": round(n_one / dt_one, 2), "pairs": len(ptrs) * args.steps,
                              "what": "the timed region's pairs (steps x pairs per step) handed to the unselected pool in ONE poppy_hip_pool_morph_pairs call instead of one call per step: "
                                      "informational — the pool without any step structure; not `value`"},
        "pool_selection": pool_selection,
        "step_ms": {"min": round(min(step_ms), 3), "median": round(sorted(step_ms)[len(step_ms) // 2], 3), "max": round(max(step_ms), 3),
                    "what": "the steps of `value_step_synchronous` one by one (diagnostic; the queued steps of `value` have no ends of their own)"},
        "timed_region_s": round(dt, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "scaling_note": "the --gpus N lines shard ONE fixed 480-frame 1080p morph by frame range (total work fixed: strong); the N = 1 point of that series is "
                        "`scaling_baseline_480.fps` of this line (the same job on one GPU), not this line's `value`, which is BASELINE.json configs[1]",
        "dtype": "u8+f32",
        "dtype_note": "u8 pixels and fixed-point remap, f32 pyramid and unsharp, f64 Gabor sums; bit-compatible with the reference (no FMA contraction)",
        "data": "synthetic (integer-defined shapes pairs, seeds 1234+k: poppy_amd/synth.py); point sets and mask field come from the real pair set-up",
        "run": {"hostname": __import__("socket").gethostname(), "run_id": os.environ.get("POPPY_RUN_ID"),
                "note": "run_id is set by tools/profile_round.sh: the one gpurun call that wrote profiles/<tag>_trace.md, _pmc.md and the <round>_warp_*.json this line quotes "
                        "(null: a bench run outside that script — then roofline.from_profiles quotes the committed profiles of another run, its own run_id says which)"},
        "config": {"workload": f"{W}x{H} pairs, {FRAMES}-frame morph each, default chained mode (BASELINE.json configs[1]): per step {PAIRS} pairs x "
                               "(pair set-up from the raw images + 60 chained frames handed to a writer through pinned host memory), pyramid_levels 64",
                   "pairs_per_step": PAIRS, "steps_queued": True, "contexts": args.contexts, "frames_per_pair": FRAMES, "mode": "chain", "includes": ["pair set-up", "frame loop", "writer hand-off (D2H)"],
                   "parallelism": "1 GPU"},
        "roofline": roof,
    }
    if args.headline_only:
        ctx.close()
        return out

    # ---- extras, all outside the timed region -------------------------------------------------------------------------
    ta, tb = pairs_dev[0]
    a_h, b_h = pairs_host[0]
    ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
    p1, p2 = ctx.pair_points()
    out["config"]["point_pairs_pair0"] = int(len(p1))
    shapes = np.array([capi.lib().poppy_frame_ratio(j, FRAMES, -1.0) for j in range(FRAMES)])
    reps = max(args.steps, 10)
    # the same step on ONE context, pair after pair (the reference's own order)
    def seq_step():
        n = 0
        for ta_, tb_ in pairs_dev:
            ctx.pair_begin_device(ta_.data_ptr(), tb_.data_ptr(), W, H)
            n += ctx.morph_frames_counted(-1.0)
        return n
    seq_step()
    t1 = time.perf_counter()
    k = 0
    for _ in range(max(3, args.steps // 4)):
        k += seq_step()
    out["sequential_fps"] = round(k / (time.perf_counter() - t1), 1)
    ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
    # per-frame operator on the resident pair, frames left in HBM (round 1's `value`)
    ctx.reset(); ctx.render_many(shapes, chain=True); ctx.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.reset(); ctx.render_many(shapes, chain=True)
    ctx.sync()
    out["resident_pair_fps"] = round(reps * FRAMES / (time.perf_counter() - t1), 1)
    out["roofline"]["isolated"] = roofline_isolated(ctx, shapes, W, H)
    # the same with the writer hand-off
    ctx.reset(); ctx.render_many_counted(shapes, chain=True)
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.reset(); ctx.render_many_counted(shapes, chain=True)
    ctx.sync()
    out["resident_pair_fps_with_writer"] = round(reps * FRAMES / (time.perf_counter() - t1), 1)
    # pair set-up alone
    t1 = time.perf_counter()
    for _ in range(5):
        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
    out["pair_setup_ms"] = round((time.perf_counter() - t1) / 5 * 1e3, 2)
    # N = 1 point of the sharded workload (configs[3]): a 60-frame phase-mode morph incl. set-up and writer
    ph = np.arange(FRAMES) / float(FRAMES)

    def phase_step():
        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
        return ctx.render_phases(ph, counted=True)        # frame 0 (t = 0) is a copy of image 1 (src/poppy.hpp:54-62)
    phase_step()
    t1 = time.perf_counter()
    for _ in range(reps):
        phase_step()
    ctx.sync()
    dtp = time.perf_counter() - t1
    ctx.render_many(ph[1:], chain=False); ctx.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.render_many(ph[1:], chain=False)
    ctx.sync()
    out["scaling_baseline"] = {"workload": f"one {FRAMES}-frame phase-mode morph per step, set-up and writer included",
                               "fps": round(reps * FRAMES / dtp, 1), "frames_only_fps": round(reps * (FRAMES - 1) / (time.perf_counter() - t1), 1)}
    # THE denominator of the --gpus N lines (north_star: ">= 7x at 8 GPUs vs 1 GPU on a 480-frame 1080p morph"): the same fixed job —
    # one TOTAL-frame phase-mode morph, set-up from the raw pair + every frame handed to the writer — on this one GPU
    TOTAL = args.total_frames
    ph480 = np.arange(TOTAL) / float(TOTAL)

    def job480():
        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
        return ctx.render_phases(ph480, counted=True)
    job480()
    r480 = max(3, min(reps, 8))
    t1 = time.perf_counter()
    for _ in range(r480):
        job480()
    ctx.sync()
    dt480 = time.perf_counter() - t1
    out["scaling_baseline_480"] = {"workload": f"ONE {TOTAL}-frame phase-mode morph of a {W}x{H} pair on one GPU: pair set-up from the raw images + {TOTAL} frames handed to the writer "
                                               "(the job every --gpus N line shards by frame range; its `value` / this `fps` = the speed-up)",
                                   "fps": round(r480 * TOTAL / dt480, 1), "ms_per_job": round(dt480 / r480 * 1e3, 3), "jobs_timed": r480,
                                   "serial_part_ms": out["pair_setup_ms"],
                                   "amdahl_note": "with the set-up serial, N GPUs cannot beat (set-up + frames) / (set-up + frames / N)"}
    if not args.headline_only:
        # parity of THAT job (outside every timed region): the whole job once more through a capturing writer; its middle frame against the oracle's
        # frame from the same pair state, frame 0 against image 1 (the phase == 0 copy, src/poppy.hpp:54-70)
        import oracle_lib as O
        mid = TOTAL // 2
        kept, k480 = {}, [0]

        def keep480(f):
            if k480[0] in (0, mid):
                kept[k480[0]] = f.copy()
            k480[0] += 1
        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
        ctx.render_phases(ph480, write=keep480)
        a0_h, b0_h = ta.cpu().numpy().reshape(H, W, 3), tb.cpu().numpy().reshape(H, W, 3)
        p1_, p2_ = ctx.pair_points()
        want, _ = O.morph_images(a0_h, b0_h, ctx.fetch("gabor2"), p1_, p2_, float(ph480[mid]), float(ph480[mid]), 64)
        out["scaling_baseline_480"]["parity_check"] = {
            "frames_written": k480[0], "frame0_is_image1": bool(np.array_equal(kept.get(0), a0_h)),
            "frame": mid, "equal": bool(np.array_equal(kept.get(mid), want)),
            "what": f"frame {mid} (t = {float(ph480[mid])}) of the {TOTAL}-frame job as handed to a writer vs the oracle's phase-mode frame from the same pair state"}
    out["scaling_baseline_480"]["predicted_speedup"] = predicted_speedups(dt480 / r480 * 1e3, out["pair_setup_ms"], capi.pair_state_bytes(W, H))
    # the headline's step on photographs (round-5 review): six pairs = the reference's own sample pair (images/amir1.jpg / amir2.jpg, committed as pixels, upscaled in
    # integers) six times — pairs are independent —, the same pool form, the same number of steps; outside `value`'s timed region
    try:
        from poppy_amd import synth as _synth
        pa, pb = _synth.photo_pair(W, H)
        tpa, tpb = torch.from_numpy(pa).to(dev), torch.from_numpy(pb).to(dev)
        pptrs = [(tpa.data_ptr(), tpb.data_ptr())] * PAIRS
        ppool = capi.Pool([local], contexts_per_device=args.contexts, number_of_frames=FRAMES)
        ppool.morph_pairs_device_counted(pptrs, W, H, -1.0)
        nph, dph, _ = timed_region(ppool, False, pptrs)
        ppool.close()
        out["value_photo"] = {"fps": round(nph / dph, 2), "ms_per_step": round(dph / args.steps * 1e3, 3),
                              "what": f"`value`'s step ({PAIRS} pairs x (set-up + {FRAMES} chained frames + writer), {args.contexts} contexts, a pool from plain poppy_hip_pool_create: compare `value_unselected`) "
                                      "on the reference's sample photographs instead of the synthetic shapes: their medians stay on k_median_u8 (set-up ~4.0 ms against ~2.6)"}
        del tpa, tpb
    except Exception as e:      # (the fixture is data under tests/golden; a tree without it still benches)
        out["value_photo"] = {"error": str(e)}
    # the headline step once more with the raw pairs copied from pinned host memory inside the step (the reference's morph() takes host images)
    pinned = [(torch.from_numpy(a).pin_memory(), torch.from_numpy(b).pin_memory()) for a, b in pairs_host]
    pool2 = capi.Pool([local], contexts_per_device=args.contexts, number_of_frames=FRAMES)

    def step_h2d():
        for (da, db), (ha, hb) in zip(pairs_dev, pinned):
            da.copy_(ha, non_blocking=True); db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
        return pool2.morph_pairs_device_counted(ptrs, W, H, -1.0)
    step_h2d()
    rh = max(3, args.steps // 4)
    t1 = time.perf_counter()
    kh = 0
    for _ in range(rh):
        kh += step_h2d()
    dth = time.perf_counter() - t1
    t1 = time.perf_counter()
    for _ in range(rh):
        for (da, db), (ha, hb) in zip(pairs_dev, pinned):
            da.copy_(ha, non_blocking=True); db.copy_(hb, non_blocking=True)
        torch.cuda.synchronize()
    # the headline itself from HOST images: the queued steps of `value`, but every pair handed to the pool as two pinned host images (poppy_hip_morph per pair: the
    # set-up uploads them on its own streams, beside other pairs' frames — the link is full duplex — and behind the same set-up gate)
    hptrs = [(ha.data_ptr(), hb.data_ptr()) for ha, hb in pinned]
    pool2.submit_pairs_device_counted(hptrs, W, H, -1.0, on_device=False); pool2.wait()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        pool2.submit_pairs_device_counted(hptrs, W, H, -1.0, on_device=False)
    nhost = pool2.wait()
    torch.cuda.synchronize()
    dthost = time.perf_counter() - t1
    assert nhost == args.steps * PAIRS * FRAMES, (nhost, args.steps * PAIRS * FRAMES)
    out["value_from_host_images"] = {"fps": round(nhost / dthost, 2), "ms_per_step": round(dthost / args.steps * 1e3, 3),
                                     "what": f"`value`'s timed region (the same {args.steps} queued steps of {PAIRS} pairs on a pool from plain poppy_hip_pool_create: compare `value_unselected`) with every pair "
                                             "handed over as two images in pinned HOST memory, as the reference's morph() is: the uploads (2 x %.1f MB per pair) are inside, on the set-ups' own streams" % (P * 3 / 1e6)}
    out["h2d"] = {"value_incl_h2d": round(kh / dth, 1), "h2d_ms_per_pair": round((time.perf_counter() - t1) / rh / PAIRS * 1e3, 3),
                  "what": "the step of `value_step_synchronous` (compare with that figure) with the raw pairs (2 x %.1f MB each) copied host -> device inside the step, serially before the pool starts; the timed region of `value` starts with them resident in HBM" % (P * 3 / 1e6)}
    pool2.close()
    out["config"]["workload"] += (f"; the {PAIRS} pairs of a step are rendered by a pool of {args.contexts} contexts (pooled headline); the same step on ONE context, "
                                  f"pair after pair, runs at sequential_fps = {out['sequential_fps']} frames/s")
    # per-kernel breakdown: one untimed sequence with events around every kernel group
    ctx.set_timing(1)
    ctx.reset(); ctx.render_many(shapes, chain=True); ctx.sync()
    kernels = {}
    tot = 0.0
    for name, ms, cnt in ctx.timing_summary():
        per = ms / max(cnt, 1)
        tot += ms
        ab = ALGO_BYTES_PER_PX.get(name, 0.0) * P
        kernels[name] = {"avg_ms": round(per, 5), "launch_groups": cnt, "algo_MB": round(ab / 1e6, 3),
                         "GBps": round(ab / (per * 1e-3) / 1e9, 1) if per > 0 else None}
    ctx.set_timing(0)
    out["kernels"] = kernels
    out["kernel_groups_ms_per_frame"] = round(tot / FRAMES, 4)

    if not args.no_cpu_baseline:
        try:
            ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
            got = []
            ctx.reset()
            ctx.render_many(shapes[:20], chain=True, write=lambda f: got.append(f.copy()))
            out["cpu_baseline"], out["parity_check"] = cpu_baseline_and_parity(ctx, a_h, b_h, got, 20)
            if not args.no_cpu_end_to_end:
                out["cpu_baseline"]["end_to_end"] = cpu_end_to_end()
        except Exception as e:   # the checker is optional for the measurement, never for parity
            out["cpu_baseline"] = {"value": None, "error": str(e)}
    try:
        out["content_sensitivity"] = content_sensitivity(capi, ctx, W, H)
    except Exception as e:
        out["content_sensitivity"] = {"error": str(e)}
    ctx.close()
    try:
        out["odd_width"] = odd_width(capi, local)
    except Exception as e:
        out["odd_width"] = {"error": str(e)}
    try:
        out["ceilings"] = measure_ceilings(torch, dev, W, H)
        cap = out["ceilings"]["pinned_d2h_frames"]["frames_per_s_cap"]
        out["ceilings"]["value_frac_of_d2h_cap"] = round(out["value"] / cap, 3)
        out["ceilings"]["scaling_baseline_480_frac_of_d2h_cap"] = round(out["scaling_baseline_480"]["fps"] / cap, 3)
        out["roofline"]["frac_of_device_copy_ceiling"] = round(out["roofline"]["achieved"] / out["ceilings"]["device_copy_hbm"]["GBps"], 4)
    except Exception as e:
        out["ceilings"] = {"error": str(e)}
    if not args.no_4k and (W, H) == (1920, 1080):
        try:
            out["cfg3_4k"] = run_cfg3_4k(capi, torch, dev, max(3, min(args.steps, 10)), check=not args.no_cpu_baseline)
            if "error" not in out.get("ceilings", {}):
                c4 = measure_ceilings(torch, dev, 3840, 2160)["pinned_d2h_frames"]
                out["cfg3_4k"]["d2h_cap"] = c4
                out["cfg3_4k"]["value_frac_of_d2h_cap"] = round(out["cfg3_4k"]["value"] / c4["frames_per_s_cap"], 3)
                if "value" in out["cfg3_4k"].get("pooled", {}):
                    out["cfg3_4k"]["pooled"]["frac_of_d2h_cap"] = round(out["cfg3_4k"]["pooled"]["value"] / c4["frames_per_s_cap"], 3)
                out["cfg3_4k"]["roofline"]["frac_of_device_copy_ceiling"] = round(out["cfg3_4k"]["roofline"]["achieved"] / out["ceilings"]["device_copy_hbm"]["GBps"], 4)
        except Exception as e:
            out["cfg3_4k"] = {"error": str(e)}
    return out


def bench_sharded(args, torch, dist, capi, sharding, dev, local, rank, world, rehearsal):
    """configs[3]: ONE args.total_frames-frame phase-mode morph for every N; set-up on rank 0, one broadcast of the pair state, frame shares per rank."""
    P = W * H
    total = args.total_frames
    cdev = torch.device("cpu") if rehearsal else dev
    ctx = capi.Context(local, number_of_frames=1)
    if rank == 0:
        a, b = synth_pair(W, H, 0)
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ts = sharding.phase_share(rank, world, total)
    link = sharding.PairLink(torch, dist, capi, ctx, rank, world, W, H, cdev, use_library=not rehearsal)
    t_setup = [0.0]; t_bcast = [0.0]; t_frames = [0.0]
    comm_info = ctx.comm_info() if link.library else None
    if comm_info is not None and comm_info[3] >= 0:          # the communicator the library exchanges the pair through IS this job: ncclCommCount == N on every rank
        assert comm_info[3] == world and comm_info[1] == world, ("ncclCommCount != --gpus", comm_info, world)
    # every rank's share of the job's frames (the partition is a pure function of (rank, world, total): checked here across the ranks)
    shares = torch.zeros(world, dtype=torch.int32, device=cdev)
    shares[rank] = len(ts)
    dist.all_reduce(shares)
    frame_shares = [int(x) for x in shares.cpu().tolist()]
    assert sum(frame_shares) == total, (frame_shares, total)

    shard_setup = link.library and args.shard_setup and not args.no_shard_setup

    def fence():
        ctx.sync(); torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # The set-up has two forms — spread over ranks 0-2 through the library's communicator, or all of it on rank 0 followed by one broadcast of the
    # pair state.  Which is faster depends on the links' collective latency; both are run before the warm-up, the point lists they leave on rank 0
    # are compared, and the faster one (max over ranks) is the one the timed region uses.  Everything is agreed through reductions: no rank
    # decides alone.
    variants = None
    if shard_setup:
        def run_form(sharded, reps=3):
            fence()
            t0 = time.perf_counter()
            for _ in range(reps):
                if sharded:
                    ctx.pair_begin_sharded(ta.data_ptr() if rank == 0 else None, tb.data_ptr() if rank == 0 else None, W, H, 0)
                else:
                    if rank == 0:
                        ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
                    link.broadcast(root=0)
            fence()
            return link.max_time(time.perf_counter() - t0) / reps
        run_form(True, 1); run_form(False, 1)                          # first calls allocate
        ms_sharded = run_form(True) * 1e3
        pts_sharded = ctx.pair_points()
        ms_rank0 = run_form(False) * 1e3
        pts_rank0 = ctx.pair_points()
        same = int(all(np.array_equal(x, y) for x, y in zip(pts_sharded, pts_rank0)))
        flag = torch.tensor([same], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        same = bool(int(flag.item()))
        variants = {"sharded_ms": round(ms_sharded, 3), "rank0_then_broadcast_ms": round(ms_rank0, 3), "same_point_lists_on_every_rank": same,
                    "sharded_protocol_runs_on_rank0": capi.sharded_setups()}
        shard_setup = same and ms_sharded <= ms_rank0
        variants["used"] = "sharded" if shard_setup else "rank 0 + broadcast"

    def step():
        t0 = time.perf_counter()
        if shard_setup:          # the set-up itself spread over the ranks (image 1 on rank 0, image 2 on rank 1, the mask field on rank 2), one collective call
            ctx.pair_begin_sharded(ta.data_ptr() if rank == 0 else None, tb.data_ptr() if rank == 0 else None, W, H, 0)
            t1 = t2 = time.perf_counter()
        else:
            if rank == 0:
                ctx.pair_begin_device(ta.data_ptr(), tb.data_ptr(), W, H)
            t1 = time.perf_counter()
            link.broadcast(root=0)
            t2 = time.perf_counter()
        n = ctx.render_phases(ts, counted=True)        # global frame 0 (t = 0) is the reference's phase == 0 short-circuit: a copy of image 1
        t3 = time.perf_counter()
        t_setup[0] += t1 - t0; t_bcast[0] += t2 - t1; t_frames[0] += t3 - t2
        return n

    for _ in range(args.warmup):
        step()
    ctx.set_timing(2)          # HIP events on the roofline kernel's own dispatch (one launch in seven), as at N = 1; rank 0's launches are reported
    t_setup[0] = t_bcast[0] = t_frames[0] = 0.0
    fence()
    t0 = time.perf_counter()
    n = 0
    for _ in range(args.steps):
        n += step()
    fence()
    dt = time.perf_counter() - t0
    warp_ms, warp_n = next(((ms, c) for nm, ms, c in ctx.timing_summary() if nm == "warp"), (0.0, 0))
    ctx.set_timing(0)
    assert n == args.steps * len(ts)
    dt_max = link.max_time(dt)
    # steady state: the broadcast pair resident, frames only (in HBM)
    ctx.render_many(ts[1:] if len(ts) and ts[0] == 0.0 else ts, chain=False)
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        ctx.render_many(ts[1:] if len(ts) and ts[0] == 0.0 else ts, chain=False)
    fence()
    dt_f = link.max_time(time.perf_counter() - t1)
    # configs[4]: independent pairs spread over the GPUs (8 per GPU: 64 pairs on 8), each a whole chained poppy::morph from the raw
    # images, handed out to the contexts of the library's pool on that GPU; no communication at all
    ppg = args.pairs_per_gpu
    mine = sharding.pair_range(rank, world, ppg * world)
    pool = capi.Pool([local], contexts_per_device=args.contexts, number_of_frames=FRAMES)
    dev_pairs = [tuple(torch.from_numpy(x).to(dev) for x in synth_pair(W, H, k)) for k in mine]
    ptrs = [(a_.data_ptr(), b_.data_ptr()) for a_, b_ in dev_pairs]
    pool.morph_pairs_device_counted(ptrs, W, H, -1.0)            # untimed: every context of the pool allocates its pair state and frame slots
    fence()
    t2 = time.perf_counter()
    reps = max(1, args.steps // 4)
    for _ in range(reps):                                          # queued, as the --gpus 1 line's `value` is: one wait at the end
        pool.submit_pairs_device_counted(ptrs, W, H, -1.0)
    nfr = pool.wait()
    fence()
    dt_p = link.max_time(time.perf_counter() - t2)
    pool.close()
    assert nfr == reps * len(ptrs) * FRAMES
    out = None
    if rank == 0:
        fps = args.steps * total / dt_max
        k = float(args.steps)
        out = {
            "metric": "morph frames/sec at 1080p; Mpix/s warped (one %d-frame phase-mode morph sharded by frame range)" % total,
            "value": round(fps, 2), "unit": "frames/s", "mpix_per_s": round(fps * P / 1e6, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8+f32",
            "data": "synthetic (integer-defined shapes pair, poppy_amd/synth.py); pair state from the real pair set-up on rank 0",
            "config": {"workload": f"{W}x{H} pair, ONE {total}-frame phase-mode morph (BASELINE.json configs[3]), the same job for every N: per step the pair set-up "
                                   + ("spread over ranks 0-2 (image 1 / image 2 / mask field; raw pair, detail values, keypoints, point sets and mask field exchanged through the library's RCCL communicator), "
                                      if shard_setup else f"on rank 0, one broadcast of the pair state ({link.how}), ") +
                                   f"then each GPU renders its contiguous share of the {total} frames and hands them to a writer",
                       "total_frames": total, "frames_per_gpu": len(ts), "mode": "phase", "parallelism": f"frame-range x{world}", "setup": "sharded over ranks 0-2" if shard_setup else "rank 0", "broadcast": link.how,
                       "broadcast_bytes": link.nbytes},
            "roofline": roofline_of(ctx, warp_ms, warp_n, W, H),
            "frames_only_fps": round(args.steps * total / dt_f, 1),
            "setup_forms": variants if variants is not None else {"used": "rank 0 + broadcast", "note": "the library's communicator is not in use (or --no-shard-setup): one form only"},
            "amdahl": {"rank0_setup_ms": round(t_setup[0] / k * 1e3, 3), "rank0_broadcast_ms": round(t_bcast[0] / k * 1e3, 3),
                       "rank0_frames_ms": round(t_frames[0] / k * 1e3, 3),
                       "note": "host-side wall time on rank 0 per step; the set-up and the broadcast are the serial part of the job: N GPUs cannot beat "
                               "(set-up + frames) / (set-up + broadcast + frames / N) over the --gpus 1 line's scaling_baseline_480"},
            "frame_shares": frame_shares,
            "communicator": ({"rank_world_given": list(comm_info[:2]), "nccl_rank_count": list(comm_info[2:]), "count_equals_gpus": comm_info[3] == world} if comm_info is not None
                             else {"note": "torch.distributed transport (no library communicator in this run)"}),
            # what THIS run's own rank-0 terms predict for the job against one GPU doing everything: (set-up + N x this rank's frames) / (set-up + broadcast + this rank's frames);
            # read beside value / scaling_baseline_480.fps of the --gpus 1 line (the measured speed-up) and that line's predicted_speedup
            "predicted_speedup_from_this_runs_terms": round((t_setup[0] + world * t_frames[0]) / max(t_setup[0] + t_bcast[0] + t_frames[0], 1e-9), 2),
            "cfg4_note": "value = configs[3] (one %d-frame morph sharded by frame range); speed-up = value / scaling_baseline_480.fps of the --gpus 1 line" % total,
            "cfg5_pairs": {"workload": f"BASELINE.json configs[4]: {ppg * world} independent {W}x{H} pairs x {FRAMES} chained frames, {ppg} pairs per GPU on "
                                       f"{args.contexts} contexts each, set-up from the raw images and the writer hand-off included, no communication",
                           "value": round(reps * ppg * world * FRAMES / dt_p, 2), "unit": "frames/s", "pairs": ppg * world, "steps": reps,
                           "ms_per_step": round(dt_p / reps * 1e3, 3), "scaling": "weak", "scaling_baseline": "the `value` of the --gpus 1 line (same per-GPU work)"},
            "scaling_note": "read `value` against `scaling_baseline_480.fps` of the --gpus 1 line (the same fixed job on one GPU), not against its chained `value`; `cfg5_pairs.value` against that chained `value` x N",
        }
    ctx.close()
    return out


if __name__ == "__main__":
    main()
