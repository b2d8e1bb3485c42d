#!/usr/bin/env python3
"""bench.py — morph frames/sec at 1080p (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one 60-frame morph of a synthetic 1080p pair on every rank:
  N = 1 : the reference's default CLI mode (chained: frame j warps frame j-1; src/poppy.hpp:177-219);
  N > 1 : frame-range sharding.  The job is ONE 60*N-frame phase-mode morph (N = 8 -> the 480-frame morph of
          BASELINE.json configs[3]); rank r renders frames [60r, 60r+60) with phase t_j = j / (60 N), each
          equal to the reference call morph(img1, img2, ..., phase = t_j) with number_of_frames = 1.  The source
          pair and the mask field are broadcast once from rank 0 over RCCL (torch.distributed "nccl"); there
          is no data-path collective afterwards.  Per-GPU work is fixed => "scaling": "weak".
Frames stay in HBM (inputs are resident before the timed region, outputs are not copied back); the PCIe-inclusive
rate is printed as an extra key, never as `value`.

Output: ONE JSON line on rank 0 (see the driver contract), extended with
  roofline     : the fused map+remap kernel (k_warp_tile; k_warp4 for odd geometries): 16 B/px (faithful path) + 8 B/px for the lbmask it
                 computes on the way (m2 in, mask out); the id map is frame-tagged and never cleared,
                 average launch duration measured live with HIP events on the library's stream (one launch in seven of
                 the timed region carries the stamps: `launches_timed`);
  kernels      : the same for every kernel group of the frame (one extra untimed step; "warp" is the timed region's);
  cpu_baseline : oracle/ (CPU restatement, "port") timed on this box's host cores on a bounded sample.
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

W, H, FRAMES, NPTS = 1920, 1080, 60, 436
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic HBM bytes per frame of each kernel group, per full-resolution pixel P
ALGO_BYTES_PER_PX = {
    "upload+clear": 0.0,                       # wait for the plan upload (the id map is frame-tagged: no clear)
    "raster": 4.0,                             # every pixel's id written once
    "warp": 24.0,                              # triMap 4 + c1 3 + c2 3 in, trImg1 3 + trImg2 3 out; lbmask rider: m2 4 in, mask 4 out
    "pyrdown": (6 + 4) + (24 + 4) / 4 * (4 / 3),      # level 0: u8 L,R + mask in; quarter-size f32 L,R,M out; geometric tail
    "pyr_tail": 0.0,
    "collapse": (6 + 4 + 12) + (36 / 4) * (4 / 3) + 12 * (1 / 3),   # G_i (u8 at level 0), mask, lower level L,R,B in; B_i out
    "unsharp": 12 + 3,                         # fused tile kernel: lapBlend f32x3 in, frame u8x3 out
}


def synth_inputs():
    from poppy_amd import synth
    a, b = synth.gen_pair(W, H)
    g = synth.unit_field(W, H, 11)
    p1, p2 = synth.point_pairs(W, H, NPTS, seed=5, dup=0, oob=0)       # 436 matches + 4 corners
    return a, b, g, p1, p2


def cpu_baseline(a, b, g, p1, p2, frames=20):
    """oracle/ timed on the host: `frames` chained 1080p frames of the same workload, one thread."""
    import oracle_lib as O
    from poppy_amd import capi
    O.lib()
    L = capi.lib()
    cur, pts = a, p1
    t0 = time.perf_counter()
    for j in range(1, frames + 1):
        s = L.poppy_frame_ratio(j, FRAMES, -1.0)
        cur, pts = O.morph_images(cur, b, g, pts, p2, s, s, 64)
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{frames} chained 1080p frames (j=1..{frames} of the 60-frame sequence) through oracle/liboracle.so, "
                      f"{dt:.1f} s on 1 of {os.cpu_count()} host threads"}


def cpu_baseline_threads(a, b, g, p1, p2, threads, per_thread=2):
    """The same oracle on independent phase-mode frames (t_j = j / 60), one host thread each: what the CPU path does with all
    the cores it is given (ctypes releases the GIL for the duration of a frame)."""
    import threading
    import oracle_lib as O
    O.lib()

    def work(k):
        for i in range(per_thread):
            t = ((k * per_thread + i) % (FRAMES - 1) + 1) / float(FRAMES)
            O.morph_images(a, b, g, p1, p2, t, t, 64)
    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": threads * per_thread / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{threads * per_thread} independent phase-mode 1080p frames, {per_thread} per thread on {threads} of "
                      f"{os.cpu_count()} host threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["chain", "phase"], default=None, help="default: chain at N=1, phase at N>1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-setup", action="store_true", help="skip the pair set-up timing (poppy_hip_pair_begin)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed region and the per-kernel step (no phase-mode / batched / download / set-up extras): "
                         "the command the rocprofv3 summaries under profiles/ are taken from")
    ap.add_argument("--width", type=int, default=None, help="override the 1080p headline size (e.g. 3840 for BASELINE configs[2])")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 60)")
    args = ap.parse_args()
    global W, H, FRAMES
    if args.width and args.height:
        W, H = args.width, args.height
    if args.frames:
        FRAMES = args.frames

    import torch
    import torch.distributed as dist
    from poppy_amd import capi, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    # POPPY_BENCH_ONE_GPU=1 is a rehearsal of the N > 1 path on a single-GPU box (every rank on GPU 0, gloo as the
    # backend, the pair staged through host memory for the broadcast); its numbers mean nothing
    rehearsal = os.environ.get("POPPY_BENCH_ONE_GPU") == "1" and world > 1
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    mode = args.mode or ("chain" if world == 1 else "phase")
    if mode == "chain" and world > 1:
        raise SystemExit("chained mode is sequential by construction (SURVEY.md F5); use --mode phase for N>1")

    # ---- inputs: generated on rank 0, broadcast once over RCCL, resident in HBM before timing --------------
    if rehearsal:
        cpu = torch.device("cpu")
        host = sharding.pair_tensors(torch, cpu, W, H, NPTS + 4, synth_inputs() if rank == 0 else None)
        sharding.broadcast_pair(dist, host, src=0)
        ta, tb, tg, tp = [t.to(dev) for t in host]
    else:
        ta, tb, tg, tp = sharding.pair_tensors(torch, dev, W, H, NPTS + 4, synth_inputs() if rank == 0 else None)
        if world > 1:
            sharding.broadcast_pair(dist, (ta, tb, tg, tp), src=0)
    torch.cuda.synchronize()
    pts = tp.cpu().numpy()
    p1r, p2r = np.ascontiguousarray(pts[0]), np.ascontiguousarray(pts[1])

    ctx = capi.Context(local, number_of_frames=FRAMES)
    ctx.pair_load_device(ta.data_ptr(), tb.data_ptr(), tg.data_ptr(), W, H, p1r, p2r)

    total_frames = FRAMES * world
    if mode == "chain":
        shapes = np.array([capi.lib().poppy_frame_ratio(j, FRAMES, -1.0) for j in range(FRAMES)])
    else:
        shapes = sharding.phase_schedule(rank, world, FRAMES)

    def step():
        ctx.reset()
        ctx.render_many(shapes, chain=(mode == "chain"))

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_timing(2)          # HIP events on the roofline kernel's own dispatch (k_warp_tile), one launch in seven, on the stream it is launched on
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    warp_ms, warp_n = next(((ms, cnt) for name, ms, cnt in ctx.timing_summary() if name == "warp"), (0.0, 0))
    # per-kernel breakdown: one extra, untimed step with events around every kernel group (frames are then issued
    # launch by launch instead of through the captured graph)
    ctx.set_timing(1)
    step()
    ctx.sync()
    summary = ctx.timing_summary()
    ctx.set_timing(0)

    dt_max = sharding.max_over_ranks(torch, dist, dt, torch.device("cpu") if rehearsal else dev) if world > 1 else dt

    # N = 1 only: the same 60 frames as independent phase-mode frames t_j = j/60 (what every rank of an N > 1 run
    # does), so the driver's N-sweep has a like-for-like single-GPU point beside the chained headline
    phase_fps = None
    if world == 1 and mode == "chain" and not args.headline_only:
        ph = sharding.phase_schedule(0, 1, FRAMES)
        ctx.reset(); ctx.render_many(ph, chain=False); ctx.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ctx.reset(); ctx.render_many(ph, chain=False)
        ctx.sync()
        phase_fps = args.steps * FRAMES / (time.perf_counter() - t1)

    # N = 1 only: BASELINE.json configs[4] in miniature — several independent pairs on this GPU at once, each a chained
    # 60-frame morph driven by its own host thread and context (pairs never depend on each other)
    batched_fps, batched_pairs = None, 3
    if world == 1 and mode == "chain" and not args.no_setup and not args.headline_only:
        import threading
        extra = []
        for _ in range(batched_pairs - 1):
            cx = capi.Context(local, number_of_frames=FRAMES)
            cx.pair_load_device(ta.data_ptr(), tb.data_ptr(), tg.data_ptr(), W, H, p1r, p2r)
            extra.append(cx)
        allc = [ctx] + extra

        def run_pair(cx, n):
            for _ in range(n):
                cx.reset(); cx.render_many(shapes, chain=True)
            cx.sync()
        for cx in allc:
            run_pair(cx, 1)
        t1 = time.perf_counter()
        th = [threading.Thread(target=run_pair, args=(cx, args.steps)) for cx in allc]
        for t in th:
            t.start()
        for t in th:
            t.join()
        batched_fps = batched_pairs * args.steps * FRAMES / (time.perf_counter() - t1)
        for cx in extra:
            cx.close()

    # PCIe-inclusive rate: every frame handed to a writer callback from pinned host memory (the library pipelines the
    # downloads behind the rendering), rank 0, not `value`
    pcie_fps = None
    if rank == 0 and not args.headline_only:
        seen = [0]

        def sink(frame):
            seen[0] += 1
        ctx.reset(); ctx.render_many(shapes, chain=(mode == "chain"), write=sink); ctx.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ctx.reset(); ctx.render_many(shapes, chain=(mode == "chain"), write=sink)
        ctx.sync()
        pcie_fps = args.steps * FRAMES / (time.perf_counter() - t1)
        assert seen[0] == (args.steps + 1) * FRAMES

    # measured device-copy ceiling: a plain device-to-device copy moving as many bytes as the warp kernel does per
    # launch (14 B/px read + 14 B/px written), timed with events on torch's stream; the vendor peak is not reachable by any kernel
    copy_ceiling = None
    if rank == 0:
        try:
            nbytes = 14 * W * H
            src_t = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
            dst_t = torch.empty_like(src_t)
            for _ in range(3):
                dst_t.copy_(src_t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                dst_t.copy_(src_t)
            e1.record(); torch.cuda.synchronize()
            copy_ceiling = 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src_t, dst_t
        except Exception as e:
            print(f"bench.py: copy ceiling failed: {e}", file=sys.stderr)

    # pair set-up from the raw images (pre-ORB chain, ORB, matcher, gabor2; once per pair), rank 0, outside the timed
    # region: reported beside `value`, which is the per-frame operator on a resident pair
    setup_ms = None
    if rank == 0 and not args.no_setup and not args.headline_only:
        try:
            a_h, b_h = ta.cpu().numpy(), tb.cpu().numpy()
            c2 = capi.Context(local, number_of_frames=FRAMES)
            c2.pair_begin(a_h, b_h)
            t1 = time.perf_counter()
            c2.pair_begin(a_h, b_h)
            setup_ms = (time.perf_counter() - t1) * 1e3
            c2.close()
        except Exception as e:
            setup_ms = None
            print(f"bench.py: pair_begin failed: {e}", file=sys.stderr)

    if rank == 0:
        fps = args.steps * total_frames / dt_max
        P = W * H
        kernels = {}
        group_ms_per_frame = sum(ms for _, ms, _ in summary) / FRAMES
        summary = [(n, ms, cnt) if n != "warp" else (n, warp_ms, warp_n) for n, ms, cnt in summary]
        for name, ms, cnt in summary:
            per_launch_ms = ms / max(cnt, 1)
            ab = ALGO_BYTES_PER_PX.get(name, 0.0) * P
            kernels[name] = {"avg_ms": round(per_launch_ms, 5), "launch_groups": cnt,
                             "algo_MB": round(ab / 1e6, 3), "GBps": round(ab / (per_launch_ms * 1e-3) / 1e9, 1) if per_launch_ms > 0 else None}
        wk = kernels.get("warp", {})
        achieved = wk.get("GBps") or 0.0
        traffic, traffic_src = None, None            # PMC counters need their own rocprofv3 passes: quoted from profiles/
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_m_warp_pmc.json"))).get(f"{W}x{H}", {})
            pm = pm.get("k_warp_tile") if ctx.last_warp_kind() == 1 else pm.get("k_warp4")
            if pm:
                traffic = pm["fetch_bytes"] + pm["write_bytes"]
                traffic_src = "profiles/r01_m_warp.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
        except (OSError, ValueError):
            pass
        out = {
            "metric": "morph frames/sec at 1080p, 60-frame sequence; Mpix/s warped" if (W, H) == (1920, 1080) else f"morph frames/sec at {W}x{H}; Mpix/s warped",
            "value": round(fps, 2), "unit": "frames/s",
            "mpix_per_s": round(fps * P / 1e6, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8+f32",
            "dtype_note": "u8 pixels and fixed-point remap, f32 pyramid and unsharp; bit-compatible with the reference (no FMA contraction)",
            "data": "synthetic (integer-defined shapes pair, synthetic mask field and 436+4 point pairs; poppy_amd/synth.py)",
            "config": {"workload": f"{W}x{H} pair, {FRAMES}-frame morph per GPU ({total_frames} frames total), "
                                   f"{'default chained mode' if mode == 'chain' else 'phase-mode frame-range sharding'}, "
                                   "pyramid_levels 64, per-frame operator on a resident pair",
                       "frames_per_gpu": FRAMES, "mode": mode, "points": NPTS + 4, "parallelism": f"frame-range x{world}"},
            "roofline": {"bound": "hbm", "kernel": ("k_warp_tile" if ctx.last_warp_kind() == 1 else "k_warp4") + " (fused create_map + remap of both sources + lbmask)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algo_bytes_per_launch": int(ALGO_BYTES_PER_PX["warp"] * P), "avg_launch_ms": wk.get("avg_ms"), "launches_timed": warp_n,
                         "copy_ceiling_GBps": round(copy_ceiling, 1) if copy_ceiling else None,
                         "copy_ceiling_note": "device-to-device copy of 14+14 B/px (buffers that fit the MALL; tools/micro/warp_skeleton.hip measures the kernel's own access mix on cold data: profiles/r01_m_warp.md), measured in this run; "
                                              "traffic / launch time is the figure to hold against it"},
            "kernels": kernels,
            "kernel_groups_ms_per_frame": round(group_ms_per_frame, 4),
            "pcie_inclusive_fps": round(pcie_fps, 1) if pcie_fps else None,
            "phase_mode_fps": round(phase_fps, 1) if phase_fps else None,
            "batched_pairs_fps": {"pairs": batched_pairs, "value": round(batched_fps, 1)} if batched_fps else None,
            "pair_setup_ms": round(setup_ms, 2) if setup_ms else None,
            "fps_including_setup": round(FRAMES / (setup_ms * 1e-3 + FRAMES / fps), 1) if setup_ms and world == 1 else None,
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                a_h, b_h, g_h = ta.cpu().numpy(), tb.cpu().numpy(), tg.cpu().numpy()
                out["cpu_baseline"] = cpu_baseline(a_h, b_h, g_h, p1r, p2r)
                if not args.headline_only and (W, H) == (1920, 1080):
                    out["cpu_baseline_all_cores"] = cpu_baseline_threads(a_h, b_h, g_h, p1r, p2r, max(1, min(64, (os.cpu_count() or 1) // 2)))
            except Exception as e:   # the checker is optional for the measurement, never for parity
                out["cpu_baseline"] = {"value": None, "error": str(e)}
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
