#!/usr/bin/env python3
"""Writes profiles/<round>_warp_facts.json, which bench.py quotes next to the live measurement (roofline.valu, roofline.from_profiles):
  * the vector-issue cycles of k_warp_bin's main path: the kernel compiled with its two rare paths left out (-DPOPPY_WARP_COUNT_MAIN:
    the border redo and the overflow-record loads), every VALU instruction of the listing priced by the measured gfx950 issue costs
    (profiles/r03_notes.md section 1; tools/micro/valu_rate.hip): 2 cycles full rate, 4 half rate, 8 quarter rate;
  * the kernel's average duration in the round's rocprofv3 kernel traces (gpurun_out/<tag>_chain_<W>/*.db, tools/profile_round.sh).
usage: tools/warp_facts.py <tag>          e.g. r03_x"""
import glob, json, os, re, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_lo_u16", "v_add_u16", "v_not_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_mad_u16", "v_fma_f16", "v_exp_f32", "v_log_f32"}


def issue_cycles(asm_text, kernel_prefix):
    lines = asm_text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel_prefix) and ":" in l and not l.startswith("\t"))
    cyc = n = 0
    hist = {}
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.startswith("\t.size"):
            break
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if not m:
            continue
        op = m.group(1)
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        c = 4 if op.endswith(("_dpp", "_sdwa")) else 2 if base in FULL else 8 if base in QUARTER else 4
        cyc += c; n += 1
        hist[base] = hist.get(base, 0) + 1
    return cyc, n, hist


def main():
    tag = sys.argv[1]
    rnd = tag.split("_")[0]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
             "-DPOPPY_WARP_COUNT_MAIN", "-x", "hip", "--cuda-device-only", "-S"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "wb.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(ROOT, "poppy_amd", "csrc", "kernels_warp_bin.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    facts = {"run_id": os.environ.get("POPPY_RUN_ID"),
             "note": "issue cycles: static listing of the main path priced at 2 / 4 / 8 cycles per wave-instruction (profiles/r03_notes.md section 1); "
                     "trace: rocprofv3 --kernel-trace --stats of tools/experiments/frames_only.py W H 60 chain 3 (tools/profile_round.sh)"}
    for (w, h), tw in (((1920, 1080), 64), ((3840, 2160), 128)):
        cyc, n, hist = issue_cycles(asm, f"_ZN9poppy_hip10k_warp_binILi{tw}ELb1E")      # the instantiation for widths that are multiples of 4
        entry = {"issue_cycles_per_wave": cyc, "instructions_per_wave": n, "clock_GHz": 2.4,
                 "top_instructions": dict(sorted(hist.items(), key=lambda kv: -kv[1])[:12])}
        dbs = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_chain_{w}", "*.db"))
        if dbs:
            db = sqlite3.connect(dbs[0])
            row = db.execute("select average, total_calls from top_kernels where name like '%k_warp_bin%'").fetchone()
            if row:
                entry["trace_avg_us"] = round(row[0] / 1e3, 3) if row[0] > 1000 else round(row[0], 3)
                entry["trace_calls"] = row[1]
                entry["trace_source"] = f"profiles/{tag}_trace.md (gpurun_out/{tag}_chain_{w})"
        facts[f"{w}x{h}"] = {"k_warp_bin": entry}
        # the other full-resolution kernels of a frame: trace duration and the algorithmic bytes DESIGN.md's kernel table states for them
        others = {"k_unsharp": ("unsharp_mask + convertTo (k_unsharp_tile below 4 Mpx, k_unsharp_stream above)", 15.0),
                  "k_collapse_level<true>": ("level-0 collapse: three pyrUps + Laplacian + mix + add", 31.0),
                  "k_pyrdown_level<true>": ("level-0 pyrDown of both warped sources and the mask", 17.0)}
        if dbs:
            for pat, (what, bpp) in others.items():
                row = db.execute("select name, average, total_calls from top_kernels where name like ? order by total_duration desc", ("%" + pat + "%",)).fetchone()
                if row:
                    us = row[1] / 1e3 if row[1] > 1000 else row[1]
                    nb = bpp * w * h
                    facts[f"{w}x{h}"][pat] = {"what": what, "kernel": row[0].split("(")[0].replace("void ", "").replace("poppy_hip::", "").replace("(anonymous namespace)::", ""),
                                             "algo_bytes_per_px": bpp, "algo_bytes_per_launch": int(nb), "trace_avg_us": round(us, 3), "trace_calls": row[2],
                                             "achieved_GBps": round(nb / (us * 1e-6) / 1e9, 1), "frac_of_8_TBps": round(nb / (us * 1e-6) / 8e12, 4)}
        print(f"{w}x{h}: k_warp_bin<{tw}> {n} VALU instructions, {cyc} issue cycles per wave; trace {entry.get('trace_avg_us')} us")
    json.dump(facts, open(os.path.join(ROOT, "profiles", f"{rnd}_warp_facts.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
