#!/usr/bin/env python3
"""Per-kernel table from the raw pipe-counter dumps of tools/pmc_pipes.sh (gpurun_out/<tag>_pipes.md):
   tools/pipes_table.py <tag> [<tag> ...] > profiles/<name>.md
Columns: waves per launch; VALU instructions per wave; kernel duration in cycles (SQ_BUSY_CYCLES is summed over the 32 shader engines);
VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (duration x 1024 SIMDs) — a packed fp32 instruction counts once here although it takes two
issue slots, so kernels with packed arithmetic (warp, unsharp) are busier than the column says; waves resident per SIMD on average
(SQ_WAVE_CYCLES x 4 / (duration x 1024)); LDS bank-conflict cycles relative to LDS active cycles; share of its resident time an average
wave spends waiting."""
import re, sys
FRAME = ["k_upload", "k_tile_expand", "k_warp_bin", "k_pyrdown_level<true>", "k_pyrdown_level<false>", "k_pyrdown2", "k_pyr_tail", "k_collapse2",
         "k_collapse_level<false>", "k_collapse_level<true>", "k_unsharp_tile", "k_unsharp_stream"]
for tag in sys.argv[1:]:
    vals = {}
    head = ""
    for line in open(f"gpurun_out/{tag}_pipes.md"):
        if line.startswith("#"): head = line.strip("# \n"); continue
        m = re.match(r"- `([^`]+)`: (.*)", line)
        if not m: continue
        d = vals.setdefault(m.group(1), {})
        for kv in m.group(2).split(", "):
            k, v = kv.split("="); d[k] = float(v)
    print(f"## {head}\n")
    print("| kernel | waves | VALU instr / wave | duration (cycles) | VALU busy | waves / SIMD | bank-conflict / LDS active | wave waiting |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for k in vals:
        base = k.split("<")[0] if k.split("<")[0] in ("k_warp_bin", "k_tile_expand") else k
        if base not in FRAME: continue
        d = vals[k]
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        w = d.get("SQ_WAVES", 0) or 1
        lds_act = d.get("SQ_ACTIVE_INST_LDS", 0)
        dur = d.get("SQ_BUSY_CYCLES", 0) / 32 or 1
        print(f"| `{k}` | {w:.0f} | {d.get('SQ_INSTS_VALU', 0) / w:.0f} | {dur:.0f} | {100 * d.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (dur * 1024):.0f} % | "
              f"{wc * 4 / (dur * 1024):.1f} | {(100 * d.get('SQ_LDS_BANK_CONFLICT', 0) / (4 * lds_act)) if lds_act else 0:.0f} % | {100 * d.get('SQ_WAIT_ANY', 0) / wc:.0f} % |")
    print()
