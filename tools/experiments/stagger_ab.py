#!/usr/bin/env python3
"""Splits every kernel's dispatches of a rocprofv3 kernel-trace database by PARITY of their order: under POPPY_STAGGER_AB=<mask> (a -DPOPPY_EXPERIMENTS
build) the kernels of the mask run with the wave-priority stagger on every second launch, so even / odd dispatches are the two forms side by side in one
process.  usage: stagger_ab.py <results.db> [name filter]"""
import sqlite3, sys, statistics
db = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else "k_"
rows = list(db.execute("select name, grid_x, duration from kernels order by start"))
by = {}
for name, gx, dur in rows:
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
    if filt not in short:
        continue
    by.setdefault((short, gx), []).append(dur / 1e3)
print(f"{'kernel':34s} {'grid':>9s} {'n':>5s} {'off (even)':>11s} {'on (odd)':>11s} {'on/off':>7s}   medians")
for (k, gx), d in sorted(by.items()):
    if len(d) < 20:
        continue
    d = d[len(d) // 10:]                      # skip the warm-up launches
    ev, od = d[0::2], d[1::2]
    # which parity is "on" depends on the launch counter: the first launch of a kernel is off (counter 0), and the skipped prefix has even length or not
    if (len(by[(k, gx)]) // 10) % 2:
        ev, od = od, ev
    print(f"{k:34s} {gx:9d} {len(d):5d} {statistics.mean(ev):11.2f} {statistics.mean(od):11.2f} {statistics.mean(od) / statistics.mean(ev):7.3f}   {statistics.median(ev):.2f} / {statistics.median(od):.2f}")
