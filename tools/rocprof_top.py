#!/usr/bin/env python3
"""Prints the kernel rows of a rocprofv3 --stats kernel_stats csv as a markdown table: rocprof_top.py DIR [max rows]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | total (us) | avg (us) | % |\n|---|---:|---:|---:|---:|")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = r["Name"].split("(")[0].replace("void poppy_hip::", "").replace("poppy_hip::", "").replace("(anonymous namespace)::", "")
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e3:.1f} | {float(r['AverageNs']) / 1e3:.3f} | {float(r['Percentage']):.2f} |")
