#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel summary kept under profiles/.

usage: tools/rocprof_summary.py <results.db> [<out.md>] [--title "..."]
"""
import sqlite3
import sys


def main():
    db_path = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    title = sys.argv[sys.argv.index("--title") + 1] if "--title" in sys.argv else db_path
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats: {title}", "",
             "| kernel | calls | total (us) | avg (us) | % |", "|---|---:|---:|---:|---:|"]
    for name, calls, total, avg, pct in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
        lines.append(f"| `{short}` | {calls} | {total:.1f} | {avg:.3f} | {pct:.2f} |")
    if "--by-grid" in sys.argv:     # one row per (kernel, launch geometry): separates the pyramid levels
        lines += ["", "| kernel | grid | block | calls | avg (us) | vgpr | lds |", "|---|---:|---:|---:|---:|---:|---:|"]
        q = ("select name, grid_x, grid_y, workgroup_x, workgroup_y, count(*), avg(duration), max(vgpr_count), max(lds_size) "
             "from kernels group by name, grid_x, grid_y, workgroup_x order by name, grid_x * grid_y desc")
        for name, gx, gy, wx, wy, n, avg, vg, ldsz in db.execute(q):
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
            lines.append(f"| `{short}` | {gx}x{gy} | {wx}x{wy} | {n} | {avg / 1e3:.2f} | {vg} | {ldsz} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "a").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
