#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite output).

usage: tools/pmc_traffic.py <fetch.db> <write.db> [--px W*H]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Per MI355X_MICROARCH.md (HBM section) this image's
FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane) coalesced streaming reads at 64 B, i.e. reports half
of their bytes; other access widths and WRITE_SIZE are uncalibrated.  The table therefore prints the raw value and
"fetch x2" (the guide's correction, an upper bound for kernels that mix narrow accesses), and the calibration rows
at the bottom compare kernels whose byte counts are known exactly.
"""
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    q = ("select kernel_name, grid_size, count(*), avg(value) from counters_collection where counter_name = ? "
         "group by kernel_name, grid_size order by kernel_name, grid_size desc")
    out = {}
    for name, grid, n, avg in db.execute(q, (counter,)):
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
        out[(short, grid)] = (n, avg * 1024.0)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    px = eval(sys.argv[sys.argv.index("--px") + 1]) if "--px" in sys.argv else None
    print("| kernel | grid (threads) | launches | FETCH_SIZE raw (MB) | fetch x2 (MB) | WRITE_SIZE raw (MB) |" + (" raw B/px (f, w) |" if px else ""))
    print("|---|---:|---:|---:|---:|---:|" + ("---:|" if px else ""))
    for key in sorted(set(fetch) | set(write)):
        n, f = fetch.get(key, (0, 0.0))
        n2, w = write.get(key, (0, 0.0))
        if max(n, n2) < 4 and not key[0].startswith("k_"):
            continue
        row = f"| `{key[0]}` | {key[1]} | {max(n, n2)} | {f / 1e6:.2f} | {2 * f / 1e6:.2f} | {w / 1e6:.2f} |"
        if px:
            row += f" {f / px:.2f}, {w / px:.2f} |"
        print(row)


if __name__ == "__main__":
    main()
