#!/usr/bin/env python3
"""Average of every collected counter per kernel from rocprofv3 --pmc output (rocpd sqlite).

usage: tools/pmc_dump.py <dir-or-db> [kernel-name-substring]
"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db_path in dbs:
        db = sqlite3.connect(db_path)
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        rows = {}
        for name, counter, n, avg in db.execute(q):
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("poppy_hip::", "").replace("void ", "")
            if want in short:
                rows.setdefault(short, []).append(f"{counter}={avg:.6g}")
        for k, v in rows.items():
            print(f"- `{k}`: " + ", ".join(v))


if __name__ == "__main__":
    main()
