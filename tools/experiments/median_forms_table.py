"""Table of median kernel times from a rocprofv3 kernel trace of median_forms.py: rows = window, columns = content x form (us, last repetition)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
forms = [1, 2, 3]
reps = 2
for line in open(sys.argv[2]):
    if "forms [" in line:
        forms = [int(x) for x in line.split("forms [")[1].split("]")[0].split(",")]
        break
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start = "start" if "start" in cols else [c for c in cols if "start" in c][0]
rows = [(n, d) for n, s, d in db.execute(f"select name, {start}, duration from kernels order by {start}") if "k_median_u8" in n or "k_median_cols" in n]
contents = ["shapes", "photo", "texture"]
need = len(contents) * 11 * len(forms) * reps
print(f"{len(rows)} median launches in the trace, {need} expected")
it = iter(rows)
table = {}
for c in contents:
    for i in range(1, 12):
        for f in forms:
            for r in range(reps):
                n, d = next(it)
                assert ("k_median_u8" in n) == (f == 1), (c, i, f, n)
                table[(c, i, f)] = d / 1e3
print("| ksize | " + " | ".join(f"{c} f{f}" for c in contents for f in forms) + " |")
print("|---|" + "---:|" * (len(contents) * len(forms)))
for i in range(1, 12):
    print(f"| {8 * i + 1} | " + " | ".join(f"{table[(c, i, f)]:.1f}" for c in contents for f in forms) + " |")
print("| sum | " + " | ".join(f"{sum(table[(c, i, f)] for i in range(1, 12)):.1f}" for c in contents for f in forms) + " |")
