#!/usr/bin/env python3
"""Checks freshly generated reference dumps (oracle/_dumps/cases/*/out) against the COMMITTED fixtures: every array's dtype, shape and
sha256 must equal tests/golden/manifest.json's entry.  CONTAINER-ONLY (run_all.sh --verify); writes nothing under tests/.

verify.py            every case found under oracle/_dumps/cases
verify.py CASE ...   only these
Exit code 0 = every dumped array of every checked case is byte-identical to what the manifest pins; cases of the manifest that were
not regenerated are listed as "not regenerated" (and fail the run unless cases were named)."""
import glob
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from pack import CASES, GOLD, load_raw  # noqa: E402


def main():
    only = sys.argv[1:]
    manifest = json.load(open(os.path.join(GOLD, "manifest.json")))["cases"]
    bad = 0
    seen = set()
    n_arrays = 0
    for cdir in sorted(glob.glob(os.path.join(CASES, "*"))):
        case = os.path.basename(cdir)
        if only and case not in only:
            continue
        outs = sorted(glob.glob(os.path.join(cdir, "out", "*.bin")))
        if not outs:
            continue
        seen.add(case)
        if case not in manifest:
            print(f"{case}: NEW (not in the committed manifest)")
            continue
        want = manifest[case]
        got = {}
        for path in outs:
            name, arr = load_raw(path)
            got[name] = {"dtype": str(arr.dtype), "shape": list(arr.shape), "sha256": hashlib.sha256(arr.tobytes()).hexdigest()}
        diff = [n for n in sorted(want) if n not in got or any(got[n][k] != want[n][k] for k in ("dtype", "shape", "sha256"))]
        extra = sorted(set(got) - set(want))          # dumped by the generator but never packed for this case: not pinned, not a failure
        n_arrays += len(want)
        if diff:
            bad += 1
            print(f"{case}: {len(diff)} of {len(want)} arrays DIFFER: {diff[:8]}")
        else:
            print(f"{case}: {len(want)} arrays byte-identical to the manifest" + (f" (+{len(extra)} dumped but not in the manifest: {extra[:4]})" if extra else ""))
    missing = sorted(set(manifest) - seen) if not only else sorted(set(only) - seen)
    if missing:
        print("not regenerated:", " ".join(missing))
    print(f"verified {len(seen)} cases, {n_arrays} arrays; {bad} cases differ; {len(missing)} not regenerated")
    return 1 if bad or missing else 0


if __name__ == "__main__":
    sys.exit(main())
