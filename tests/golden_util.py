"""Access to the committed reference fixtures (tests/golden) and the inputs they were made from."""
import hashlib
import json
import os
import sys
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "golden_gen"))
import make_inputs  # noqa: E402  (case tables + input builders; pure numpy, no reference needed)

_manifest = None
_npz = {}


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLD, "manifest.json")) as f:
            _manifest = json.load(f)
    return _manifest


def entries(case):
    return manifest()["cases"][case]


def full(case, name):
    """The reference array if it was committed in full, else None."""
    if case not in _npz:
        _npz[case] = np.load(os.path.join(GOLD, case + ".npz"))
    z = _npz[case]
    return z[name] if name in z.files else None


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def check(case, name, arr, what=""):
    """Bit-exact comparison of `arr` with the reference output `name` of `case`."""
    e = entries(case)[name]
    arr = np.ascontiguousarray(arr)
    assert list(arr.shape) == e["shape"], f"{case}/{name}: shape {arr.shape} != {e['shape']} {what}"
    assert str(arr.dtype) == e["dtype"], f"{case}/{name}: dtype {arr.dtype} != {e['dtype']}"
    ref = full(case, name)
    if ref is not None:
        if arr.dtype.kind == "f":
            same = arr.view(np.uint32 if arr.dtype == np.float32 else np.uint64) == ref.view(np.uint32 if arr.dtype == np.float32 else np.uint64)
        else:
            same = arr == ref
        if not same.all():
            bad = np.argwhere(~same)
            diff = np.abs(arr.astype(np.float64) - ref.astype(np.float64)).max()
            raise AssertionError(f"{case}/{name}: {len(bad)} of {arr.size} elements differ (max abs {diff:g}); first at {bad[0]} {what}")
    else:
        assert sha(arr) == e["sha256"], f"{case}/{name}: sha256 mismatch (hash-only fixture) {what}"


def bstage_inputs(case):
    return make_inputs.bstage_inputs(case)


def orb_inputs(case):
    return make_inputs.orb_inputs(case)


def match_inputs(case):
    return make_inputs.match_inputs(case)


def astage_inputs(case):
    return make_inputs.astage_inputs(case)


def prims_inputs():
    return make_inputs.prims_inputs()
