"""Builds poppy_amd/libpoppy_hip.so (gfx950 code objects + host C++) with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only container as the build check and
the resulting .so travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpoppy_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference arithmetic rounds after every multiply and add (no FMA in its
# SSE3-baseline build); contraction would flip low bits of map coordinates and pyramid sums.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-I" + os.path.join(HERE, "..", "include")]


# kernels whose inner loops the SLP vectoriser turns into packed fp32 instructions + the moves that pair their operands: a packed
# instruction costs its two scalar halves on gfx950 (profiles/r03_notes.md section 1), the moves come on top
PER_FILE_FLAGS = {"kernels_unsharp_stream.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "poppy_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, experiments=False):
    """experiments=True: -DPOPPY_EXPERIMENTS, the build in which the result-changing measurement switches (POPPY_MED_COLS_SKIP, POPPY_GABOR_NO_REDO,
    POPPY_GABOR_BAND) are read at all; it is written to libpoppy_hip_experiments.so, never to the shipped library's name."""
    if experiments:
        return _build_experiments(verbose)
    if not force and not needs_build():
        return OUT
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(
                os.path.getmtime(src), *[os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))],
                os.path.getmtime(os.path.join(HERE, "..", "include", "poppy_hip.h"))):
            continue
        cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def _build_experiments(verbose=True):
    objdir = os.path.join(HERE, "build", "experiments")
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(HERE, "libpoppy_hip_experiments.so")
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-DPOPPY_EXPERIMENTS"] + PER_FILE_FLAGS.get(os.path.basename(src), []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    return out


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(build(experiments=True))
    else:
        build(force="--force" in sys.argv)
