import ctypes as C, os, sys, time
if os.environ.get('POOL_WITH_TORCH'): import torch      # the bench process's runtime (frame copies as blit kernels)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from poppy_amd import capi, synth
steps, contexts, PAIRS, writer = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
DISTINCT = int(sys.argv[5]) if len(sys.argv) > 5 else PAIRS        # PAIRS pairs per call drawn from DISTINCT different ones (one call with steps x 6 pairs: the contexts run out of step)
W, H = 1920, 1080
L = capi.lib(); hip = C.CDLL("libamdhip64.so")
ptrs = []
for k in range(DISTINCT):
    a, b = synth.gen_pair(W, H, seed=1234 + k); pp = []
    for img in (a, b):
        d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(img.nbytes)) == 0
        assert hip.hipMemcpy(d, img.ctypes.data_as(C.c_void_p), C.c_size_t(img.nbytes), 1) == 0; pp.append(d.value)
    ptrs.append(tuple(pp))
pool = capi.Pool([0], contexts_per_device=contexts, number_of_frames=60)
def src(user, p, device, pa, sa, pb, sb):
    q = p % DISTINCT; pa[0] = ptrs[q][0]; sa[0] = W * 3; pb[0] = ptrs[q][1]; sb[0] = W * 3; return 0
fs = capi.PAIR_SOURCE_CB(src); n = C.c_longlong(0); err = C.create_string_buffer(512)
def run():
    rc = L.poppy_hip_pool_morph_pairs(pool.h, PAIRS, W, H, -1.0, 1, C.cast(fs, C.c_void_p),
        C.cast(L.poppy_count_pair_frames_cb, C.c_void_p) if writer else None, C.cast(C.byref(n), C.c_void_p), err, 512)
    assert rc == 0, err.value
for _ in range(3): run()
hip.hipDeviceSynchronize(); t0 = time.perf_counter()
for _ in range(steps): run()
hip.hipDeviceSynchronize(); dt = time.perf_counter() - t0
print(f"contexts {contexts} pairs {PAIRS} writer {writer}: {steps*PAIRS*60/dt:.0f} frames/s")
