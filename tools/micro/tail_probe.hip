// Where does k_pyr_tail spend its time?  Builds the pyramid level table of a W x H frame with 64 levels, runs the tail kernel
// (kernels_pyramid_tail.hip, compiled in here with its stamp macro on) warm — back to back — and cold — after a kernel that sweeps the
// caches — and prints the kernel time (events) and the per-phase stamps of thread 0 (100 MHz counter).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I poppy_amd/csrc -I include tools/micro/tail_probe.hip -o tools/micro/tp
//   gpurun -- ./tools/micro/tp 1920 1080
#define POPPY_TAIL_STAMPS 1
#include "../../poppy_amd/csrc/kernels_pyramid_tail.hip"
#include <cstdio>
#include <vector>

using namespace poppy_hip;

__global__ void k_sweep(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080, L = 64;
    const int tail_px = argc > 3 ? atoi(argv[3]) : 600;
    std::vector<PyrLevel> lv(L + 1);
    size_t o3 = 0, o1 = 0;
    int w = W, h = H;
    for (int i = 0; i <= L; ++i) { lv[i] = {w, h, o3, o1}; o3 += (size_t)w * h * 3; o1 += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
    int first = L;
    for (int i = 1; i <= L; ++i) if ((size_t)lv[i].w * lv[i].h <= (size_t)tail_px) { first = i; break; }
    const PyrTailPlan plan = build_pyr_tail_plan(lv.data(), first, L);
    printf("%dx%d: tail from level %d (%dx%d), %d multi-pixel steps, %d single-pixel reductions, %d descriptors, %zu bytes of LDS, ok %d\n", W, H, first,
           lv[first].w, lv[first].h, plan.args.n_wide, plan.args.nl, plan.args.n_desc, plan.lds_bytes, (int)plan.ok);
    float *L_, *R_, *M_, *B_, *junk; void* dlv;
    const size_t junk_n = (size_t)256 << 20;
    hipMalloc(&L_, o3 * 4); hipMalloc(&R_, o3 * 4); hipMalloc(&B_, o3 * 4); hipMalloc(&M_, o1 * 4); hipMalloc(&junk, junk_n * 4);
    hipMalloc(&dlv, plan.desc.size() * 4 + 16);
    hipMemcpy(dlv, plan.desc.data(), plan.desc.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> init(o3);
    for (size_t i = 0; i < o3; ++i) init[i] = (float)((i * 2654435761u) & 0xffff) / 65535.f;
    hipMemcpy(L_, init.data(), o3 * 4, hipMemcpyHostToDevice); hipMemcpy(R_, init.data() + 7, (o3 - 7) * 4, hipMemcpyHostToDevice);
    hipMemcpy(M_, init.data(), o1 * 4, hipMemcpyHostToDevice); hipMemset(junk, 0, junk_n * 4);
    if (!prepare_pyr_tail(plan.lds_bytes)) { printf("LDS limit\n"); return 1; }
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto show = [&](const char* what) {
        long long st[64]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_tail_stamps), sizeof(st));
        printf("%s stamps (us since kernel start):", what);
        for (int i = 1; i <= 30; ++i) if (st[i] >= st[0] && st[i] - st[0] < 100000) printf(" [%d]%.2f", i, (st[i] - st[0]) / 100.0);
        printf("\n");
    };
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 20; ++r) {
            if (mode) { hipLaunchKernelGGL(k_sweep, dim3(4096), dim3(256), 0, s, junk, junk_n); }
            hipEventRecord(e0, s);
            launch_pyr_tail(L_, R_, M_, B_, dlv, plan.args, plan.lds_bytes, s);
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%s: best %.2f us, mean %.2f us (event to event)\n", mode ? "cold (after a 1 GB sweep)" : "warm (back to back)", best * 1e3f, sum / 18 * 1e3f);
        show(mode ? "cold" : "warm");
    }
    return 0;
}
