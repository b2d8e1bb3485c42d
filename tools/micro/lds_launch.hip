// Microbenchmark (round 6): what a launch of many workgroups costs as a function of the LDS each one asks for — k_unsharp_tile (4080 workgroups x 256 threads, 25.8 KB of
// LDS) takes 22-24 us at 1080p with its loads, stores and arithmetic all left out (tools/experiments/unsharp_skip.sh).  Kernels that do nothing but touch their LDS and meet
// at `barriers` barriers.   hipcc --offload-arch=gfx950 -O3 lds_launch.hip -o ll && ./ll
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_lds(float* out, int lds_floats, int barriers, int touch) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    if (touch) for (int i = tid; i < lds_floats; i += 256) lds[i] = (float)i;
    for (int b = 0; b < barriers; ++b) __syncthreads();
    if (touch && lds[(tid * 7) % lds_floats] == -1.f) out[blockIdx.x] = 1.f;
}

int main() {
    float* out; CHK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("%8s %8s %9s %6s   us per launch (back to back, 50 launches)\n", "blocks", "LDS KB", "barriers", "touch");
    for (int blocks : {2040, 4080, 8160})
        for (int kb : {0, 4, 8, 16, 26, 40, 64})
            for (int cfg = 0; cfg < 3; ++cfg) {
                const int barriers = cfg == 0 ? 0 : 4, touch = cfg == 2;
                if (kb == 0 && touch) continue;
                const int lf = kb * 256;
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), kb * 1024, 0, out, lf > 0 ? lf : 1, barriers, touch);
                CHK(hipDeviceSynchronize());
                CHK(hipEventRecord(e0));
                for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), kb * 1024, 0, out, lf > 0 ? lf : 1, barriers, touch);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                printf("%8d %8d %9d %6d   %7.2f\n", blocks, kb, barriers, touch, ms / 50 * 1e3);
            }
    return 0;
}
