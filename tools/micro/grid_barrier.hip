// Microbenchmark: cost of a grid-wide barrier (atomic counter + agent-scope fences) between dependent phases of a
// persistent kernel on MI355X, for G workgroups of 256 threads.  Each phase reads what a *different* workgroup wrote
// in the previous phase, so the fences are really needed.   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_phases(float* a, float* b, unsigned* counter, int phases, int n) {
    const int G = gridDim.x;
    float* src = a; float* dst = b;
    for (int p = 0; p < phases; ++p) {
        const int other = (blockIdx.x + 1 + p) % G;                // read another workgroup's slice
        for (int i = threadIdx.x; i < n; i += 256) dst[blockIdx.x * n + i] = src[other * n + i] + 1.f;
        grid_barrier(counter, (unsigned)(G * (p + 1)));
        float* t = src; src = dst; dst = t;
    }
}

__global__ void k_one(float* a, float* b, int n) {
    for (int i = threadIdx.x; i < n; i += 256) b[blockIdx.x * n + i] = a[((blockIdx.x + 1) % gridDim.x) * n + i] + 1.f;
}

int main() {
    const int n = 1024, phases = 200;
    for (int G : {8, 32, 64, 128, 256}) {
        float *a, *b; unsigned* c;
        hipMalloc(&a, G * n * 4); hipMalloc(&b, G * n * 4); hipMalloc(&c, 4);
        hipMemset(a, 0, G * n * 4); hipMemset(c, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(c, 0, 4);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_phases, dim3(G), dim3(256), 0, 0, a, b, c, phases, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> h(G * n);
        hipMemcpy(h.data(), (phases % 2) ? b : a, G * n * 4, hipMemcpyDeviceToHost);
        bool ok = true; for (float v : h) ok = ok && v == (float)phases;
        // same dependent chain as separate launches
        hipEventRecord(e0);
        for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, 0, (p & 1) ? b : a, (p & 1) ? a : b, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms2; hipEventElapsedTime(&ms2, e0, e1);
        printf("G=%3d: persistent %.2f us/phase (%s)   separate launches %.2f us/phase\n", G, ms * 1e3 / phases, ok ? "ok" : "WRONG", ms2 * 1e3 / phases);
        hipFree(a); hipFree(b); hipFree(c);
    }
    return 0;
}
