// chain_inlaunch.hip — the whole chain of dependent links in ONE launch (round 6): link l = workgroups [l * B, (l + 1) * B), every workgroup waits for the flags of the
// two workgroups of link l - 1 it reads from (agent-scope loads), reads their data with agent-scope loads, writes its own with agent-scope (write-through) stores and
// raises its flag after s_waitcnt — no fence anywhere (chain_overlap2.hip: a release fence per workgroup costs ~36 ns, serialised; write-through stores cost nothing
// measurable).  Workgroups are dispatched in increasing order per XCD and consumers have higher numbers than their producers: no deadlock, whatever the grid's size.
// Compared with the same links as separate ordered launches (form 0).   build: hipcc --offload-arch=gfx950 -O3 -o ci tools/micro/chain_inlaunch.hip ; ./ci [B] [work] [links]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void link_body(const float* in, float* out, unsigned n, unsigned b, int work, bool coherent, float* s) {
    const unsigned idx = b * 256 + threadIdx.x;
    const unsigned src = (idx + 31u * 256u + 17u) % n;
    const float v = coherent ? __hip_atomic_load(&in[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : in[src];
    s[threadIdx.x] = v;
    __syncthreads();
    float a = s[(threadIdx.x + 1) & 255] * 0.f + v;
    for (int i = 0; i < work; ++i) a = a * 1.0000001f + 0.f;
    const float r = (float)((int)(a + 0.5f)) + 1.f;
    if (coherent) __hip_atomic_store(&out[idx], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else out[idx] = r;
}

__global__ void __launch_bounds__(256) k_one_link(const float* in, float* out, unsigned n, int work) {
    __shared__ float s[256];
    link_body(in, out, n, blockIdx.x, work, false, s);
}

__global__ void __launch_bounds__(256) k_all_links(float* big, unsigned n, unsigned B, int work, unsigned* flags, unsigned* given_up) {
    __shared__ float s[256];
    const unsigned link = blockIdx.x / B, b = blockIdx.x - link * B;
    if (link > 0) {
        if (threadIdx.x < 2) {
            const unsigned pb = (b + 31u + threadIdx.x) % B;
            unsigned polls = 0;
            while (__hip_atomic_load(&flags[(link - 1) * B + pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++polls < (1u << 20)) __builtin_amdgcn_s_sleep(1);
            if (polls >= (1u << 20)) atomicAdd(given_up, 1u);
        }
        __syncthreads();
    }
    link_body(big + (size_t)link * n, big + (size_t)(link + 1) * n, n, b, work, true, s);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[link * B + b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const unsigned B = argc > 1 ? atoi(argv[1]) : 512;
    const int work = argc > 2 ? atoi(argv[2]) : 200, links = argc > 3 ? atoi(argv[3]) : 200;
    const unsigned n = B * 256;
    float* big; unsigned *flags, *given;
    CK(hipMalloc(&big, (size_t)(links + 1) * n * 4)); CK(hipMalloc(&flags, (size_t)links * B * 4)); CK(hipMalloc(&given, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<float> h(n);
    for (int form = 0; form < 2; ++form)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(big, 0, (size_t)n * 4, s)); CK(hipMemsetAsync(flags, 0, (size_t)links * B * 4, s)); CK(hipMemsetAsync(given, 0, 4, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            if (form == 0) for (int i = 0; i < links; ++i) hipLaunchKernelGGL(k_one_link, dim3(B), dim3(256), 0, s, big + (size_t)i * n, big + (size_t)(i + 1) * n, n, work);
            else hipLaunchKernelGGL(k_all_links, dim3(B * links), dim3(256), 0, s, big, n, B, work, flags, given);
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipMemcpy(h.data(), big + (size_t)links * n, n * 4, hipMemcpyDeviceToHost));
            unsigned bad = 0; for (unsigned i = 0; i < n; ++i) bad += h[i] != (float)links;
            unsigned g = 0; CK(hipMemcpy(&g, given, 4, hipMemcpyDeviceToHost));
            if (rep) printf("%s: %d links of %u workgroups, work %d: %.2f us per link, %u wrong elements, %u waits given up\n", form ? "one launch, flags" : "a launch per link ", links, B, work, us / links, bad, g);
        }
    return 0;
}
