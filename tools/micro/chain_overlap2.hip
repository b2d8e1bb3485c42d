// chain_overlap2.hip — what each part of an in-kernel dependency costs (follow-up of chain_overlap.hip, whose ONE counter cost ~75 ns per workgroup: 154 us per link of 2040).
//   sig 0  nothing (plain ordered launches)                       sig 1  release fence + atomic add on one counter, ordered launches, no wait
//   sig 2  release fence only                                      sig 3  atomic add only (no fence)
//   sig 4  a FLAG per workgroup (release store of the link number), ordered launches, the consumer polls the flags of the two workgroups it reads from + acquire fence
//   sig 5  sig 4 with hipExtAnyOrderLaunch: the links really overlap; the flags are the only ordering
//   sig 7  NO fences: the data itself leaves by agent-scope atomic stores (write-through, sc1) and is read by agent-scope atomic loads; a flag per workgroup after s_waitcnt; ordered launches
//   sig 8  sig 7 with hipExtAnyOrderLaunch (every link writes its own buffer: no write-after-read hazards between overlapping links)
//   sig 6  sig 5 without the producer's release FENCE (store-release only on the flag)... same thing spelled as a fence-less atomic store: checks whether wrong elements appear
// build: hipcc --offload-arch=gfx950 -O3 -o co2 tools/micro/chain_overlap2.hip ; run: ./co2 [blocks] [work] [links]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) k_link(const float* __restrict__ in, float* __restrict__ out, unsigned n, int work, int sig,
                                              unsigned long long* ctr, unsigned* flags, unsigned link, unsigned blocks, unsigned* given_up) {
    __shared__ float s[256];
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;
    const unsigned src = (idx + 1031u * 256u + 17u) % n;                 // another workgroup's element
    if (sig >= 7 && link > 0) {
        if (threadIdx.x < 2) {
            const unsigned pb = ((blockIdx.x + 1031u + threadIdx.x) % blocks);
            unsigned polls = 0;
            while (__hip_atomic_load(&flags[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < link && ++polls < (1u << 18)) __builtin_amdgcn_s_sleep(2);
            if (polls >= (1u << 18)) atomicAdd(given_up, 1u);
        }
        __syncthreads();
    } else if (sig >= 4 && link > 0) {
        if (threadIdx.x < 2) {                                           // the two producer tiles this workgroup's reads fall into
            const unsigned pb = ((blockIdx.x + 1031u + threadIdx.x) % blocks);
            unsigned polls = 0;
            while (__hip_atomic_load(&flags[pb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < link && ++polls < (1u << 18)) __builtin_amdgcn_s_sleep(2);
            if (polls >= (1u << 18)) atomicAdd(given_up, 1u);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float v = sig >= 7 ? __hip_atomic_load(&in[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : in[src];
    s[threadIdx.x] = v;
    __syncthreads();
    float a = s[(threadIdx.x + 1) & 255] * 0.f + v;
    for (int i = 0; i < work; ++i) a = a * 1.0000001f + 0.f;
    if (sig >= 7) {
        __hip_atomic_store(&out[idx], (float)((int)(a + 0.5f)) + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                                   // (vmcnt(0): the write-through stores have been acknowledged)
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], link + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    out[idx] = (float)((int)(a + 0.5f)) + 1.f;
    if (sig == 1 || sig == 2 || sig == 4 || sig == 5) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (sig) __syncthreads();
    if (threadIdx.x == 0) {
        if (sig == 1 || sig == 3) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig == 4 || sig == 5) __hip_atomic_store(&flags[blockIdx.x], link + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig == 6) __hip_atomic_store(&flags[blockIdx.x], link + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const unsigned blocks = argc > 1 ? atoi(argv[1]) : 2040;
    const int work = argc > 2 ? atoi(argv[2]) : 2000, links = argc > 3 ? atoi(argv[3]) : 600;
    const unsigned n = blocks * 256;
    float* buf[2]; float* big; unsigned long long* ctr; unsigned *flags, *given;
    CK(hipMalloc(&buf[0], n * 4)); CK(hipMalloc(&buf[1], n * 4)); CK(hipMalloc(&big, (size_t)(links + 1) * n * 4)); CK(hipMalloc(&ctr, 64)); CK(hipMalloc(&flags, blocks * 4)); CK(hipMalloc(&given, 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<float> h(n);
    for (int sig = 0; sig <= 8; ++sig) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemsetAsync(buf[0], 0, n * 4, s)); CK(hipMemsetAsync(buf[1], 0, n * 4, s)); CK(hipMemsetAsync(ctr, 0, 64, s));
            CK(hipMemsetAsync(flags, 0, blocks * 4, s)); CK(hipMemsetAsync(given, 0, 4, s));
            if (sig >= 7) CK(hipMemsetAsync(big, 0, (size_t)n * 4, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < links; ++i) {
                const float* in = sig >= 7 ? big + (size_t)i * n : buf[i & 1]; float* out = sig >= 7 ? big + (size_t)(i + 1) * n : buf[(i + 1) & 1];
                if (sig == 5 || sig == 6 || sig == 8) hipExtLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, in, out, n, work, sig, ctr, flags, (unsigned)i, blocks, given);
                else hipLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, in, out, n, work, sig, ctr, flags, (unsigned)i, blocks, given);
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipMemcpy(h.data(), sig >= 7 ? big + (size_t)links * n : buf[links & 1], n * 4, hipMemcpyDeviceToHost));
            unsigned bad = 0; for (unsigned i = 0; i < n; ++i) bad += h[i] != (float)links;
            unsigned g = 0; CK(hipMemcpy(&g, given, 4, hipMemcpyDeviceToHost));
            if (rep) printf("sig %d: %d links of %u workgroups, work %d: %.2f us per link, %u wrong elements, %u waits given up\n", sig, links, blocks, work, us / links, bad, g);
        }
    }
    return 0;
}
