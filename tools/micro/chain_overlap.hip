// chain_overlap.hip — can a chain of DEPENDENT kernels on one stream overlap its launch gaps?  (round 6)
// Kernel i reads what kernel i-1 wrote (an element of ANOTHER workgroup's tile) and writes its own buffer.  Forms:
//   0  plain launches: the stream's barrier bit orders them (what the frame chain does today)
//   1  hipExtAnyOrderLaunch (no barrier bit: the next kernel's workgroups are dispatched as soon as the previous kernel's have all been DISPATCHED) + a
//      device counter: every workgroup of kernel i-1 releases (fence + atomic add) when its stores are out, every workgroup of kernel i does its
//      independent prologue, then waits for the counter to reach the previous kernel's workgroup count (acquire) before its first dependent load.
//      Deadlock-free on one queue: a waiting workgroup can only hold a slot once every producer workgroup is resident.
//   2  form 1's kernels with plain (ordered) launches: what the counters themselves cost
// The value each element ends with is the number of kernels in the chain iff every read saw its producer's store.
// build: hipcc --offload-arch=gfx950 -O3 -o co tools/micro/chain_overlap.hip ; run: ./co [blocks] [work] [links]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) k_link(const float* __restrict__ in, float* __restrict__ out, unsigned n, int work,
                                              unsigned long long* wait_ctr, unsigned long long expect, unsigned long long* done_ctr) {
    __shared__ float s[256];
    const unsigned idx = blockIdx.x * 256 + threadIdx.x;
    const unsigned src = (idx + 1031u * 256u + 17u) % n;                 // another workgroup's element
    if (wait_ctr) {
        if (threadIdx.x == 0) {
            unsigned polls = 0;                                            // (a bound, so that a wrong assumption about dispatch order ends in an error count, not in a hung GPU)
            while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && ++polls < (1u << 18)) __builtin_amdgcn_s_sleep(4);
            if (polls >= (1u << 18)) __hip_atomic_fetch_add(wait_ctr + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float v = in[src];
    s[threadIdx.x] = v;
    __syncthreads();
    float a = s[(threadIdx.x + 1) & 255] * 0.f + v;
    for (int i = 0; i < work; ++i) a = a * 1.0000001f + 0.f;              // a dependent chain of `work` multiply-adds (kept exact: x * (1 + 2^-23) rounds back for small integers)
    out[idx] = (float)((int)(a + 0.5f)) + 1.f;
    if (done_ctr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const unsigned blocks = argc > 1 ? atoi(argv[1]) : 2040;
    const int work = argc > 2 ? atoi(argv[2]) : 2000, links = argc > 3 ? atoi(argv[3]) : 600;
    const unsigned n = blocks * 256;
    float* buf[2]; unsigned long long* ctr;
    CK(hipMalloc(&buf[0], n * 4)); CK(hipMalloc(&buf[1], n * 4)); CK(hipMalloc(&ctr, 8 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<float> h(n);
    for (int form = 0; form < 3; ++form) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(buf[0], 0, n * 4, s)); CK(hipMemsetAsync(buf[1], 0, n * 4, s)); CK(hipMemsetAsync(ctr, 0, 32, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < links; ++i) {
                const float* in = buf[i & 1]; float* out = buf[(i + 1) & 1];
                unsigned long long* w = form && i ? ctr : nullptr;
                unsigned long long* d = form ? ctr : nullptr;
                const unsigned long long expect = (unsigned long long)i * blocks;
                if (form == 1) hipExtLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, in, out, n, work, w, expect, d);
                else hipLaunchKernelGGL(k_link, dim3(blocks), dim3(256), 0, s, in, out, n, work, w, expect, d);
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipMemcpy(h.data(), buf[links & 1], n * 4, hipMemcpyDeviceToHost));
            unsigned bad = 0; for (unsigned i = 0; i < n; ++i) bad += h[i] != (float)links;
            unsigned long long hc[4]; CK(hipMemcpy(hc, ctr, 32, hipMemcpyDeviceToHost));
            printf("form %d: %d links of %u workgroups, work %d: %.2f us per link, %u wrong elements, %llu waits given up%s\n", form, links, blocks, work, us / links, bad, hc[1], rep ? "" : " (first)");
        }
    }
    return 0;
}
