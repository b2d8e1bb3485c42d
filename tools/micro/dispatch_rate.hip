// Microbenchmark: what a grid costs before its threads do anything — the workgroup / wave launch rate of the chip.
// An (almost) empty kernel is launched with the same number of threads cut into workgroups of 64 ... 1024 threads,
// with and without an LDS allocation and a VGPR budget, and with a fixed amount of arithmetic per wave.
//   hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dr && ./dr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kThreads, int kLds, int kWork, int kStore>
__global__ void __launch_bounds__(kThreads) k_empty(unsigned* out, unsigned flag) {
    __shared__ unsigned s[kLds > 0 ? kLds / 4 : 1];
    unsigned v = threadIdx.x + flag;
    if (kLds > 0) { s[threadIdx.x % (kLds / 4)] = v; __syncthreads(); v += s[(threadIdx.x * 7) % (kLds / 4)]; }
    float a = (float)v, b = 1.0001f;
#pragma unroll 16
    for (int i = 0; i < kWork; ++i) a = __builtin_fmaf(a, b, 0.5f);
    v += (unsigned)a;
    if (kStore == 1) out[(size_t)blockIdx.x * kThreads + threadIdx.x] = v;          // 4 B per thread
    else if (kStore == 2) { if (flag == 12345u) out[threadIdx.x] = v; }               // never
}

template <int kThreads, int kLds, int kWork, int kStore>
static void run(const char* what, size_t total_threads, unsigned* out) {
    const unsigned blocks = (unsigned)(total_threads / kThreads);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_empty<kThreads, kLds, kWork, kStore>), dim3(blocks), dim3(kThreads), 0, 0, out, 1u);
    CHK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_empty<kThreads, kLds, kWork, kStore>), dim3(blocks), dim3(kThreads), 0, 0, out, 1u);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("  %-44s threads/WG %4d  WGs %7u  %8.2f us  %6.2f WG/ns  %6.2f waves/ns\n", what, kThreads, blocks, us, blocks / us / 1e3,
           total_threads / 64.0 / us / 1e3);
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
}

int main() {
    unsigned* out;
    CHK(hipMalloc(&out, (size_t)3840 * 2160 * 4 + 4096));
    for (size_t px : {(size_t)1920 * 1080 / 4, (size_t)3840 * 2160 / 4, (size_t)3840 * 2160}) {
        printf("%zu threads\n", px);
        run<64, 0, 0, 2>("empty", px, out);
        run<128, 0, 0, 2>("empty", px, out);
        run<256, 0, 0, 2>("empty", px, out);
        run<512, 0, 0, 2>("empty", px, out);
        run<1024, 0, 0, 2>("empty", px, out);
        run<256, 8192, 0, 2>("8 KB LDS + barrier", px, out);
        run<256, 32768, 0, 2>("32 KB LDS + barrier", px, out);
        run<1024, 32768, 0, 2>("32 KB LDS + barrier", px, out);
        run<256, 0, 0, 1>("4 B store per thread", px, out);
        run<1024, 0, 0, 1>("4 B store per thread", px, out);
        run<256, 0, 256, 2>("256 dependent fma per thread", px, out);
        run<256, 0, 1024, 2>("1024 dependent fma per thread", px, out);
        run<1024, 0, 1024, 2>("1024 dependent fma per thread", px, out);
        run<256, 8192, 1024, 1>("8 KB LDS + 1024 fma + store", px, out);
    }
    return 0;
}
