// Which streams of one process wait for each other?  A process's streams are mapped onto a few hardware queues, and a hardware queue runs its kernels in order: a short
// kernel on stream j, launched just after a long one-wave sleeper on stream i, finishes at once if j has a queue of its own and after the sleeper if it shares i's.
// Prints, for every sleeper stream i, the completion latency of a tiny kernel (and of a 6 MB device-to-host copy) on every other stream.
//   hipcc --offload-arch=gfx950 -O3 queue_map.hip -o qm && ./qm [streams]      (also under GPU_MAX_HW_QUEUES=n)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k_sleep(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64); }
__global__ void k_tiny(int* p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 12;
    std::vector<hipStream_t> st(n);
    for (auto& s : st) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CHK(hipMalloc(&d, 4)); CHK(hipMemset(d, 0, 4));
    char *dbuf, *hbuf; const size_t nb = 6220800;
    CHK(hipMalloc(&dbuf, nb)); CHK(hipHostMalloc(&hbuf, nb));
    for (auto& s : st) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d); CHK(hipMemcpyAsync(hbuf, dbuf, nb, hipMemcpyDeviceToHost, s)); }   // every stream gets its queue now
    CHK(hipDeviceSynchronize());
    const long long ticks = 3 * 100000;                           // 3 ms at 100 MHz
    for (int mode = 0; mode < 2; ++mode) {
        printf("%s on stream j while a 3 ms sleeper runs on stream i: latency in ms (rows i, columns j)\n", mode ? "6 MB D2H copy" : "tiny kernel");
        for (int i = 0; i < n; ++i) {
            printf("i=%2d:", i);
            for (int j = 0; j < n; ++j) {
                if (j == i) { printf("    - "); continue; }
                hipLaunchKernelGGL(k_sleep, dim3(1), dim3(64), 0, st[i], ticks);
                const double t0 = now_ms();
                if (mode) CHK(hipMemcpyAsync(hbuf, dbuf, nb, hipMemcpyDeviceToHost, st[j])); else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st[j], d);
                CHK(hipStreamSynchronize(st[j]));
                printf(" %5.2f", now_ms() - t0);
                CHK(hipStreamSynchronize(st[i]));
            }
            printf("\n"); fflush(stdout);
        }
    }
    return 0;
}
