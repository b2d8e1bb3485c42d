// Microbenchmark: what an LDS atomic costs on gfx950 as a function of the lanes that take part.
// The median kernels are bound by ds_add_u32 to per-lane histograms ([bin][lane], conflict-free); this measures the LDS pipe's time per
// instruction with all 64, 32, 16, 8, 1 lanes active (the rest masked off by exec), contiguous or strided, against ds_read / ds_write.
// One workgroup of 256 threads per CU x k (k = 1, 4 workgroups per CU); wall clock per launch / instructions per CU.
//   hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lar && ./lar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 2048, kUnroll = 16;

// mode 0: ds_add_u32 (no return), 1: ds_read_b32, 2: ds_write_b32, 3: ds_add_u32 on 16-bit-shared dwords (lanes l and l+32 hit the same dword)
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(unsigned* out, unsigned long long active_mask, unsigned seed) {
    __shared__ unsigned lds[64 * 128];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 128; i += 256) lds[i] = 0;
    __syncthreads();
    unsigned acc = 0;
    const bool on = (active_mask >> lane) & 1ull;
    unsigned bin = (lane * 7u + seed + wv * 13u) & 31u;
    if (on) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const unsigned b = (bin + u * 5u) & 31u;
                unsigned* p = MODE == 3 ? &lds[(wv * 32 + b) * 32 + (lane & 31)] : &lds[(wv * 32 + b) * 64 + lane];
                if (MODE == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 3) __hip_atomic_fetch_add(p, 1u << ((lane >> 5) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 1) acc += *(volatile unsigned*)p;
                else *(volatile unsigned*)p = acc + u;
            }
            bin = (bin + 3u) & 31u;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE>
static void run(const char* name, unsigned* d_out, int wgs_per_cu) {
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0xffull, 1ull, 0x5555555555555555ull, 0x1111111111111111ull, 0x0101010101010101ull, 0x0000000100000001ull};
    const char* mnames[] = {"64 lanes", "32 low", "16 low", "8 low", "1 lane", "every 2nd (32)", "every 4th (16)", "every 8th (8)", "lanes 0 + 32"};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int m = 0; m < 9; ++m) {
        const int grid = 256 * wgs_per_cu;
        hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, d_out, masks[m], 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 0, 0, d_out, masks[m], (unsigned)r);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_cu = 5.0 * wgs_per_cu * 4 * (double)kIters * kUnroll;     // wave instructions per CU
        printf("%-28s %d WG/CU  %-16s %7.2f ns per wave instruction per CU  (%.1f cycles at 2.4 GHz)\n", name, wgs_per_cu, mnames[m], ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    }
}
int main() {
    unsigned* d_out; CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    for (int k : {1, 4}) {
        run<0>("ds_add_u32 [bin][lane]", d_out, k);
        run<3>("ds_add_u32 l / l+32 share", d_out, k);
        run<1>("ds_read_b32", d_out, k);
        run<2>("ds_write_b32", d_out, k);
    }
    return 0;
}
