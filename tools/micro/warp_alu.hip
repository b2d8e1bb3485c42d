// Microbenchmark: the ARITHMETIC of the fused warp kernel alone — k_warp_bin's map arithmetic (warp_taps) and blends (blend_fast) on register
// data, records from LDS, no global memory in the loop — run kReps times per wave with k workgroups of 256 threads per CU.  What it answers:
// how many shader cycles of a SIMD one tile-wave of that instruction stream really takes (the static listing priced by valu_rate.hip says
// ~1950; the kernel's duration at 4K corresponds to ~3000-3500 per wave).  mode 0: taps + blends, 1: taps only, 2: blends only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../poppy_amd/csrc -I../../include warp_alu.hip -o wa && ./wa
#include "warp_fast_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace poppy_hip;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kReps = 64;

template <int kMode>
__global__ void __launch_bounds__(256, 8) k_alu(const float4* __restrict__ recs, unsigned* out, unsigned long long* stamps, int W, int H, unsigned seed) {
    __shared__ float4 s_rec[160];
    const int tid = threadIdx.x;
    if (tid < 160) s_rec[tid] = recs[tid];
    __syncthreads();
    uint32_t ids = (tid * 2654435761u + seed) & 0x03030303u;     // entries 0..3
    int x0 = (tid & 31) * 4 + (seed & 1023), y = (tid >> 5) + (seed & 511);
    uint32_t acc = 0;
    uint32_t s0 = tid * 77u + seed, s1 = seed ^ 0x5bd1e995u, s2 = tid + 12345u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < kReps; ++r) {
        asm volatile("" : "+v"(ids), "+v"(x0), "+v"(y));
        FastTap t[2][4];
        const float fy = (float)y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (kMode != 2) {
                const int li = (int)((ids >> (8 * k)) & 255u);
                const float4 A = s_rec[li * 5], B = s_rec[li * 5 + 1], C = s_rec[li * 5 + 2], D = s_rec[li * 5 + 3], e4 = s_rec[li * 5 + 4];
                warp_taps(A, B, C, D, f2{e4.x, e4.y}, (float)(x0 + k), fy, W, H, t[0][k], t[1][k]);
            } else {
                t[0][k].wt = ids + k; t[0][k].wb = ids ^ k; t[0][k].off = x0 + k; t[0][k].inside = true;
                t[1][k].wt = ids * 3 + k; t[1][k].wb = ids ^ (k + 7); t[1][k].off = y + k; t[1][k].inside = true;
            }
        }
        if (kMode != 1) {
            // the footprints "arrive" here: 48 registers written by plain moves (full rate: ~100 issue cycles the real kernel does not have)
            u3v ra[2][4], rb[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int im = 0; im < 2; ++im) {
                    uint32_t v[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(v[q]) : "v"(q % 3 == 0 ? s0 : q % 3 == 1 ? s1 : s2));
                    ra[im][k] = u3v{v[0], v[1], v[2]}; rb[im][k] = u3v{v[3], v[4], v[5]};
                }
            uint32_t p[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int im = 0; im < 2; ++im) {
                    const uint32_t bs = t[im][k].off & 3u;
                    const u3v a3 = ra[im][k], b3 = rb[im][k];
                    const u2v a = {__builtin_amdgcn_alignbyte(a3.y, a3.x, bs), __builtin_amdgcn_alignbyte(a3.z, a3.y, bs)};
                    const u2v b = {__builtin_amdgcn_alignbyte(b3.y, b3.x, bs), __builtin_amdgcn_alignbyte(b3.z, b3.y, bs)};
                    p[im][k] = blend_fast(t[im][k], a, b);
                }
            const u3v o1 = {p[0][0] | (p[0][1] << 24), (p[0][1] >> 8) | (p[0][2] << 16), (p[0][2] >> 16) | (p[0][3] << 8)};
            const u3v o2 = {p[1][0] | (p[1][1] << 24), (p[1][1] >> 8) | (p[1][2] << 16), (p[1][2] >> 16) | (p[1][3] << 8)};
            acc ^= o1.x ^ o1.y ^ o1.z ^ o2.x ^ o2.y ^ o2.z;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc ^= t[0][k].wt ^ t[0][k].wb ^ t[0][k].off ^ t[1][k].wt ^ t[1][k].wb ^ t[1][k].off ^ (t[0][k].inside ? 1u : 0u) ^ (t[1][k].inside ? 2u : 0u);
        }
        ids = (ids + 0x01010101u) & 0x03030303u; x0 += 4; y += (r & 1); s0 += acc; s1 ^= s0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + tid] = acc;
    if ((tid & 63) == 0) { stamps[(blockIdx.x * 4 + tid / 64) * 2] = t0; stamps[(blockIdx.x * 4 + tid / 64) * 2 + 1] = t1; }
}

template <int kMode>
static void run(const char* what, const float4* recs, unsigned* out, unsigned long long* stamps, std::vector<unsigned long long>& h) {
    printf("%-34s", what);
    for (int wg : {1, 2, 4, 8}) {
        const int blocks = 256 * wg;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_alu<kMode>, dim3(blocks), dim3(256), 0, 0, recs, out, stamps, 3840, 2160, 1u);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_alu<kMode>, dim3(blocks), dim3(256), 0, 0, recs, out, stamps, 3840, 2160, 2u);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
        std::vector<double> d;
        for (int w = 0; w < blocks * 4; ++w) d.push_back((double)(h[2 * w + 1] - h[2 * w]));
        std::sort(d.begin(), d.end());
        // per SIMD: wg waves each ran kReps tile-waves in d ticks
        printf("  k=%d: %7.0f ticks per tile-wave per SIMD, wall %.3f us per tile-wave per SIMD", wg, d[d.size() / 2] / kReps / wg, ms * 1e3 / kReps / wg);
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    printf("\n"); fflush(stdout);
}

int main() {
    std::vector<float> rec(160 * 4, 0.f);
    for (int e = 0; e < 32; ++e) {                               // near-identity records in pack_warp_records' layout (A, B, C, D, E)
        float* r = &rec[e * 20];
        const float d = 1e-3f * (e + 1);
        r[0] = 1.f + d; r[1] = 1.f - d; r[2] = d; r[3] = -d; r[4] = 3.f * e; r[5] = -2.f * e; r[6] = d; r[7] = -d; r[8] = 1.f - d; r[9] = 1.f + d; r[10] = 1.5f * e; r[11] = 2.5f * e;
        r[12] = 1e-9f * e; r[13] = -1e-9f * e; r[14] = 2e-9f * e; r[15] = 1e-9f; r[16] = 1.f; r[17] = 1.f;
    }
    float4* d_rec; unsigned* out; unsigned long long* stamps;
    CHK(hipMalloc(&d_rec, 160 * 16)); CHK(hipMemcpy(d_rec, rec.data(), 160 * 16, hipMemcpyHostToDevice));
    CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHK(hipMalloc(&stamps, 256 * 8 * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(256 * 8 * 8);
    run<0>("map arithmetic + blends", d_rec, out, stamps, h);
    run<1>("map arithmetic only", d_rec, out, stamps, h);
    run<2>("blends only", d_rec, out, stamps, h);
    return 0;
}
