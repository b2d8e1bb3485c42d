// Microbenchmark: what MIXED instruction streams cost per SIMD on gfx950 — the question valu_rate.hip (one kind of instruction per stream)
// left open after round 4's persistent warp kernels lost: is a kernel's time the sum of its VALU instructions' own costs (2 / 4 / 8
// cycles), or does every instruction — s_nop, s_waitcnt, scalar ALU, a full-rate v_fma beside a half-rate v_perm — take an issue slot?
// Same scheme as valu_rate.hip: 256 x k workgroups of 256 threads, every wave runs kIters x 4 copies of one 8-to-16-instruction body on 8
// independent register chains, lane 0 stamps s_memtime around the loop; reported: shader cycles per BODY per SIMD (k waves sharing it).
//   hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o im && ./im
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 512;

#define KERNEL(NAME, BODY) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* stamps, unsigned seed) { \
    unsigned a[8]; \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; \
    unsigned b = threadIdx.x * 7u + 3u + seed, c = 0x3f800001u + threadIdx.x; \
    __shared__ unsigned lds[512]; lds[threadIdx.x] = b; lds[threadIdx.x + 256] = c; __syncthreads(); \
    unsigned la = (threadIdx.x & 255) * 4; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int it = 0; it < kIters; ++it) { \
        asm volatile(BODY BODY BODY BODY : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c), "v"(la) : "s10", "s11", "vcc"); } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= a[i]; \
    out[blockIdx.x * 256 + threadIdx.x] = s; \
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t0; stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = t1; } \
}

#define FMA(n)  "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define DOT2(n) "v_dot2_u32_u16 %" #n ", %8, %9, %" #n "\n"
#define NOP     "s_nop 0\n"
#define SMOV    "s_mov_b32 s10, 7\n"
#define SADD    "s_add_u32 s10, s10, 3\n"
#define WAITV   "s_waitcnt vmcnt(0)\n"
#define WAITL   "s_waitcnt lgkmcnt(0)\n"
#define DSR(n)  "ds_read_b32 %" #n ", %10\n"
// packed fp32 on register pairs (0,1) (2,3) (4,5) (6,7)
#define PKF(lo, hi) "v_pk_fma_f32 v[%" #lo ":%" #hi "], v[%" #lo ":%" #hi "], v[%" #lo ":%" #hi "], v[%" #lo ":%" #hi "]\n"

KERNEL(k_fma8,        FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7))
KERNEL(k_perm8,       PERM(0) PERM(1) PERM(2) PERM(3) PERM(4) PERM(5) PERM(6) PERM(7))
KERNEL(k_fma4perm4,   FMA(0) PERM(1) FMA(2) PERM(3) FMA(4) PERM(5) FMA(6) PERM(7))
KERNEL(k_fma4perm4g,  FMA(0) FMA(2) FMA(4) FMA(6) PERM(1) PERM(3) PERM(5) PERM(7))
KERNEL(k_fma8addu8,   FMA(0) ADDU(1) FMA(2) ADDU(3) FMA(4) ADDU(5) FMA(6) ADDU(7) FMA(1) ADDU(0) FMA(3) ADDU(2) FMA(5) ADDU(4) FMA(7) ADDU(6))
KERNEL(k_fma8nop8,    FMA(0) NOP FMA(1) NOP FMA(2) NOP FMA(3) NOP FMA(4) NOP FMA(5) NOP FMA(6) NOP FMA(7) NOP)
KERNEL(k_perm8nop8,   PERM(0) NOP PERM(1) NOP PERM(2) NOP PERM(3) NOP PERM(4) NOP PERM(5) NOP PERM(6) NOP PERM(7) NOP)
KERNEL(k_perm8smov8,  PERM(0) SMOV PERM(1) SMOV PERM(2) SMOV PERM(3) SMOV PERM(4) SMOV PERM(5) SMOV PERM(6) SMOV PERM(7) SMOV)
KERNEL(k_fma8smov8,   FMA(0) SMOV FMA(1) SMOV FMA(2) SMOV FMA(3) SMOV FMA(4) SMOV FMA(5) SMOV FMA(6) SMOV FMA(7) SMOV)
KERNEL(k_perm8sadd8,  PERM(0) SADD PERM(1) SADD PERM(2) SADD PERM(3) SADD PERM(4) SADD PERM(5) SADD PERM(6) SADD PERM(7) SADD)
KERNEL(k_perm8wait8,  PERM(0) WAITV PERM(1) WAITV PERM(2) WAITV PERM(3) WAITV PERM(4) WAITV PERM(5) WAITV PERM(6) WAITV PERM(7) WAITV)
KERNEL(k_perm8dsr4,   PERM(0) DSR(1) PERM(2) PERM(3) DSR(1) PERM(4) PERM(5) DSR(1) PERM(6) PERM(7) DSR(1) PERM(0) WAITL)
KERNEL(k_dot2perm,    DOT2(0) PERM(1) DOT2(2) PERM(3) DOT2(4) PERM(5) DOT2(6) PERM(7))
KERNEL(k_dot2chain,   DOT2(0) DOT2(0) PERM(0) NOP DOT2(1) DOT2(1) PERM(1) NOP)
// the compiler's division chain: dependent v_pk_fma with the hazard nop between them, against two chains interleaved (no nop needed)
#define KERNEL64(NAME, BODY) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* stamps, unsigned seed) { \
    double a[8]; \
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + (threadIdx.x + i + seed) * 1e-3; \
    double b = 1.0000001 + seed * 1e-9, c = 1e-9; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int it = 0; it < kIters; ++it) { \
        asm volatile(BODY BODY BODY BODY : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c)); } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)__double2int_rn(s); \
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t0; stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = t1; } \
}
#define PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
KERNEL64(k_pkdep_nop, PK(0) NOP PK(0) NOP PK(0) NOP PK(0) NOP)
KERNEL64(k_pk2chains, PK(0) PK(1) PK(0) PK(1) PK(0) PK(1) PK(0) PK(1))
KERNEL64(k_pk8,       PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7))
KERNEL(k_fmadep,      "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %0, %0, %8, %9\n")
KERNEL(k_fma2dep,     "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\n")
KERNEL(k_permdep,     "v_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\nv_perm_b32 %0, %0, %8, %9\n")

typedef void (*kern_t)(unsigned*, unsigned long long*, unsigned);

static void run(const char* name, const char* what, kern_t k, unsigned* out, unsigned long long* stamps, std::vector<unsigned long long>& h) {
    printf("%-14s %-62s", name, what);
    for (int wg : {1, 2, 4, 8}) {
        const int blocks = 256 * wg;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, stamps, 1u);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, stamps, 2u);
        CHK(hipDeviceSynchronize());
        CHK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
        std::vector<double> d;
        for (int w = 0; w < blocks * 4; ++w) d.push_back((double)(h[2 * w + 1] - h[2 * w]));
        std::sort(d.begin(), d.end());
        printf("  k=%d: %6.1f", wg, d[d.size() / 2] / ((double)kIters * 4) / wg);
    }
    printf("   cycles per body per SIMD\n"); fflush(stdout);
}

int main() {
    unsigned* out; unsigned long long* stamps;
    CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHK(hipMalloc(&stamps, 256 * 8 * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(256 * 8 * 8);
#define R(K, WHAT) run(#K, WHAT, K, out, stamps, h);
    R(k_fma8, "8 v_fma_f32 (independent)")
    R(k_perm8, "8 v_perm_b32")
    R(k_fma4perm4, "4 v_fma + 4 v_perm alternating (sum of costs: 24)")
    R(k_fma4perm4g, "4 v_fma then 4 v_perm")
    R(k_fma8addu8, "8 v_fma + 8 v_add_u32 alternating (sum: 32)")
    R(k_fma8nop8, "8 v_fma + 8 s_nop 0")
    R(k_perm8nop8, "8 v_perm + 8 s_nop 0")
    R(k_perm8smov8, "8 v_perm + 8 s_mov_b32")
    R(k_fma8smov8, "8 v_fma + 8 s_mov_b32")
    R(k_perm8wait8, "8 v_perm + 8 s_waitcnt vmcnt(0) (nothing outstanding)")
    R(k_perm8dsr4, "9 v_perm + 4 ds_read_b32 + s_waitcnt lgkmcnt(0)")
    R(k_dot2perm, "4 v_dot2 + 4 v_perm alternating, independent")
    R(k_dot2chain, "2 x (dot2, dot2 accumulate, perm of the sum, s_nop)")
    R(k_pkdep_nop, "4 dependent v_pk_fma_f32 with s_nop 0 between (one chain)")
    R(k_pk2chains, "8 v_pk_fma_f32, two chains interleaved, no nop")
    R(k_pk8, "8 independent v_pk_fma_f32")
    R(k_fmadep, "8 dependent v_fma_f32 (one chain)")
    R(k_fma2dep, "8 v_fma_f32, two chains interleaved")
    R(k_permdep, "8 dependent v_perm_b32 (one chain)")
    return 0;
}
