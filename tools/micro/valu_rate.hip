// Microbenchmark: issue rate of the vector instructions the frame kernels are made of, per SIMD, on gfx950.
// One launch = 256 x k workgroups of 256 threads (one wave per SIMD per workgroup, k workgroups per CU), every wave
// runs kIters x 8 independent chains of ONE instruction; lane 0 stamps s_memtime around the loop.  Reported: shader
// cycles per instruction per SIMD = k waves' instructions / the wave's own duration, for k = 1, 2, 4, 8.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o vr && ./vr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 512;

#define BODY8(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));

#define KERNEL32(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* stamps, unsigned seed) { \
    unsigned a[8]; \
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; \
    unsigned b = threadIdx.x * 7u + 3u + seed, c = 0x3f800001u + threadIdx.x; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int it = 0; it < kIters; ++it) { BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= a[i]; \
    out[blockIdx.x * 256 + threadIdx.x] = s; \
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t0; stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = t1; } \
}

// 64-bit operand kernels (packed fp32, fp64): a[] are register pairs
#define BODY8P(INS) \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));

#define KERNEL64(NAME, INS) \
__global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned long long* stamps, unsigned seed) { \
    double a[8]; \
    for (int i = 0; i < 8; ++i) a[i] = 1.0 + (threadIdx.x + i + seed) * 1e-3; \
    double b = 1.0000001 + seed * 1e-9, c = 1e-9; \
    const unsigned long long t0 = __builtin_readcyclecounter(); \
    for (int it = 0; it < kIters; ++it) { BODY8P(INS) BODY8P(INS) BODY8P(INS) BODY8P(INS) } \
    const unsigned long long t1 = __builtin_readcyclecounter(); \
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; \
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)__double2int_rn(s); \
    if ((threadIdx.x & 63) == 0) { stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t0; stamps[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = t1; } \
}

#define I_FMA(n)   "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MUL(n)   "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n)   "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_AND(n)   "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_ADDU(n)  "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_PERM(n)  "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ALIGN(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n"
#define I_DOT2(n)  "v_dot2_u32_u16 %" #n ", %8, %9, %" #n "\n"
#define I_DOT4(n)  "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define I_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define I_RCP(n)   "v_rcp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n)  "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_CVTUB(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define I_CVTI(n)  "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_MED3(n)  "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MIN3(n)  "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAX(n)   "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_BFE(n)   "v_bfe_u32 %" #n ", %" #n ", 3, 5\n"
#define I_LSHR(n)  "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMPCND(n) "v_cmp_lt_u32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n"
#define I_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define I_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_MOV(n)   "v_mov_b32 %" #n ", %8\n"
#define I_PKMOV(n) "v_pk_mov_b32 %" #n ", %" #n ", %8\n"


#define I_OR(n)    "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_XOR(n)   "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_LSHL(n)  "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_ASHR(n)  "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define I_SUBU(n)  "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MINU(n)  "v_min_u32 %" #n ", %" #n ", %8\n"
#define I_MAXI(n)  "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MINF(n)  "v_min_f32 %" #n ", %" #n ", %8\n"
#define I_SUBF(n)  "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_FMAC(n)  "v_fmac_f32 %" #n ", %8, %9\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_OR3(n)   "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD3(n)  "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %8\n"
#define I_BFI(n)   "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define I_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I_RNDNE(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define I_FRACT(n) "v_fract_f32 %" #n ", %" #n "\n"
#define I_CVTU(n)  "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define I_CVTFU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_CVTFI(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %8, 1, %" #n "\n"
#define I_LDEXP(n) "v_ldexp_f32 %" #n ", %" #n ", 1\n"
#define I_CMPF(n)  "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CMPU(n)  "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_CMPU64(n) "v_cmp_lt_u32 s[10:11], %" #n ", %8\n"
#define I_CND2(n)  "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define I_MADU16(n) "v_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO16(n) "v_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define I_ADDU16(n) "v_add_u16 %" #n ", %" #n ", %8\n"
#define I_PKADDU16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_PKMULLO16(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define I_PKLSHL16(n) "v_pk_lshlrev_b16 %" #n ", 1, %" #n "\n"
#define I_SADU8(n)  "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define I_CVTPKRTZ(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define I_FMAF16(n) "v_fma_f16 %" #n ", %" #n ", %8, %9\n"
#define I_PKFMAF16(n) "v_pk_fma_f16 %" #n ", %" #n ", %8, %9\n"
#define I_DOT2F16(n) "v_dot2_f32_f16 %" #n ", %8, %9, %" #n "\n"
#define I_DOT4I8(n) "v_dot4_i32_i8 %" #n ", %8, %9, %" #n "\n"
#define I_DOT8U4(n) "v_dot8_u32_u4 %" #n ", %8, %9, %" #n "\n"
#define I_MOVDPP(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(n) "v_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVSDWA(n) "v_mov_b32_sdwa %" #n ", %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n"
#define I_ORSDWA(n) "v_or_b32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
#define I_ADDSDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_MUL24SDWA(n) "v_mul_u32_u24_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0\n"
#define I_CVTUBSDWA(n) "v_cvt_f32_u32_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
#define I_ADDFSDWA(n) "v_add_f32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define I_READLANE(n) "v_readlane_b32 s10, %" #n ", 3\n"
#define I_READFIRST(n) "v_readfirstlane_b32 s10, %" #n "\n"
#define I_SWAP(n)  "v_swap_b32 %" #n ", %8\n"
#define I_BCNT(n)  "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %8\n"
#define I_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAX3(n)  "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MED3I(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_MAD_U64(n) "v_mad_u64_u32 %" #n ", s[10:11], %8, %9, %" #n "\n"
#define I_LSHL64(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n"
#define I_FMA_SGPR(n) "v_fma_f32 %" #n ", %" #n ", s4, %9\n"
#define I_MULLIT(n) "v_mul_f32 %" #n ", 0x3f7ffff0, %" #n "\n"

KERNEL32(k_fma, I_FMA) KERNEL32(k_mul, I_MUL) KERNEL32(k_add, I_ADD) KERNEL32(k_and, I_AND) KERNEL32(k_addu, I_ADDU)
KERNEL32(k_lshladd, I_LSHLADD) KERNEL32(k_mad24, I_MAD24) KERNEL32(k_mul24, I_MUL24) KERNEL32(k_mullo, I_MULLO)
KERNEL32(k_perm, I_PERM) KERNEL32(k_align, I_ALIGN) KERNEL32(k_dot2, I_DOT2) KERNEL32(k_dot4, I_DOT4) KERNEL32(k_pkmad, I_PKMAD)
KERNEL32(k_rcp, I_RCP) KERNEL32(k_sqrt, I_SQRT) KERNEL32(k_cvtub, I_CVTUB) KERNEL32(k_cvti, I_CVTI) KERNEL32(k_med3, I_MED3)
KERNEL32(k_min3, I_MIN3) KERNEL32(k_max, I_MAX) KERNEL32(k_bfe, I_BFE) KERNEL32(k_lshr, I_LSHR) KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_cmpcnd, I_CMPCND) KERNEL32(k_mov, I_MOV)
KERNEL64(k_pkfma, I_PKFMA) KERNEL64(k_pkmul, I_PKMUL) KERNEL64(k_pkadd, I_PKADD) KERNEL64(k_fma64, I_FMA64) KERNEL64(k_mul64, I_MUL64)
KERNEL64(k_add64, I_ADD64) KERNEL64(k_pkmov, I_PKMOV)

KERNEL32(k_or, I_OR) KERNEL32(k_xor, I_XOR) KERNEL32(k_lshl, I_LSHL) KERNEL32(k_ashr, I_ASHR) KERNEL32(k_subu, I_SUBU) KERNEL32(k_minu, I_MINU) KERNEL32(k_maxi, I_MAXI) KERNEL32(k_minf, I_MINF) KERNEL32(k_subf, I_SUBF) KERNEL32(k_fmac, I_FMAC) KERNEL32(k_andor, I_ANDOR) KERNEL32(k_or3, I_OR3) KERNEL32(k_add3, I_ADD3) KERNEL32(k_lshlor, I_LSHLOR) KERNEL32(k_bfi, I_BFI) KERNEL32(k_mulhi, I_MULHI) KERNEL32(k_floor, I_FLOOR) KERNEL32(k_rndne, I_RNDNE) KERNEL32(k_fract, I_FRACT) KERNEL32(k_cvtu, I_CVTU) KERNEL32(k_cvtfu, I_CVTFU) KERNEL32(k_cvtfi, I_CVTFI) KERNEL32(k_cvtpku8, I_CVTPKU8) KERNEL32(k_ldexp, I_LDEXP) KERNEL32(k_cmpf, I_CMPF) KERNEL32(k_cmpu, I_CMPU) KERNEL32(k_cmpu64, I_CMPU64) KERNEL32(k_cnd2, I_CND2) KERNEL32(k_madi24, I_MADI24) KERNEL32(k_madu16, I_MADU16) KERNEL32(k_mullo16, I_MULLO16) KERNEL32(k_addu16, I_ADDU16) KERNEL32(k_pkaddu16, I_PKADDU16) KERNEL32(k_pkmullo16, I_PKMULLO16) KERNEL32(k_pklshl16, I_PKLSHL16) KERNEL32(k_sadu8, I_SADU8) KERNEL32(k_cvtpkrtz, I_CVTPKRTZ) KERNEL32(k_fmaf16, I_FMAF16) KERNEL32(k_pkfmaf16, I_PKFMAF16) KERNEL32(k_dot2f16, I_DOT2F16) KERNEL32(k_dot4i8, I_DOT4I8) KERNEL32(k_dot8u4, I_DOT8U4) KERNEL32(k_movdpp, I_MOVDPP) KERNEL32(k_adddpp, I_ADDDPP) KERNEL32(k_movsdwa, I_MOVSDWA) KERNEL32(k_orsdwa, I_ORSDWA) KERNEL32(k_addsdwa, I_ADDSDWA) KERNEL32(k_mul24sdwa, I_MUL24SDWA) KERNEL32(k_cvtubsdwa, I_CVTUBSDWA) KERNEL32(k_addfsdwa, I_ADDFSDWA) KERNEL32(k_readlane, I_READLANE) KERNEL32(k_readfirst, I_READFIRST) KERNEL32(k_bcnt, I_BCNT) KERNEL32(k_mbcnt, I_MBCNT) KERNEL32(k_fmamix, I_FMAMIX) KERNEL32(k_max3, I_MAX3) KERNEL32(k_med3i, I_MED3I) KERNEL32(k_fma_sgpr, I_FMA_SGPR) KERNEL32(k_mullit, I_MULLIT)
KERNEL64(k_lshl64, I_LSHL64)

typedef void (*kern_t)(unsigned*, unsigned long long*, unsigned);

static void run(const char* name, kern_t k, int per_instr, unsigned* out, unsigned long long* stamps, std::vector<unsigned long long>& h) {
    printf("%-22s", name);
    for (int wg : {1, 2, 4, 8}) {
        const int blocks = 256 * wg;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, stamps, 1u);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, stamps, 2u);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
        std::vector<double> d;
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < blocks * 4; ++w) { d.push_back((double)(h[2 * w + 1] - h[2 * w])); lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
        std::sort(d.begin(), d.end());
        const double med = d[d.size() / 2];
        const double n = (double)kIters * 32 * per_instr;
        // cycles per instruction per SIMD with wg waves sharing it; counter = 100 MHz constant clock on gfx950 (s_memtime is the
        // shader clock on this part per the micro-architecture guide); wall-derived figure beside it
        printf("  k=%d: %6.2f cyc/instr/SIMD (wall %.1f us, span %.0f ticks)", wg, med / n / wg, ms * 1e3, (double)(hi - lo));
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    printf("\n");
}

int main() {
    unsigned* out; unsigned long long* stamps;
    CHK(hipMalloc(&out, 256 * 8 * 256 * 4));
    CHK(hipMalloc(&stamps, 256 * 8 * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(256 * 8 * 8);
#define R(NAME, K, N) run(NAME, K, N, out, stamps, h);
    R("v_fma_f32", k_fma, 1) R("v_mul_f32", k_mul, 1) R("v_add_f32", k_add, 1) R("v_mov_b32", k_mov, 1) R("v_and_b32", k_and, 1) R("v_add_u32", k_addu, 1)
    R("v_lshl_add_u32", k_lshladd, 1) R("v_mad_u32_u24", k_mad24, 1) R("v_mul_u32_u24", k_mul24, 1) R("v_mul_lo_u32", k_mullo, 1)
    R("v_perm_b32", k_perm, 1) R("v_alignbyte_b32", k_align, 1) R("v_dot2_u32_u16", k_dot2, 1) R("v_dot4_u32_u8", k_dot4, 1)
    R("v_pk_mad_u16", k_pkmad, 1) R("v_rcp_f32", k_rcp, 1) R("v_sqrt_f32", k_sqrt, 1) R("v_cvt_f32_ubyte0", k_cvtub, 1) R("v_cvt_i32_f32", k_cvti, 1)
    R("v_med3_f32", k_med3, 1) R("v_min3_f32", k_min3, 1) R("v_max_f32", k_max, 1) R("v_bfe_u32", k_bfe, 1) R("v_lshrrev_b32", k_lshr, 1)
    R("v_cndmask_b32", k_cndmask, 1) R("v_cmp+v_cndmask", k_cmpcnd, 2)
    R("v_pk_fma_f32", k_pkfma, 1) R("v_pk_mul_f32", k_pkmul, 1) R("v_pk_add_f32", k_pkadd, 1) R("v_pk_mov_b32", k_pkmov, 1)
    R("v_fma_f64", k_fma64, 1) R("v_mul_f64", k_mul64, 1) R("v_add_f64", k_add64, 1)
    R("or", k_or, 1) R("xor", k_xor, 1) R("lshl", k_lshl, 1) R("ashr", k_ashr, 1) R("subu", k_subu, 1) R("minu", k_minu, 1) R("maxi", k_maxi, 1) R("minf", k_minf, 1) R("subf", k_subf, 1) R("fmac", k_fmac, 1) R("andor", k_andor, 1) R("or3", k_or3, 1) R("add3", k_add3, 1) R("lshlor", k_lshlor, 1) R("bfi", k_bfi, 1) R("mulhi", k_mulhi, 1) R("floor", k_floor, 1) R("rndne", k_rndne, 1) R("fract", k_fract, 1) R("cvtu", k_cvtu, 1) R("cvtfu", k_cvtfu, 1) R("cvtfi", k_cvtfi, 1) R("cvtpku8", k_cvtpku8, 1) R("ldexp", k_ldexp, 1) R("cmpf", k_cmpf, 1) R("cmpu", k_cmpu, 1) R("cmpu64", k_cmpu64, 1) R("cnd2", k_cnd2, 1) R("madi24", k_madi24, 1) R("madu16", k_madu16, 1) R("mullo16", k_mullo16, 1) R("addu16", k_addu16, 1) R("pkaddu16", k_pkaddu16, 1) R("pkmullo16", k_pkmullo16, 1) R("pklshl16", k_pklshl16, 1) R("sadu8", k_sadu8, 1) R("cvtpkrtz", k_cvtpkrtz, 1) R("fmaf16", k_fmaf16, 1) R("pkfmaf16", k_pkfmaf16, 1) R("dot2f16", k_dot2f16, 1) R("dot4i8", k_dot4i8, 1) R("dot8u4", k_dot8u4, 1) R("movdpp", k_movdpp, 1) R("adddpp", k_adddpp, 1) R("movsdwa", k_movsdwa, 1) R("orsdwa", k_orsdwa, 1) R("addsdwa", k_addsdwa, 1) R("mul24sdwa", k_mul24sdwa, 1) R("cvtubsdwa", k_cvtubsdwa, 1) R("addfsdwa", k_addfsdwa, 1) R("readlane", k_readlane, 1) R("readfirst", k_readfirst, 1) R("bcnt", k_bcnt, 1) R("mbcnt", k_mbcnt, 1) R("fmamix", k_fmamix, 1) R("max3", k_max3, 1) R("med3i", k_med3i, 1) R("fma_sgpr", k_fma_sgpr, 1) R("mullit", k_mullit, 1) R("lshl64", k_lshl64, 1)
    return 0;
}
