// Does v_cvt_pk_u8_f32 return what saturate_cast<uchar>(cvRound(x)) returns (the frame's final conversion,
// sat_u8(cv_round_x86(x)) in pyramid_device.h) for EVERY float bit pattern?  Counts the differing patterns and prints the first few.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I poppy_amd/csrc tools/micro/cvt_u8_probe.hip -o tools/micro/cvtp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../poppy_amd/csrc/pyramid_device.h"
using namespace poppy_hip;
__global__ void k(unsigned long long* n_bad, uint32_t* first) {
    const uint32_t bits0 = (blockIdx.x * 256u + threadIdx.x) << 8;
    for (uint32_t j = 0; j < 256; ++j) {
        const uint32_t b = bits0 + j;
        const float x = __uint_as_float(b);
        const uint32_t ref = sat_u8(cv_round_x86(x));
        const uint32_t got = __builtin_amdgcn_cvt_pk_u8_f32(x, 0, 0u);
        if (ref != got) { const unsigned long long i = atomicAdd(n_bad, 1ull); if (i < 16) { first[2 * i] = b; first[2 * i + 1] = ref | got << 8; } }
    }
}
int main() {
    unsigned long long* d; uint32_t* f; hipMalloc(&d, 8); hipMalloc(&f, 128); hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(1u << 16), dim3(256), 0, 0, d, f);
    unsigned long long n; uint32_t h[32]; hipMemcpy(&n, d, 8, hipMemcpyDeviceToHost); hipMemcpy(h, f, 128, hipMemcpyDeviceToHost);
    printf("differing bit patterns: %llu of 2^32\n", n);
    for (unsigned i = 0; i < (n < 16 ? n : 16); ++i) { float x; memcpy(&x, &h[2 * i], 4); printf("  %08x (%g): ref %u, cvt_pk_u8 %u\n", h[2 * i], x, h[2 * i + 1] & 255, h[2 * i + 1] >> 8); }
    return 0;
}
