// Which XCDs / CUs does a stream created with hipExtStreamCreateWithCUMask run on?  For a few masks: 4096 workgroups record
// HW_REG_XCC_ID and HW_REG_HW_ID; prints workgroups per XCD and the number of distinct (XCD, SE, SH, CU) places.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cumask_probe.hip -o tools/micro/cump ; gpurun -- ./tools/micro/cump
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
__global__ void k(uint32_t* out) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        out[blockIdx.x] = xcc << 28 | (hw & 0x0fffffffu);
    }
    // keep the block alive a little so that the grid spreads over everything it may use
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(1);
}
int main() {
    const int N = 4096;
    uint32_t* d; hipMalloc(&d, N * 4);
    std::vector<uint32_t> h(N);
    struct M { const char* name; std::vector<uint32_t> w; };
    std::vector<M> masks;
    masks.push_back({"all 256 bits", std::vector<uint32_t>(8, 0xffffffffu)});
    { std::vector<uint32_t> w(8, 0); for (int i = 0; i < 4; ++i) w[i] = 0xffffffffu; masks.push_back({"bits 0..127", w}); }
    { std::vector<uint32_t> w(8, 0); for (int i = 4; i < 8; ++i) w[i] = 0xffffffffu; masks.push_back({"bits 128..255", w}); }
    masks.push_back({"even bits", std::vector<uint32_t>(8, 0x55555555u)});
    masks.push_back({"bits with (i % 8) < 4", std::vector<uint32_t>(8, 0x0f0f0f0fu)});
    masks.push_back({"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}});
    for (auto& m : masks) {
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)m.w.size(), m.w.data()) != hipSuccess) { printf("%s: create failed\n", m.name); continue; }
        hipMemsetAsync(d, 0xff, N * 4, s);
        hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, N * 4, hipMemcpyDeviceToHost);
        int per[16] = {0}; std::set<uint32_t> places;
        for (uint32_t v : h) { per[v >> 28]++; places.insert((v >> 28) << 16 | ((v >> 8) & 0xff) | ((v >> 13) & 7) << 8); }
        printf("%-24s: workgroups per XCD", m.name);
        for (int x = 0; x < 8; ++x) printf(" %4d", per[x]);
        printf("   distinct places %zu\n", places.size());
        hipStreamDestroy(s);
    }
    return 0;
}
