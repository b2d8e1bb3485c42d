// Microbenchmark: what the memory system gives a kernel with k_warp_tile's access mix and NONE of its arithmetic.
// Same geometry (256 threads, 4 pixels per thread, 128 x 8 or 64 x 16 tiles, XCD-interleaved tile order), same
// streams per 4-pixel group: ids 16 B read + 16 B written (clear), m2 16 B read, mask 16 B written, two sources read
// as footprint rows (mode 1: one 12-byte row per source, i.e. a plain copy; mode 2: 4 pixels x 2 rows x 12 bytes per
// source at the identity positions = the real kernel's 16 gathers), two outputs 12 B written.  Buffers rotate over
// `sets` copies so that nothing is found in the 256 MB MALL from the launch before.
//   hipcc --offload-arch=gfx950 -O3 warp_skeleton.hip -o ws && ./ws 3840 2160
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u3v __attribute__((ext_vector_type(3)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

struct Set { unsigned* ids; float* m2; float* mask; unsigned char* c1; unsigned char* c2; unsigned char* o1; unsigned char* o2; };

template <int kTileW, int kMode>
__global__ void __launch_bounds__(256) k_skeleton(Set s, int W, int H, int tiles_x) {
    constexpr int kTileH = 1024 / kTileW, kTileTx = kTileW / 4;
    const int tid = threadIdx.x;
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned per = nb / 8, rem = nb % 8, x = b % 8, q = b / 8;                 // XCD x takes a contiguous run of tiles
    const int tile = (int)(x * per + (x < rem ? x : rem) + q);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * kTileW + (tid % kTileTx) * 4, y = ty * kTileH + tid / kTileTx;
    if (x0 >= W || y >= H) return;
    const unsigned g = (unsigned)y * (unsigned)(W >> 2) + (unsigned)(x0 >> 2);
    const unsigned npx = (unsigned)W * H, pitch = (unsigned)W * 3u;
    const auto rmap = rsrc(s.ids, npx * 4u), rm2 = rsrc(s.m2, npx * 4u), rmask = rsrc(s.mask, npx * 4u);
    const auto rs1 = rsrc(s.c1, npx * 3u + 16u), rs2 = rsrc(s.c2, npx * 3u + 16u), ro1 = rsrc(s.o1, npx * 3u), ro2 = rsrc(s.o2, npx * 3u);
    const u4v ids = __builtin_amdgcn_raw_buffer_load_b128(rmap, g * 16u, 0, 0);
    const u4v m2 = __builtin_amdgcn_raw_buffer_load_b128(rm2, g * 16u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(u4v{0u, 0u, 0u, 0u}, rmap, g * 16u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(m2 + ids, rmask, g * 16u, 0, 0);
    u3v a1 = {0u, 0u, 0u}, a2 = {0u, 0u, 0u};
    if (kMode == 1) {
        a1 = __builtin_amdgcn_raw_buffer_load_b96(rs1, g * 12u, 0, 0);
        a2 = __builtin_amdgcn_raw_buffer_load_b96(rs2, g * 12u, 0, 0);
    } else {
        const int yy = y < H - 1 ? y : H - 2;
        u3v r[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned o = ((unsigned)yy * (unsigned)W + (unsigned)(x0 + k)) * 3u & ~3u;
            r[4 * k + 0] = __builtin_amdgcn_raw_buffer_load_b96(rs1, o, 0, 0);
            r[4 * k + 1] = __builtin_amdgcn_raw_buffer_load_b96(rs1, o, (int)pitch, 0);
            r[4 * k + 2] = __builtin_amdgcn_raw_buffer_load_b96(rs2, o, 0, 0);
            r[4 * k + 3] = __builtin_amdgcn_raw_buffer_load_b96(rs2, o, (int)pitch, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1 += r[4 * k] ^ r[4 * k + 1]; a2 += r[4 * k + 2] ^ r[4 * k + 3]; }
    }
    __builtin_amdgcn_raw_buffer_store_b96(a1, ro1, g * 12u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b96(a2, ro2, g * 12u, 0, 0);
}

template <int TW, int MODE>
static float run(const std::vector<Set>& sets, int W, int H, int iters, hipStream_t st) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + 1024 / TW - 1) / (1024 / TW);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_skeleton<TW, MODE>), dim3(tiles_x * tiles_y), dim3(256), 0, st, sets[i % sets.size()], W, H, tiles_x);
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_skeleton<TW, MODE>), dim3(tiles_x * tiles_y), dim3(256), 0, st, sets[i % sets.size()], W, H, tiles_x);
    CHK(hipEventRecord(e1, st));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160;
    const size_t npx = (size_t)W * H;
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    for (int nsets : {1, 6}) {
        std::vector<Set> sets(nsets);
        for (auto& s : sets) {
            CHK(hipMalloc(&s.ids, npx * 4)); CHK(hipMalloc(&s.m2, npx * 4)); CHK(hipMalloc(&s.mask, npx * 4));
            CHK(hipMalloc(&s.c1, npx * 3 + 16)); CHK(hipMalloc(&s.c2, npx * 3 + 16)); CHK(hipMalloc(&s.o1, npx * 3)); CHK(hipMalloc(&s.o2, npx * 3));
            CHK(hipMemset(s.ids, 0, npx * 4)); CHK(hipMemset(s.m2, 0, npx * 4)); CHK(hipMemset(s.c1, 7, npx * 3 + 16)); CHK(hipMemset(s.c2, 9, npx * 3 + 16));
        }
        const double mb = npx * 28.0 / 1e6;
        const float t1a = run<128, 1>(sets, W, H, 60, st), t1b = run<64, 1>(sets, W, H, 60, st);
        const float t2a = run<128, 2>(sets, W, H, 60, st), t2b = run<64, 2>(sets, W, H, 60, st);
        printf("%dx%d, %d buffer set(s) (%.0f MB each), 28 B/px = %.0f MB per launch\n", W, H, nsets, npx * 24.0 / 1e6, mb);
        printf("  streams only          : tile 128x8 %.1f us (%.2f TB/s)   tile 64x16 %.1f us (%.2f TB/s)\n", t1a, mb / t1a, t1b, mb / t1b);
        printf("  streams + 16 gathers  : tile 128x8 %.1f us (%.2f TB/s)   tile 64x16 %.1f us (%.2f TB/s)\n", t2a, mb / t2a, t2b, mb / t2b);
        for (auto& s : sets) { hipFree(s.ids); hipFree(s.m2); hipFree(s.mask); hipFree(s.c1); hipFree(s.c2); hipFree(s.o1); hipFree(s.o2); }
    }
    return 0;
}
