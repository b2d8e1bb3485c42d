// Microbenchmark (round 6): the SKELETON of k_warp_bin in today's layout and nothing else — a thread's id bytes (one coalesced dword per 4 pixels), the
// tile's 32 record slots (2560 B), the records of the thread's pixels, two 12-byte stores per 4 pixels; no division, no weights, no blend, no gather.
// The round-5 ablation (profiles/r05_warp_ablation.txt, mask 15) left 30.2 us at 4K / 15.2 us at 1080p for exactly this; the question the round-5 review
// asked: is there a form of the skeleton that is much faster?  Swept here:
//   threads per workgroup x pixels per thread for a 1024-pixel tile: 256 x 4 (today), 128 x 8, 64 x 16; larger tiles 256 x 8 (2048 px), 256 x 16 (4096 px)
//   records:  L  slots -> LDS -> barrier -> five ds_read_b128 per pixel (today)
//             S  no LDS, no barrier: the wave's ids are uniform (the benchmark's ids ARE: upper bound of the wave-uniform fast path) and the record
//                comes through scalar loads
//             G  no LDS, no barrier: five per-lane 16-byte loads per pixel straight from the slots (L1 / L2 hits)
//             N  no records at all (ids -> stores): the floor of the store stream
//   stores:   one buffer_store_dwordx3 per output and 4 pixels, or three dword stores
// Buffers rotate over `sets` copies so that nothing is found in the 256 MB MALL from the launch before.
//   hipcc --offload-arch=gfx950 -O3 warp_skeleton2.hip -o ws2 && ./ws2 3840 2160 && ./ws2 1920 1080
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
__device__ __forceinline__ int xcd_swizzle(int b, int n) {
    const int per = n >> 3, rem = n & 7;
    const int x = b & 7, s = b >> 3;
    return x * per + (x < rem ? x : rem) + s;
}

struct Set { const unsigned char* data; unsigned char* o1; unsigned char* o2; };
constexpr int kSlots = 32, kEntry = 80, kSlotBytes = kSlots * kEntry;

enum { REC_L = 0, REC_S = 1, REC_G = 2, REC_N = 3 };

// TILE_PX pixels per workgroup as TILE_W x (TILE_PX / TILE_W); a thread owns PX / 4 groups of 4 pixels (group q = tid + k * THREADS, row-major in the tile)
template <int THREADS, int PX, int TILE_W, int REC, int ST3>
__global__ void __launch_bounds__(THREADS) k_skel(Set s, int W, int H, int tiles_x, unsigned data_bytes, unsigned n_tiles) {
    constexpr int TILE_PX = THREADS * PX, TILE_H = TILE_PX / TILE_W, GX = TILE_W / 4, NG = PX / 4;
    constexpr int ID_BYTES = TILE_PX;
    __shared__ float4 s_rec[REC == REC_L ? kSlots * 5 : 1];
    const int tid = threadIdx.x;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const __amdgpu_buffer_rsrc_t rdata = rsrc(s.data, data_bytes);
    const unsigned npx = (unsigned)W * H;
    const __amdgpu_buffer_rsrc_t ro1 = rsrc(s.o1, npx * 3u), ro2 = rsrc(s.o2, npx * 3u);
    unsigned ids[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) ids[k] = __builtin_amdgcn_raw_buffer_load_b32(rdata, (unsigned)tile * ID_BYTES + (unsigned)(tid + k * THREADS) * 4u, 0, 0);
    const unsigned slot_base = n_tiles * ID_BYTES + (unsigned)tile * kSlotBytes;
    if (REC == REC_L) {
        for (int i = tid; i < kSlots * 5; i += THREADS)
            s_rec[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, slot_base + (unsigned)i * 16u, 0, 0));
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int q = tid + k * THREADS;
        const int row = q / GX, xg = q - row * GX;
        const int x0 = tx * TILE_W + xg * 4, y = ty * TILE_H + row;
        if (x0 >= W || y >= H) continue;
        unsigned a[3] = {ids[k], ids[k] >> 3, ids[k] >> 5}, b[3] = {ids[k] >> 1, ids[k] >> 2, ids[k] >> 4};
        if (REC != REC_N) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned li = (ids[k] >> (8 * p)) & 31u;
                float4 A, B, C, D, E;
                if (REC == REC_L) {
                    A = s_rec[li * 5]; B = s_rec[li * 5 + 1]; C = s_rec[li * 5 + 2]; D = s_rec[li * 5 + 3]; E = s_rec[li * 5 + 4];
                } else if (REC == REC_S) {
                    const unsigned lu = __builtin_amdgcn_readfirstlane(li);
                    const float4* r = (const float4*)(s.data + slot_base + lu * kEntry);       // uniform address: scalar loads
                    A = r[0]; B = r[1]; C = r[2]; D = r[3]; E = r[4];
                } else {
                    const unsigned o = slot_base + li * kEntry;
                    A = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, o, 0, 0));
                    B = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, o + 16, 0, 0));
                    C = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, o + 32, 0, 0));
                    D = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, o + 48, 0, 0));
                    E = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rdata, o + 64, 0, 0));
                }
                a[p % 3] ^= __float_as_uint(A.x) ^ __float_as_uint(B.y) ^ __float_as_uint(C.z);
                b[p % 3] ^= __float_as_uint(D.w) ^ __float_as_uint(E.x) ^ __float_as_uint(A.w);
            }
        }
        const unsigned g = (unsigned)y * (unsigned)(W >> 2) + (unsigned)(x0 >> 2);
        if (ST3) {
            __builtin_amdgcn_raw_buffer_store_b96(u3v{a[0], a[1], a[2]}, ro1, g * 12u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b96(u3v{b[0], b[1], b[2]}, ro2, g * 12u, 0, 0);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                __builtin_amdgcn_raw_buffer_store_b32(a[c], ro1, g * 12u + 4u * c, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(b[c], ro2, g * 12u + 4u * c, 0, 0);
            }
        }
    }
}

struct Bench {
    int W, H, sets, reps;
    std::vector<Set> dev;           // per tile size: rebuilt
    hipEvent_t e0, e1;
};

template <int THREADS, int PX, int TILE_W, int REC, int ST3>
static float run(int W, int H, int sets, int reps, const char* name) {
    constexpr int TILE_PX = THREADS * PX, TILE_H = TILE_PX / TILE_W;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H, n_tiles = tiles_x * tiles_y;
    const size_t data_bytes = (size_t)n_tiles * (TILE_PX + kSlotBytes) + 256, img = (size_t)W * H * 3;
    std::vector<unsigned char> h(data_bytes);
    unsigned st = 12345u;
    for (int t = 0; t < n_tiles; ++t) {                     // ids uniform per 64-pixel run of a tile row (a wave of mode S sees one id), 3-8 distinct per tile
        for (int i = 0; i < TILE_PX; i += 256) { st = st * 1664525u + 1013904223u; const unsigned char v = (unsigned char)((st >> 24) % 7); for (int j = 0; j < 256 && i + j < TILE_PX; ++j) h[(size_t)t * TILE_PX + i + j] = v; }
    }
    for (size_t i = (size_t)n_tiles * TILE_PX; i < data_bytes; ++i) { st = st * 1664525u + 1013904223u; h[i] = (unsigned char)(st >> 24); }
    std::vector<Set> dev(sets);
    for (Set& s : dev) {
        unsigned char* d; CHK(hipMalloc(&d, data_bytes)); CHK(hipMemcpy(d, h.data(), data_bytes, hipMemcpyHostToDevice));
        s.data = d; CHK(hipMalloc(&s.o1, img + 16)); CHK(hipMalloc(&s.o2, img + 16));
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto launch = [&](int i) {
        hipLaunchKernelGGL((k_skel<THREADS, PX, TILE_W, REC, ST3>), dim3(n_tiles), dim3(THREADS), 0, 0, dev[i % sets], W, H, tiles_x, (unsigned)data_bytes, (unsigned)n_tiles);
    };
    for (int i = 0; i < sets; ++i) launch(i);
    CHK(hipDeviceSynchronize());
    // every launch timed on its own (event pairs around single launches cost ~2 us of their own: the same for every form)
    float total = 0.f, best = 1e9f;
    for (int i = 0; i < reps; ++i) {
        CHK(hipEventRecord(e0)); launch(i); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); total += ms; best = ms < best ? ms : best;
    }
    // back to back (what the stream sees without the event packets)
    CHK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) launch(i); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float msb; CHK(hipEventElapsedTime(&msb, e0, e1));
    const double bytes = (double)n_tiles * (TILE_PX + (REC == REC_N ? 0 : kSlotBytes)) + 2.0 * img;
    printf("%-34s tiles %6d  avg %7.2f us  best %7.2f us  back-to-back %7.2f us  (%.2f TB/s of %.1f MB)\n", name, n_tiles, total / reps * 1e3, best * 1e3, msb / reps * 1e3,
           bytes / (msb / reps * 1e-3) / 1e12, bytes / 1e6);
    for (Set& s : dev) { CHK(hipFree((void*)s.data)); CHK(hipFree(s.o1)); CHK(hipFree(s.o2)); }
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return msb / reps * 1e3f;
}

#define RUN(T, P, TW, R, S) run<T, P, TW, R, S>(W, H, sets, reps, #T " thr x " #P " px, tile_w " #TW ", " #R ", " #S)

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160;
    const int sets = 6, reps = 60;
    printf("# warp skeleton, %d x %d, %d buffer sets, %d launches per form\n", W, H, sets, reps);
    if (W >= 3000) {
        RUN(256, 4, 128, REC_L, 1); RUN(256, 4, 128, REC_L, 0);
        RUN(128, 8, 128, REC_L, 1); RUN(64, 16, 128, REC_L, 1);
        RUN(256, 8, 128, REC_L, 1); RUN(256, 16, 128, REC_L, 1); RUN(256, 8, 256, REC_L, 1); RUN(256, 16, 256, REC_L, 1);
        RUN(256, 4, 128, REC_S, 1); RUN(256, 8, 128, REC_S, 1); RUN(256, 16, 128, REC_S, 1); RUN(64, 16, 128, REC_S, 1);
        RUN(256, 4, 128, REC_G, 1); RUN(256, 8, 128, REC_G, 1);
        RUN(256, 4, 128, REC_N, 1); RUN(256, 8, 128, REC_N, 1); RUN(256, 16, 128, REC_N, 1); RUN(256, 4, 128, REC_N, 0);
    } else {
        RUN(256, 4, 64, REC_L, 1); RUN(256, 4, 64, REC_L, 0);
        RUN(128, 8, 64, REC_L, 1); RUN(64, 16, 64, REC_L, 1);
        RUN(256, 8, 64, REC_L, 1); RUN(256, 16, 64, REC_L, 1); RUN(256, 8, 128, REC_L, 1); RUN(256, 16, 128, REC_L, 1);
        RUN(256, 4, 64, REC_S, 1); RUN(256, 8, 64, REC_S, 1); RUN(256, 16, 64, REC_S, 1); RUN(64, 16, 64, REC_S, 1);
        RUN(256, 4, 64, REC_G, 1); RUN(256, 8, 64, REC_G, 1);
        RUN(256, 4, 64, REC_N, 1); RUN(256, 8, 64, REC_N, 1); RUN(256, 16, 64, REC_N, 1); RUN(256, 4, 64, REC_N, 0);
    }
    return 0;
}
