// Checks that glibc hypotf(x, y) == (float)sqrt((double)x*x + (double)y*y) bit for bit on 4e8 inputs of several distributions
// (the formula k_rotation_pairs uses on the device).  gcc -O2 hypotf_check.c -lm && ./a.out   -> "0 of ... differ" (glibc 2.35).
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static uint64_t s = 88172645463325252ull;
static uint64_t nx(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(void) {
    long bad = 0, n = 0;
    for (long i = 0; i < 400000000L; ++i) {
        float x, y;
        uint64_t r = nx();
        int mode = i & 3;
        if (mode == 0) { x = (float)((int)(r & 0xFFFFF) - 524288) / 256.f; y = (float)((int)((r >> 20) & 0xFFFFF) - 524288) / 256.f; }
        else if (mode == 1) { uint32_t a = (uint32_t)r, b = (uint32_t)(r >> 32); memcpy(&x, &a, 4); memcpy(&y, &b, 4); if (!isfinite(x) || !isfinite(y)) continue; }
        else if (mode == 2) { x = (float)((int)(r & 0xFFFF) - 32768) / 8.f; y = (float)((int)((r >> 16) & 0xFFFF) - 32768) / 8.f; }
        else { x = ldexpf((float)(r & 0xFFFFFF), -(int)((r >> 24) & 31)); y = ldexpf((float)((r >> 32) & 0xFFFFFF), -(int)((r >> 56) & 31)); }
        float h = hypotf(x, y);
        float g = (float)sqrt((double)x * (double)x + (double)y * (double)y);
        ++n;
        if (h != g && !(h != h && g != g)) { if (bad < 5) printf("mismatch %a %a: %a vs %a\n", x, y, h, g); ++bad; }
    }
    printf("%ld of %ld differ\n", bad, n);
    return 0;
}
