// dft_exact.cpp — see dft_exact.h.  Host only.
#include "dft_exact.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace poppy_hip {

int dft_optimal_size(int n) {
    for (int m = n;; ++m) {
        int t = m;
        while (t % 2 == 0) t /= 2;
        while (t % 3 == 0) t /= 3;
        while (t % 5 == 0) t /= 5;
        if (t == 1) return m;
    }
}

static unsigned char bitrev8(unsigned v) {
    unsigned r = 0;
    for (int b = 0; b < 8; ++b) r |= ((v >> b) & 1u) << (7 - b);
    return (unsigned char)r;
}

// DFTFactorize (dxt.cpp:158-200): the power of two first, then odd factors ascending, then the odd part reversed
static int factorize(int n, int* factors) {
    int nf = 0, f;
    if (n <= 5) { factors[0] = n; return 1; }
    f = (((n - 1) ^ n) + 1) >> 1;
    if (f > 1) { factors[nf++] = f; n = f == n ? 1 : n / f; }
    for (f = 3; n > 1;) {
        const int d = n / f;
        if (d * f == n) { factors[nf++] = f; n = d; }
        else { f += 2; if (f * f > n) break; }
    }
    if (n > 1) factors[nf++] = n;
    f = (factors[0] & 1) == 0;
    for (int i = f; i < (nf + f) / 2; i++) { const int t = factors[i]; factors[i] = factors[nf - i - 1 + f]; factors[nf - i - 1 + f] = t; }
    return nf;
}

// DFTInit (dxt.cpp:202-400) for the forward, non-inverted permutation table and float twiddles
void dft_make_plan(int n0, DftPlanHost& p) {
    p.n = n0;
    p.nf = factorize(n0, p.factors);
    p.itab.assign(n0, 0);
    p.wave.assign((size_t)n0 * 2, 0.f);
    int* itab = p.itab.data();
    const int* factors = p.factors;
    const int nf = p.nf;
    int digits[34], radix[34];
    int n = factors[0], m = 0;
    int i, j, k;
    if (n0 <= 5) {
        itab[0] = 0; itab[n0 - 1] = n0 - 1;
        if (n0 != 4) { for (i = 1; i < n0 - 1; i++) itab[i] = i; }
        else { itab[1] = 2; itab[2] = 1; }
        if (n0 == 5) { p.wave[0] = 1.f; p.wave[1] = 0.f; }
        if (n0 != 4) return;
        m = 2;
    } else {
        if (nf >= 34) throw std::runtime_error("dft_make_plan: too many factors");
        radix[nf] = 1; digits[nf] = 0;
        for (i = 0; i < nf; i++) { digits[i] = 0; radix[nf - i - 1] = radix[nf - i] * factors[nf - i - 1]; }
        if ((n & 1) == 0) {
            const int a = radix[1], na2 = n * a >> 1, na4 = na2 >> 1;
            for (m = 0; (unsigned)(1 << m) < (unsigned)n; m++) {}
            if (n <= 2) { itab[0] = 0; itab[1] = na2; }
            else if (n <= 256) {
                const int shift = 10 - m;
                for (i = 0; i <= n - 4; i += 4) {
                    j = (bitrev8(i >> 2) >> shift) * a;
                    itab[i] = j; itab[i + 1] = j + na2; itab[i + 2] = j + na4; itab[i + 3] = j + na2 + na4;
                }
            } else {
                const int shift = 34 - m;
                for (i = 0; i < n; i += 4) {
                    const unsigned i4 = (unsigned)(i >> 2);
                    const unsigned rev = ((unsigned)bitrev8(i4 & 255) << 24) + ((unsigned)bitrev8((i4 >> 8) & 255) << 16) +
                                         ((unsigned)bitrev8((i4 >> 16) & 255) << 8) + (unsigned)bitrev8(i4 >> 24);
                    j = (int)(rev >> shift) * a;
                    itab[i] = j; itab[i + 1] = j + na2; itab[i + 2] = j + na4; itab[i + 3] = j + na2 + na4;
                }
            }
            digits[1]++;
            if (nf >= 2) {
                for (i = n, j = radix[2]; i < n0;) {
                    for (k = 0; k < n; k++) itab[i + k] = itab[k] + j;
                    if ((i += n) >= n0) break;
                    j += radix[2];
                    for (k = 1; ++digits[k] >= factors[k]; k++) { digits[k] = 0; j += radix[k + 2] - radix[k]; }
                }
            }
        } else {
            for (i = 0, j = 0;;) {
                itab[i] = j;
                if (++i >= n0) break;
                j += radix[1];
                for (k = 0; ++digits[k] >= factors[k]; k++) { digits[k] = 0; j += radix[k + 2] - radix[k]; }
            }
        }
    }
    double wre, wim, w1re, w1im, t;
    if ((n0 & (n0 - 1)) == 0) {
        // DFTTab[m] (dxt.cpp:89-124) holds (cos, sin)(2 pi / 2^m) as decimal literals with 17 digits after the point, i.e. the
        // doubles one gets by printing the C library's values with "%.17f" and reading them back (checked against all 32
        // entries of the reference table in the container)
        const double ang = 2 * M_PI / (double)(1u << m);
        auto lit = [](double v) { char b[64]; snprintf(b, sizeof b, "%.17f", v); return strtod(b, nullptr); };
        const double c = m == 0 ? 1.0 : m == 1 ? -1.0 : m == 2 ? 0.0 : lit(std::cos(ang));
        // (one entry does not follow the rule: the table's sine for m = 7 ends in ...802 where the rule gives ...801)
        const double s = m <= 1 ? 0.0 : m == 2 ? 1.0 : m == 7 ? 0.04906767432741802 : lit(std::sin(ang));
        wre = w1re = c; wim = w1im = -s;
    } else {
        t = -M_PI * 2 / n0;
        wim = w1im = std::sin(t);
        wre = w1re = std::sqrt(1. - w1im * w1im);
    }
    n = (n0 + 1) / 2;
    float* wave = p.wave.data();
    wave[0] = 1.f; wave[1] = 0.f;
    if ((n0 & 1) == 0) { wave[2 * n] = -1.f; wave[2 * n + 1] = 0.f; }
    for (i = 1; i < n; i++) {
        wave[2 * i] = (float)wre; wave[2 * i + 1] = (float)wim;
        wave[2 * (n0 - i)] = (float)wre; wave[2 * (n0 - i) + 1] = (float)-wim;
        t = wre * w1re - wim * w1im;
        wim = wre * w1im + wim * w1re;
        wre = t;
    }
}

}  // namespace poppy_hip
