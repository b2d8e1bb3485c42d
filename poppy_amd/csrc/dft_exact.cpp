// dft_exact.cpp — see dft_exact.h.  Host only.
//
// Written from the definition of the transform the kernels run — a decimation-in-time mixed-radix FFT whose passes go
// "whole power of two (radix 4, then at most one radix 2), then the odd primes from the largest to the smallest" — not from
// the reference's table-building code.  What has to coincide with cv::dft (OCV/core/src/dxt.cpp) for dft_detail2's bytes
// to coincide is the RESULT of the plan, and tests/test_host_plan.py::test_dft_plan_tables pins exactly that:
//   * the pass order: the output index i is read as digits d_0, d_1, ... (least significant first) over the factors in pass
//     order f_0, f_1, ...;
//   * the load permutation that makes every pass work in place: the element that ends up in slot i comes from
//     sum_k d_k * (n / (f_0 ... f_k)) — digit reversal — with the power-of-two digit additionally bit-reversed inside its
//     own range (that factor is itself carried out as radix-4 / radix-2 sub-passes);
//   * the twiddles exp(-2 pi i k / n) as FLOATS obtained by the recurrence w_{k+1} = w_k * w_1 in double, where w_1 comes from
//     a constant table for powers of two (dft_pow2_roots.inc, published constants) and from libm otherwise
//     (sin(2 pi / n), cos = sqrt(1 - sin^2)): the float roundings of this particular recurrence are what the reference
//     multiplies by, so a mathematically better table would give different low bits.
#include "dft_exact.h"
#include <cmath>
#include <stdexcept>

namespace poppy_hip {

namespace {

const double kPow2Roots[32][2] = {
#include "dft_pow2_roots.inc"
};

bool is_pow2(int n) { return (n & (n - 1)) == 0; }

int log2_exact(int n) { int m = 0; while ((1 << m) < n) ++m; return m; }

int bit_reverse(int v, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) if (v & (1 << b)) r |= 1 << (bits - 1 - b);
    return r;
}

}  // namespace

int dft_optimal_size(int n) {
    for (int m = n;; ++m) {
        int t = m;
        for (int f : {2, 3, 5}) while (t % f == 0) t /= f;
        if (t == 1) return m;
    }
}

void dft_make_plan(int n, DftPlanHost& p) {
    if (n < 1) throw std::runtime_error("dft_make_plan: length must be positive");
    p.n = n;
    // ---- pass order ---------------------------------------------------------------------------------------------------
    std::vector<int> order;
    const int two = n & -n;                                       // the whole power of two is one entry
    if (two > 1) order.push_back(two);
    {
        std::vector<int> primes;
        int rest = n / two;
        for (int f = 3; rest > 1; f += 2)
            while (rest % f == 0) { primes.push_back(f); rest /= f; }
        order.insert(order.end(), primes.rbegin(), primes.rend());   // largest odd prime first
    }
    if (order.empty()) order.push_back(1);                        // n == 1: a single trivial pass
    if (order.size() >= 34) throw std::runtime_error("dft_make_plan: too many factors");
    p.nf = (int)order.size();
    for (int k = 0; k < 34; ++k) p.factors[k] = k < p.nf ? order[k] : 0;

    // ---- load permutation: digit reversal over `order`, bit reversal inside the power of two --------------------------
    const int two_bits = two > 1 ? log2_exact(two) : 0;
    p.itab.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        int rem = i, place = n, from = 0;
        for (size_t k = 0; k < order.size(); ++k) {
            int digit = rem % order[k];
            rem /= order[k];
            place /= order[k];
            if (k == 0 && two > 1) digit = bit_reverse(digit, two_bits);
            from += digit * place;
        }
        p.itab[i] = from;
    }

    // ---- twiddles ---------------------------------------------------------------------------------------------------------
    double c1, s1;                                                // w_1 = c1 + i s1 = exp(-2 pi i / n)
    if (is_pow2(n)) {
        const int m = log2_exact(n);
        c1 = kPow2Roots[m][0]; s1 = -kPow2Roots[m][1];
    } else {
        s1 = std::sin(-M_PI * 2 / n);
        c1 = std::sqrt(1. - s1 * s1);
    }
    p.wave.assign((size_t)n * 2, 0.f);
    auto put = [&](int k, double re, double im) { p.wave[2 * (size_t)k] = (float)re; p.wave[2 * (size_t)k + 1] = (float)im; };
    put(0, 1.0, 0.0);
    const int half = (n + 1) / 2;
    if (n % 2 == 0) put(half, -1.0, 0.0);
    double re = c1, im = s1;
    for (int k = 1; k < half; ++k) {
        put(k, re, im);
        put(n - k, re, -im);                                      // the upper half mirrors the lower one
        const double nre = re * c1 - im * s1;
        im = re * s1 + im * c1;
        re = nre;
    }
}

}  // namespace poppy_hip
