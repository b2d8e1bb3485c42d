// dft_exact.h — plan (factorisation, permutation table, twiddles) of cv::dft's mixed-radix complex float transform,
// OCV/core/src/dxt.cpp:158-400.  The reference's results depend on this exact factor order, on the twiddle table built
// by a double-precision recurrence, and on the butterflies' operation order (kernels_prefilter2.hip restates those).
#pragma once
#include <cstdint>
#include <vector>

namespace poppy_hip {

struct DftPlanHost {
    int n = 0, nf = 0;
    int factors[34] = {0};
    std::vector<int> itab;            // n entries
    std::vector<float> wave;          // n complex values (re, im)
};

void dft_make_plan(int n, DftPlanHost& plan);
int dft_optimal_size(int n);          // cv::getOptimalDFTSize for the 2^a 3^b 5^c sizes it tabulates

}  // namespace poppy_hip
