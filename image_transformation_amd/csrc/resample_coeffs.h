// Host-side tables for one resample axis (see resample_coeffs.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace mic {

struct AxisTable {
    int out_size = 0;
    int ksize = 0;                 // taps allotted per output sample
    std::vector<int32_t> bounds;   // [out_size][2]: first input index, tap count
    std::vector<int32_t> coeffs;   // [out_size][ksize], 22-bit fixed point, unused taps zero
};

AxisTable build_axis_table(int in_size, int out_size, int filter);
std::vector<int32_t> transpose_coeffs(const AxisTable &t);  // -> [ksize][out_size]
void thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h);

}  // namespace mic
