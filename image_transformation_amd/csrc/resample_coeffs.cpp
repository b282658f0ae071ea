// Host-side resampling tables for the device passes (kernels_resample.hip).
//
// The reference resizes with Pillow (compositor.py:20: obj.resize((w, h), Image.LANCZOS)).  To be
// bit-exact with it the taps must be Pillow's own: Resample.c precompute_coeffs evaluates the
// normalised filter in double precision, normalize_coeffs_8bpc rounds each tap to 22-bit fixed
// point.  Doing that here, once per (in, out, filter) axis, leaves pure int32 work to the GPU.
#include "resample_coeffs.h"

#include <cmath>

namespace mic {

namespace {
inline double sinc(double x) {
    if (x == 0.0) return 1.0;
    x *= M_PI;
    return std::sin(x) / x;
}
inline double lanczos3(double x) { return (-3.0 <= x && x < 3.0) ? sinc(x) * sinc(x / 3) : 0.0; }
inline double triangle(double x) {
    x = std::fabs(x);
    return x < 1.0 ? 1.0 - x : 0.0;
}
}  // namespace

AxisTable build_axis_table(int in_size, int out_size, int filter) {
    AxisTable t;
    const bool bil = filter == 1;
    const double scale = static_cast<double>(in_size) / out_size;
    const double fscale = scale < 1.0 ? 1.0 : scale;
    const double support = (bil ? 1.0 : 3.0) * fscale;
    t.ksize = static_cast<int>(std::ceil(support)) * 2 + 1;
    t.out_size = out_size;
    t.bounds.assign(static_cast<size_t>(out_size) * 2, 0);
    t.coeffs.assign(static_cast<size_t>(out_size) * t.ksize, 0);
    std::vector<double> w(t.ksize);
    const double inv = 1.0 / fscale;
    for (int o = 0; o < out_size; ++o) {
        const double center = (o + 0.5) * scale;
        int first = static_cast<int>(center - support + 0.5);
        if (first < 0) first = 0;
        int last = static_cast<int>(center + support + 0.5);
        if (last > in_size) last = in_size;
        const int n = last - first;
        double total = 0.0;
        for (int k = 0; k < n; ++k) {
            const double arg = (k + first - center + 0.5) * inv;
            w[k] = bil ? triangle(arg) : lanczos3(arg);
            total += w[k];
        }
        int32_t *row = &t.coeffs[static_cast<size_t>(o) * t.ksize];
        for (int k = 0; k < n; ++k) {
            const double v = total != 0.0 ? w[k] / total : w[k];
            row[k] = v < 0 ? static_cast<int32_t>(-0.5 + v * (1 << 22))
                           : static_cast<int32_t>(0.5 + v * (1 << 22));
        }
        t.bounds[2 * o] = first;
        t.bounds[2 * o + 1] = n;
    }
    return t;
}

std::vector<int32_t> transpose_coeffs(const AxisTable &t) {
    std::vector<int32_t> out(t.coeffs.size());
    for (int o = 0; o < t.out_size; ++o)
        for (int k = 0; k < t.ksize; ++k)
            out[static_cast<size_t>(k) * t.out_size + o] = t.coeffs[static_cast<size_t>(o) * t.ksize + k];
    return out;
}

// Pillow Image.thumbnail's size rule (macro_placement_test.py:194 calls thumbnail((256,256), LANCZOS)).
void thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h) {
    int x = req_w, y = req_h;
    if (x >= w && y >= h) {
        *out_w = w;
        *out_h = h;
        return;
    }
    const double aspect = static_cast<double>(w) / h;
    auto pick = [](double v, auto key) {
        const long lo = static_cast<long>(std::floor(v)), hi = static_cast<long>(std::ceil(v));
        const long best = key(hi) < key(lo) ? hi : lo;  // min(floor, ceil, key=...) keeps floor on ties
        return static_cast<int>(best < 1 ? 1 : best);
    };
    if (static_cast<double>(x) / y >= aspect) {
        x = pick(y * aspect, [&](long n) { return std::fabs(aspect - static_cast<double>(n) / y); });
    } else {
        y = pick(x / aspect, [&](long n) {
            return n == 0 ? 0.0 : std::fabs(aspect - static_cast<double>(x) / n);
        });
    }
    *out_w = x;
    *out_h = y;
}

}  // namespace mic
