// Host-side resampling tables for the device passes (kernels_resample.hip).
//
// The reference resizes with Pillow (compositor.py:20: obj.resize((w, h), Image.LANCZOS)).  To be
// bit-exact with it the taps must be Pillow's own: Resample.c precompute_coeffs evaluates the
// normalised filter in double precision, normalize_coeffs_8bpc rounds each tap to 22-bit fixed
// point.  Doing that here, once per (in, out, filter) axis, leaves pure int32 work to the GPU.
#include "resample_coeffs.h"

#include <algorithm>
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

AxisTable identity_axis_table(int size) {
    AxisTable t;
    t.out_size = size;
    t.ksize = 1;
    t.bounds.resize(static_cast<size_t>(size) * 2);
    t.coeffs.assign(static_cast<size_t>(size), 1 << 22);  // clip8((2^21 + s * 2^22) >> 22) == s
    for (int o = 0; o < size; ++o) {
        t.bounds[2 * o] = o;
        t.bounds[2 * o + 1] = 1;
    }
    return t;
}

namespace {
// first tap / one past the last tap of the 16 outputs of a tile
inline void tile_span(const AxisTable &t, int tile, int *lo_out, int *hi_out) {
    const int o0 = tile * 16, o1 = std::min(t.out_size, o0 + 16);
    int lo = t.bounds[2 * o0], hi = lo;
    for (int o = o0; o < o1; ++o) {
        lo = std::min(lo, t.bounds[2 * o]);
        hi = std::max(hi, t.bounds[2 * o] + t.bounds[2 * o + 1]);
    }
    *lo_out = lo;
    *hi_out = hi;
}
}  // namespace

size_t axis_frags_layout(const AxisTable &t, AxisFrags *f, int form) {
    f->tiles = (t.out_size + 15) / 16;
    f->max_chunks = 0;
    f->meta.assign(static_cast<size_t>(f->tiles) * 4, 0);
    size_t chunks = 0;
    if (form == kFragsLaneH) {
        for (int tile = 0; tile < f->tiles;) {
            int lo, hi;
            tile_span(t, tile, &lo, &hi);
            const int ws = lo & ~15;
            int n = 1;
            if (hi - ws <= 64 && tile + 1 < f->tiles) {
                int lo2, hi2;
                tile_span(t, tile + 1, &lo2, &hi2);
                if (hi2 - ws <= 64) n = 2;
            }
            for (int j = 0; j < n; ++j) {
                int lo_j, hi_j;
                tile_span(t, tile + j, &lo_j, &hi_j);
                const int n_chunks = std::max(1, (hi_j - ws + 63) / 64);  // (> 1 only for a lone tile with a wide window)
                f->meta[4 * (tile + j) + 0] = ws;
                f->meta[4 * (tile + j) + 1] = j == 0 ? n : 0;
                f->meta[4 * (tile + j) + 2] = static_cast<int32_t>(chunks);
                f->meta[4 * (tile + j) + 3] = hi_j;
                f->max_chunks = std::max(f->max_chunks, n_chunks);
                chunks += 1;  // one fragment per tile, whatever: a disqualified axis is never read in this form
            }
            tile += n;
        }
        return chunks;
    }
    for (int tile = 0; tile < f->tiles; ++tile) {
        int lo, hi;
        tile_span(t, tile, &lo, &hi);
        if (form == kFragsLaneV) {
            const int b_first = lo >> 4, b_last = (hi - 1) >> 4;
            int need = 0;
            for (int b = b_first; b <= b_last; ++b) need |= 1 << (b & 3);
            f->meta[4 * tile + 0] = lo;
            f->meta[4 * tile + 1] = b_last | (need << 24);
            f->meta[4 * tile + 2] = static_cast<int32_t>(chunks);
            f->meta[4 * tile + 3] = hi;
            f->max_chunks = std::max(f->max_chunks, b_last - b_first > 3 ? 2 : 1);
            chunks += 1;
            continue;
        }
        const int ws = lo & ~15;
        const int n_chunks = std::max(1, (hi - ws + 63) / 64);
        f->meta[4 * tile + 0] = ws;
        f->meta[4 * tile + 1] = n_chunks;
        f->meta[4 * tile + 2] = static_cast<int32_t>(chunks);
        f->meta[4 * tile + 3] = hi;
        f->max_chunks = std::max(f->max_chunks, n_chunks);
        chunks += (size_t)n_chunks;
    }
    return chunks;
}

void fill_axis_frags(const AxisTable &t, const AxisFrags &f, int32_t *bias, int8_t *frags, size_t chunks_total, int form) {
    std::fill(bias, bias + static_cast<size_t>(f.tiles) * 16, 0);
    std::fill(frags, frags + chunks_total * 3 * 64 * 16, 0);
    const bool lane = form == kFragsLaneH || form == kFragsLaneV;
    for (int tile = 0; tile < f.tiles; ++tile) {
        const int o0 = tile * 16, o1 = std::min(t.out_size, o0 + 16);
        const int ws = f.meta[4 * tile + 0];
        const size_t chunks = (size_t)f.meta[4 * tile + 2];  // (this tile's first chunk)
        for (int o = o0; o < o1; ++o) {
            const int first = t.bounds[2 * o], n = t.bounds[2 * o + 1];
            const int32_t *row = &t.coeffs[static_cast<size_t>(o) * t.ksize];
            int64_t sum = 0;
            for (int k = 0; k < n; ++k) {
                const int32_t c = row[k];
                sum += c;
                const int32_t d0 = ((c + 128) & 255) - 128;
                const int32_t c1 = (c - d0) >> 8;
                const int32_t d1 = ((c1 + 128) & 255) - 128;
                const int32_t d2 = (c1 - d1) >> 8;  // |c| < 2^23 keeps it a signed byte
                int chunk, h, j;
                if (form == kFragsLaneV) {
                    const int r = first + k;  // absolute source row: ring word (r >> 4) & 3 of lane quarter (r & 15) >> 2
                    chunk = 0; h = (r & 15) >> 2; j = 4 * ((r >> 4) & 3) + (r & 3);
                } else {
                    const int pos = first + k - ws;  // window position of this tap
                    chunk = pos / 64; h = (pos % 64) / 16; j = pos % 16;
                }
                if (lane && chunk > 0) continue;  // (a disqualified axis: its table is never read in this form)
                const int lane_i = 16 * h + (o - o0);
                int8_t *base = &frags[((chunks + chunk) * 3 * 64 + lane_i) * 16 + j];
                base[0 * 64 * 16] = static_cast<int8_t>(d0);
                base[1 * 64 * 16] = static_cast<int8_t>(d1);
                base[2 * 64 * 16] = static_cast<int8_t>(d2);
            }
            bias[o] = static_cast<int32_t>((1 << 21) + 128 * sum);
        }
    }
}

AxisFrags build_axis_frags(const AxisTable &t, int form) {
    AxisFrags f;
    const size_t chunks = axis_frags_layout(t, &f, form);
    f.bias.resize(static_cast<size_t>(f.tiles) * 16);
    f.frags.resize(chunks * 3 * 64 * 16);
    fill_axis_frags(t, f, f.bias.data(), f.frags.data(), chunks, form);
    return f;
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
