// Median background colour: replaces _median_color_nontransparent (background_resizing.py:11-22),
// i.e. np.median over the R, G, B of pixels with alpha > 0 (over all pixels when none has), then
// int() truncation.
//
// An exact median of bytes needs no sort: a 256-bin histogram per channel plus a cumulative scan
// gives the order statistics (n-1)//2 and n//2 whose mean np.median returns.  One streaming read
// of the image (4 B/px): HBM-bound.
//
// Histogram kernel: per-wavefront private LDS histograms.  Backgrounds are mostly flat, so within
// a wave all lanes often hit the same bin; instead of letting 64 same-address LDS atomics serialise,
// the wave checks for that case with readlane/__ballot and lets one lane add the population count.
// Otherwise (noisy regions, mostly distinct bins) every lane issues its own LDS atomic.  Per-block
// totals are flushed with one global atomic per non-empty bin.
#include <algorithm>

#include "mic_internal.h"

namespace mic {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHistWaves = 4;

// hist layout (uint32): [set][channel][256], set 0 = alpha > 0, set 1 = all pixels;
// then counts[2] at kCountOff.
constexpr int kCountOff = 2 * 3 * 256;

__device__ __forceinline__ void wave_hist_add(uint32_t *h, uint32_t v, bool valid, int lane) {
    // Backgrounds are mostly flat: if every participating lane of the wave holds the same value, one
    // lane adds the population count (readfirstlane + compare + ballot); otherwise every lane issues
    // its own LDS atomic (noise spreads over the bins, so those rarely collide).
    const unsigned long long todo = __ballot(valid);
    if (todo == 0) return;
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lv = (uint32_t)__builtin_amdgcn_readlane((int)v, leader);
    if (__ballot(valid && v != lv) == 0) {
        if (lane == leader) atomicAdd(&h[lv], (uint32_t)__popcll(todo));
    } else if (valid) {
        atomicAdd(&h[v], 1u);
    }
}

// mode 0: pixels with alpha > 0.  mode 1: all pixels, skipped entirely unless counts[0] == 0.
__global__ __launch_bounds__(64 * kHistWaves) void median_hist_kernel(
    const uint32_t *__restrict__ px, size_t n_px, uint32_t *__restrict__ hist, int mode) {
    if (mode == 1 && hist[kCountOff] != 0) return;
    __shared__ uint32_t lh[kHistWaves][3][256];
    __shared__ uint32_t lcount[kHistWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kHistWaves * 3 * 256; i += blockDim.x) (&lh[0][0][0])[i] = 0;
    if (threadIdx.x < kHistWaves) lcount[threadIdx.x] = 0;
    __syncthreads();

    uint32_t *h0 = lh[wave][0], *h1 = lh[wave][1], *h2 = lh[wave][2];
    uint32_t my_count = 0;
    // wave-uniform trip count: every lane of a wave stays in the loop (ballot/readlane inside)
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t base = ((size_t)blockIdx.x * blockDim.x + (size_t)wave * 64) * 4; base < n_px;
         base += stride) {
        const size_t i = base + (size_t)lane * 4;
        uint32_t p[4];
        bool ok[4];
        if (i + 4 <= n_px) {
            u32x4 v;
            __builtin_memcpy(&v, px + i, 16);
            p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
            ok[0] = ok[1] = ok[2] = ok[3] = true;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[j] = i + j < n_px;
                p[j] = ok[j] ? px[i + j] : 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool take = ok[j] && (mode == 1 || (p[j] >> 24) != 0u);
            wave_hist_add(h0, p[j] & 255u, take, lane);
            wave_hist_add(h1, (p[j] >> 8) & 255u, take, lane);
            wave_hist_add(h2, (p[j] >> 16) & 255u, take, lane);
            my_count += take ? 1u : 0u;
        }
    }
    // wave-level count reduction (shuffle), one LDS add per wave
    for (int off = 32; off > 0; off >>= 1) my_count += __shfl_down(my_count, off);
    if (lane == 0) lcount[wave] = my_count;
    __syncthreads();

    uint32_t *gh = hist + (size_t)mode * 3 * 256;
    for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < kHistWaves; ++w) s += (&lh[w][0][0])[i];
        if (s) atomicAdd(&gh[i], s);
    }
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < kHistWaves; ++w) s += lcount[w];
        if (s) atomicAdd(&hist[kCountOff + mode], s);
    }
}

// One block: cumulative scan of each channel's 256 bins, pick order statistics (n-1)//2 and n//2,
// write int((lo + hi) / 2) -- np.median's mean of the two middle values, truncated by int().
__global__ __launch_bounds__(256) void median_select_kernel(const uint32_t *__restrict__ hist,
                                                            uint32_t *__restrict__ out_rgba) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t res[3][2];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int set = hist[kCountOff] != 0 ? 0 : 1;
    const uint32_t n = hist[kCountOff + set];
    if (n == 0) {  // empty image
        if (t == 0) out_rgba[0] = 0xff000000u;
        return;
    }
    const uint32_t klo = (n - 1) / 2, khi = n / 2;
    for (int c = 0; c < 3; ++c) {
        const uint32_t v = hist[(set * 3 + c) * 256 + t];
        uint32_t incl = v;  // inclusive scan within the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        incl += before;
        const uint32_t excl = incl - v;
        if (excl <= klo && klo < incl) res[c][0] = (uint32_t)t;
        if (excl <= khi && khi < incl) res[c][1] = (uint32_t)t;
        __syncthreads();
    }
    if (t == 0) {
        const uint32_t r = (res[0][0] + res[0][1]) >> 1;
        const uint32_t g = (res[1][0] + res[1][1]) >> 1;
        const uint32_t b = (res[2][0] + res[2][1]) >> 1;
        out_rgba[0] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

hipError_t launch_median(const void *rgba, size_t n_px, uint32_t *hist_dev, uint32_t *out_rgba_dev,
                         hipStream_t stream) {
    hipError_t e = hipMemsetAsync(hist_dev, 0, kMedianScratchWords * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    if (n_px > 0) {
        size_t blocks = (n_px + 256 * 4 * 8 - 1) / (256 * 4 * 8);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        const uint32_t *p = reinterpret_cast<const uint32_t *>(rgba);
        hipLaunchKernelGGL(median_hist_kernel, dim3((unsigned)blocks), dim3(64 * kHistWaves), 0, stream,
                           p, n_px, hist_dev, 0);
        // all-pixels fallback: exits at once unless the image had no pixel with alpha > 0
        hipLaunchKernelGGL(median_hist_kernel, dim3((unsigned)std::min<size_t>(blocks, 512)), dim3(64 * kHistWaves),
                           0, stream, p, n_px, hist_dev, 1);
    }
    hipLaunchKernelGGL(median_select_kernel, dim3(1), dim3(256), 0, stream, hist_dev, out_rgba_dev);
    return hipGetLastError();
}

}  // namespace mic
