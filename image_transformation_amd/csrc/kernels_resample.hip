// Pillow-exact separable resampling of RGBA cutouts (Image.resize(size, LANCZOS), call site
// compositor.py:20; thumbnails macro_placement_test.py:194).
//
// Restates Pillow's 8 bpc path: premultiply (Convert.c rgbA2rgba) -> horizontal pass -> 8-bit
// intermediate -> vertical pass -> unpremultiply (Convert.c rgba2rgbA).  Each output sample is
//   clip8((2^21 + sum_k in[first + k] * K[k]) >> 22)
// with int32 fixed-point coefficients K built on the host in double precision exactly as
// Resample.c precompute_coeffs / normalize_coeffs_8bpc do (resample_coeffs.cpp), so the device
// passes are pure integer arithmetic and bit-exact.
//
// One launch processes every resampled layer of a composite call (blockIdx.z = layer).
#include "mic_internal.h"

namespace mic {

__device__ __forceinline__ uint32_t div255_shift(uint32_t t) { return ((t >> 8) + t) >> 8; }

__device__ __forceinline__ uint32_t premultiply(uint32_t p) {
    const uint32_t a = p >> 24;
    const uint32_t r = div255_shift((p & 255u) * a + 128u);
    const uint32_t g = div255_shift(((p >> 8) & 255u) * a + 128u);
    const uint32_t b = div255_shift(((p >> 16) & 255u) * a + 128u);
    return r | (g << 8) | (b << 16) | (a << 24);
}

__device__ __forceinline__ uint32_t unpremultiply(uint32_t p) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const uint32_t r = min(255u, (255u * (p & 255u)) / a);
    const uint32_t g = min(255u, (255u * ((p >> 8) & 255u)) / a);
    const uint32_t b = min(255u, (255u * ((p >> 16) & 255u)) / a);
    return r | (g << 8) | (b << 16) | (a << 24);
}

__device__ __forceinline__ uint32_t clip8(int32_t v) {
    v >>= kPrecisionBits;  // arithmetic shift, like Pillow's clip8 lookup index
    // hipcc (ROCm 7.2, gfx950) fuses "shift, clamp to 0..255, pack" into v_ashr_pk_u8_i32 and then
    // ORs the 16-bit result as if the destination's upper half were zero; it is not (the
    // instruction only writes D[15:0]), which corrupted blue/alpha on the MI355X.  The empty asm
    // keeps the shift and the clamp apart so the clamp lowers to v_med3_i32.
    asm volatile("" : "+v"(v));
    return (uint32_t)min(255, max(0, v));
}

__device__ __forceinline__ uint32_t pack_clip(int32_t s0, int32_t s1, int32_t s2, int32_t s3) {
    return clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16) | (clip8(s3) << 24);
}

// Horizontal pass: one thread per output pixel (x', y).  Coefficients are stored TRANSPOSED
// ([ksize][out_w]) so that the 64 lanes of a wave read consecutive words for each tap.
__global__ __launch_bounds__(256) void resample_h_kernel(const RsJob *__restrict__ jobs) {
    const RsJob J = jobs[blockIdx.z];
    const int xx = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (y >= J.in_h || xx >= J.out_w) return;
    gciptr bounds = reinterpret_cast<gciptr>(J.bounds);
    gciptr kt = reinterpret_cast<gciptr>(J.coeffs) + xx;
    const int first = bounds[2 * xx], n = bounds[2 * xx + 1];
    gcptr row = reinterpret_cast<gcptr>(J.src) + (size_t)y * J.in_w + first;
    const bool pre = (J.flags & kRsPremultiplyOnLoad) != 0;
    int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < n; ++i) {
        uint32_t p = row[i];
        if (pre) p = premultiply(p);
        const int32_t k = kt[(size_t)i * J.out_w];
        s0 += (int32_t)(p & 255u) * k;
        s1 += (int32_t)((p >> 8) & 255u) * k;
        s2 += (int32_t)((p >> 16) & 255u) * k;
        s3 += (int32_t)(p >> 24) * k;
    }
    uint32_t o = pack_clip(s0, s1, s2, s3);
    if (J.flags & kRsUnpremultiplyOnStore) o = unpremultiply(o);
    reinterpret_cast<gptr>(J.dst)[(size_t)y * J.out_w + xx] = o;
}

// Vertical pass: one thread per output pixel (x, y'); taps walk down a column, lanes are
// adjacent columns (coalesced).  Coefficients [out_h][ksize] are wave-uniform per row.
__global__ __launch_bounds__(256) void resample_v_kernel(const RsJob *__restrict__ jobs) {
    const RsJob J = jobs[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yy = blockIdx.y;
    if (yy >= J.out_h || x >= J.out_w) return;
    gciptr bounds = reinterpret_cast<gciptr>(J.bounds);
    gciptr k = reinterpret_cast<gciptr>(J.coeffs) + (size_t)yy * J.ksize;
    const int first = bounds[2 * yy], n = bounds[2 * yy + 1];
    gcptr col = reinterpret_cast<gcptr>(J.src) + (size_t)first * J.in_w + x;
    const bool pre = (J.flags & kRsPremultiplyOnLoad) != 0;
    int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < n; ++i) {
        uint32_t p = col[(size_t)i * J.in_w];
        if (pre) p = premultiply(p);
        const int32_t c = k[i];
        s0 += (int32_t)(p & 255u) * c;
        s1 += (int32_t)((p >> 8) & 255u) * c;
        s2 += (int32_t)((p >> 16) & 255u) * c;
        s3 += (int32_t)(p >> 24) * c;
    }
    uint32_t o = pack_clip(s0, s1, s2, s3);
    if (J.flags & kRsUnpremultiplyOnStore) o = unpremultiply(o);
    reinterpret_cast<gptr>(J.dst)[(size_t)yy * J.out_w + x] = o;
}

hipError_t launch_resample_h(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_rows,
                             hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_out_w + 255) / 256), (unsigned)max_rows, (unsigned)n_jobs);
    hipLaunchKernelGGL(resample_h_kernel, grid, dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

hipError_t launch_resample_v(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_out_h,
                             hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_out_w + 255) / 256), (unsigned)max_out_h, (unsigned)n_jobs);
    hipLaunchKernelGGL(resample_v_kernel, grid, dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

}  // namespace mic
