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
#include <algorithm>
#include <atomic>

#include "resample_device.h"

namespace mic {

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
        mac4(s0, s1, s2, s3, p, k);
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
        mac4(s0, s1, s2, s3, p, c);
    }
    uint32_t o = pack_clip(s0, s1, s2, s3);
    if (J.flags & kRsUnpremultiplyOnStore) o = unpremultiply(o);
    reinterpret_cast<gptr>(J.dst)[(size_t)yy * J.out_w + x] = o;
}

// ---- resident planar copy of an atlas ------------------------------------------------------------
// Premultiplying and planarising a cutout is a pure function of the cutout, and the atlas stays
// resident across composites / refine iterations / batches: the first resample that touches an atlas
// builds, once, a planar copy of every cutout -- four planes [row][column] of premultiplied samples
// stored as signed bytes (s - 128), row pitch = width rounded up to 16, padding = premultiplied zero --
// and phase 1 of the MFMA kernel becomes a 16-byte copy per lane instead of ~25 instructions per pixel
// (times the 1.7x tile halo).  mic_resize's arbitrary source pointers keep the interleaved loader.
__global__ __launch_bounds__(256) void planarize_kernel(const PlanarJob *__restrict__ jobs) {
    const PlanarJob J = jobs[blockIdx.y];
    const int groups = J.pitch >> 2;  // groups of 4 columns per row (pitch is a multiple of 16)
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= (int64_t)groups * J.h) return;
    const int y = (int)(item / groups), x = 4 * (int)(item - (int64_t)y * groups);
    gcptr src = reinterpret_cast<gcptr>(J.src) + (size_t)y * J.w + x;
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) px[j] = x + j < J.w ? src[j] : 0u;
    uint32_t rb[4], ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = px[j] >> 24;
        rb[j] = premultiply2_hi(px[j] & 0x00FF00FFu, a);
        ga[j] = premultiply2_hi(byte_perm(px[j], px[j], 0x0c0d0c01u), a);
    }
    const uint32_t rb01 = byte_perm(rb[1], rb[0], 0x07030501u), rb23 = byte_perm(rb[3], rb[2], 0x07030501u);
    const uint32_t ga01 = byte_perm(ga[1], ga[0], 0x07030501u), ga23 = byte_perm(ga[3], ga[2], 0x07030501u);
    MIC_GLOBAL uint32_t *dst = reinterpret_cast<MIC_GLOBAL uint32_t *>(J.dst + (size_t)y * J.pitch + x);
    const size_t plane = (size_t)J.h * J.pitch / 4;  // words
    dst[0 * plane] = byte_perm(rb23, rb01, 0x05040100u) ^ 0x80808080u;  // R
    dst[1 * plane] = byte_perm(ga23, ga01, 0x05040100u) ^ 0x80808080u;  // G
    dst[2 * plane] = byte_perm(rb23, rb01, 0x07060302u) ^ 0x80808080u;  // B
    dst[3 * plane] = byte_perm(ga23, ga01, 0x07060302u) ^ 0x80808080u;  // A
}

hipError_t launch_planarize(const PlanarJob *jobs_dev, int n_jobs, int64_t max_items, hipStream_t stream) {
    if (n_jobs <= 0 || max_items <= 0) return hipSuccess;
    for (int first = 0; first < n_jobs; first += 65535) {
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(planarize_kernel, dim3((unsigned)((max_items + 255) / 256), (unsigned)n), dim3(256), 0, stream,
                           jobs_dev + first);
    }
    return hipGetLastError();
}

// Every tile of both axes of a layer that comes here has its taps inside ONE 64-sample window (any scale down to
// ~1/3: the host checks it per layer; deeper shrinks, single images and small calls take the tile kernel of
// kernels_resample_tile.hip) -- no chunk loops, their registers or their branches.
__global__ __launch_bounds__(256, MIC_RS_WAVES) void resample_march_kernel(const RsMarch *__restrict__ jobs) {
    const RsMarch J = jobs[blockIdx.y];
    const int bx = (int)blockIdx.x, tid = threadIdx.x;
#include "resample_march_body.inc"
}

hipError_t launch_resample_march(const RsMarch *jobs_dev, int n_jobs, size_t lds_bytes, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    // opt the kernel in for more than 64 KB of dynamic LDS, once per device of this process
    // (atomic flags: two threads racing here both set the same attribute, which is harmless)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_march_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsMarchMaxLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    for (int first = 0; first < n_jobs; first += 65535) {  // grid.y limit
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(resample_march_kernel, dim3((unsigned)kRsUnitsPerEntry, (unsigned)n), dim3(256), lds_bytes, stream,
                           jobs_dev + first);
    }
    return hipGetLastError();
}

// ---- known-answer canary ------------------------------------------------------------------------------------------
// hipcc 7.2 miscompiled "shift, clamp to 0..255, pack" (it ORs v_ashr_pk_u8_i32's 16-bit result as if the destination's
// upper half were zero); clip8's empty asm and the explicit builtin + 16-bit truncation in clip8x4 work around it.
// Nothing but the parity tests would notice a ROCm update that changes the lowering again, so the library carries a
// tiny kernel that pushes known values through exactly these helpers -- every lane a different mix of negative,
// in-range, saturating and boundary sums -- and mic_selftest compares with plain host arithmetic.
__global__ void selftest_clip_kernel(uint32_t *__restrict__ out) {
    const int l = threadIdx.x;
    // four sums per lane around the interesting boundaries of clip8((bias + sum) >> 22) and sat8(v >> 6)
    const int32_t base[8] = {INT32_MIN, -(1 << 22) - 1, -1, 0, (1 << 22) - 1, 255 << 22, (256 << 22) + 5, INT32_MAX};
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)  // (unsigned arithmetic: the sums wrap by definition)
        v[j] = (int32_t)((uint32_t)base[(l + 3 * j) & 7] +
                         (uint32_t)((int32_t)((uint32_t)l * 0x9E3779B1u >> (j + 3)) * ((l + j) & 1 ? 1 : -1) / 7));
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));  // values the compiler cannot fold
    out[l * 4 + 0] = pack_clip(v[0], v[1], v[2], v[3]);
    const v4i q = {v[0] >> 16, v[1] >> 16, v[2] >> 16, v[3] >> 16};  // the MFMA epilogue's range: sat8(q >> 6)
    out[l * 4 + 1] = clip8x4(q);
    out[l * 4 + 2] = clip8x4_signed(q);
    out[l * 4 + 3] = premultiply((uint32_t)v[0]) ^ unpremultiply((uint32_t)v[1]);
}

static int32_t selftest_value(int l, int j) {
    const int32_t base[8] = {INT32_MIN, -(1 << 22) - 1, -1, 0, (1 << 22) - 1, 255 << 22, (256 << 22) + 5, INT32_MAX};
    return (int32_t)((uint32_t)base[(l + 3 * j) & 7] +
                     (uint32_t)((int32_t)((uint32_t)l * 0x9E3779B1u >> (j + 3)) * ((l + j) & 1 ? 1 : -1) / 7));
}

// Runs the canary on `stream`, waits for it, and returns the number of mismatching words (0 = the helpers compute
// what Pillow's arithmetic says); *first_bad (optional) names the first one.
hipError_t run_selftest_clip(hipStream_t stream, int *mismatches, int *first_bad) {
    uint32_t *dev = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dev), 64 * 4 * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(selftest_clip_kernel, dim3(1), dim3(64), 0, stream, dev);
    uint32_t got[256];
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(got, dev, sizeof got, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return e;
    auto c8 = [](int64_t v) { return (uint32_t)std::min<int64_t>(255, std::max<int64_t>(0, v)); };
    auto c8s = [](int64_t v) { return (uint32_t)(std::min<int64_t>(127, std::max<int64_t>(-128, v)) & 255); };
    auto d255 = [](uint32_t t) { return ((t >> 8) + t) >> 8; };
    auto premul = [&](uint32_t p) {
        const uint32_t a = p >> 24;
        return d255((p & 255u) * a + 128u) | (d255(((p >> 8) & 255u) * a + 128u) << 8) | (d255(((p >> 16) & 255u) * a + 128u) << 16) | (a << 24);
    };
    auto unpremul = [](uint32_t p) {
        const uint32_t a = p >> 24;
        if (a == 0u || a == 255u) return p;
        auto ch = [&](uint32_t c) { return std::min<uint32_t>(255u, 255u * c / a); };
        return ch(p & 255u) | (ch((p >> 8) & 255u) << 8) | (ch((p >> 16) & 255u) << 16) | (a << 24);
    };
    int bad = 0, first = -1;
    for (int l = 0; l < 64; ++l) {
        int32_t v[4];
        for (int j = 0; j < 4; ++j) v[j] = selftest_value(l, j);
        uint32_t want[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            want[0] |= c8((int64_t)v[j] >> 22) << (8 * j);
            want[1] |= c8(((int64_t)v[j] >> 16) >> 6) << (8 * j);
            want[2] |= c8s(((int64_t)v[j] >> 16) >> 6) << (8 * j);
        }
        want[3] = premul((uint32_t)v[0]) ^ unpremul((uint32_t)v[1]);
        for (int k = 0; k < 4; ++k)
            if (got[l * 4 + k] != want[k]) {
                if (first < 0) first = l * 4 + k;
                ++bad;
            }
    }
    *mismatches = bad;
    if (first_bad) *first_bad = first;
    return hipSuccess;
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
