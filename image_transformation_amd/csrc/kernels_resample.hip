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

// Convert.c rgba2rgbA: c = min(255, 255*c' / a) for 0 < a < 255.  The integer division is replaced
// by one v_mul_hi_u32 with R[a] = ceil(255 * 2^24 / a): floor(c' * R[a] / 2^24) == floor(255*c'/a)
// exactly for c' in 0..255 (the excess is < 2^-16, the quotient's fractional part is a multiple
// of 1/a <= 1 - 1/254); checked exhaustively in tests/test_blend_identities.py.
struct UnpremulTable {
    uint32_t r[256];
    constexpr UnpremulTable() : r{} {
        for (uint32_t a = 1; a < 256; ++a) r[a] = (uint32_t)((((uint64_t)255 << 24) + a - 1) / a);
    }
};
__device__ __constant__ UnpremulTable kUnpremul{};

__device__ __forceinline__ uint32_t unpremultiply(uint32_t p) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const uint32_t R = kUnpremul.r[a];
    const uint32_t r = min(255u, __umulhi((p & 255u) << 8, R));
    const uint32_t g = min(255u, __umulhi(p & 0xFF00u, R));
    const uint32_t b = min(255u, __umulhi((p >> 8) & 0xFF00u, R));
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

// acc += channel * tap for the four channels of one pixel.  Both factors fit 24 bits (bytes; taps
// are 22-bit fixed point, |k| < 2^23), so v_mad_i32_i24 is exact -- and is what must be asked for:
// a plain int32 multiply-add made hipcc emit 64-bit v_mad_u64_u32, several times slower.
__device__ __forceinline__ void mac4(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, uint32_t p, int32_t k) {
    s0 += __mul24((int)(p & 255u), k);
    s1 += __mul24((int)((p >> 8) & 255u), k);
    s2 += __mul24((int)((p >> 16) & 255u), k);
    s3 += __mul24((int)(p >> 24), k);
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

// Fused resize: one workgroup produces a tx x ty tile of the FINAL image.
//   1. the tile's slices of both coefficient tables go to LDS (horizontal taps transposed to
//      [k][x] so that lanes = adjacent columns read adjacent words);
//   2. the source window the tile depends on (its rows' vertical taps x its columns' horizontal
//      taps) is read from HBM once, premultiplied once per pixel, and kept in LDS;
//   3. horizontal pass LDS -> LDS: the 8-bit intermediate Pillow keeps between its two passes
//      (ImagingResampleHorizontal_8bpc output) for exactly the window rows;
//   4. vertical pass LDS -> registers, unpremultiply, one coalesced store per output row segment.
// Arithmetic is identical to the two-pass kernels above (and to Pillow): bit-exact.  The host picks
// tx/ty per layer so that the window fits LDS (mic_api.hip: choose_fused); layers shrunk so hard
// that even a 16x1 tile's window does not fit fall back to the two-pass kernels through HBM.
__global__ __launch_bounds__(256) void resample_fused_kernel(const RsFused *__restrict__ jobs) {
    extern __shared__ uint32_t lds[];
    const RsFused J = jobs[blockIdx.y];
    const int tile = blockIdx.x;
    if (tile >= J.tiles_x * J.tiles_y) return;
    const int tyi = tile / J.tiles_x, txi = tile - tyi * J.tiles_x;
    const int ox0 = txi * J.tx, oy0 = tyi * J.ty;
    const int tw = min(J.tx, J.dw - ox0), th = min(J.ty, J.dh - oy0);
    const bool need_h = J.kx > 0, need_v = J.ky > 0;
    const int tid = threadIdx.x, lane = tid & 63, wy = tid >> 6;

    uint32_t *srcT = lds;                                       // [max_r][max_c]
    uint32_t *mid = srcT + (size_t)J.max_r * J.max_c;           // [max_r][tx]
    int32_t *cH = reinterpret_cast<int32_t *>(mid + (size_t)J.max_r * J.tx);  // [kx][tx]
    int32_t *cV = cH + J.kx * J.tx;                             // [ty][ky]
    int32_t *bH = cV + J.ty * J.ky;                             // [tx][2]
    int32_t *bV = bH + 2 * J.tx;                                // [ty][2]

    if (need_h) {
        gciptr hb = reinterpret_cast<gciptr>(J.hbounds) + 2 * ox0;
        gciptr hc = reinterpret_cast<gciptr>(J.hcoeffs) + (size_t)ox0 * J.kx;
        for (int i = tid; i < 2 * tw; i += 256) bH[i] = hb[i];
        for (int i = tid; i < tw * J.kx; i += 256) {
            const int x = i / J.kx, k = i - x * J.kx;
            cH[k * J.tx + x] = hc[i];
        }
    }
    if (need_v) {
        gciptr vb = reinterpret_cast<gciptr>(J.vbounds) + 2 * oy0;
        gciptr vc = reinterpret_cast<gciptr>(J.vcoeffs) + (size_t)oy0 * J.ky;
        for (int i = tid; i < 2 * th; i += 256) bV[i] = vb[i];
        for (int i = tid; i < th * J.ky; i += 256) cV[i] = vc[i];
    }
    __syncthreads();

    // source window [r0, r1) x [c0, c1)
    const int c0 = need_h ? bH[0] : ox0;
    const int c1 = need_h ? bH[2 * (tw - 1)] + bH[2 * (tw - 1) + 1] : ox0 + tw;
    const int r0 = need_v ? bV[0] : oy0;
    const int r1 = need_v ? bV[2 * (th - 1)] + bV[2 * (th - 1) + 1] : oy0 + th;
    const int C = c1 - c0, R = r1 - r0;
    gcptr src = reinterpret_cast<gcptr>(J.src);
    for (int rr = wy; rr < R; rr += 4) {
        gcptr row = src + (size_t)(r0 + rr) * J.sw + c0;
        for (int cc = lane; cc < C; cc += 64) srcT[rr * J.max_c + cc] = premultiply(row[cc]);
    }
    __syncthreads();

    // horizontal pass: window rows -> 8-bit intermediate
    for (int rr = wy; rr < R; rr += 4) {
        const uint32_t *srow = srcT + rr * J.max_c;
        for (int xx = lane; xx < tw; xx += 64) {
            uint32_t o;
            if (need_h) {
                const int first = bH[2 * xx] - c0, n = bH[2 * xx + 1];
                int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
                for (int k = 0; k < n; ++k) {
                    const uint32_t p = srow[first + k];
                    const int32_t c = cH[k * J.tx + xx];
                    mac4(s0, s1, s2, s3, p, c);
                }
                o = pack_clip(s0, s1, s2, s3);
            } else {
                o = srow[xx];
            }
            mid[rr * J.tx + xx] = o;
        }
    }
    __syncthreads();

    // vertical pass + unpremultiply + store
    gptr dst = reinterpret_cast<gptr>(J.dst);
    for (int yy = wy; yy < th; yy += 4) {
        for (int xx = lane; xx < tw; xx += 64) {
            uint32_t o;
            if (need_v) {
                const int first = bV[2 * yy] - r0, n = bV[2 * yy + 1];
                const int32_t *kv = cV + yy * J.ky;
                int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
                for (int k = 0; k < n; ++k) {
                    const uint32_t p = mid[(first + k) * J.tx + xx];
                    const int32_t c = kv[k];
                    mac4(s0, s1, s2, s3, p, c);
                }
                o = pack_clip(s0, s1, s2, s3);
            } else {
                o = mid[yy * J.tx + xx];
            }
            dst[(size_t)(oy0 + yy) * J.dw + ox0 + xx] = unpremultiply(o);
        }
    }
}

hipError_t launch_resample_fused(const RsFused *jobs_dev, int n_jobs, int max_tiles, size_t lds_bytes,
                                 hipStream_t stream) {
    if (n_jobs <= 0 || max_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(resample_fused_kernel, dim3((unsigned)max_tiles, (unsigned)n_jobs), dim3(256), lds_bytes, stream,
                       jobs_dev);
    return hipGetLastError();
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
