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

// The same in float, for the MFMA kernel (v_mul_hi_u32 is a quarter-rate instruction, and this runs
// once per output pixel): floor(c' * F[a]) == floor(255 c' / a) for every c' in 0..255 once clamped to
// 255, with F[a] = 255/a rounded to float and bumped up one ulp -- the product can only exceed the
// exact quotient, by < 2^-14, and the quotient's fractional part is <= 1 - 1/254 (or the value is
// >= 256 and clamps).  Checked exhaustively in tests/test_blend_identities.py with numpy float32.
__device__ __forceinline__ float unpremul_factor(uint32_t a) {
    return __uint_as_float(__float_as_uint(__fdiv_rn(255.0f, (float)a)) + 1u);
}
__device__ __forceinline__ uint32_t unpremultiply_with(uint32_t p, const float *table) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const float F = table[a];
    const uint32_t r = min(255u, (uint32_t)((float)(p & 255u) * F));
    const uint32_t g = min(255u, (uint32_t)((float)((p >> 8) & 255u) * F));
    const uint32_t b = min(255u, (uint32_t)((float)((p >> 16) & 255u) * F));
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

// Fused resize on the matrix cores: one workgroup produces a (16 tx16) x (16 ty16) tile of the FINAL
// image.
//
// A separable resample is a banded matrix product per axis -- out = in x K^T with K[x][k] the taps
// of output sample x -- and it is exact integer arithmetic, so it maps onto v_mfma_i32_16x16x64_i8
// without touching the result: the 8-bit samples are stored as signed bytes (s - 128, the constant
// 128 * sum(taps) goes into the accumulator's initial value together with Pillow's 2^21 rounding
// term) and each 22-bit tap is split into three signed-byte digits, c = d0 + 256 d1 + 65536 d2,
// one MFMA per digit; acc0 + (acc1 << 8) + (acc2 << 16) is then exactly Pillow's int32 sum.  A
// 16x16x64 MFMA covers 16 output samples and a 64-sample window -- wider than the band for any
// scale down to ~1/3 -- so the zeros outside the band are free.  (The VALU version of this kernel
// spent ~12 instructions per tap per pixel and was bound by integer issue: 0.20 ms for the 32
// layers of the C3 placements workload.)
//
//   1. source window rows x columns -> LDS, premultiplied once per pixel, split into four channel
//      planes [row][column] of signed bytes;
//   2. horizontal pass: A = 16 window rows x 64 columns of one plane (ds_read_b128 per lane),
//      B = the x-tile's tap digits (host-built fragments, resample_coeffs.cpp), D = 16 rows x 16
//      outputs; clip8 -> the 8-bit intermediate Pillow keeps between its passes, written
//      transposed into planes [x][row] so that the next pass again reads 16 consecutive bytes;
//   3. vertical pass: A = the y-tile's tap digits, B = 64 intermediate rows x 16 columns,
//      D = 16 output rows x 16 columns; clip8, interleave the planes, unpremultiply, store.
// The k index of both operands is defined by the same (lane >> 4, byte) -> window position map, so
// the result does not depend on the hardware's internal k order; C/D follow the documented
// col = lane & 15, row = 4 (lane >> 4) + reg map.  Bit-exact with the two-pass kernels above.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4i *gv4ptr;

// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {hi (bytes 4..7), lo (bytes 0..3)}.
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}

// Convert.c rgbA2rgba on two channels at once: x holds two bytes in 16-bit lanes (0x00XX00YY); the
// two results div255(c * a + 128) = (t + (t >> 8)) >> 8 are left in BYTES 1 AND 3 of the returned
// word (bytes 0 and 2 are rounding residue) -- the planarising v_perm picks them from there, which
// saves the final shift+mask.  No lane can carry into the other: c * a + 128 <= 65153 and adding
// (t >> 8) <= 254 stays below 65536.  Three instructions for two channels.
__device__ __forceinline__ uint32_t premultiply2_hi(uint32_t x, uint32_t a) {
    const uint32_t t = __umul24(x, a) + 0x00800080u;            // v_mad_u32_u24
    return t + byte_perm(t, t, 0x0c030c01u);                     // + {t.b1, 0, t.b3, 0}
}

// clip8 of four 32-bit sums -> four bytes of one word, byte i from v[i].  v_ashr_pk_u8_i32 shifts,
// saturates to 0..255 and packs two values per instruction; it writes only D[15:0] (which is what
// hipcc's own use of it gets wrong, see clip8), so the halves are masked/shifted explicitly.
__device__ __forceinline__ uint32_t clip8x4(int v0, int v1, int v2, int v3) {
    uint32_t lo, hi;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "=v"(lo) : "v"(v0), "v"(v1));
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 22" : "=v"(hi) : "v"(v2), "v"(v3));
    return (lo & 0xFFFFu) | (hi << 16);
}

// acc0 + (acc1 << 8) + (acc2 << 16) in Horner form, two v_lshl_add_u32 (hipcc re-associates the C
// expression into two shifts and a three-operand add).
__device__ __forceinline__ int combine(int a0, int a1, int a2) {
    int t;
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(a2), "v"(a1));
    asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(t) : "v"(t), "v"(a0));
    return t;
}

// Phase 1 of the MFMA kernel: R rows x C columns of the source, starting at pixel index `origin`,
// go to LDS premultiplied and split into four planes of signed bytes (s - 128).  Items = (row, group
// of 4 columns), dealt round-robin to the 256 threads, indices advanced incrementally (no per-item
// multiply or divide); four 16-byte loads are in flight per thread and the loads are unconditional
// (index clamped to last4 = pixels - 4; the host keeps images smaller than 4 px off this kernel).
// EDGE: C is not a multiple of 4 (only when the window ends at the image's right edge): the last
// group of each row is re-read pixel by pixel.
template <bool EDGE>
__device__ __forceinline__ uint32_t load_window(gcptr src, int sw, int last4, int origin, int R, int C, uint8_t *srcP,
                                                int pitch_c, int plane_s, int tid) {
    uint32_t seen = 0;  // OR of the pixels this thread handled: bits 24-31 say whether any had alpha > 0
    const int G = (C + 3) >> 2;
    const int dq = 256 / G, dr = 256 - dq * G;
    int rr = tid / G, g = tid - rr * G;
    int gi = origin + rr * sw + 4 * g;   // pixel index in the source image
    int lo = rr * pitch_c + 4 * g;       // byte offset in a plane
    const int gi_step = dq * sw + 4 * dr, lo_step = dq * pitch_c + 4 * dr;
    const int gi_wrap = sw - 4 * G, lo_wrap = pitch_c - 4 * G;
    while (rr < R) {
        int irr[4], ig[4], igi[4], ilo[4];
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            irr[k] = rr; ig[k] = g; igi[k] = gi; ilo[k] = lo;
            __builtin_memcpy(&v[k], (const void *)(src + min(gi, last4)), 16);
            rr += dq; g += dr; gi += gi_step; lo += lo_step;
            if (g >= G) { g -= G; ++rr; gi += gi_wrap; lo += lo_wrap; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (irr[k] >= R) break;
            uint32_t px[4] = {v[k][0], v[k][1], v[k][2], v[k][3]};
            if (EDGE && ig[k] == G - 1) {
                const int left = C - 4 * ig[k];  // 1..3 valid pixels
                px[0] = src[igi[k]];
                px[1] = left > 1 ? src[igi[k] + 1] : 0u;
                px[2] = left > 2 ? src[igi[k] + 2] : 0u;
                px[3] = 0u;
            }
            seen |= px[0] | px[1] | px[2] | px[3];
            uint32_t rb[4], ga[4];  // premultiplied {R, B} and {G, A} in bytes 1 and 3
            // Cutouts are mostly binary-alpha (the reference's bundles have no partial alpha at all):
            // when every pixel this wave holds has alpha 0 or 255, premultiplying is a select.
            bool binary = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) binary = binary && ((px[j] >> 24) == 0u || (px[j] >> 24) == 255u);
            if (__all(binary)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t keep = (uint32_t)((int32_t)px[j] >> 31);  // alpha 255 -> all ones, 0 -> zero
                    const uint32_t q = px[j] & keep;
                    rb[j] = (q & 0x00FF00FFu) << 8;                           // bytes 1 and 3, like the general path
                    ga[j] = q & 0xFF00FF00u;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a = px[j] >> 24;
                    rb[j] = premultiply2_hi(px[j] & 0x00FF00FFu, a);
                    // {G, 255}: div255(255 a + 128) == a keeps the alpha byte itself
                    ga[j] = premultiply2_hi(byte_perm(px[j], px[j], 0x0c0d0c01u), a);
                }
            }
            // 4 px x 2 lanes -> one word per plane: {c0, c1, c2, c3} of the four pixels
            const uint32_t rb01 = byte_perm(rb[1], rb[0], 0x07030501u), rb23 = byte_perm(rb[3], rb[2], 0x07030501u);
            const uint32_t ga01 = byte_perm(ga[1], ga[0], 0x07030501u), ga23 = byte_perm(ga[3], ga[2], 0x07030501u);
            uint32_t *dst = reinterpret_cast<uint32_t *>(srcP + ilo[k]);
            dst[0 * (plane_s >> 2)] = byte_perm(rb23, rb01, 0x05040100u) ^ 0x80808080u;  // R
            dst[1 * (plane_s >> 2)] = byte_perm(ga23, ga01, 0x05040100u) ^ 0x80808080u;  // G
            dst[2 * (plane_s >> 2)] = byte_perm(rb23, rb01, 0x07060302u) ^ 0x80808080u;  // B
            dst[3 * (plane_s >> 2)] = byte_perm(ga23, ga01, 0x07060302u) ^ 0x80808080u;  // A
        }
    }
    return seen;
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

// Phase 1 from the planar copy: rows [r0, r0 + R) x 16-byte chunks [c_lo, c_lo + 16 G) of each plane go
// to LDS as they are (an item = one chunk position, its four planes loaded back to back).  Returns
// non-zero iff some pixel of the window has alpha > 0 (alpha bytes are stored as alpha ^ 0x80).
__device__ __forceinline__ uint32_t load_window_planar(uint64_t planar, int pitch, size_t plane_bytes, int r0, int c_lo,
                                                       int R, int G, uint8_t *srcP, int pitch_c, int plane_s, int tid) {
    uint32_t seen = 0;
    const int dq = 256 / G, dr = 256 - dq * G;
    int rr = tid / G, g = tid - rr * G;
    while (rr < R) {
        const uint64_t gsrc = planar + (size_t)(r0 + rr) * pitch + c_lo + 16 * g;
        v4i v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<gv4ptr>(gsrc + c * plane_bytes);
        uint8_t *dst = srcP + rr * pitch_c + 16 * g;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<v4i *>(dst + c * plane_s) = v[c];
        const int k = (int)0x80808080u;
        seen |= (uint32_t)((v[3][0] ^ k) | (v[3][1] ^ k) | (v[3][2] ^ k) | (v[3][3] ^ k));
        rr += dq;
        g += dr;
        if (g >= G) { g -= G; ++rr; }
    }
    return seen;
}

// One 16 x 16 output tile: acc[channel][digit] = bias + sum over the window's 64-sample chunks of
// data x tap-digit fragments.  DATA_IS_A: the LDS bytes are the A operand (horizontal pass: rows of
// a source plane), otherwise B (vertical pass: columns of an intermediate plane).  f = the first
// chunk's fragments (kept in registers by the caller), fbase = where the tile's fragments start.
template <bool DATA_IS_A>
__device__ __forceinline__ void tile_mfma(v4i (&acc)[4][3], const uint8_t *data, int plane, const v4i (&f)[3],
                                          gv4ptr fbase, int n_chunks, v4i bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const v4i d = *reinterpret_cast<const v4i *>(data + c * plane);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const v4i init = k == 0 ? bias : v4i{0, 0, 0, 0};
            acc[c][k] = DATA_IS_A ? __builtin_amdgcn_mfma_i32_16x16x64_i8(d, f[k], init, 0, 0, 0)
                                  : __builtin_amdgcn_mfma_i32_16x16x64_i8(f[k], d, init, 0, 0, 0);
        }
    }
    for (int ch = 1; ch < n_chunks; ++ch) {  // windows wider than 64 samples (shrinks below ~1/3)
        const v4i e[3] = {fbase[(ch * 3 + 0) * 64], fbase[(ch * 3 + 1) * 64], fbase[(ch * 3 + 2) * 64]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4i d = *reinterpret_cast<const v4i *>(data + 64 * ch + c * plane);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                acc[c][k] = DATA_IS_A ? __builtin_amdgcn_mfma_i32_16x16x64_i8(d, e[k], acc[c][k], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_i32_16x16x64_i8(e[k], d, acc[c][k], 0, 0, 0);
        }
    }
}

// Accumulators -> per channel one word holding the clipped bytes of the lane's 4 rows.
__device__ __forceinline__ void tile_words(const v4i (&acc)[4][3], uint32_t (&w)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
        w[c] = clip8x4(combine(acc[c][0][0], acc[c][1][0], acc[c][2][0]), combine(acc[c][0][1], acc[c][1][1], acc[c][2][1]),
                       combine(acc[c][0][2], acc[c][1][2], acc[c][2][2]), combine(acc[c][0][3], acc[c][1][3], acc[c][2][3]));
}

// 4 channels x 4 rows -> 4 RGBA pixels (byte transpose), unpremultiply, store column ox of rows oy..oy+3.
__device__ __forceinline__ void store_pixels(const uint32_t (&w)[4], gptr dst, int dw, int dh, int ox, int oy,
                                             const float *recip) {
    const uint32_t rg01 = byte_perm(w[1], w[0], 0x05010400u), rg23 = byte_perm(w[1], w[0], 0x07030602u);
    const uint32_t ba01 = byte_perm(w[3], w[2], 0x05010400u), ba23 = byte_perm(w[3], w[2], 0x07030602u);
    const uint32_t px[4] = {byte_perm(ba01, rg01, 0x05040100u), byte_perm(ba01, rg01, 0x07060302u),
                            byte_perm(ba23, rg23, 0x05040100u), byte_perm(ba23, rg23, 0x07060302u)};
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (ox < dw && oy + r < dh) dst[(uint32_t)((oy + r) * dw + ox)] = unpremultiply_with(px[r], recip);  // < 2^31 px
}

template <bool BANDED>
__global__ __launch_bounds__(256) void resample_mfma_kernel(const RsMfma *__restrict__ jobs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
    const RsMfma J = jobs[blockIdx.y];
    // XCD-aware tile order (see RsMfma): blockIdx.x & 7 is the XCD this workgroup lands on
    const int tile = (((int)blockIdx.x + J.xcd_rot) & 7) * (4 * J.n_entries) + 4 * J.entry + ((int)blockIdx.x >> 3);
    if (tile >= J.tiles_x * J.tiles_y) return;
    const int tyi = tile / J.tiles_x, txi = tile - tyi * J.tiles_x;
    const int xt0 = txi * J.tx16, yt0 = tyi * J.ty16;
    const int n_xt = min(J.tx16, ((J.dw + 15) >> 4) - xt0), n_yt = min(J.ty16, ((J.dh + 15) >> 4) - yt0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lh = lane >> 4;

    gv4ptr hmeta = reinterpret_cast<gv4ptr>(J.hmeta), vmeta = reinterpret_cast<gv4ptr>(J.vmeta);
    const int c_lo = hmeta[xt0][0], c_hi = min(J.sw, hmeta[xt0 + n_xt - 1][3]);
    const int r_lo = vmeta[yt0][0], r_hi = min(J.sh, vmeta[yt0 + n_yt - 1][3]);
    const int R = r_hi - r_lo;
    // columns are loaded in groups of 4: round the window up to that, inside the image
    const int C = min((c_hi - c_lo + 3) & ~3, J.sw - c_lo);
    const int plane_s = J.rows16 * J.pitch_c;       // bytes per source plane
    const int plane_m = 16 * J.tx16 * J.pitch_r;    // bytes per intermediate plane
    uint8_t *srcP = lds8;                           // [4][rows16][pitch_c]
    uint8_t *midT = lds8 + 4 * plane_s;             // [4][16 tx16][pitch_r]
    __shared__ float recip[256];                    // unpremultiply factors 255/a: an LDS read per pixel
    recip[tid] = unpremul_factor((uint32_t)tid);

    // ---- 1 + 2, per band of J.rows16 window rows (all of them at once unless the window is too tall
    // for LDS -- deep shrinks -- in which case the source planes hold one band at a time and only the
    // 8-bit intermediate covers the whole window).
    // (Two instantiations: the banded loop keeps the loader's and the pass's registers alive together
    // -- 192 VGPRs, two waves per SIMD -- which the common whole-window case must not pay for.)
    int band0 = 0;
    do {
        const int Rb = BANDED ? min(J.rows16, R - band0) : R;
        if (BANDED && band0 > 0) __syncthreads();  // the previous band's horizontal pass is done reading srcP

        // ---- 1. source rows -> premultiplied signed-byte planes
        uint32_t seen;
        if (J.planar_pitch > 0)  // the atlas' resident planar copy: a straight 16-byte copy per lane
            seen = load_window_planar(J.src, J.planar_pitch, (size_t)J.planar_pitch * J.sh, r_lo + band0, c_lo, Rb,
                                      (c_hi - c_lo + 15) >> 4, srcP, J.pitch_c, plane_s, tid) ? 0xFF000000u : 0u;
        else if ((C & 3) == 0)
            seen = load_window<false>(reinterpret_cast<gcptr>(J.src), J.sw, J.sw * J.sh - 4, (r_lo + band0) * J.sw + c_lo,
                                      Rb, C, srcP, J.pitch_c, plane_s, tid);
        else  // the window ends at the image's right edge in the middle of a group of 4 columns
            seen = load_window<true>(reinterpret_cast<gcptr>(J.src), J.sw, J.sw * J.sh - 4, (r_lo + band0) * J.sw + c_lo,
                                     Rb, C, srcP, J.pitch_c, plane_s, tid);
        if (!BANDED) {
            // A window without a single pixel of alpha > 0 (the corners around a cutout's shape)
            // premultiplies to all zeros, and both passes of zeros give clip8(2^21 >> 22) = 0: the tile is
            // transparent black.
            if (!__syncthreads_or((seen >> 24) != 0u)) {
                gptr dst = reinterpret_cast<gptr>(J.dst);
                const int ox0 = xt0 * 16, oy0 = yt0 * 16;
                const int tw = min(16 * n_xt, J.dw - ox0), th = min(16 * n_yt, J.dh - oy0);
                for (int yy = wave; yy < th; yy += 4)
                    for (int xx = lane; xx < tw; xx += 64) dst[(uint32_t)((oy0 + yy) * J.dw + ox0 + xx)] = 0u;
                return;
            }
        } else {
            __syncthreads();
        }

        // ---- 2. horizontal pass: window rows -> 8-bit intermediate, transposed.  A wave keeps one
        // x-tile (its tap fragments stay in registers) and walks the row tiles two at a time: the second
        // tile's MFMAs run in the matrix pipe while the VALU does the first tile's epilogue.
        const int groups = 4 / n_xt;  // waves per x-tile (n_xt <= 4)
        if (wave < n_xt * groups) {
            const int xi = wave % n_xt, sub = wave / n_xt;
            const int n_rt = (Rb + 15) >> 4;
            const v4i m = hmeta[xt0 + xi];
            const int b = reinterpret_cast<gciptr>(J.hbias)[(xt0 + xi) * 16 + l15];
            const v4i bias = {b, b, b, b};
            gv4ptr fbase = reinterpret_cast<gv4ptr>(J.hfrag) + (size_t)m[2] * 3 * 64 + lane;
            const v4i f[3] = {fbase[0], fbase[64], fbase[128]};
            const uint8_t *a0 = srcP + l15 * J.pitch_c + (m[0] - c_lo) + 16 * lh;      // + 16 rt pitch_c
            uint8_t *m0 = midT + (xi * 16 + l15) * J.pitch_r + band0 + 4 * lh;         // + 16 rt
            for (int rt = sub; rt < n_rt; rt += 2 * groups) {
                const int rt2 = rt + groups;
                const bool two = rt2 < n_rt;  // wave-uniform
                v4i acc[4][3], acc2[4][3];
                tile_mfma<true>(acc, a0 + rt * 16 * J.pitch_c, plane_s, f, fbase, m[1], bias);
                if (two) tile_mfma<true>(acc2, a0 + rt2 * 16 * J.pitch_c, plane_s, f, fbase, m[1], bias);
                // D[row = 4 lh + reg (window row)][col = l15 (x)]: 4 consecutive rows of one column
                uint32_t w[4];
                tile_words(acc, w);
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<uint32_t *>(m0 + rt * 16 + c * plane_m) = w[c] ^ 0x80808080u;
                if (two) {
                    tile_words(acc2, w);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<uint32_t *>(m0 + rt2 * 16 + c * plane_m) = w[c] ^ 0x80808080u;
                }
            }
        }
        band0 += J.rows16;
    } while (BANDED && band0 < R);
    __syncthreads();

    // ---- 3. vertical pass + unpremultiply + store: a wave keeps one y-tile, walks the x-tiles
    {
        const int groups = 4 / n_yt;
        if (wave < n_yt * groups) {
            const int yi = wave % n_yt, sub = wave / n_yt;
            const v4i m = vmeta[yt0 + yi];
            const v4i bias = *reinterpret_cast<gv4ptr>(reinterpret_cast<gciptr>(J.vbias) + (yt0 + yi) * 16 + 4 * lh);
            gv4ptr fbase = reinterpret_cast<gv4ptr>(J.vfrag) + (size_t)m[2] * 3 * 64 + lane;
            const v4i f[3] = {fbase[0], fbase[64], fbase[128]};
            const uint8_t *b0 = midT + l15 * J.pitch_r + (m[0] - r_lo) + 16 * lh;      // + 16 xi pitch_r
            gptr dst = reinterpret_cast<gptr>(J.dst);
            const int oy0 = (yt0 + yi) * 16 + 4 * lh;
            for (int xi = sub; xi < n_xt; xi += 2 * groups) {
                const int xi2 = xi + groups;
                const bool two = xi2 < n_xt;
                v4i acc[4][3], acc2[4][3];
                tile_mfma<false>(acc, b0 + xi * 16 * J.pitch_r, plane_m, f, fbase, m[1], bias);
                if (two) tile_mfma<false>(acc2, b0 + xi2 * 16 * J.pitch_r, plane_m, f, fbase, m[1], bias);
                // D[row = 4 lh + reg (output row)][col = l15 (x)]: per channel the bytes of 4 rows
                uint32_t w[4];
                tile_words(acc, w);
                store_pixels(w, dst, J.dw, J.dh, (xt0 + xi) * 16 + l15, oy0, recip);
                if (two) {
                    tile_words(acc2, w);
                    store_pixels(w, dst, J.dw, J.dh, (xt0 + xi2) * 16 + l15, oy0, recip);
                }
            }
        }
    }
}

// jobs_dev[0, n_whole) keep their whole source window in LDS, jobs_dev[n_whole, n_jobs) are banded.
hipError_t launch_resample_mfma(const RsMfma *jobs_dev, int n_jobs, int n_whole, size_t lds_bytes,
                                hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    // opt the kernels in for more than 64 KB of dynamic LDS, once per device of this process
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_mfma_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsMfmaMaxLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_mfma_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsMfmaMaxLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    for (int first = 0; first < n_whole; first += 65535) {  // grid.y limit
        const int n = std::min(65535, n_whole - first);
        hipLaunchKernelGGL(resample_mfma_kernel<false>, dim3((unsigned)kRsTilesPerEntry, (unsigned)n), dim3(256),
                           lds_bytes, stream, jobs_dev + first);
    }
    for (int first = n_whole; first < n_jobs; first += 65535) {
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(resample_mfma_kernel<true>, dim3((unsigned)kRsTilesPerEntry, (unsigned)n), dim3(256),
                           lds_bytes, stream, jobs_dev + first);
    }
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
