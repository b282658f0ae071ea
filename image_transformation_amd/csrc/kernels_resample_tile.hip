// Fused resize on the matrix cores, one workgroup per tile of the FINAL image (round 1's kernel, kept for what
// the marching kernel of kernels_resample.hip is bad at): single images handed to mic_resize (no planar copy to
// amortise: this kernel premultiplies and planarises the window while loading it), deep shrinks (a handful of
// output tiles fed by hundreds of source rows: the BANDED instantiation keeps one band of source rows in LDS at a
// time and every tile is its own workgroup) and calls too small to fill the chip with marching units.
// Pillow-exact like everything else here (Image.resize(size, LANCZOS), compositor.py:20; thumbnails
// macro_placement_test.py:194); see kernels_resample.hip for the arithmetic.
#include <algorithm>
#include <atomic>

#include "mic_internal.h"

namespace mic {
namespace {

__device__ __forceinline__ float unpremul_factor(uint32_t a) {
    return __uint_as_float(__float_as_uint(__fdiv_rn(255.0f, (float)a)) + 1u);
}
// floor(c' * F[a]) == floor(255 c' / a) once clamped to 255 (tests/test_blend_identities.py)
__device__ __forceinline__ uint32_t unpremultiply_with(uint32_t p, const float *table) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const float F = table[a];
    const uint32_t r = min(255u, (uint32_t)((float)(p & 255u) * F));
    const uint32_t g = min(255u, (uint32_t)((float)((p >> 8) & 255u) * F));
    const uint32_t b = min(255u, (uint32_t)((float)((p >> 16) & 255u) * F));
    return r | (g << 8) | (b << 16) | (a << 24);
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

// clip8 of four 32-bit sums -> four bytes of one word, byte i from v[i]: v_ashr_pk_u8_i32 shifts, saturates to
// 0..255 and packs two values per instruction.  Through the compiler's builtin: these values come straight out of
// MFMAs, and the wait states between an MFMA and a VALU read of its result are inserted by the hazard recogniser,
// which does not look inside asm statements.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t clip8x4(int v0, int v1, int v2, int v3) {
    const u16x2 p = {__builtin_amdgcn_ashr_pk_u8_i32(v0, v1, 22), __builtin_amdgcn_ashr_pk_u8_i32(v2, v3, 22)};
    return __builtin_bit_cast(uint32_t, p);
}

// acc0 + (acc1 << 8) + (acc2 << 16) (two v_lshl_add_u32)
__device__ __forceinline__ int combine(int a0, int a1, int a2) {
    return (int)((((((uint32_t)a2) << 8) + (uint32_t)a1) << 8) + (uint32_t)a0);
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

// What a wave needs of one axis' tables for the tile it keeps: the tile's table entry, the accumulators' initial
// values, where its fragments start and the first chunk's three digit fragments.
struct AxisOperands {
    v4i m, bias;
    gv4ptr fbase;
    v4i f[3];
};
// horizontal: the bias is per output column (the C/D column of a lane)
__device__ __forceinline__ AxisOperands h_operands(const RsTile &J, int xt, int lane) {
    AxisOperands o;
    o.m = reinterpret_cast<gv4ptr>(J.hmeta)[xt];
    const int b = reinterpret_cast<gciptr>(J.hbias)[xt * 16 + (lane & 15)];
    o.bias = v4i{b, b, b, b};
    o.fbase = reinterpret_cast<gv4ptr>(J.hfrag) + (size_t)o.m[2] * 3 * 64 + lane;
    o.f[0] = o.fbase[0]; o.f[1] = o.fbase[64]; o.f[2] = o.fbase[128];
    return o;
}
// vertical: per output row (the lane's four C/D rows)
__device__ __forceinline__ AxisOperands v_operands(const RsTile &J, int yt, int lane) {
    AxisOperands o;
    o.m = reinterpret_cast<gv4ptr>(J.vmeta)[yt];
    o.bias = *reinterpret_cast<gv4ptr>(reinterpret_cast<gciptr>(J.vbias) + yt * 16 + 4 * (lane >> 4));
    o.fbase = reinterpret_cast<gv4ptr>(J.vfrag) + (size_t)o.m[2] * 3 * 64 + lane;
    o.f[0] = o.fbase[0]; o.f[1] = o.fbase[64]; o.f[2] = o.fbase[128];
    return o;
}

template <bool BANDED>
__device__ __forceinline__ void resample_tile(const RsTile &J) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
    // XCD-aware tile order (see RsTile): blockIdx.x & 7 is the XCD this workgroup lands on
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
        // (Requesting this tile's and the vertical pass' tap fragments up front, together with the window, instead of
        // behind the barriers was built and measured in round 5: 24 more registers, and the small calls it was meant
        // for did not move -- 8.28 us against 8.28 for the reference-sized call, 7.40 against 7.44 for thumbnails.
        // profiles/r05_small_calls.txt, r05_tile_eager.patch.)
        const int groups = 4 / n_xt;  // waves per x-tile (n_xt <= 4)
        if (wave < n_xt * groups) {
            const int xi = wave % n_xt, sub = wave / n_xt;
            const int n_rt = (Rb + 15) >> 4;
            const AxisOperands H = h_operands(J, xt0 + xi, lane);
            const v4i m = H.m, bias = H.bias;
            gv4ptr fbase = H.fbase;
            const v4i f[3] = {H.f[0], H.f[1], H.f[2]};
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
            const AxisOperands V = v_operands(J, yt0 + yi, lane);
            const v4i m = V.m, bias = V.bias;
            gv4ptr fbase = V.fbase;
            const v4i f[3] = {V.f[0], V.f[1], V.f[2]};
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

template <bool BANDED>
__global__ __launch_bounds__(256) void resample_tile_kernel(const RsTile *__restrict__ jobs) {
    const RsTile J = jobs[blockIdx.y];
    resample_tile<BANDED>(J);
}
// A handful of whole-window entries, handed over BY VALUE in the kernel arguments (like the composite kernel's single
// job): the reference's own call -- three or four cutouts resized onto one small canvas, compositor.py:18-21 -- then
// stages and uploads nothing at all, and a workgroup's entry comes through scalar loads from the argument segment.
__global__ __launch_bounds__(256) void resample_tile_args_kernel(const RsTileArgs args) {
    const RsTile J = args.t[blockIdx.y];
    resample_tile<false>(J);
}

}  // namespace

// jobs_dev[0, n_whole) keep their whole source window in LDS, jobs_dev[n_whole, n_jobs) are banded.
// jobs_host: the same entries on the host; when they all fit the argument block (rs_tile_in_args) the launch carries them
// and jobs_dev is not read.
hipError_t launch_resample_tile(const RsTile *jobs_dev, int n_jobs, int n_whole, size_t lds_bytes,
                                hipStream_t stream, const RsTile *jobs_host) {
    if (n_jobs <= 0) return hipSuccess;
    // opt the kernels in for more than 64 KB of dynamic LDS, once per device of this process
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        const void *forms[3] = {reinterpret_cast<const void *>(resample_tile_kernel<false>),
                                reinterpret_cast<const void *>(resample_tile_kernel<true>),
                                reinterpret_cast<const void *>(resample_tile_args_kernel)};
        for (const void *f : forms) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsTileMaxLds);
            if (e != hipSuccess) return e;
        }
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    if (jobs_host && rs_tile_in_args(n_jobs, n_whole)) {
        RsTileArgs args{};
        std::copy(jobs_host, jobs_host + n_jobs, args.t);
        hipLaunchKernelGGL(resample_tile_args_kernel, dim3((unsigned)kRsTilesPerEntry, (unsigned)n_jobs), dim3(256), lds_bytes,
                           stream, args);
        return hipGetLastError();
    }
    for (int first = 0; first < n_whole; first += 65535) {  // grid.y limit
        const int n = std::min(65535, n_whole - first);
        hipLaunchKernelGGL(resample_tile_kernel<false>, dim3((unsigned)kRsTilesPerEntry, (unsigned)n), dim3(256),
                           lds_bytes, stream, jobs_dev + first);
    }
    for (int first = n_whole; first < n_jobs; first += 65535) {
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(resample_tile_kernel<true>, dim3((unsigned)kRsTilesPerEntry, (unsigned)n), dim3(256),
                           lds_bytes, stream, jobs_dev + first);
    }
    return hipGetLastError();
}

}  // namespace mic
