// Device helpers shared by the MFMA resample kernels (kernels_resample.hip: marching kernel; kernels_resample_lane.hip:
// wave-private lane kernel): the digit chain on v_mfma_i32_16x16x64_i8, the saturating packs, Convert.c's unpremultiply on
// four pixels at once.  Pillow's arithmetic exactly; the derivations are with each function.
#pragma once
#include "mic_internal.h"

namespace mic {

// The same in float, for the MFMA kernel (v_mul_hi_u32 is a quarter-rate instruction, and this runs
// once per output pixel): floor(c' * F[a]) == floor(255 c' / a) for every c' in 0..255 once clamped to
// 255, with F[a] = 255/a rounded to float and bumped up one ulp -- the product can only exceed the
// exact quotient, by < 2^-14, and the quotient's fractional part is <= 1 - 1/254 (or the value is
// >= 256 and clamps).  Checked exhaustively in tests/test_blend_identities.py with numpy float32.
__device__ __forceinline__ float unpremul_factor(uint32_t a) {
    return __uint_as_float(__float_as_uint(__fdiv_rn(255.0f, (float)a)) + 1u);
}
typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4i *gv4ptr;
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;  // 16-byte access, 4-byte alignment

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

// sat8(v >> 6) of the four sums of an accumulator -> four bytes of one word, byte i from v[i].
// v_ashr_pk_u8_i32 shifts, saturates to 0..255 and packs two values per instruction into D[15:0].  It is
// issued through the compiler's builtin, not inline asm: the values come straight out of an MFMA, and the
// wait states between an MFMA and a VALU read of its result are inserted by the compiler's hazard
// recogniser, which does not look inside asm statements (an asm version read the accumulator early and
// produced saturated garbage).  The builtin returns 16 bits, so the upper half's stale bits -- what
// hipcc's own pattern-matched use of the instruction gets wrong, see clip8 -- are dropped explicitly.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t clip8x4(v4i v) {
    // (two 16-bit halves built as a vector: the compiler then gathers them with ONE v_perm_b32; widening the
    // halves to 32 bits first costs a mask each)
    const u16x2 p = {__builtin_amdgcn_ashr_pk_u8_i32(v[0], v[1], 6), __builtin_amdgcn_ashr_pk_u8_i32(v[2], v[3], 6)};
    return __builtin_bit_cast(uint32_t, p);
}
// The same with SIGNED saturation to -128..127: with 128 << 22 taken off the bias beforehand this is
// clip8(...) - 128, the signed-byte form the next pass' MFMA operand wants (clamp(x, 0, 255) - 128 ==
// clamp(x - 128, -128, 127)), without the xor 0x80 per word.
__device__ __forceinline__ uint32_t clip8x4_signed(v4i v) {
    const u16x2 p = {__builtin_amdgcn_ashr_pk_i8_i32(v[0], v[1], 6), __builtin_amdgcn_ashr_pk_i8_i32(v[2], v[3], 6)};
    return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ v4i shr8(v4i v) { return v4i{v[0] >> 8, v[1] >> 8, v[2] >> 8, v[3] >> 8}; }

// One 16 x 16 tile of all four channels through the digit chain -> per channel one word of clipped bytes.
// load(c) returns the data operand (A) of channel c, f = the tile's tap digits (B).  The four channels' chains
// are written side by side: each MFMA's result is needed three MFMAs later, so the dependent shifts need no
// s_nop padding.  SIGNED: clip to signed bytes (the horizontal pass, see clip8x4_signed).
// DIGITS = 1: an axis that keeps its size (one tap of weight 1.0 = 2^22: the two low digits of every tap are zero, so
// the chain collapses to its last link -- acc = (bias >> 16) + data x digit 2): the pass Pillow skips (Resample.c
// ImagingResample: need_horizontal / need_vertical) costs one MFMA per channel, which is also the transposition the
// operand layouts need, and no floor shifts.
template <bool SIGNED, int DIGITS = 3, class Load>
__device__ __forceinline__ void tile4(Load load, const v4i (&f)[3], v4i bias, uint32_t (&w)[4]) {
    static_assert(DIGITS == 3 || DIGITS == 1, "digit chain: all three digits, or the last one alone");
    v4i a[4], acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = load(c);
    if (DIGITS == 1) {
        const v4i b2 = v4i{bias[0] >> 16, bias[1] >> 16, bias[2] >> 16, bias[3] >> 16};  // ((bias >> 8) >> 8: floor shifts compose)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[c], f[2], b2, 0, 0, 0);
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[c], f[d], d == 0 ? bias : acc[c], 0, 0, 0);
            if (d < 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = shr8(acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) w[c] = SIGNED ? clip8x4_signed(acc[c]) : clip8x4(acc[c]);
}

// Convert.c rgba2rgbA on the four pixels a lane holds after the vertical pass, straight from the
// per-channel words (w[c] = channel c of pixels 0..3): c = min(255, floor(255 c' / a)) for 0 < a < 255,
// the pixel as it is for a = 0 and a = 255.  Per channel one v_cvt_f32_ubyteN (which also picks the
// byte), one fma and one v_cvt_pk_u8_f32 (round to nearest even, saturating, written into byte c of the
// pixel word: profiles/r02_ubench_isa.txt), so the planar -> interleaved transposition costs nothing:
//     RNE(c' * F[a] - 0.5 + 2^-9) == floor(255 c' / a)   whenever that is < 256, and >= 255.5 otherwise,
// with F[a] = 255/a rounded to float and bumped up one ulp, F[0] = F[255] = 1 (then it returns c'):
// the product exceeds the exact quotient by < 2^-13, the quotient's fractional part is a multiple of
// 1/a <= 1 - 1/254, and 2^-9 sits strictly between the two.  Checked exhaustively over (a, c') in
// tests/test_blend_identities.py with float32 arithmetic.
__device__ __forceinline__ u32x4 unpremultiply4(const uint32_t (&w)[4], const float *recip) {
    const float K = -0.5f + 0.001953125f;
    u32x4 px;
#define MIC_UNPREMUL_PX(X)                                                                              \
    {                                                                                                   \
        const uint32_t a = (w[3] >> (8 * X)) & 255u;                                                    \
        const float F = recip[a];                                                                       \
        uint32_t p = a << 24;                                                                           \
        p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)((w[0] >> (8 * X)) & 255u), F, K), 0, p); \
        p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)((w[1] >> (8 * X)) & 255u), F, K), 1, p); \
        p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)((w[2] >> (8 * X)) & 255u), F, K), 2, p); \
        px[X] = p;                                                                                      \
    }
    MIC_UNPREMUL_PX(0) MIC_UNPREMUL_PX(1) MIC_UNPREMUL_PX(2) MIC_UNPREMUL_PX(3)
#undef MIC_UNPREMUL_PX
    return px;
}

// The same four pixels when no lane of the wave holds a partial alpha: a 4 x 4 byte transpose.
__device__ __forceinline__ u32x4 interleave4(const uint32_t (&w)[4]) {
    const uint32_t rg01 = byte_perm(w[1], w[0], 0x05010400u), rg23 = byte_perm(w[1], w[0], 0x07030602u);
    const uint32_t ba01 = byte_perm(w[3], w[2], 0x05010400u), ba23 = byte_perm(w[3], w[2], 0x07030602u);
    return u32x4{byte_perm(ba01, rg01, 0x05040100u), byte_perm(ba01, rg01, 0x07060302u),
                 byte_perm(ba23, rg23, 0x05040100u), byte_perm(ba23, rg23, 0x07060302u)};
}

// uniform 64-bit base + 32-bit per-lane byte offset, written as pointer arithmetic so that the load takes
// the scalar-base addressing form (global_load v, v_off, s[base:base+1]) instead of 64-bit vector address maths
template <class T>
__device__ __forceinline__ const MIC_GLOBAL T *at(uint64_t base, uint32_t byte_off) {
    return reinterpret_cast<const MIC_GLOBAL T *>(reinterpret_cast<const MIC_GLOBAL char *>(base) + byte_off);
}

}  // namespace mic
