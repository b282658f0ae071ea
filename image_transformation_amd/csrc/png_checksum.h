// CRC-32 (slicing-by-8) and Adler-32 (SSE2) shared by libmic's PNG writer and reader (png_encode.cpp, png_decode.cpp).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mic {

inline uint32_t g_crc[8][256];
inline std::once_flag g_crc_once;

inline void crc_init() {
    std::call_once(g_crc_once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            g_crc[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) g_crc[s][i] = g_crc[0][g_crc[s - 1][i] & 255] ^ (g_crc[s - 1][i] >> 8);
    });
}

inline uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {  // slicing-by-8; crc is the running (inverted) state
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        crc = g_crc[0][(crc ^ *p++) & 255] ^ (crc >> 8);
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= crc;
        crc = g_crc[7][v & 255] ^ g_crc[6][(v >> 8) & 255] ^ g_crc[5][(v >> 16) & 255] ^ g_crc[4][(v >> 24) & 255] ^
              g_crc[3][(v >> 32) & 255] ^ g_crc[2][(v >> 40) & 255] ^ g_crc[1][(v >> 48) & 255] ^ g_crc[0][v >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) crc = g_crc[0][(crc ^ *p++) & 255] ^ (crc >> 8);
    return crc;
}

constexpr uint32_t kAdlerMod = 65521;

// Adler-32 of n bytes continuing from (a, b).  After k more bytes p[0..k): a' = a + S, b' = b + k a + sum (k - i) p[i];
// in chunks of 16 that weighted sum is 16 * (sum over chunks of the bytes BEFORE the chunk) + sum of the chunks' own
// (16 - j)-weighted sums, which is what the vector loop accumulates (psadbw for the plain sums, pmaddwd for the weights).
inline void adler_update(uint32_t *pa, uint32_t *pb, const uint8_t *p, size_t n) {
    uint64_t a = *pa, b = *pb;
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16);  // (e7 .. e0): byte 0 weighs 16
    const __m128i w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
    while (n >= 16) {
        const size_t k = std::min<size_t>(n, 5552) & ~(size_t)15;
        n -= k;
        __m128i v_s1 = zero, v_ps = zero, v_w = zero;
        for (size_t i = 0; i < k; i += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p + i));
            v_ps = _mm_add_epi32(v_ps, v_s1);
            v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(v, zero));
            v_w = _mm_add_epi32(v_w, _mm_madd_epi16(_mm_unpacklo_epi8(v, zero), w_lo));
            v_w = _mm_add_epi32(v_w, _mm_madd_epi16(_mm_unpackhi_epi8(v, zero), w_hi));
        }
        p += k;
        uint32_t t[4];
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), v_s1);
        const uint64_t s1 = (uint64_t)t[0] + t[2];
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), v_ps);
        const uint64_t ps = (uint64_t)t[0] + t[2];
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), v_w);
        const uint64_t w = (uint64_t)t[0] + t[1] + t[2] + t[3];
        b = (b + (uint64_t)k * a + 16 * ps + w) % kAdlerMod;
        a = (a + s1) % kAdlerMod;
    }
#endif
    while (n) {
        size_t k = std::min<size_t>(n, 5552);
        n -= k;
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= kAdlerMod;
        b %= kAdlerMod;
    }
    *pa = (uint32_t)a;
    *pb = (uint32_t)b;
}

}  // namespace mic
