// PNG decoder of libmic.so (host only): the other half of SURVEY 8(f1).  Replaces Pillow behind
// load_object_images / open_rgba (compositor.py:25-35: Image.open(path).convert("RGBA"), re-run every iteration at
// macro_placement_test.py:1493, 1679) for the files that path meets: 8-bit RGBA / RGB / grey / grey+alpha and palette
// (1-8 bit, with or without tRNS) PNGs, non-interlaced.  Output = exactly what Image.open(f).convert("RGBA") holds.
// Everything else -- 16-bit samples, Adam7, tRNS on RGB / grey, APNG, a broken checksum, trailing garbage inside the
// stream -- is DECLINED (kPngUnsupported / kPngMalformed) and the Python binding hands the file to Pillow, which then
// decodes it or raises its own error: this decoder never guesses.
//
// inflate: one pass into a buffer of the exact raw size (height x (1 + row bytes)), so matches never wrap and every
// bound is known up front; 64-bit bit buffer, two-level canonical Huffman tables (10-bit primary for literals /
// lengths, 8-bit for distances), an unchecked fast loop while >= 8 input bytes and >= 274 output bytes remain.
// Unfilter: Sub / Up / Average / Paeth with SSE2 for 4-byte pixels (one pixel per step in a register, as libpng's
// filter_sse2 does), scalar otherwise.  CRC-32 of every chunk and the stream's Adler-32 are verified.
#include "png_decode.h"

#include "png_checksum.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mic {
namespace {

// ---------------------------------------------------------------------------------------- checksums: png_checksum.h
uint32_t crc32(const uint8_t *p, size_t n) { return crc32_update(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu; }
uint32_t adler32(const uint8_t *p, size_t n) {
    uint32_t a = 1, b = 0;
    adler_update(&a, &b, p, n);
    return a | (b << 16);
}
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// ---------------------------------------------------------------------------------------- inflate
// Decode tables, two levels.  Entry (32 bits):
//   bits 0..7   bits this entry consumes: the codeword's length at this level (0 = no such code);
//   bits 8..11  extra bits that follow the codeword (lengths, distances) -- or the subtable's index bits for a link;
//   bits 12..15 kind: kLit, kEob, kSub (a link to a subtable), 0 = a length / distance;
//   bits 16..31 the literal, the length / distance BASE, or the subtable's offset in the table.
constexpr int kLitRoot = 11, kDistRoot = 8;
constexpr uint32_t kLit = 1u << 12, kEob = 1u << 13, kSub = 1u << 14;
constexpr int kMaxLitTable = (1 << kLitRoot) + 1024, kMaxDistTable = (1 << kDistRoot) + 512;  // (subtables: < 2^(15 - root) entries per long prefix, bounded by the code space)

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

enum TableKind { kCodeLengths, kLitLen, kDistances };
constexpr uint32_t kNoSymbol = 0xFFFFFFFFu;  // payload_of: the code exists but must not occur -> a table entry of 0 ("no such code")

inline uint32_t payload_of(TableKind kind, int sym) {
    if (kind == kCodeLengths) return kLit | ((uint32_t)sym << 16);
    if (kind == kLitLen) {
        if (sym < 256) return kLit | ((uint32_t)sym << 16);
        if (sym == 256) return kEob;
        if (sym > 285) return kNoSymbol;  // (286, 287: codes of the fixed table that may not occur in the data)
        return ((uint32_t)kLenExtra[sym - 257] << 8) | ((uint32_t)kLenBase[sym - 257] << 16);
    }
    if (sym > 29) return kNoSymbol;
    return ((uint32_t)kDistExtra[sym] << 8) | ((uint32_t)kDistBase[sym] << 16);
}

// Canonical code from lens[0..n) into t (capacity cap entries).  false: over-subscribed, or incomplete other than the
// cases zlib accepts (a single code of length 1; allow_incomplete: the fixed distance table's 30 codes of 32).
bool build(const uint8_t *lens, int n, int root, TableKind kind, uint32_t *t, int cap, bool allow_incomplete = false) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int max_len = 15;
    while (max_len > 0 && count[max_len] == 0) --max_len;
    const uint32_t root_n = (uint32_t)1 << root;
    memset(t, 0, sizeof(uint32_t) * root_n);
    if (max_len == 0) return true;  // no codes: every lookup is invalid (legal for a distance tree nobody uses)
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
    }
    if (left > 0 && !allow_incomplete && !(max_len == 1 && count[1] == 1)) return false;
    uint32_t next[16];
    next[1] = 0;
    for (int l = 1; l < 15; ++l) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
    // codes in canonical order = by (length, symbol): symbols sorted by length
    uint16_t sorted[288];
    int offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (int i = 0; i < n; ++i)
        if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
    const int n_codes = offs[15];
    auto reversed = [](uint32_t c, int l) {  // deflate codes are read LSB first
        uint32_t r = 0;
        for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
        return r;
    };
    int k = 0;
    for (; k < n_codes && lens[sorted[k]] <= root; ++k) {
        const int sym = sorted[k], l = lens[sym];
        const uint32_t pl = payload_of(kind, sym), e = pl == kNoSymbol ? 0u : ((uint32_t)l | pl);
        for (uint32_t at = reversed(next[l]++, l); at < root_n; at += (uint32_t)1 << l) t[at] = e;
    }
    // the longer codes: one subtable per root prefix, sized by the longest code under it (canonical order keeps the
    // codes of one prefix together, longest last... not: sized in a first pass)
    if (k < n_codes) {
        uint8_t sub_bits[1 << kLitRoot];
        memset(sub_bits, 0, root_n);
        uint32_t probe[16];
        memcpy(probe, next, sizeof probe);
        for (int j = k; j < n_codes; ++j) {
            const int l = lens[sorted[j]];
            const uint32_t r = reversed(probe[l]++, l), pfx = r & (root_n - 1);
            sub_bits[pfx] = (uint8_t)std::max<int>(sub_bits[pfx], l - root);
        }
        uint32_t total = root_n;
        for (uint32_t pfx = 0; pfx < root_n; ++pfx)
            if (sub_bits[pfx]) {
                const uint32_t sz = (uint32_t)1 << sub_bits[pfx];
                if (total + sz > (uint32_t)cap) return false;
                t[pfx] = (uint32_t)root | ((uint32_t)sub_bits[pfx] << 8) | kSub | (total << 16);
                memset(t + total, 0, sizeof(uint32_t) * sz);
                total += sz;
            }
        for (int j = k; j < n_codes; ++j) {
            const int sym = sorted[j], l = lens[sym];
            const uint32_t r = reversed(next[l]++, l), pfx = r & (root_n - 1);
            const uint32_t base = t[pfx] >> 16, sb = (t[pfx] >> 8) & 15u;
            const uint32_t pl = payload_of(kind, sym), e = pl == kNoSymbol ? 0u : ((uint32_t)(l - root) | pl);
            for (uint32_t at = r >> root; at < ((uint32_t)1 << sb); at += (uint32_t)1 << (l - root)) t[base + at] = e;
        }
    }
    return true;
}

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int n = 0;  // bits of buf accounted for by p (bits above n mirror the bytes at p when >= 8 of them were readable)
    bool over = false;  // a read went past the end of the input
    inline void refill_fast() {  // needs end - p >= 8; afterwards 56 <= n <= 63
        uint64_t v;
        memcpy(&v, p, 8);
        buf |= v << n;
        p += (63 - n) >> 3;
        n |= 56;
    }
    inline void refill() {
        if (end - p >= 8) {
            refill_fast();
        } else {
            while (n <= 56 && p < end) {
                buf |= (uint64_t)*p++ << n;
                n += 8;
            }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & (((uint64_t)1 << k) - 1)); }
    inline void drop(int k) {
        if (k > n) { over = true; k = n; }
        buf >>= k;
        n -= k;
    }
    inline uint32_t get(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
};

// One literal / length / distance (or code-length) symbol -> its table entry, codeword consumed; 0 = no such code.
inline uint32_t lookup(Bits &b, const uint32_t *t, int root) {
    uint32_t e = t[b.peek(root)];
    if (e & kSub) {
        b.drop((int)(e & 255u));
        e = t[(e >> 16) + b.peek((int)((e >> 8) & 15u))];
    }
    if ((e & 255u) == 0) return 0;
    b.drop((int)(e & 255u));
    return e;
}

inline void copy_match(uint8_t *dst, uint32_t d, uint32_t len, bool room8) {
    const uint8_t *src = dst - d;
    if (d >= 8 && room8) {  // 8 bytes at a time (writes up to 7 bytes past len: the caller has checked the room)
        for (uint32_t k = 0; k < len; k += 8) memcpy(dst + k, src + k, 8);
    } else if (d == 1) {
        memset(dst, *src, len);
    } else if (d >= 4 && room8) {  // a 4-byte pixel repeated: the commonest short distance in filtered RGBA rows
        for (uint32_t k = 0; k < len; k += 4) memcpy(dst + k, src + k, 4);
    } else {
        for (uint32_t k = 0; k < len; ++k) dst[k] = src[k];
    }
}

// zlib stream -> exactly out_n bytes; the Adler-32 trailer is verified.  0 ok, else kPngMalformed.
int inflate_exact(const uint8_t *in, size_t in_n, uint8_t *out, size_t out_n, bool verify, std::string *err) {
    auto bad = [&](const char *what) {
        if (err) *err = std::string("png: ") + what;
        return kPngMalformed;
    };
    if (in_n < 6) return bad("zlib stream too short");
    if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return bad("bad zlib header");
    Bits b{in + 2, in + in_n};
    size_t o = 0;
    uint32_t lit[kMaxLitTable], dist[kMaxDistTable], cl_t[128];
    struct Fixed {
        uint32_t lit[kMaxLitTable], dist[kMaxDistTable];
        Fixed() {
            uint8_t l[288];
            for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            build(l, 288, kLitRoot, kLitLen, lit, kMaxLitTable);
            for (int i = 0; i < 30; ++i) l[i] = 5;
            build(l, 30, kDistRoot, kDistances, dist, kMaxDistTable, /*allow_incomplete=*/true);
        }
    };
    static const Fixed fixed;
    for (bool last = false; !last;) {
        b.refill();
        last = b.get(1) != 0;
        const uint32_t type = b.get(2);
        if (type == 0) {
            b.drop(b.n & 7);  // to the byte boundary
            b.refill();
            const uint32_t len = b.get(16), nlen = b.get(16);
            if (b.over || (len ^ 0xFFFFu) != nlen) return bad("bad stored block");
            size_t k = len;
            while (k && b.n >= 8) {  // the bytes still in the bit buffer belong to the block
                if (o >= out_n) return bad("more data than the image holds");
                out[o++] = (uint8_t)b.get(8);
                --k;
            }
            if (k) {
                b.buf = 0;  // (n == 0 here; the bits a fast refill left above n mirror bytes at the OLD position)
                b.n = 0;
                if ((size_t)(b.end - b.p) < k) return bad("truncated stored block");
                if (out_n - o < k) return bad("more data than the image holds");
                memcpy(out + o, b.p, k);
                b.p += k;
                o += k;
            }
            continue;
        }
        const uint32_t *L, *D;
        if (type == 1) {
            L = fixed.lit;
            D = fixed.dist;
        } else if (type == 2) {
            b.refill();
            const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
            if (hlit > 286 || hdist > 30) return bad("bad code counts");
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                b.refill();
                cl[order[i]] = (uint8_t)b.get(3);
            }
            if (!build(cl, 19, 7, kCodeLengths, cl_t, 128)) return bad("bad code-length code");
            uint8_t lens[286 + 30 + 8] = {0};
            for (int i = 0; i < hlit + hdist;) {
                b.refill();
                const uint32_t e = lookup(b, cl_t, 7);
                if (!e) return bad("bad code-length symbol");
                const uint32_t sym = e >> 16;
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                } else {
                    int rep;
                    uint8_t v = 0;
                    if (sym == 16) {
                        if (i == 0) return bad("repeat without a previous length");
                        v = lens[i - 1];
                        rep = 3 + (int)b.get(2);
                    } else if (sym == 17) {
                        rep = 3 + (int)b.get(3);
                    } else {
                        rep = 11 + (int)b.get(7);
                    }
                    if (i + rep > hlit + hdist) return bad("code lengths overrun");
                    while (rep--) lens[i++] = v;
                }
            }
            if (b.over) return bad("truncated stream");
            if (lens[256] == 0) return bad("no end-of-block code");
            if (!build(lens, hlit, kLitRoot, kLitLen, lit, kMaxLitTable) ||
                !build(lens + hlit, hdist, kDistRoot, kDistances, dist, kMaxDistTable))
                return bad("bad Huffman code");
            L = lit;
            D = dist;
        } else {
            return bad("bad block type");
        }
        // ---- the block's symbols.  Fast loop: while 16 input bytes (two 8-byte refills per turn) and 272 output bytes (3 literals + a 258-byte match
        // + 8 bytes of copy slack + 3) remain nothing is bounds-checked per symbol; the careful loop below finishes.
        bool done = false;
        while (!done && (size_t)(b.end - b.p) >= 16 && out_n - o >= 272) {
            b.refill_fast();
            uint32_t e = L[b.buf & ((1u << kLitRoot) - 1)];
            if (e & kLit) {  // up to three literals out of one refill (15 bits each at most)
                out[o++] = (uint8_t)(e >> 16);
                b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
                e = L[b.buf & ((1u << kLitRoot) - 1)];
                if (e & kLit) {
                    out[o++] = (uint8_t)(e >> 16);
                    b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
                    e = L[b.buf & ((1u << kLitRoot) - 1)];
                    if (e & kLit) {
                        out[o++] = (uint8_t)(e >> 16);
                        b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
                        continue;
                    }
                }
                b.refill_fast();  // (>= 9 readable bytes still: p has moved by at most 7 since the check of 16)
            }
            if (e & kSub) {
                b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
                e = L[(e >> 16) + (uint32_t)(b.buf & ((1u << ((e >> 8) & 15u)) - 1))];
                if (e & kLit) {
                    out[o++] = (uint8_t)(e >> 16);
                    b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
                    continue;
                }
            }
            if ((e & 255u) == 0) return bad("bad literal/length code");
            b.buf >>= (e & 255u); b.n -= (int)(e & 255u);
            if (e & kEob) { done = true; break; }
            const uint32_t xl = (e >> 8) & 15u;
            const uint32_t len = (e >> 16) + (uint32_t)(b.buf & ((1u << xl) - 1));
            b.buf >>= xl; b.n -= (int)xl;
            // (consumed since the last refill: <= 15 + 5 bits -> >= 36 left; a distance needs <= 15 + 13)
            uint32_t de = D[b.buf & ((1u << kDistRoot) - 1)];
            if (de & kSub) {
                b.buf >>= (de & 255u); b.n -= (int)(de & 255u);
                de = D[(de >> 16) + (uint32_t)(b.buf & ((1u << ((de >> 8) & 15u)) - 1))];
            }
            if ((de & 255u) == 0) return bad("bad distance code");
            b.buf >>= (de & 255u); b.n -= (int)(de & 255u);
            const uint32_t xd = (de >> 8) & 15u;
            const uint32_t d = (de >> 16) + (uint32_t)(b.buf & ((1u << xd) - 1));
            b.buf >>= xd; b.n -= (int)xd;
            if (d > o) return bad("distance beyond the start of the stream");
            copy_match(out + o, d, len, true);
            o += len;
        }
        while (!done) {
            b.refill();
            if (b.over) return bad("truncated stream");
            const uint32_t e = lookup(b, L, kLitRoot);
            if (!e) return bad("bad literal/length code");
            if (e & kLit) {
                if (o >= out_n) return bad("more data than the image holds");
                out[o++] = (uint8_t)(e >> 16);
                continue;
            }
            if (e & kEob) break;
            const uint32_t len = (e >> 16) + b.get((int)((e >> 8) & 15u));
            if (b.n < 32) b.refill();
            const uint32_t de = lookup(b, D, kDistRoot);
            if (!de) return bad("bad distance code");
            if (b.n < 16) b.refill();
            const uint32_t d = (de >> 16) + b.get((int)((de >> 8) & 15u));
            if (b.over) return bad("truncated stream");
            if (d > o) return bad("distance beyond the start of the stream");
            if (out_n - o < len) return bad("more data than the image holds");
            copy_match(out + o, d, len, out_n - o >= (size_t)len + 8);
            o += len;
        }
        if (b.over || b.n < 0) return bad("truncated stream");
    }
    if (o != out_n) return bad("less data than the image holds");
    // Adler-32 trailer: the bytes behind the last block, after the bits left in the buffer are given back
    b.drop(b.n & 7);
    const uint8_t *tail = b.p - (b.n >> 3);
    if (b.end - tail < 4) return bad("missing Adler-32");
    if (verify && be32(tail) != adler32(out, out_n)) return bad("Adler-32 mismatch");
    if (b.end - tail != 4) return bad("data after the zlib stream");  // (Pillow tolerates some of this; declined: let it decide)
    return 0;
}

// ---------------------------------------------------------------------------------------- unfilter
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// cur: the filtered row (n bytes) -> in place the reconstructed row; prev: the row above (nullptr: zeros); bpp 1..4
void unfilter_row(int ft, uint8_t *cur, const uint8_t *prev, size_t n, int bpp) {
    switch (ft) {
        case 0: return;
        case 1:
#if defined(__SSE2__)
            if (bpp == 4 && n >= 4) {
                __m128i a = _mm_setzero_si128();
                size_t i = 0;
                for (; i + 4 <= n; i += 4) {
                    int32_t v;
                    memcpy(&v, cur + i, 4);
                    a = _mm_add_epi8(a, _mm_cvtsi32_si128(v));
                    v = _mm_cvtsi128_si32(a);
                    memcpy(cur + i, &v, 4);
                }
                return;
            }
#endif
            for (size_t i = (size_t)bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            return;
        case 2:
            if (!prev) return;
            {
                size_t i = 0;
#if defined(__SSE2__)
                for (; i + 16 <= n; i += 16)
                    _mm_storeu_si128(reinterpret_cast<__m128i *>(cur + i),
                                     _mm_add_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(cur + i)),
                                                  _mm_loadu_si128(reinterpret_cast<const __m128i *>(prev + i))));
#endif
                for (; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            }
            return;
        case 3:
#if defined(__SSE2__)
            if (bpp == 4 && n >= 4) {
                const __m128i z = _mm_setzero_si128();
                __m128i a = z;  // the pixel to the left, 16-bit lanes
                for (size_t i = 0; i + 4 <= n; i += 4) {
                    int32_t v, u = 0;
                    memcpy(&v, cur + i, 4);
                    if (prev) memcpy(&u, prev + i, 4);
                    const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(u), z);
                    const __m128i avg = _mm_srli_epi16(_mm_add_epi16(a, b), 1);
                    a = _mm_and_si128(_mm_add_epi16(_mm_unpacklo_epi8(_mm_cvtsi32_si128(v), z), avg), _mm_set1_epi16(255));
                    v = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
                    memcpy(cur + i, &v, 4);
                }
                return;
            }
#endif
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            return;
        case 4:
#if defined(__SSE2__)
            if (bpp == 4 && n >= 4 && prev) {
                const __m128i z = _mm_setzero_si128();
                __m128i a = z, c = z;  // left, upper-left (16-bit lanes)
                for (size_t i = 0; i + 4 <= n; i += 4) {
                    int32_t v, u;
                    memcpy(&v, cur + i, 4);
                    memcpy(&u, prev + i, 4);
                    const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(u), z);
                    // pa = |b - c|, pb = |a - c|, pc = |a + b - 2c|
                    const __m128i pa0 = _mm_sub_epi16(b, c), pb0 = _mm_sub_epi16(a, c), pc0 = _mm_add_epi16(pa0, pb0);
                    auto abs16 = [&](__m128i x) { return _mm_max_epi16(x, _mm_sub_epi16(z, x)); };
                    const __m128i pa = abs16(pa0), pb = abs16(pb0), pc = abs16(pc0);
                    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
                    // a if pa is smallest, else b if pb is, else c (the order of the PNG specification)
                    const __m128i ma = _mm_cmpeq_epi16(smallest, pa), mb = _mm_cmpeq_epi16(smallest, pb);
                    const __m128i pick = _mm_or_si128(_mm_and_si128(ma, a),
                                                      _mm_andnot_si128(ma, _mm_or_si128(_mm_and_si128(mb, b), _mm_andnot_si128(mb, c))));
                    c = b;
                    a = _mm_and_si128(_mm_add_epi16(_mm_unpacklo_epi8(_mm_cvtsi32_si128(v), z), pick), _mm_set1_epi16(255));
                    v = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
                    memcpy(cur + i, &v, 4);
                }
                return;
            }
#endif
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0,
                          c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            return;
        default: return;
    }
}

struct Parsed {
    int32_t w = 0, h = 0;
    int depth = 0, ctype = 0;
    std::vector<uint8_t> idat;        // the concatenated zlib stream (one IDAT: a view would do; files are small)
    const uint8_t *idat_one = nullptr;  // ... or the only IDAT chunk's payload, in place
    size_t idat_n = 0;
    uint8_t pal[256][4];
    int n_pal = 0;
    bool has_trns = false;
};

int parse(const uint8_t *d, size_t n, bool verify, bool want_idat, Parsed *P, std::string *err) {
    auto bad = [&](int code, const char *what) {
        if (err) *err = std::string("png: ") + what;
        return code;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) return bad(kPngMalformed, "not a PNG file");
    crc_init();
    size_t pos = 8;
    bool seen_ihdr = false, seen_iend = false, idat_open = false, idat_closed = false;
    int n_idat = 0;
    for (int i = 0; i < 256; ++i) { P->pal[i][0] = P->pal[i][1] = P->pal[i][2] = 0; P->pal[i][3] = 255; }
    while (pos + 12 <= n && !seen_iend) {
        const uint32_t len = be32(d + pos);
        if (len > 0x7fffffffu || (size_t)len > n - pos - 12) return bad(kPngMalformed, "chunk runs past the end of the file");
        const uint8_t *type = d + pos + 4, *body = d + pos + 8;
        if (verify && crc32(type, (size_t)len + 4) != be32(body + len)) return bad(kPngMalformed, "chunk CRC mismatch");
        const uint32_t tag = be32(type);
        if (!seen_ihdr && tag != 0x49484452u) return bad(kPngMalformed, "IHDR is not the first chunk");
        if (tag == 0x49484452u) {  // IHDR
            if (seen_ihdr || len != 13) return bad(kPngMalformed, "bad IHDR");
            seen_ihdr = true;
            const uint32_t w = be32(body), h = be32(body + 4);
            P->depth = body[8];
            P->ctype = body[9];
            if (w == 0 || h == 0 || w > 0x7fffffffu || h > 0x7fffffffu || body[10] != 0 || body[11] != 0 || body[12] > 1)
                return bad(kPngMalformed, "bad IHDR");
            if (w > 65535 || h > 65535) return bad(kPngUnsupported, "image larger than 65535 on a side");
            P->w = (int32_t)w;
            P->h = (int32_t)h;
            if (body[12] != 0) return bad(kPngUnsupported, "interlaced");
            const int dp = P->depth, ct = P->ctype;
            const bool ok = (ct == 6 && dp == 8) || (ct == 2 && dp == 8) || (ct == 0 && dp == 8) || (ct == 4 && dp == 8) ||
                            (ct == 3 && (dp == 1 || dp == 2 || dp == 4 || dp == 8));
            if (!ok) {
                const bool legal = (ct == 0 && (dp == 1 || dp == 2 || dp == 4 || dp == 16)) || ((ct == 2 || ct == 4 || ct == 6) && dp == 16);
                return bad(legal ? kPngUnsupported : kPngMalformed, legal ? "sample depth left to Pillow" : "bad colour type / depth");
            }
        } else if (tag == 0x504c5445u) {  // PLTE
            if (idat_open || P->n_pal || len % 3 != 0 || len > 768 || len == 0) return bad(kPngMalformed, "bad PLTE");
            P->n_pal = (int)(len / 3);
            for (int i = 0; i < P->n_pal; ++i) { P->pal[i][0] = body[3 * i]; P->pal[i][1] = body[3 * i + 1]; P->pal[i][2] = body[3 * i + 2]; }
        } else if (tag == 0x74524e53u) {  // tRNS
            if (idat_open || P->has_trns) return bad(kPngMalformed, "bad tRNS");
            if (P->ctype != 3) return bad(kPngUnsupported, "tRNS on a non-palette image");  // (a colour key: Pillow's business)
            if (!P->n_pal || (int)len > P->n_pal) return bad(kPngMalformed, "bad tRNS");
            P->has_trns = true;
            for (uint32_t i = 0; i < len; ++i) P->pal[i][3] = body[i];
        } else if (tag == 0x49444154u) {  // IDAT
            if (idat_closed) return bad(kPngMalformed, "IDAT chunks are not consecutive");
            idat_open = true;
            ++n_idat;
            if (want_idat) {
                if (n_idat == 1) {
                    P->idat_one = body;
                    P->idat_n = len;
                } else {
                    if (n_idat == 2) P->idat.assign(P->idat_one, P->idat_one + P->idat_n);
                    P->idat.insert(P->idat.end(), body, body + len);
                }
            }
        } else if (tag == 0x49454e44u) {  // IEND
            if (len != 0) return bad(kPngMalformed, "bad IEND");
            seen_iend = true;
        } else {
            if (idat_open) idat_closed = true;
            if (tag == 0x6163544cu || tag == 0x6663544cu || tag == 0x66644154u)  // acTL / fcTL / fdAT
                return bad(kPngUnsupported, "animated PNG");
            if (!(type[0] & 0x20)) return bad(kPngUnsupported, "unknown critical chunk");
        }
        if (tag != 0x49444154u && idat_open) idat_closed = true;
        pos += 12 + (size_t)len;
    }
    if (!seen_ihdr || !seen_iend || n_idat == 0) return bad(kPngMalformed, "truncated file");
    if (P->ctype == 3 && P->n_pal == 0) return bad(kPngMalformed, "palette image without PLTE");
    if (n_idat > 1) {
        P->idat_one = P->idat.data();
        P->idat_n = P->idat.size();
    }
    return 0;
}

}  // namespace

int png_decode_info(const uint8_t *data, size_t n, int32_t *w, int32_t *h, std::string *err) {
    Parsed P;
    if (int rc = parse(data, n, /*verify=*/true, /*want_idat=*/false, &P, err)) return rc;
    *w = P.w;
    *h = P.h;
    return 0;
}

int png_decode_rows(const uint8_t *data, size_t n, uint8_t *const *rows, int32_t w, int32_t h, bool verify, std::string *err) {
    Parsed P;
    if (int rc = parse(data, n, verify, /*want_idat=*/true, &P, err)) return rc;
    if (P.w != w || P.h != h) {
        if (err) *err = "png: the image is not of the size the caller allocated";
        return kPngMalformed;
    }
    const int channels = P.ctype == 6 ? 4 : P.ctype == 2 ? 3 : P.ctype == 4 ? 2 : 1;
    const size_t row_bytes = ((size_t)w * (size_t)channels * (size_t)P.depth + 7) / 8;
    const int bpp = std::max(1, channels * P.depth / 8);
    const size_t raw_n = (size_t)h * (row_bytes + 1);
    // DEFLATE expands at most 1032 : 1 (a 258-byte match per 2 bits): a stream too short to hold the scanlines the
    // header promises is refused BEFORE h * (row_bytes + 1) bytes are allocated for it (a 65535 x 65535 IHDR in front
    // of a few bytes of IDAT must not cost 17 GB)
    if (P.idat_n < 6 || raw_n / 1032 > P.idat_n) {
        if (err) *err = "png: the IDAT stream cannot hold the image the header declares";
        return kPngMalformed;
    }
    std::unique_ptr<uint8_t[]> raw(new (std::nothrow) uint8_t[raw_n + 16]);
    if (!raw) {
        if (err) *err = "png: out of memory";
        return kPngNoMem;
    }
    if (int rc = inflate_exact(P.idat_one, P.idat_n, raw.get(), raw_n, verify, err)) return rc;
    const uint8_t *prev = nullptr;
    for (int32_t y = 0; y < h; ++y) {
        uint8_t *line = raw.get() + (size_t)y * (row_bytes + 1);
        const int ft = line[0];
        if (ft > 4) {
            if (err) *err = "png: bad filter type";
            return kPngMalformed;
        }
        uint8_t *cur = line + 1;
        unfilter_row(ft, cur, prev, row_bytes, bpp);
        prev = cur;
        uint8_t *o = rows[y];
        switch (P.ctype) {
            case 6: memcpy(o, cur, (size_t)w * 4); break;
            case 2:
                for (int32_t x = 0; x < w; ++x) { o[4 * x] = cur[3 * x]; o[4 * x + 1] = cur[3 * x + 1]; o[4 * x + 2] = cur[3 * x + 2]; o[4 * x + 3] = 255; }
                break;
            case 0:
                for (int32_t x = 0; x < w; ++x) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = cur[x]; o[4 * x + 3] = 255; }
                break;
            case 4:
                for (int32_t x = 0; x < w; ++x) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = cur[2 * x]; o[4 * x + 3] = cur[2 * x + 1]; }
                break;
            default: {  // palette, 1 / 2 / 4 / 8 bits per index, most significant bits first
                const int dp = P.depth, per = 8 / dp, mask = (1 << dp) - 1;
                for (int32_t x = 0; x < w; ++x) {
                    const int idx = dp == 8 ? cur[x] : (cur[x / per] >> (8 - dp - (x % per) * dp)) & mask;
                    // (an index beyond the palette: Pillow pads its palette with zeros -> black, opaque unless tRNS covers it;
                    // P.pal holds exactly that)
                    memcpy(o + 4 * (size_t)x, P.pal[idx], 4);
                }
            }
        }
    }
    return 0;
}

int png_decode_many(int n, const uint8_t *const *datas, const size_t *sizes, uint8_t *const *const *rows, const int32_t *ws,
                    const int32_t *hs, int threads, int *status, std::string *err) {
    if (n <= 0) return 0;
    crc_init();
    std::vector<std::string> errs((size_t)n);
    auto one = [&](int i) { status[i] = png_decode_rows(datas[i], sizes[i], rows[i], ws[i], hs[i], true, &errs[(size_t)i]); };
    const int T = std::max(1, std::min(std::min(threads <= 0 ? 8 : threads, n), 16));
    if (T == 1) {
        for (int i = 0; i < n; ++i) one(i);
    } else {
        // the biggest files first, dealt round-robin: a bundle is a few cutouts of very different sizes
        std::vector<int> order((size_t)n);
        for (int i = 0; i < n; ++i) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] > sizes[b]; });
        std::vector<std::thread> pool;
        auto work = [&](int t) {
            for (int k = t; k < n; k += T) one(order[(size_t)k]);
        };
        try {
            for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
        } catch (...) {  // no more threads: this one does what the missing ones would have
            for (int t = (int)pool.size() + 1; t < T; ++t) work(t);
        }
        work(0);
        for (auto &th : pool) th.join();
    }
    int worst = 0;
    for (int i = 0; i < n; ++i)
        if (status[i] != 0 && worst == 0) {
            worst = status[i];
            if (err) *err = errs[(size_t)i];
        }
    return worst;
}

}  // namespace mic
