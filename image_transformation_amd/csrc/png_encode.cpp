// PNG writer (host side of libmic.so): RGBA8 rows -> a PNG file / byte string.  No dependency (own deflate,
// CRC-32, Adler-32).  It replaces PIL's encoder behind the reference's `draft.save(path)` /
// `canvas_img.save(canvas_path)` / the overlay save (macro_placement_test.py:1428-1430, 1513-1514, 1699-1700): once
// the pixels come off the GPU, zlib level 6 on one thread is 95 % of the deterministic loop's wall time.
//
// Design: the image is cut into horizontal stripes, one per worker thread.  A stripe is filtered (per row the
// cheaper of PNG's Sub and Up filters, by the sum of absolute residuals), LZ77-tokenised (one-probe hash of four
// bytes, greedy, a fast path for runs of one byte -- what a filtered solid background is -- and LZ4-style skipping
// through incompressible data), and written as deflate blocks of its own (per block the cheapest of dynamic
// Huffman, fixed Huffman and stored), closed by an empty stored block so that it ends on a byte boundary.  Every
// stripe becomes one IDAT chunk (its CRC is computed by the same thread); the zlib stream is their concatenation,
// its Adler-32 the combination of the stripes' checksums in a last 4-byte IDAT chunk.  Nothing is serial but the
// final gather of the compressed pieces.  Decoders see an ordinary 8-bit RGBA, non-interlaced PNG.
#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "png_encode.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "png_checksum.h"

namespace mic {
namespace {

// ---------------------------------------------------------------------------------------- checksums: png_checksum.h
// adler32(A || B) from adler32(A) = (a1, b1), adler32(B) = (a2, b2) and len(B)  (zlib's adler32_combine)
uint32_t adler_combine(uint32_t ad1, uint32_t ad2, uint64_t len2) {
    const uint32_t rem = (uint32_t)(len2 % kAdlerMod);
    uint32_t sum1 = ad1 & 0xffff;
    uint32_t sum2 = (uint32_t)(((uint64_t)rem * sum1) % kAdlerMod);
    sum1 += (ad2 & 0xffff) + kAdlerMod - 1;
    sum2 += ((ad1 >> 16) & 0xffff) + ((ad2 >> 16) & 0xffff) + kAdlerMod - rem;
    if (sum1 >= kAdlerMod) sum1 -= kAdlerMod;
    if (sum1 >= kAdlerMod) sum1 -= kAdlerMod;
    if (sum2 >= ((uint32_t)kAdlerMod << 1)) sum2 -= ((uint32_t)kAdlerMod << 1);
    if (sum2 >= kAdlerMod) sum2 -= kAdlerMod;
    return sum1 | (sum2 << 16);
}

// ---------------------------------------------------------------------------------------- deflate tables
constexpr int kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr int kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr int kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr int kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
    uint8_t len_sym[259];    // match length 3..258 -> length symbol index 0..28
    uint8_t dist_sym[512];   // distance -> symbol: [d - 1] for d <= 256, [256 + ((d - 1) >> 7)] above
    Tables() {
        for (int s = 0; s < 29; ++s)
            for (int l = kLenBase[s]; l < (s == 28 ? 259 : kLenBase[s + 1]) && l <= 258; ++l) len_sym[l] = (uint8_t)s;
        len_sym[258] = 28;
        for (int d = 1; d <= 32768; ++d) {
            int s = 29;
            while (kDistBase[s] > d) --s;
            if (d <= 256) dist_sym[d - 1] = (uint8_t)s;
            else dist_sym[256 + ((d - 1) >> 7)] = (uint8_t)s;
        }
    }
};
const Tables kT;

inline int dist_symbol(int d) { return d <= 256 ? kT.dist_sym[d - 1] : kT.dist_sym[256 + ((d - 1) >> 7)]; }

inline uint32_t bit_reverse(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// Canonical codes (already bit-reversed for LSB-first output) from code lengths.
void make_codes(const uint8_t *lens, int n, uint16_t *codes) {
    int count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + count[b - 1]) << 1;
        next[b] = code;
    }
    for (int i = 0; i < n; ++i)
        codes[i] = lens[i] ? (uint16_t)bit_reverse((uint32_t)next[lens[i]]++, lens[i]) : 0;
}

// Length-limited Huffman code lengths: plain Huffman by two-queue merging of the sorted frequencies, then the
// classic overflow repair (Kraft sum brought back to 1 by lengthening the cheapest symbols).
void huff_lengths(const uint32_t *freq, int n, int max_len, uint8_t *lens, bool force_two = false) {
    struct Node { uint64_t w; int l, r; };
    int idx[288];
    int m = 0;
    for (int i = 0; i < n; ++i) {
        lens[i] = 0;
        if (freq[i]) idx[m++] = i;
    }
    if (m == 0) return;
    if (m == 1) {
        lens[idx[0]] = 1;
        // the code-length code must be complete even with one symbol in use (inflate's CODES table): a second,
        // unused symbol of length 1 completes it
        if (force_two) lens[idx[0] == 0 ? 1 : 0] = 1;
        return;
    }
    std::sort(idx, idx + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    Node nodes[2 * 288];
    for (int i = 0; i < m; ++i) nodes[i] = {freq[idx[i]], -1, -1};
    int leaf = 0, inner = m, made = m;
    auto take = [&]() {
        if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
        return inner++;
    };
    while ((m - leaf) + (made - inner) > 1) {
        const int a = take(), b = take();
        nodes[made] = {nodes[a].w + nodes[b].w, a, b};
        ++made;
    }
    // depths, root = made - 1 (children always have smaller indices than their parent)
    int depth[2 * 288];
    depth[made - 1] = 0;
    for (int i = made - 1; i >= m; --i) {
        depth[nodes[i].l] = depth[i] + 1;
        depth[nodes[i].r] = depth[i] + 1;
    }
    int bl_count[64] = {0};
    for (int i = 0; i < m; ++i) ++bl_count[std::min(depth[i], 63)];
    // fold everything deeper than max_len into max_len, then repair the Kraft inequality
    int overflow = 0;
    for (int d = 63; d > max_len; --d) {
        overflow += bl_count[d];
        bl_count[max_len] += bl_count[d];
        bl_count[d] = 0;
    }
    if (overflow > 0) {
        // Kraft sum in units of 2^-max_len: the unclamped tree was complete (sum == 1), folding made it larger.  Each
        // step below (zlib's gen_bitlen repair) turns a leaf of the deepest level above max_len into an inner node whose
        // children are that leaf and one of the leaves parked at max_len: the sum drops by exactly one unit, so the
        // repaired code is complete again (inflate rejects over- and under-subscribed literal/length codes).
        int64_t kraft = 0;
        for (int d = 1; d <= max_len; ++d) kraft += (int64_t)bl_count[d] << (max_len - d);
        for (int64_t excess = kraft - ((int64_t)1 << max_len); excess > 0; --excess) {
            int d = max_len - 1;
            while (bl_count[d] == 0) --d;
            --bl_count[d];
            bl_count[d + 1] += 2;
            --bl_count[max_len];
        }
    }
    // hand the lengths out: most frequent symbols get the shortest codes
    int pos = m - 1;
    for (int d = 1; d <= max_len; ++d)
        for (int k = 0; k < bl_count[d]; ++k) lens[idx[pos--]] = (uint8_t)d;
}

// ---------------------------------------------------------------------------------------- buffers
// Growable byte buffer without value-initialisation; contents are NOT kept across ensure().
struct Buf {
    std::unique_ptr<uint8_t[]> mem;
    size_t cap = 0;
    uint8_t *ensure(size_t n) {
        if (cap < n) {
            // rounded up generously: the stripes of one image differ by a row, the images of one run by little --
            // a pooled buffer should fit the next request (a fresh one is an mmap plus page faults, which serialise
            // the workers on the process's address-space lock)
#ifdef MIC_PNG_EXACT_ALLOC  // sanitizer builds: no slack, so that a bound that is too small is an ASan report
            const size_t want = n;
#else
            const size_t want = (n + n / 8 + ((size_t)256 << 10)) & ~(((size_t)64 << 10) - 1);
#endif
            mem.reset();
            mem.reset(new uint8_t[want + 64]);
            cap = want;
        }
        return mem.get();
    }
};

// ---------------------------------------------------------------------------------------- bit writer
// Writes into memory the caller has sized for the worst case (deflate_bound): no capacity checks per put.
struct BitWriter {
    uint8_t *p;
    uint64_t acc = 0;
    int n = 0;
    explicit BitWriter(uint8_t *at) : p(at) {}
    inline void put(uint32_t bits, int count) {  // count <= 32, n < 32 on entry
        acc |= (uint64_t)bits << n;
        n += count;
        if (n >= 32) {
            const uint32_t w = (uint32_t)acc;
            memcpy(p, &w, 4);
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    void align() {
        while (n > 0) {
            *p++ = (uint8_t)acc;
            acc >>= 8;
            n -= 8;
        }
        acc = 0;
        n = 0;
    }
    void bytes(const void *src, size_t k) {  // only when aligned
        memcpy(p, src, k);
        p += k;
    }
};

// Most bytes deflate_stripe can produce for n input bytes.  Every block is at most its stored form (the cheapest of
// the three encodings is emitted, priced exactly) plus one byte of bit alignment behind a Huffman block; the stored
// form of a block of r bytes is r + 5 * ceil(r / 65535).  A block closes after kBlockTokens = 65535 tokens or
// kBlockRaw input bytes, so every block but a stripe's last covers >= 65535 bytes: at most n / 65535 + 1 blocks, and
// over all of them sum(5 * ceil(r / 65535) + 1) <= 5 * (n / 65535 + blocks) + blocks <= 11 * (n / 65535 + 1) + 6.
// (+ 64: the zlib header, the closing empty stored block, the bit writer's 4-byte stores.)
inline size_t deflate_bound(size_t n) { return n + 12 * (n / 65535 + 2) + 64; }

// ---------------------------------------------------------------------------------------- block emission
// token: literal = byte value (< 256); match = 0x80000000 | (len - 3) << 16 | (dist - 1)
constexpr uint32_t kMatchFlag = 0x80000000u;

struct FixedCodes {
    uint8_t ll_len[288], d_len[30];
    uint16_t ll_code[288], d_code[30];
    FixedCodes() {
        for (int i = 0; i < 288; ++i) ll_len[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        for (int i = 0; i < 30; ++i) d_len[i] = 5;
        make_codes(ll_len, 288, ll_code);
        make_codes(d_len, 30, d_code);
    }
};
const FixedCodes kFixed;

void emit_tokens(BitWriter &bw, const uint32_t *tok, size_t n, const uint8_t *ll_len, const uint16_t *ll_code,
                 const uint8_t *d_len, const uint16_t *d_code) {
    // per length symbol / distance symbol: the code with room for the extra bits above it, so that a match is two
    // puts (length + its extra, distance + its extra: at most 15 + 5 and 15 + 13 bits) instead of four
    uint32_t lcode[29], dcode[30];
    uint8_t lbits[29], dbits[30];
    for (int s = 0; s < 29; ++s) {
        lcode[s] = ll_code[257 + s];
        lbits[s] = ll_len[257 + s];
    }
    for (int s = 0; s < 30; ++s) {
        dcode[s] = d_code[s];
        dbits[s] = d_len[s];
    }
    for (size_t i = 0; i < n; ++i) {
        const uint32_t t = tok[i];
        if (!(t & kMatchFlag)) {
            bw.put(ll_code[t], ll_len[t]);
            continue;
        }
        const int len = (int)((t >> 16) & 0x7fff) + 3, dist = (int)(t & 0xffff) + 1;
        const int ls = kT.len_sym[len];
        bw.put(lcode[ls] | ((uint32_t)(len - kLenBase[ls]) << lbits[ls]), lbits[ls] + kLenExtra[ls]);
        const int ds = dist_symbol(dist);
        bw.put(dcode[ds] | ((uint32_t)(dist - kDistBase[ds]) << dbits[ds]), dbits[ds] + kDistExtra[ds]);
    }
    bw.put(ll_code[256], ll_len[256]);
}

void emit_stored(BitWriter &bw, const uint8_t *data, size_t n, bool final_block) {
    size_t off = 0;
    do {  // at least one block, also for n == 0
        const size_t k = std::min<size_t>(n - off, 65535);
        const bool last = off + k == n;
        bw.put((final_block && last) ? 1u : 0u, 1);
        bw.put(0, 2);
        bw.align();
        const uint16_t len = (uint16_t)k, nlen = (uint16_t)~len;
        bw.bytes(&len, 2);
        bw.bytes(&nlen, 2);
        if (k) bw.bytes(data + off, k);
        off += k;
    } while (off < n);
}

// One deflate block for tokens tok[0..n) that cover data[0..raw_n): the cheapest of dynamic, fixed, stored.
void emit_block(BitWriter &bw, const uint32_t *tok, size_t n, const uint32_t *ll_freq_in, const uint32_t *d_freq_in,
                const uint8_t *data, size_t raw_n, bool final_block, bool allow_dynamic) {
    uint32_t ll_freq[286], d_freq[30];
    memcpy(ll_freq, ll_freq_in, sizeof ll_freq);
    memcpy(d_freq, d_freq_in, sizeof d_freq);
    ll_freq[256] = 1;
    uint64_t extra_bits = 0;
    for (int s = 0; s < 29; ++s) extra_bits += (uint64_t)ll_freq[257 + s] * kLenExtra[s];
    for (int s = 0; s < 30; ++s) extra_bits += (uint64_t)d_freq[s] * kDistExtra[s];
    uint64_t fixed_bits = 3 + extra_bits;
    for (int i = 0; i < 286; ++i) fixed_bits += (uint64_t)ll_freq[i] * kFixed.ll_len[i];
    for (int i = 0; i < 30; ++i) fixed_bits += (uint64_t)d_freq[i] * 5;
    const uint64_t stored_bits = (uint64_t)raw_n * 8 + 40 * ((raw_n + 65534) / 65535 + (raw_n == 0)) ;

    uint8_t ll_len[286], d_len[30];
    uint64_t dyn_bits = ~(uint64_t)0;
    uint8_t cl_len[19];
    std::vector<uint16_t> cl_seq;  // code-length symbols (low 5 bits) with their extra-bit values (high bits)
    int hlit = 0, hdist = 0, hclen = 0;
    if (allow_dynamic) {
        huff_lengths(ll_freq, 286, 15, ll_len);
        huff_lengths(d_freq, 30, 15, d_len);
        bool any_dist = false;
        for (int i = 0; i < 30; ++i) any_dist |= d_len[i] != 0;
        if (!any_dist) d_len[0] = 1;  // a complete-enough distance code for decoders that insist on one
        hlit = 286;
        while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
        hdist = 30;
        while (hdist > 1 && d_len[hdist - 1] == 0) --hdist;
        uint8_t all[286 + 30];
        memcpy(all, ll_len, (size_t)hlit);
        memcpy(all + hlit, d_len, (size_t)hdist);
        const int total = hlit + hdist;
        uint32_t cl_freq[19] = {0};
        for (int i = 0; i < total;) {
            int run = 1;
            while (i + run < total && all[i + run] == all[i]) ++run;
            const int v = all[i];
            int left = run;
            if (v == 0) {
                while (left >= 11) {
                    const int k = std::min(left, 138);
                    cl_seq.push_back((uint16_t)(18 | ((k - 11) << 5)));
                    ++cl_freq[18];
                    left -= k;
                }
                if (left >= 3) {
                    cl_seq.push_back((uint16_t)(17 | ((left - 3) << 5)));
                    ++cl_freq[17];
                    left = 0;
                }
            } else {
                cl_seq.push_back((uint16_t)v);
                ++cl_freq[v];
                --left;
                while (left >= 3) {
                    const int k = std::min(left, 6);
                    cl_seq.push_back((uint16_t)(16 | ((k - 3) << 5)));
                    ++cl_freq[16];
                    left -= k;
                }
            }
            while (left-- > 0) {
                cl_seq.push_back((uint16_t)v);
                ++cl_freq[v];
            }
            i += run;
        }
        huff_lengths(cl_freq, 19, 7, cl_len, /*force_two=*/true);
        hclen = 19;
        while (hclen > 4 && cl_len[kClOrder[hclen - 1]] == 0) --hclen;
        dyn_bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen + extra_bits;
        for (int i = 0; i < 19; ++i) dyn_bits += (uint64_t)cl_freq[i] * cl_len[i];
        dyn_bits += 2 * (uint64_t)cl_freq[16] + 3 * (uint64_t)cl_freq[17] + 7 * (uint64_t)cl_freq[18];
        for (int i = 0; i < 286; ++i) dyn_bits += (uint64_t)ll_freq[i] * ll_len[i];
        for (int i = 0; i < 30; ++i) dyn_bits += (uint64_t)d_freq[i] * d_len[i];
    }
    if (stored_bits <= fixed_bits && stored_bits <= dyn_bits) {
        emit_stored(bw, data, raw_n, final_block);
        return;
    }
    bw.put(final_block ? 1u : 0u, 1);
    if (fixed_bits <= dyn_bits) {
        bw.put(1, 2);
        emit_tokens(bw, tok, n, kFixed.ll_len, kFixed.ll_code, kFixed.d_len, kFixed.d_code);
        return;
    }
    bw.put(2, 2);
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) bw.put(cl_len[kClOrder[i]], 3);
    uint16_t cl_code[19];
    make_codes(cl_len, 19, cl_code);
    for (uint16_t s : cl_seq) {
        const int sym = s & 31, ex = s >> 5;
        bw.put(cl_code[sym], cl_len[sym]);
        if (sym == 16) bw.put((uint32_t)ex, 2);
        else if (sym == 17) bw.put((uint32_t)ex, 3);
        else if (sym == 18) bw.put((uint32_t)ex, 7);
    }
    uint16_t ll_code[286], d_code[30];
    make_codes(ll_len, 286, ll_code);
    make_codes(d_len, 30, d_code);
    emit_tokens(bw, tok, n, ll_len, ll_code, d_len, d_code);
}

// ---------------------------------------------------------------------------------------- LZ77 of one stripe
constexpr int kHashBits = 15;
// (65535, not 65536: a block of incompressible data is that many literals, and its stored form is then ONE stored
// sub-block of 65535 bytes instead of 65535 + 1)
constexpr size_t kBlockTokens = 65535;
constexpr size_t kBlockRaw = 1 << 18;  // a block also ends after this many input bytes (stored fallback granularity)

inline uint32_t load32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
inline uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - kHashBits); }

inline int match_len(const uint8_t *a, const uint8_t *b, int max_len) {
    int l = 0;
    while (l + 8 <= max_len) {
        const uint64_t x = load64(a + l) ^ load64(b + l);
        if (x) return l + (__builtin_ctzll(x) >> 3);
        l += 8;
    }
    while (l < max_len && a[l] == b[l]) ++l;
    return l;
}

// Per-worker scratch, kept between calls (a free list below): fresh allocations of this size are page faults the
// next call would pay again -- in a VM they cost more than the encoding itself.
struct Scratch {
    Buf filt, row, out;
    std::unique_ptr<uint32_t[]> tok;
    std::unique_ptr<int32_t[]> head;
    size_t out_len = 0;
};

// Deflate `data[0..n)` (the filtered bytes of one stripe) to `dst` (deflate_bound(n) bytes); level 0 = stored blocks
// only.  Returns the number of bytes written.
size_t deflate_stripe(const uint8_t *data, size_t n, int level, bool last_stripe, uint8_t *dst, Scratch *sc) {
    BitWriter bw(dst);
    if (level <= 0 || n < 64) {
        emit_stored(bw, data, n, last_stripe);  // (stored blocks end on a byte boundary)
        return (size_t)(bw.p - dst);
    }
    if (!sc->tok) sc->tok.reset(new uint32_t[kBlockTokens + 8]);
    if (!sc->head) sc->head.reset(new int32_t[(size_t)1 << kHashBits]);
    uint32_t *tok = sc->tok.get();
    int32_t *head = sc->head.get();
    memset(head, 0xff, sizeof(int32_t) << kHashBits);
    uint32_t ll_freq[286], d_freq[30];
    size_t pos = 0;
    while (pos < n) {
        memset(ll_freq, 0, sizeof ll_freq);
        memset(d_freq, 0, sizeof d_freq);
        const size_t block_start = pos;
        const size_t block_limit = std::min(n, pos + kBlockRaw);
        size_t nt = 0;
        uint32_t misses = 0;
        while (pos < block_limit && nt < kBlockTokens) {
            const size_t left = n - pos;
            if (left < 8) {
                ++ll_freq[data[pos]];
                tok[nt++] = data[pos++];
                continue;
            }
            const int max_len = (int)std::min<size_t>(258, left);
            // run of one byte value (a filtered flat area): distance 1, no hashing
            if (pos > 0 && data[pos] == data[pos - 1] && load32(data + pos) == load32(data + pos - 1)) {
                const int l = match_len(data + pos, data + pos - 1, max_len);
                if (l >= 4) {
                    ++ll_freq[257 + kT.len_sym[l]];
                    ++d_freq[0];
                    tok[nt++] = kMatchFlag | ((uint32_t)(l - 3) << 16);
                    pos += (size_t)l;
                    misses = 0;
                    continue;
                }
            }
            const uint32_t v = load32(data + pos);
            const uint32_t h = hash4(v);
            const int32_t cand = head[h];
            head[h] = (int32_t)pos;
            if (cand >= 0 && pos - (size_t)cand <= 32768 && load32(data + cand) == v) {
                const int l = match_len(data + pos, data + cand, max_len);
                if (l >= 4) {
                    const int dist = (int)(pos - (size_t)cand);
                    ++ll_freq[257 + kT.len_sym[l]];
                    ++d_freq[dist_symbol(dist)];
                    tok[nt++] = kMatchFlag | ((uint32_t)(l - 3) << 16) | (uint32_t)(dist - 1);
                    // index a few positions inside the match so that later data can find it
                    if (l <= 32)
                        for (int k = 1; k < l && pos + (size_t)k + 4 <= n; k += 2) head[hash4(load32(data + pos + k))] = (int32_t)(pos + (size_t)k);
                    pos += (size_t)l;
                    misses = 0;
                    continue;
                }
            }
            // literal(s); after many misses in a row step faster through what is evidently incompressible
            const size_t step = 1 + (misses >> 5);
            ++misses;
            const size_t stop = std::min(std::min(pos + step, block_limit), pos + (kBlockTokens - nt));
            while (pos < stop) {
                ++ll_freq[data[pos]];
                tok[nt++] = data[pos++];
            }
        }
        const bool final_block = last_stripe && pos >= n;
        emit_block(bw, tok, nt, ll_freq, d_freq, data + block_start, pos - block_start, final_block, level >= 1);
    }
    if (!last_stripe) {  // end on a byte boundary: an empty stored block (zlib's "sync flush")
        bw.put(0, 3);
        bw.align();
        const uint8_t tail[4] = {0, 0, 0xff, 0xff};
        bw.bytes(tail, 4);
    } else {
        bw.align();
    }
    return (size_t)(bw.p - dst);
}

// ---------------------------------------------------------------------------------------- PNG filtering
// Row y of the image, filtered into dst[0 .. 1 + 4w): the cheaper of Sub (1) and Up (2) by the sum of absolute
// residuals (as signed bytes).  (Noise costs the same under every filter; the LZ stage skips through it and the
// block falls back to stored.)
void filter_row(const uint8_t *row, const uint8_t *prev, size_t nbytes, uint8_t *dst, uint8_t *scratch) {
    uint8_t *sub = dst + 1, *up = scratch;
    uint64_t cs = 0, cu = 0;
    auto mag = [](uint8_t v) { return (uint32_t)std::min<int>(v, 256 - v); };  // |v as a signed byte|
    size_t i = 0;
    for (; i < 4 && i < nbytes; ++i) {  // Sub: x - a, a = the byte 4 to the left (0 for the first pixel)
        sub[i] = row[i];
        cs += mag(sub[i]);
        if (prev) {
            up[i] = (uint8_t)(row[i] - prev[i]);
            cu += mag(up[i]);
        }
    }
#if defined(__SSE2__)
    {
        const __m128i zero = _mm_setzero_si128();
        __m128i acc_s = zero, acc_u = zero;
        for (; i + 16 <= nbytes; i += 16) {
            const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(row + i));
            const __m128i l = _mm_loadu_si128(reinterpret_cast<const __m128i *>(row + i - 4));
            const __m128i ds = _mm_sub_epi8(v, l);
            _mm_storeu_si128(reinterpret_cast<__m128i *>(sub + i), ds);
            acc_s = _mm_add_epi64(acc_s, _mm_sad_epu8(_mm_min_epu8(ds, _mm_sub_epi8(zero, ds)), zero));
            if (prev) {
                const __m128i du = _mm_sub_epi8(v, _mm_loadu_si128(reinterpret_cast<const __m128i *>(prev + i)));
                _mm_storeu_si128(reinterpret_cast<__m128i *>(up + i), du);
                acc_u = _mm_add_epi64(acc_u, _mm_sad_epu8(_mm_min_epu8(du, _mm_sub_epi8(zero, du)), zero));
            }
        }
        uint64_t t[2];
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), acc_s);
        cs += t[0] + t[1];
        _mm_storeu_si128(reinterpret_cast<__m128i *>(t), acc_u);
        cu += t[0] + t[1];
    }
#endif
    for (; i < nbytes; ++i) {
        sub[i] = (uint8_t)(row[i] - row[i - 4]);
        cs += mag(sub[i]);
        if (prev) {
            up[i] = (uint8_t)(row[i] - prev[i]);
            cu += mag(up[i]);
        }
    }
    if (prev && cu < cs) {
        dst[0] = 2;
        memcpy(dst + 1, up, nbytes);
    } else {
        dst[0] = 1;
    }
}

struct Stripe {
    int y0 = 0, y1 = 0;
    Scratch *sc = nullptr;  // sc->out[0 .. sc->out_len) is a complete IDAT chunk: length, "IDAT", data, CRC
    uint32_t adler = 1;
    uint64_t raw = 0;
};

void put_be32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

// c[0 .. n) = 8 header bytes (length left open, type) + data; writes length and CRC, returns n + 4
size_t finish_chunk(uint8_t *c, size_t n) {
    put_be32(c, (uint32_t)(n - 8));
    put_be32(c + n, ~crc32_update(0xffffffffu, c + 4, n - 4));
    return n + 4;
}

std::mutex g_pool_mu;
std::vector<Scratch *> g_pool;
constexpr size_t kPoolKeep = 32;                    // scratch objects kept between calls
constexpr size_t kPoolKeepBytes = (size_t)24 << 20;  // ... unless one grew beyond this (an 8K image on few threads)

Scratch *scratch_get() {
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (!g_pool.empty()) {
            Scratch *s = g_pool.back();
            g_pool.pop_back();
            return s;
        }
    }
    return new Scratch();
}

void scratch_put(Scratch *s) {
    if (!s) return;
    if (s->filt.cap + s->out.cap <= kPoolKeepBytes) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (g_pool.size() < kPoolKeep) {
            g_pool.push_back(s);
            return;
        }
    }
    delete s;
}

void encode_stripe(const uint8_t *const *rows, int32_t w, Stripe *s, bool first, bool last, int level) {
    const size_t nbytes = (size_t)w * 4;
    const int h = s->y1 - s->y0;
    Scratch *sc = s->sc = scratch_get();
    const size_t raw = (size_t)h * (nbytes + 1);
    uint8_t *filt = sc->filt.ensure(raw);
    uint8_t *tmp = sc->row.ensure(nbytes + 16);
    for (int y = s->y0; y < s->y1; ++y)
        filter_row(rows[y], y > 0 ? rows[y - 1] : nullptr, nbytes, filt + (size_t)(y - s->y0) * (nbytes + 1), tmp);
    uint32_t a = 1, b = 0;
    adler_update(&a, &b, filt, raw);
    s->adler = a | (b << 16);
    s->raw = raw;
    uint8_t *c = sc->out.ensure(deflate_bound(raw) + 32);
    const uint8_t head[8] = {0, 0, 0, 0, 'I', 'D', 'A', 'T'};
    memcpy(c, head, 8);
    size_t n = 8;
    if (first) {
        c[n++] = 0x78;  // zlib header: deflate, 32 KiB window, no preset dictionary, "fastest" level hint
        c[n++] = 0x01;
    }
    n += deflate_stripe(filt, raw, level, last, c + n, sc);
    if (n > deflate_bound(raw) + 24) abort();  // the unchecked bit writer ran past what deflate_bound promised: a bug, not an input
    sc->out_len = finish_chunk(c, n);
}

int plan_threads(int32_t w, int32_t h, int threads) {
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads <= 0) threads = 1;
        threads = std::min(threads, 16);
    }
    const uint64_t bytes = (uint64_t)w * h * 4;
    const int by_size = (int)std::max<uint64_t>(1, bytes / ((uint64_t)384 << 10));  // >= 384 KiB of pixels per worker
    return std::max(1, std::min(std::min(threads, by_size), (int)h));
}

}  // namespace

size_t png_bound(int32_t w, int32_t h) {
    if (w <= 0 || h <= 0) return 0;
    const uint64_t raw = (uint64_t)h * ((uint64_t)w * 4 + 1);
    // the stripes' deflate_bound()s (12 bytes per 65535 in all, + 88 per stripe) + 12 bytes of chunk framing per
    // stripe; at most raw / 256 KiB + 1 stripes whatever `threads` says (png_encode_rows); + signature, IHDR, the
    // Adler trailer chunk, IEND
    const uint64_t stripes = std::min<uint64_t>((uint64_t)h, raw / ((uint64_t)256 << 10) + 1);
    return (size_t)(raw + raw / 65535 * 12 + stripes * 128 + 1024);
}

PngPieces::~PngPieces() {
    for (void *h : held) scratch_put(static_cast<Scratch *>(h));
}

int png_encode_rows(const uint8_t *const *rows, int32_t w, int32_t h, int level, int threads, PngPieces *out,
                    std::string *err) {
    if (!rows || w <= 0 || h <= 0 || !out) {
        if (err) *err = "png: bad arguments";
        return -1;
    }
    crc_init();
    const int T = plan_threads(w, h, threads);
    // More stripes than workers when there are several workers: a stripe full of photo costs twenty times a stripe
    // of flat background, so the workers draw stripes from a counter instead of owning one each (a stripe boundary
    // costs ~20 bytes and the matches across it).
    int S = T;
    if (T > 1) {
        const uint64_t raw = (uint64_t)h * ((uint64_t)w * 4 + 1);
        S = (int)std::min<uint64_t>(std::min<uint64_t>((uint64_t)T * 4, (uint64_t)h), std::max<uint64_t>((uint64_t)T, raw / ((uint64_t)256 << 10)));
    }
    std::vector<Stripe> stripes((size_t)S);
    for (int t = 0; t < S; ++t) {
        stripes[(size_t)t].y0 = (int)((int64_t)h * t / S);
        stripes[(size_t)t].y1 = (int)((int64_t)h * (t + 1) / S);
    }
    std::atomic<bool> failed{false};
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int t; (t = next.fetch_add(1)) < S;) {
            try {
                encode_stripe(rows, w, &stripes[(size_t)t], t == 0, t == S - 1, level);
            } catch (...) {  // bad_alloc
                failed.store(true);
            }
        }
    };
    try {
        if (T == 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            pool.reserve((size_t)T - 1);
            try {
                for (int t = 1; t < T; ++t) pool.emplace_back(work);
            } catch (const std::exception &e) {  // no more threads: the ones that started (and this one) do all the stripes
            }
            work();
            for (auto &th : pool) th.join();
        }
    } catch (const std::exception &e) {
        if (err) *err = std::string("png: ") + e.what();
        failed.store(true);
    }
    for (auto &s : stripes) out->held.push_back(s.sc);
    if (failed.load()) {
        if (err && err->empty()) *err = "png: out of memory";
        return -3;
    }
    // signature + IHDR
    out->head.assign({0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a, 0, 0, 0, 0, 'I', 'H', 'D', 'R', 0, 0, 0, 0, 0, 0, 0, 0, 8, 6, 0, 0, 0,
                      0, 0, 0, 0});
    put_be32(&out->head[16], (uint32_t)w);
    put_be32(&out->head[20], (uint32_t)h);
    finish_chunk(&out->head[8], 8 + 13);
    out->pieces.push_back({out->head.data(), out->head.size()});
    uint32_t adler = stripes[0].adler;
    for (int t = 1; t < S; ++t) adler = adler_combine(adler, stripes[(size_t)t].adler, stripes[(size_t)t].raw);
    for (auto &s : stripes) out->pieces.push_back({s.sc->out.mem.get(), s.sc->out_len});
    out->tail.assign({0, 0, 0, 0, 'I', 'D', 'A', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'I', 'E', 'N', 'D', 0, 0, 0, 0});
    put_be32(&out->tail[8], adler);
    finish_chunk(&out->tail[0], 12);
    finish_chunk(&out->tail[16], 8);
    out->pieces.push_back({out->tail.data(), out->tail.size()});
    return 0;
}

// ---------------------------------------------------------------------------------------- files
int png_write_file(const PngPieces &pieces, const char *path, std::string *err) {
    auto bad = [&](const char *what) {
        if (err) *err = std::string("mic_png_write: ") + what + " " + (path ? path : "(null)") + ": " + strerror(errno);
        return -1;
    };
    if (!path) {
        if (err) *err = "mic_png_write: null path";
        return -1;
    }
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return bad("cannot open");
    std::vector<iovec> iov;
    for (const auto &p : pieces.pieces)
        if (p.size) iov.push_back(iovec{const_cast<uint8_t *>(p.data), p.size});
    size_t i = 0;
    int rc = 0;
    while (i < iov.size()) {
        const ssize_t n = writev(fd, &iov[i], (int)std::min<size_t>(iov.size() - i, 64));
        if (n < 0) {
            if (errno == EINTR) continue;
            rc = bad("writing");
            break;
        }
        size_t left = (size_t)n;
        while (i < iov.size() && left >= iov[i].iov_len) left -= iov[i++].iov_len;
        if (i < iov.size() && left) {
            iov[i].iov_base = static_cast<char *>(iov[i].iov_base) + left;
            iov[i].iov_len -= left;
        }
    }
    if (close(fd) != 0 && rc == 0) rc = bad("closing");
    return rc;
}

// ---------------------------------------------------------------------------------------- asynchronous writes
// Encoding + writing on threads of the LIBRARY's own: a Python caller that hands its saves to Python worker threads
// pays for it in GIL hand-offs on its own critical path (pipeline.run_layouts: +0.8 ms per iteration for four
// submissions); here a submission is a queue push, the workers never touch the interpreter, and the caller only
// keeps the pixel memory alive until it has waited for the job.
namespace {

struct AsyncJob {
    std::string path;
    std::vector<const uint8_t *> rows;
    int32_t w = 0, h = 0;
    int level = 1, threads = 1;
    int status = 0;
    std::string err;
    bool done = false;
};

struct AsyncPool {
    std::mutex mu;
    std::condition_variable work, finished;
    std::deque<std::shared_ptr<AsyncJob>> queue;
    std::map<int64_t, std::shared_ptr<AsyncJob>> jobs;
    std::vector<std::thread> workers;
    int64_t next_id = 1;
    pid_t pid = 0;

    void run() {
        for (;;) {
            std::shared_ptr<AsyncJob> job;
            {
                std::unique_lock<std::mutex> lock(mu);
                work.wait(lock, [&] { return !queue.empty(); });
                job = queue.front();
                queue.pop_front();
            }
            PngPieces pieces;
            std::string err;
            int rc = png_encode_rows(job->rows.data(), job->w, job->h, job->level, job->threads, &pieces, &err);
            if (rc == 0) rc = png_write_file(pieces, job->path.c_str(), &err);
            {
                std::lock_guard<std::mutex> lock(mu);
                job->status = rc;
                job->err = std::move(err);
                job->done = true;
            }
            finished.notify_all();
        }
    }
};

AsyncPool *g_async = nullptr;
std::mutex g_async_mu;

AsyncPool *async_pool() {
    std::lock_guard<std::mutex> lock(g_async_mu);
    if (g_async && g_async->pid != getpid()) g_async = nullptr;  // a forked child: the parent's threads did not come along
    if (!g_async) {                                              // (the old object is leaked on purpose: its mutexes may be held)
        g_async = new AsyncPool();
        g_async->pid = getpid();
        unsigned n = std::thread::hardware_concurrency();
        n = std::max(2u, std::min(n ? n : 4u, 8u));
        for (unsigned i = 0; i < n; ++i) {
            g_async->workers.emplace_back([p = g_async] { p->run(); });
            g_async->workers.back().detach();  // workers live as long as the process
        }
    }
    return g_async;
}

}  // namespace

int64_t png_write_async(const char *path, const uint8_t *const *rows, int32_t w, int32_t h, int level, int threads,
                        std::string *err) {
    if (!path || !rows || w <= 0 || h <= 0) {
        if (err) *err = "mic_png_write_async: bad arguments";
        return -1;
    }
    auto job = std::make_shared<AsyncJob>();
    job->path = path;
    job->rows.assign(rows, rows + h);
    job->w = w;
    job->h = h;
    job->level = level;
    job->threads = threads <= 0 ? 1 : threads;  // a job is one queue entry; its stripes may still use a few threads
    AsyncPool *pool;
    try {
        pool = async_pool();
    } catch (const std::exception &e) {
        if (err) *err = std::string("mic_png_write_async: ") + e.what();
        return -1;
    }
    int64_t id;
    {
        std::lock_guard<std::mutex> lock(pool->mu);
        id = pool->next_id++;
        pool->jobs[id] = job;
        pool->queue.push_back(job);
    }
    pool->work.notify_one();
    return id;
}

int png_wait(int64_t id, std::string *err) {
    AsyncPool *pool = async_pool();
    std::shared_ptr<AsyncJob> job;
    {
        std::unique_lock<std::mutex> lock(pool->mu);
        auto it = pool->jobs.find(id);
        if (it == pool->jobs.end()) {
            if (err) *err = "mic_png_wait: unknown job (already waited for?)";
            return -1;
        }
        job = it->second;
        pool->finished.wait(lock, [&] { return job->done; });
        pool->jobs.erase(it);
    }
    if (job->status != 0 && err) *err = job->err;
    return job->status;
}

}  // namespace mic
