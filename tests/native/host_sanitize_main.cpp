// Host-side C++ of libmic (the Flex JSON parser/placer and the resample table builders) under
// AddressSanitizer + UndefinedBehaviorSanitizer.  Built and driven by tests/test_host_sanitizers.py
// (CPU only; GPU sanitizers are not available on the pool).
//
// stdin: u32 n_cases, then per case: u32 json_len, json bytes, u32 n_obj, n_obj x (i32 id, w, h),
//        i32 W, i32 H.  Every case must return without a sanitizer report; results are summed into
//        a checksum so that the calls cannot be optimised away.
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flex_place.h"
#include "resample_coeffs.h"

static bool rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }

int main() {
    uint32_t n_cases = 0;
    if (!rd(&n_cases, 4)) return 2;
    uint64_t sum = 0;
    int ok = 0, unsupported = 0, malformed = 0;
    for (uint32_t c = 0; c < n_cases; ++c) {
        uint32_t len = 0, n_obj = 0;
        if (!rd(&len, 4)) return 2;
        std::string json(len, '\0');
        if (len && !rd(&json[0], len)) return 2;
        if (!rd(&n_obj, 4)) return 2;
        std::vector<int32_t> ids(n_obj), ws(n_obj), hs(n_obj);
        for (uint32_t i = 0; i < n_obj; ++i) {
            int32_t t[3];
            if (!rd(t, 12)) return 2;
            ids[i] = t[0]; ws[i] = t[1]; hs[i] = t[2];
        }
        int32_t WH[2];
        if (!rd(WH, 8)) return 2;
        std::vector<int32_t> oi, ob;
        std::string err;
        // exact-size heap copy: an over-read of the text is an ASan report, not a lucky NUL
        char *text = static_cast<char *>(malloc(len ? len : 1));
        if (len) memcpy(text, json.data(), len);
        const int rc = mic::flex_place(text, len, (int)n_obj, ids.data(), ws.data(), hs.data(), WH[0], WH[1], &oi, &ob, &err);
        free(text);
        if (rc == mic::kFlexOk) ++ok; else if (rc == mic::kFlexUnsupported) ++unsupported; else ++malformed;
        for (int32_t v : oi) sum += (uint32_t)v;
        for (int32_t v : ob) sum = sum * 31 + (uint32_t)v;
    }
    // resample tables: every tap must be rebuilt by its three signed-byte digits, and the fragments
    // must stay inside their buffers (ASan checks the writes)
    static const int shapes[][2] = {{1, 1}, {1, 97}, {97, 1}, {2, 3}, {3, 2}, {1000, 256}, {256, 1000}, {37, 12}, {12, 37},
                                    {4000, 40}, {40, 4000}, {65535, 17}, {17, 65535}, {500, 500}, {501, 500}, {333, 1024}};
    for (const auto &sh : shapes) {
        for (int filter = 0; filter < 2; ++filter) {
            const mic::AxisTable t = sh[0] == sh[1] ? mic::identity_axis_table(sh[0]) : mic::build_axis_table(sh[0], sh[1], filter);
            const mic::AxisFrags f = mic::build_axis_frags(t);
            for (int tile = 0; tile < f.tiles; ++tile) {
                const int ws = f.meta[4 * tile], nch = f.meta[4 * tile + 1], first_chunk = f.meta[4 * tile + 2];
                for (int o = tile * 16; o < std::min(t.out_size, tile * 16 + 16); ++o) {
                    const int first = t.bounds[2 * o], n = t.bounds[2 * o + 1];
                    long long total = 0;
                    for (int k = 0; k < n; ++k) {
                        const int pos = first + k - ws, chunk = pos / 64, lane = 16 * ((pos % 64) / 16) + (o - tile * 16), j = pos % 16;
                        if (chunk >= nch) { fprintf(stderr, "tap outside the tile's chunks\n"); return 3; }
                        const int8_t *b = &f.frags[(((size_t)first_chunk + chunk) * 3 * 64 + lane) * 16 + j];
                        const long long c = b[0] + 256LL * b[64 * 16] + 65536LL * b[2 * 64 * 16];
                        if (c != t.coeffs[(size_t)o * t.ksize + k]) { fprintf(stderr, "digits do not rebuild the tap\n"); return 3; }
                        total += c;
                    }
                    if (f.bias[o] != (int32_t)((1 << 21) + 128 * total)) { fprintf(stderr, "bias\n"); return 3; }
                    sum += (uint64_t)total;
                }
            }
        }
    }
    // the lane kernel's forms of the same tables (resample_coeffs.h: kFragsLaneH / kFragsLaneV): where an axis qualifies
    // (max_chunks == 1) every tap sits at the k position the kernel's operand layout gives its sample, nothing else is
    // non-zero, groups / emit bands / ring-word masks say what the taps say
    for (const auto &sh : shapes) {
        for (int filter = 0; filter < 2; ++filter) {
            const mic::AxisTable t = sh[0] == sh[1] ? mic::identity_axis_table(sh[0]) : mic::build_axis_table(sh[0], sh[1], filter);
            for (const int form : {(int)mic::kFragsLaneH, (int)mic::kFragsLaneV}) {
                const mic::AxisFrags f = mic::build_axis_frags(t, form);
                if ((int)f.frags.size() != f.tiles * 3072) { fprintf(stderr, "lane form: one fragment per tile\n"); return 4; }
                if (f.max_chunks != 1) continue;  // the axis does not qualify: the table is never read in this form
                for (int tile = 0; tile < f.tiles; ++tile) {
                    const int32_t *m = &f.meta[4 * tile];
                    if (m[2] != tile) { fprintf(stderr, "lane form: chunk index\n"); return 4; }
                    long long nonzero_want = 0, nonzero_got = 0;
                    int lo = 1 << 30, hi = 0;
                    for (int o = tile * 16; o < std::min(t.out_size, tile * 16 + 16); ++o) {
                        const int first = t.bounds[2 * o], n = t.bounds[2 * o + 1];
                        lo = std::min(lo, first); hi = std::max(hi, first + n);
                        long long total = 0;
                        for (int k = 0; k < n; ++k) {
                            const int r = first + k;
                            int pos;
                            if (form == mic::kFragsLaneH) {
                                pos = r - m[0];  // the GROUP's window start
                            } else {
                                pos = 16 * ((r & 15) >> 2) + 4 * ((r >> 4) & 3) + (r & 3);
                            }
                            if (pos < 0 || pos >= 64) { fprintf(stderr, "lane form: tap outside the window\n"); return 4; }
                            const int lane = 16 * (pos / 16) + (o - tile * 16), j = pos % 16;
                            const int8_t *b = &f.frags[((size_t)tile * 3 * 64 + lane) * 16 + j];
                            const long long c = b[0] + 256LL * b[64 * 16] + 65536LL * b[2 * 64 * 16];
                            if (c != t.coeffs[(size_t)o * t.ksize + k]) { fprintf(stderr, "lane form: digits do not rebuild the tap\n"); return 4; }
                            nonzero_want += (b[0] != 0) + (b[64 * 16] != 0) + (b[2 * 64 * 16] != 0);
                            total += c;
                        }
                        if (f.bias[o] != (int32_t)((1 << 21) + 128 * total)) { fprintf(stderr, "lane form: bias\n"); return 4; }
                    }
                    for (size_t k = 0; k < 3072; ++k) nonzero_got += f.frags[(size_t)tile * 3072 + k] != 0;
                    if (nonzero_got != nonzero_want) { fprintf(stderr, "lane form: stray digits\n"); return 4; }
                    if (form == mic::kFragsLaneH) {
                        if (m[0] % 16 != 0 || hi - m[0] > 64 || m[3] != hi) { fprintf(stderr, "lane form: window\n"); return 4; }
                        if (m[1] < 0 || m[1] > 2 || (m[1] == 2 && (tile + 1 >= f.tiles || f.meta[4 * (tile + 1) + 1] != 0 || f.meta[4 * (tile + 1)] != m[0])) ||
                            (m[1] == 0 && (tile == 0 || f.meta[4 * (tile - 1) + 1] != 2))) { fprintf(stderr, "lane form: groups\n"); return 4; }
                    } else {
                        int need = 0;
                        for (int b = lo >> 4; b <= (hi - 1) >> 4; ++b) need |= 1 << (b & 3);
                        if (m[0] != lo || m[3] != hi || (m[1] & 0xFFFFFF) != ((hi - 1) >> 4) || (m[1] >> 24) != need || ((hi - 1) >> 4) - (lo >> 4) > 3) {
                            fprintf(stderr, "lane form: emit band / ring words\n"); return 4; }
                    }
                    sum += (uint64_t)nonzero_got;
                }
            }
        }
    }
    printf("ok=%d unsupported=%d malformed=%d checksum=%llu\n", ok, unsupported, malformed, (unsigned long long)sum);
    return 0;
}
