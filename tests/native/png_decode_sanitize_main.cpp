// libmic's PNG reader (csrc/png_decode.cpp) under AddressSanitizer / UBSan.  stdin: u32 count, then per case
// { u32 length, bytes, u64 expected FNV-1a of the RGBA pixels (0 = "whatever: a mutated file, only no crash") }.
// Every file goes through the header parse and the decoder twice: with the CRC-32 / Adler-32 verification (the library's
// setting) and without (so that mutated bytes reach the inflater and the unfilter instead of stopping at a checksum).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "png_decode.h"

static bool rd(void *p, size_t n) { return fread(p, 1, n, stdin) == n; }

int main() {
    uint32_t count = 0;
    if (!rd(&count, 4)) return 2;
    int ok = 0, declined = 0, wrong = 0;
    for (uint32_t c = 0; c < count; ++c) {
        uint32_t len = 0;
        if (!rd(&len, 4)) return 2;
        std::vector<uint8_t> d(len);
        if (len && !rd(d.data(), len)) return 2;
        uint64_t want = 0;
        if (!rd(&want, 8)) return 2;
        // exact-size heap copy: an over-read of the input is an ASan report
        std::vector<uint8_t> exact(d.begin(), d.end());
        int32_t w = 0, h = 0;
        std::string err;
        const int irc = mic::png_decode_info(exact.data(), exact.size(), &w, &h, &err);
        for (int verify = 1; verify >= 0; --verify) {
            int32_t ww = w, hh = h;
            if (irc != 0) {
                if (verify) { ++declined; continue; }
                // the header parse without checksums may still accept it: find the size in the IHDR bytes
                if (exact.size() < 33) continue;
                ww = (int32_t)((exact[16] << 24) | (exact[17] << 16) | (exact[18] << 8) | exact[19]);
                hh = (int32_t)((exact[20] << 24) | (exact[21] << 16) | (exact[22] << 8) | exact[23]);
                if (ww <= 0 || hh <= 0 || ww > 4096 || hh > 4096) continue;
            }
            if ((int64_t)ww * hh > (1 << 24)) continue;
            std::vector<uint8_t> px((size_t)ww * hh * 4);
            std::vector<uint8_t *> rows((size_t)hh);
            for (int y = 0; y < hh; ++y) rows[(size_t)y] = px.data() + (size_t)y * ww * 4;
            const int rc = mic::png_decode_rows(exact.data(), exact.size(), rows.data(), ww, hh, verify != 0, &err);
            if (rc == 0 && verify) {
                uint64_t hsh = 1469598103934665603ull;
                for (uint8_t b : px) hsh = (hsh ^ b) * 1099511628211ull;
                if (want != 0 && hsh != want) ++wrong;
                ++ok;
            } else if (verify) {
                ++declined;
                if (want != 0) ++wrong;  // an intact file of a supported kind must decode
            }
        }
    }
    printf("ok=%d declined=%d wrong=%d\n", ok, declined, wrong);
    return wrong ? 1 : 0;
}
