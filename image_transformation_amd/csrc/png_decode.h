// Host-side PNG decoder of libmic.so (png_decode.cpp): a PNG file in memory -> RGBA8 rows, exactly what
// Image.open(f).convert("RGBA") holds, for the kinds of file the compositor path meets; everything else is declined.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace mic {

constexpr int kPngMalformed = -5;    // not a PNG / broken (Pillow would raise): MIC_ERR_FORMAT
constexpr int kPngUnsupported = -6;  // a valid PNG this decoder leaves to the caller (16-bit, interlaced, ...): MIC_ERR_UNSUPPORTED
constexpr int kPngNoMem = -3;

// Size of the image (and whether this decoder takes the file at all: same status codes as the decode).
int png_decode_info(const uint8_t *data, size_t n, int32_t *w, int32_t *h, std::string *err);
// Decode into rows[0..h) (w * 4 bytes each).  verify = false skips the CRC-32 / Adler-32 checks (the sanitizer
// harness feeds mutated files through the parser and the inflater that way; the library always verifies).
int png_decode_rows(const uint8_t *data, size_t n, uint8_t *const *rows, int32_t w, int32_t h, bool verify, std::string *err);
// n files at once on up to `threads` worker threads (<= 0: min(n, 8)); status[i] per file; returns the first failure.
int png_decode_many(int n, const uint8_t *const *datas, const size_t *sizes, uint8_t *const *const *rows, const int32_t *ws,
                    const int32_t *hs, int threads, int *status, std::string *err);

}  // namespace mic
