// Host-side PNG writer of libmic.so (png_encode.cpp): RGBA8 rows -> the pieces of a PNG file.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace mic {

// Upper bound of the encoded size of a w x h RGBA image (any level).
size_t png_bound(int32_t w, int32_t h);

// The encoded file as a list of byte ranges (signature + IHDR, one IDAT chunk per stripe, Adler trailer + IEND) that
// stay valid while the object lives: the stripes' chunks sit in pooled worker buffers that go back to the pool with it.
struct PngPieces {
    struct Piece { const uint8_t *data; size_t size; };
    std::vector<Piece> pieces;
    std::vector<uint8_t> head, tail;
    std::vector<void *> held;
    PngPieces() = default;
    PngPieces(const PngPieces &) = delete;
    PngPieces &operator=(const PngPieces &) = delete;
    ~PngPieces();
    size_t total() const {
        size_t n = 0;
        for (const Piece &p : pieces) n += p.size;
        return n;
    }
};

// Encode rows[0..h) (each w*4 bytes of RGBA) as an 8-bit RGBA, non-interlaced PNG.  level 0: stored deflate blocks
// (no compression), >= 1: LZ77 + Huffman.  threads <= 0: one worker per ~384 KiB of pixels, at most min(cores, 16).
// 0 on success, negative on failure (*err says why).
int png_encode_rows(const uint8_t *const *rows, int32_t w, int32_t h, int level, int threads, PngPieces *out,
                    std::string *err);

// The pieces -> a file (one writev).  0, or -1 with *err.
int png_write_file(const PngPieces &pieces, const char *path, std::string *err);

// Encode + write on the library's own worker threads.  Returns a job id (> 0) at once, or -1 with *err; the row
// pointers are copied, the PIXELS must stay valid until png_wait(id) has returned.  png_wait blocks until the job is
// done and returns its status (0, or negative with *err); an id is waited for exactly once.
int64_t png_write_async(const char *path, const uint8_t *const *rows, int32_t w, int32_t h, int level, int threads,
                        std::string *err);
int png_wait(int64_t id, std::string *err);

}  // namespace mic
