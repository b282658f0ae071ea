// libmic's PNG encoder (csrc/png_encode.cpp) under AddressSanitizer / UBSan: every image kind and size class the
// bound computation (deflate_bound), the unchecked bit writer and the stripe logic have to survive -- incompressible
// noise (stored fallback), flat images (run fast path), periodic patterns (long-distance matches), one-pixel and
// one-row / one-column shapes, widths around the SSE chunking, every level / thread count.  Prints the FNV-1a of
// all encoded bytes so that a second run can be compared.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "png_encode.h"

static uint64_t g_hash = 1469598103934665603ull;
static void mix(const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; ++i) g_hash = (g_hash ^ p[i]) * 1099511628211ull;
}

int main(int argc, char **argv) {
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    const int sizes[][2] = {{1, 1}, {1, 777}, {999, 1}, {3, 3}, {4, 5}, {5, 4}, {15, 9}, {16, 9}, {17, 9}, {63, 7}, {64, 7}, {65, 7},
                            {131, 97}, {492, 492}, {1025, 300}, {37, 2500}, {4099, 33}, {16385, 3}};
    int n_ok = 0;
    for (const auto &wh : sizes) {
        const int w = wh[0], h = wh[1];
        for (int kind = 0; kind < 6; ++kind) {
            std::vector<uint8_t> img((size_t)w * h * 4);
            for (size_t i = 0; i < img.size(); ++i) {
                const size_t px = i / 4, x = px % w, y = px / w;
                switch (kind) {
                    case 0: img[i] = (uint8_t)rnd(); break;                                   // noise
                    case 1: img[i] = (uint8_t)(0x11 * (i % 4 + 1)); break;                    // flat
                    case 2: img[i] = (uint8_t)((x * 3 + y * 5 + i % 4) & 255); break;         // ramps
                    case 3: img[i] = (uint8_t)(((x & 7) * 31 + (y & 7) * 17 + i % 4) & 255); break;  // 8 x 8 tiles
                    case 4: img[i] = (y % 3 == 0) ? 255 : 0; break;                           // stripes
                    default: img[i] = (x > (size_t)w / 2) ? (uint8_t)rnd() : 7; break;        // half flat, half noise
                }
            }
            std::vector<const uint8_t *> rows((size_t)h);
            for (int y = 0; y < h; ++y) rows[(size_t)y] = img.data() + (size_t)y * w * 4;
            for (int level = 0; level <= 1; ++level)
                for (int threads : {1, 2, 3, 0}) {
                    mic::PngPieces out;
                    std::string err;
                    if (mic::png_encode_rows(rows.data(), w, h, level, threads, &out, &err) != 0) {
                        printf("encode failed: %s\n", err.c_str());
                        return 1;
                    }
                    if (out.total() > mic::png_bound(w, h)) {
                        printf("bound exceeded %dx%d kind %d\n", w, h, kind);
                        return 1;
                    }
                    for (const auto &p : out.pieces) mix(p.data, p.size);
                    ++n_ok;
                }
        }
    }
    // big incompressible images on ONE thread (a single stripe of 67 MB: ~1000 deflate blocks that all fall
    // back to stored) -- the total must stay inside png_bound and the stripe inside deflate_bound; the second pass
    // reuses the pooled scratch of the first (this binary is built with -DMIC_PNG_EXACT_ALLOC: buffers hold exactly
    // what was asked for, so a bound that is too small is an ASan report, not silent slack)
    {
        const int w = 4096, h = 4096;
        std::vector<uint8_t> img((size_t)w * h * 4);
        for (auto &b : img) b = (uint8_t)rnd();
        std::vector<const uint8_t *> rows((size_t)h);
        for (int y = 0; y < h; ++y) rows[(size_t)y] = img.data() + (size_t)y * w * 4;
        for (int pass = 0; pass < 2; ++pass)
            for (int level = 1; level >= 0; --level) {
                mic::PngPieces out;
                std::string err;
                if (mic::png_encode_rows(rows.data(), w, h, level, 1, &out, &err) != 0) { printf("encode failed: %s\n", err.c_str()); return 1; }
                if (out.total() > mic::png_bound(w, h)) {
                    printf("bound exceeded 4096x4096 noise level %d: %zu > %zu\n", level, out.total(), mic::png_bound(w, h));
                    return 1;
                }
                if (pass == 0) for (const auto &p : out.pieces) mix(p.data, p.size);
                ++n_ok;
            }
    }
    // the asynchronous writer: a burst of jobs on the library's worker threads, files into argv[1]
    if (argc > 1) {
        const int w = 300, h = 200;
        std::vector<std::vector<uint8_t>> imgs(24, std::vector<uint8_t>((size_t)w * h * 4));
        std::vector<std::vector<const uint8_t *>> tables(imgs.size());
        std::vector<int64_t> ids;
        for (size_t k = 0; k < imgs.size(); ++k) {
            for (auto &b : imgs[k]) b = (k % 3 == 0) ? (uint8_t)rnd() : (uint8_t)(k * 7);
            for (int y = 0; y < h; ++y) tables[k].push_back(imgs[k].data() + (size_t)y * w * 4);
            std::string err;
            const std::string path = std::string(argv[1]) + "/a" + std::to_string(k) + ".png";
            const int64_t id = mic::png_write_async(path.c_str(), tables[k].data(), w, h, 1, 1 + (int)(k % 3), &err);
            if (id <= 0) { printf("async submit failed: %s\n", err.c_str()); return 1; }
            ids.push_back(id);
        }
        for (int64_t id : ids) {
            std::string err;
            if (mic::png_wait(id, &err) != 0) { printf("async job failed: %s\n", err.c_str()); return 1; }
        }
        std::string err;
        if (mic::png_wait(ids[0], &err) == 0) { printf("a job was waited for twice\n"); return 1; }
        n_ok += (int)ids.size();
    }
    printf("ok=%d hash=%016llx\n", n_ok, (unsigned long long)g_hash);
    return 0;
}
