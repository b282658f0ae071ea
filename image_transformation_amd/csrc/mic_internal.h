// Internal declarations shared by the host runtime (mic_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lane_unit.h"

namespace mic {

// Device addresses travel as integers inside the job tables; casting them to address space 1
// makes the compiler emit global_load/global_store instead of flat_*.
#define MIC_GLOBAL __attribute__((address_space(1)))
typedef const MIC_GLOBAL uint32_t *gcptr;
typedef MIC_GLOBAL uint32_t *gptr;
typedef const MIC_GLOBAL int32_t *gciptr;

// ---- composite ---------------------------------------------------------------------------
// The canvas is treated as a linear stream of pixels cut into 4 KiB pages ALIGNED TO ABSOLUTE
// ADDRESS; one WAVE writes exactly one page.  Measured on MI355X (scripts/streambench.hip,
// 531 MB): a store stream reaches 6.7-6.8 TB/s only when every workgroup, taken in dispatch
// order, writes one whole 4 KiB page -- workgroups are dealt round-robin over the 8 XCDs, so each
// XCD then keeps writing one fixed residue class of pages mod 8.  2-D tiles (256 px x 4 rows per
// wave: 5.5 TB/s), fatter workgroups (8-16 KiB: 6.0), XCD-contiguous ranges (5.0) and grid-stride
// persistent loops (4.4) all lose; page offset and read-side placement/alignment do not matter.
// The composite kernel uses one wavefront per page (16 px per lane), four such waves per workgroup, so that the per-page work
// (one division, layer culling, record broadcast) is amortised over 4 KiB.
constexpr int kLaneNPx = 4;                    // adjacent pixels per lane per group (16 B)
constexpr int kWavePx = 64 * kLaneNPx;         // 256 px = 1 KiB per wave-wide access
constexpr int kGroups = 4;                     // groups per lane, 1 KiB apart
constexpr int kPagePx = kWavePx * kGroups;     // 1024 px = 4 KiB per wave
constexpr int kPagesPerWorkgroup = 4;          // waves (= consecutive pages) per composite workgroup

// One resolved placement: where the layer's pixels live and where they land on the canvas.
struct alignas(16) Layer {
    uint64_t src;      // device address of w*h RGBA pixels, row stride = w
    int32_t dx, dy;    // canvas position of the layer's top-left pixel (may be negative)
    int32_t w, h;      // layer size (box size after the max(1, .) rule)
    int32_t pad0, pad1;
};
static_assert(sizeof(Layer) == 32, "Layer layout");

// One canvas.
struct alignas(16) Job {
    uint64_t out;      // device address of the W*H canvas
    uint64_t bg;       // device address of a W*H background image, 0 = solid colour; flags & kJobColourWord: of ONE colour word
    uint32_t bg_rgba;  // solid colour, little-endian r | g<<8 | b<<16 | a<<24
    int32_t W, H;
    int32_t layer_begin, layer_count;
    int32_t px_shift;  // pixels of the canvas' first 4 KiB page that precede the canvas: (out % 4096) / 4
    int32_t n_pages;   // ceil((W*H + px_shift) / 1024)
    int32_t flags;     // kJobColourWord: the solid colour is read from device memory at `bg` (mic_job.bg_rgba_dev)
};
constexpr int32_t kJobColourWord = 1;
// set by launch_composite on the kernel-argument copy of a single job with at most kSmallCanvasPages pages: the launch
// uses ONE-wave workgroups (page = workgroup), not kPagesPerWorkgroup-wave ones.  Such a launch is one partial generation of
// waves; a quarter as many workgroups stops filling the 256 CUs evenly (the 1024 x 328 contact sheet: 88 four-wave
// workgroups 7.15 - 7.27 us, 336 one-wave ones 6.8 us; 1080p 6.33 against 6.14 - 6.25: profiles/r05_composite_pages_per_workgroup.txt).
constexpr int32_t kJobOnePagePerWorkgroup = 2;
constexpr int kSmallCanvasPages = 2048;
static_assert(sizeof(Job) == 48, "Job layout");

// ---- resample ------------------------------------------------------------------------------
constexpr int kPrecisionBits = 22;           // Pillow Resample.c: 32 - 8 - 2

enum RsFlags : uint32_t {
    kRsPremultiplyOnLoad = 1,   // input is straight RGBA: premultiply each tap (Convert.c rgbA2rgba)
    kRsUnpremultiplyOnStore = 2 // output is the final image: RGBa -> RGBA (Convert.c rgba2rgbA)
};

// One axis pass of one layer.  Horizontal: out(x', y) = sum_k in(xmin[x'] + k, y) * K[x'][k];
// vertical: the same along y.  bounds/coeffs live in the context's coefficient arena.
struct alignas(16) RsJob {
    uint64_t src, dst;
    uint64_t bounds;   // int32 [out_len][2] = (first tap, tap count)
    uint64_t coeffs;   // int32 [out_len][ksize]
    int32_t in_w, in_h;
    int32_t out_w, out_h;
    int32_t ksize;
    uint32_t flags;
    int32_t pad0, pad1;
};
static_assert(sizeof(RsJob) == 64, "RsJob layout");

// One layer resized by the marching MFMA kernel (kernels_resample.hip): both axes in one launch.  Axis
// tables are the fragment form of resample_coeffs.h (AxisFrags); an axis that keeps its size gets the
// identity table (one tap of weight 1.0: the pass Pillow skips).  Work units of a layer = `strips`
// column strips (4 tiles of 16 output columns, one per wave) x `segs` segments of `seg_tiles` tiles of
// 16 output rows; a workgroup marches one unit down the source in bands of 16 rows.
struct alignas(16) RsMarch {
    uint64_t src, dst;             // src: the cutout's planar premultiplied copy (4 planes of sh rows x planar_pitch)
    uint64_t hmeta, hbias, hfrag;  // horizontal axis: [xtiles][4] int32, [16 xtiles] int32, fragments
    uint64_t vmeta, vbias, vfrag;  // vertical axis
    int32_t sw, sh, dw, dh;
    int32_t planar_pitch;          // bytes per row of a source plane (a multiple of 16)
    int32_t tiles_x, tiles_y;      // 16-sample tiles per axis
    int32_t strips, segs, seg_tiles;
    int32_t pitch_c;               // LDS bytes per row of a source band plane: what a strip's tiles can touch
    int32_t ring16, pitch_r;       // ring of intermediate rows: 16-row slots (a power of two), bytes per column (64 ring16 + 16)
    // A layer with more than kRsUnitsPerEntry units takes several table entries (grid.y); workgroup
    // bx of entry e works on unit ((bx + xcd_rot) & 7) * 4 n_entries + 4 e + (bx >> 3): workgroups
    // are dealt round-robin over the 8 XCDs, so each XCD gets a contiguous run of the layer's units
    // (x-fastest) and neighbouring strips (which share source columns and vertical taps) share an L2.
    int32_t entry, n_entries, xcd_rot;
};
static_assert(sizeof(RsMarch) == 128, "RsMarch layout");
constexpr int kRsMaxSegTiles = 64;                 // tiles of 16 output rows per unit, at most
inline size_t rs_march_lds_bytes(int pitch_c, int pitch_r) {
    // source band: 4 planes x 16 rows x pitch_c; ring: 64 columns x pitch_r (ring16 slots x 4 channels x 16 rows + 16)
    return (size_t)4 * 16 * pitch_c + (size_t)64 * pitch_r + 64;  // + slack for chunk over-reads
}
constexpr size_t kRsMarchMaxLds = 150 * 1024;      // last resort before the two-pass fallback
constexpr int kRsUnitsPerEntry = 32;
#ifndef MIC_RS_WAVES
#define MIC_RS_WAVES 5  // waves per SIMD the marching kernel's register budget is set for
#endif

// One layer resized by the tile kernel (kernels_resample_tile.hip: single images, deep shrinks, small calls): both axes in one launch, source planes and the 8-bit
// intermediate in LDS.  Axis tables are the fragment form of resample_coeffs.h (AxisFrags); an axis
// that keeps its size gets the identity table (one tap of weight 1.0: the pass Pillow skips).
struct alignas(16) RsTile {
    uint64_t src, dst;
    uint64_t hmeta, hbias, hfrag;  // horizontal axis: [xtiles][4] int32, [16 xtiles] int32, fragments
    uint64_t vmeta, vbias, vfrag;  // vertical axis
    int32_t sw, sh, dw, dh;
    int32_t tx16, ty16;            // workgroup tile in units of 16 output samples
    int32_t tiles_x, tiles_y;
    int32_t pitch_c, pitch_r;      // bytes per row of a source plane / per column of an intermediate plane
    int32_t rows16;                // rows of a source plane (multiple of 16): the whole window of a tile, or
                                   // one band of it when the window is too tall for LDS (deep shrinks)
    // A layer with more than kRsTilesPerEntry tiles takes several table entries (grid.y); workgroup
    // bx of entry e works on tile ((bx + xcd_rot) & 7) * 4 n_entries + 4 e + (bx >> 3): workgroups
    // are dealt round-robin over the 8 XCDs, so each XCD gets a contiguous band of the layer's tiles
    // and neighbouring tiles (which share their source halo) share an L2.
    int32_t entry, n_entries, xcd_rot;
    // > 0: `src` is the atlas' planar premultiplied copy of the cutout (four planes of sh rows, this many
    // bytes per row); 0: `src` is interleaved RGBA and is premultiplied while it is loaded.
    int32_t planar_pitch;
    int32_t pad;
};
static_assert(sizeof(RsTile) == 128, "RsTile layout");
inline size_t rs_tile_lds_bytes(int rows16, int pitch_c, int tx16, int pitch_r) {
    return 4 * ((size_t)rows16 * pitch_c + (size_t)16 * tx16 * pitch_r) + 64;  // + slack for chunk over-reads
}
constexpr size_t kRsTilePreferredLds = 52 * 1024;  // three workgroups per CU (160 KB of LDS, 1 KB static each)
constexpr size_t kRsTileMaxLds = 150 * 1024;       // last resort before the two-pass fallback
// Up to this many whole-window entries ride in the kernel arguments (2 KiB of the 4 KiB segment)
constexpr int kRsTileArgJobs = 16;
struct RsTileArgs {
    RsTile t[kRsTileArgJobs];
};
inline bool rs_tile_in_args(int n_jobs, int n_whole) { return n_jobs > 0 && n_jobs == n_whole && n_jobs <= kRsTileArgJobs; }
// The launch is a (tiles per entry) x (entries) grid: a layer with more tiles takes several entries,
// so that small layers do not pad the grid out to the largest layer's tile count.
constexpr int kRsTilesPerEntry = 32;

// ---- launchers (defined next to their kernels) -----------------------------------------------
// The job table is sorted by kernel class; class_end[c] = one past the last job of class c for
// c = 0 (aligned + solid opaque background), 1 (unaligned + solid), 2 (aligned + other background);
// the rest is class 3.  pitch = max PAGES per job rounded up to 8 * kPagesPerWorkgroup; every launch (batch or
// single job) dispatches pitch / kPagesPerWorkgroup workgroups per job, and *launched_workgroups says how many.
// single != nullptr (n_jobs == 1): the job rides in the kernel arguments; single_layers_host != nullptr and at most
// kPackLayers layers: so do its layer records (host array, indexed by the job's layer_begin) -- nothing is read from
// jobs_dev / layers_dev then.
constexpr int kPackLayers = 64;
hipError_t launch_composite(const Job *jobs_dev, const Layer *layers_dev, int n_jobs, const int class_end[3],
                            int pitch, const Job *single, const Layer *single_layers_host, hipStream_t stream,
                            uint64_t *launched_workgroups);
hipError_t launch_resample_h(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_rows,
                             hipStream_t stream);
hipError_t launch_resample_v(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_out_h,
                             hipStream_t stream);
// One cutout of an atlas to premultiply + planarise (kernels_resample.hip: planarize_kernel).
struct alignas(16) PlanarJob {
    uint64_t src;  // interleaved RGBA, w x h
    uint64_t dst;  // 4 planes of h rows x pitch bytes
    int32_t w, h, pitch, pad;
};
static_assert(sizeof(PlanarJob) == 32, "PlanarJob layout");
hipError_t launch_planarize(const PlanarJob *jobs_dev, int n_jobs, int64_t max_items, hipStream_t stream);
hipError_t launch_resample_march(const RsMarch *jobs_dev, int n_jobs, size_t lds_bytes, hipStream_t stream);
hipError_t launch_resample_tile(const RsTile *jobs_dev, int n_jobs, int n_whole, size_t lds_bytes, hipStream_t stream,
                                const RsTile *jobs_host = nullptr);
#ifndef MIC_RS_LANE_WAVES
#define MIC_RS_LANE_WAVES 4  // waves per SIMD the lane kernel's register budget is set for
#endif
hipError_t launch_resample_lane(const RsLaneUnit *units_dev, int n_slots, hipStream_t stream);
// PlanarJob with pitch = 16 * (tiles per band) and dst = 4 planes of ceil(h / 16) bands of tiles
hipError_t launch_planarize_tiled(const PlanarJob *jobs_dev, int n_jobs, int64_t max_items, hipStream_t stream);
// Known-answer canary of the clip / pack / (un)premultiply helpers (kernels_resample.hip): 0 mismatches expected.
hipError_t run_selftest_clip(hipStream_t stream, int *mismatches, int *first_bad);
hipError_t launch_fill(void *out, uint32_t rgba, size_t n_px, hipStream_t stream);
// table_dev: scratch for max(W, H) <= 65535 colours (kGradientTableWords uint32)
hipError_t launch_gradient(void *out, int W, int H, const uint8_t c1[3], const uint8_t c2[3], int vertical,
                           uint32_t *table_dev, hipStream_t stream);
constexpr size_t kGradientTableWords = 65536;
// One rectangle outline of the debug overlay (kernels_overlay.hip), resolved on the host.
struct alignas(16) OutlineRect {
    int32_t x0, y0, x1, y1;  // ImageDraw.rectangle's box, inclusive
    int32_t vlo, vhi;        // rows covered by the two vertical lines (vlo > vhi: none)
    int32_t ymin, ymax;      // rows the whole outline can touch
    uint32_t rgba;
    int32_t pad[3];
};
static_assert(sizeof(OutlineRect) == 48, "OutlineRect layout");
hipError_t launch_rect_outlines(void *out, int W, int H, const OutlineRect *rects_dev, int n, int width,
                                hipStream_t stream);
// Median colour of up to kMedianMaxBatch images in one launch (kernels_median.hip).  Per image a scratch slot of
// 8 copies of { uint32 [2][3][256] + counts[2] (+ padding) } + ticket; the context holds TWO sets of slots (a double
// buffer, kMedianScratchWords in all, zeroed once by mic_create): a call works in one half and clears, in the shadow
// of its first loads, what the previous call left in the other.
constexpr int kMedianMaxBatch = 16;
constexpr size_t kMedianSlotWords = 8 * (2 * 3 * 256 + 16) + 8;
static_assert(kMedianSlotWords % 4 == 0, "the median kernel clears the scratch 16 bytes at a time");
constexpr size_t kMedianScratchWords = 2 * kMedianMaxBatch * kMedianSlotWords;
struct MedianView {
    const void *px;     // device RGBA
    int32_t w, h;
    int32_t stride_px;  // pixels between rows (== w: a packed image)
};
struct MedianState {
    uint32_t phase = 0;       // half of the double buffer the next call works in
    uint32_t prev_words = 0;  // words of the other half the previous call used
};
hipError_t launch_median_batch(int k, const MedianView *views, uint32_t *const *out_rgba_dev, uint32_t *scratch_dev,
                               MedianState *state, int two_launches /* 1, 0, or -1 = by size */, hipStream_t stream);

}  // namespace mic
