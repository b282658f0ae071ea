// The lane resample kernel's work record (kernels_resample_lane.hip) -- plain data, no HIP types: shared by the device code
// (mic_internal.h), the host-side cut (lane_partition.h) and its native sanitizer harness (tests/native).
#pragma once
#include <cstdint>

namespace mic {

// One WAVE's work in the lane kernel (kernels_resample_lane.hip): T adjacent tiles of 16 output columns whose taps fit
// one 64-column window of the cutout's TILED planar copy, marched from band band0 to band_last, emitting n_vtiles tiles
// of 16 output rows.  Everything a wave needs first is in the record: scalar loads, then one round of vector loads.
// Record s (s < slots) is the first piece of wave slot s (n_vtiles == 0: the slot has nothing to do); the few slots
// whose equal-cost chunk falls across the end of a strip chain further pieces through `next` (records >= slots).
struct alignas(16) RsLaneUnit {
    uint64_t src;        // plane 0 of the tiled planar copy, at tile (band0, first window tile)
    uint64_t dst;        // the layer's pixels (row-major RGBA, dw x dh)
    uint64_t hfrag;      // [T][3][64][16]: horizontal tap digits of the unit's x-tiles against ITS window
    uint64_t hbias;      // [T][16] int32
    uint64_t vfrag;      // [n_vtiles][3][64][16]: vertical tap digits in ring order (band b at k bytes 4 (b & 3) .. + 3)
    uint64_t vbias;      // [n_vtiles][16] int32
    uint64_t vemit;      // [n_vtiles] int32: band after which the tile can be emitted | ring words it reads << 24
    uint32_t plane_bytes, band_bytes;  // bytes between planes / between bands of tiles
    int32_t band0, band_last;
    int32_t n_vtiles, T;
    int32_t x0, row0;    // first output column / row of the unit
    int32_t dw, dh;
    uint32_t next;       // index of the wave's next piece (0: none) -- records [0, slots) are the slots' FIRST pieces
    int32_t cls;         // kLaneGeneral, or the layer keeps its width / height (below)
    int32_t pad[6];
};
static_assert(sizeof(RsLaneUnit) == 128, "RsLaneUnit layout");
// Piece classes: a layer that keeps one axis (the pass Pillow skips, Resample.c need_horizontal / need_vertical; call site
// compositor.py:20) runs that axis as ONE MFMA per channel and no floor shifts -- every tap is 2^22 or 0, so the two low
// digits are zero and the chain collapses to its last link; what is left is the transposition between the two passes'
// operand layouts.  A wave-uniform choice per piece inside one launch.
enum : int32_t { kLaneGeneral = 0, kLaneKeepsWidth = 1, kLaneKeepsHeight = 2 };
}  // namespace mic
