// Host-side tables for one resample axis (see resample_coeffs.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mic {

struct AxisTable {
    int out_size = 0;
    int ksize = 0;                 // taps allotted per output sample
    std::vector<int32_t> bounds;   // [out_size][2]: first input index, tap count
    std::vector<int32_t> coeffs;   // [out_size][ksize], 22-bit fixed point, unused taps zero
};

AxisTable build_axis_table(int in_size, int out_size, int filter);
AxisTable identity_axis_table(int size);  // one tap of weight 1.0 per sample: the pass Pillow skips

// The same axis laid out for the MFMA resample kernel (kernels_resample.hip): output samples in
// tiles of 16, each tile's taps as i8 operand fragments of v_mfma_i32_16x16x64_i8.
//   meta  [tiles][4]   : window start (first tap of the tile, rounded down to 16), 64-tap chunks,
//                        index of the tile's first chunk in `frags`, one past the tile's last tap
//   bias  [tiles * 16] : 2^21 (Pillow's rounding) + 128 * sum of the sample's taps (samples are
//                        stored as s - 128 so that they are signed bytes)
//   frags [chunks][3][64][16] bytes: digit d (c = d0 + 256 d1 + 65536 d2, each digit a signed byte)
//                        of the tap that output sample 16 t + (lane & 15) applies to window
//                        position 64 chunk + 16 (lane >> 4) + j; zero outside the sample's taps.
struct AxisFrags {
    int tiles = 0;
    int max_chunks = 0;  // most chunks any tile has
    std::vector<int32_t> meta;
    std::vector<int32_t> bias;
    std::vector<int8_t> frags;
};
// The same in two steps, for callers that place the fragments themselves (several axes into one upload buffer):
// axis_frags_layout fills tiles / max_chunks / meta and returns the number of 64-tap chunks (frags = chunks * 3072 bytes);
// fill_axis_frags writes bias[tiles * 16] and frags[chunks * 3072] (which it zeroes first) for that layout.
// `form` selects what the fragments are laid out for:
//   kFragsTile   the marching / tile kernels: every tile against its own window (start = first tap rounded down to 16),
//                as many 64-tap chunks as the window needs;
//   kFragsLaneH  the lane kernel's horizontal axis (kernels_resample_lane.hip): tiles are GROUPED, up to two adjacent
//                tiles whose taps fit one 64-column window share it (the window loads of a band serve both);
//                meta = {window start of the tile's group, tiles in the group (0 on a group's second tile), chunk, end};
//                a tile whose own window exceeds 64 columns takes two chunks, which disqualifies the axis (max_chunks);
//   kFragsLaneV  the lane kernel's vertical axis: a tile's window is the FOUR 16-row bands that end with the band of
//                its last tap row, and window row r sits at k position 16 ((r & 15) >> 2) + 4 ((r >> 4) & 3) + (r & 3) --
//                band b lives in word b & 3 of the lane's ring of intermediate rows, rows 4 q .. 4 q + 3 of a band in
//                lane quarter q; meta = {first tap row, last band | ring words read << 24, chunk, end}; a tile whose
//                taps span more than four bands disqualifies the axis (max_chunks = 2).
enum : int { kFragsTile = 0, kFragsLaneH = 2, kFragsLaneV = 3 };
AxisFrags build_axis_frags(const AxisTable &t, int form = kFragsTile);
size_t axis_frags_layout(const AxisTable &t, AxisFrags *layout, int form = kFragsTile);
void fill_axis_frags(const AxisTable &t, const AxisFrags &layout, int32_t *bias, int8_t *frags, size_t chunks, int form = kFragsTile);
std::vector<int32_t> transpose_coeffs(const AxisTable &t);  // -> [ksize][out_size]
void thumbnail_size(int w, int h, int req_w, int req_h, int *out_w, int *out_h);

}  // namespace mic
