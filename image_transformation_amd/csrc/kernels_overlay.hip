// Debug overlay: replaces _save_overlay_debug's drawing (macro_placement_test.py:967-983) --
// ImageDraw.rectangle(box, outline=colour, width) per placement, in list order, on a transparent
// RGBA image.  ImageDraw stores the ink on an RGBA image (no blending), so the result per pixel is
// the ink of the LAST rectangle whose outline covers it: a pure function of (x, y), evaluated in one
// pass that writes every pixel once (4 B/px).
//
// Coverage is the closed form of what Pillow's ImagingDrawRectangle draws (probed against the
// installed Pillow, pinned by tests/golden/overlay.npz through the oracle's line-by-line version):
//   horizontal bands  x0 <= x <= x1 and (y0 <= y < y0 + w  or  y1 - w < y <= y1)
//   vertical bands    (x0 <= x < x0 + w  or  x1 - w < x <= x1) and vlo <= y <= vhi
// where [vlo, vhi] is the run of the two vertical lines: they start at y0 + w and take |dy| steps
// towards y1 - w + 1 without reaching it, so for boxes thinner than 2 w the run points upwards and
// leaves the box (the host precomputes vlo/vhi and the outline's overall row range for culling).
#include "mic_internal.h"

namespace mic {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Does outline R reach the linear pixel run [first, last] of one wave?  Rows first; when the run lies in
// a single row (the usual case on a wide canvas) also columns: an outline only reaches it through its
// horizontal bands (any column of the box) or through its two vertical lines (w columns at either
// side) -- the interior of a box is skipped.
__device__ __forceinline__ bool outline_hits_run(const OutlineRect &R, int width, int row_first, int row_last,
                                                 int col_first, int col_last) {
    if (R.ymax < row_first || R.ymin > row_last) return false;
    if (row_first != row_last) return true;
    const int y = row_first;
    const bool band = (y >= R.y0 && y < R.y0 + width) || (y > R.y1 - width && y <= R.y1);
    const bool run = y >= R.vlo && y <= R.vhi;
    return (band && R.x0 <= col_last && R.x1 >= col_first) ||
           (run && ((R.x0 <= col_last && R.x0 + width > col_first) || (R.x1 - width < col_last && R.x1 >= col_first)));
}

// One wave = 256 consecutive pixels (4 per lane).  Culling happens inside the wave, 64 outlines at a
// time, like the composite kernel's: lane l tests outline base + l against the wave's run, a __ballot
// gives the ordered hit mask, and only the hits are broadcast (v_readlane) and evaluated per pixel.
__global__ __launch_bounds__(256) void rect_outline_kernel(uint32_t *__restrict__ out, int W, int H,
                                                           const OutlineRect *__restrict__ rects, int n, int width) {
    const int lane = threadIdx.x & 63;
    const int64_t n_px = (int64_t)W * H;
    const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kLaneNPx;
    const int64_t w0 = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63)) * kLaneNPx;  // the wave's first pixel
    if (w0 >= n_px) return;                                                             // wave-uniform
    const int64_t w1 = min(w0 + kWavePx - 1, n_px - 1);
    const int row_first = (int)(w0 / W), row_last = (int)(w1 / W);
    const int col_first = (int)(w0 - (int64_t)row_first * W), col_last = (int)(w1 - (int64_t)row_last * W);
    int y = (int)(min(q0, n_px - 1) / W), x = (int)(min(q0, n_px - 1) - (int64_t)y * W);
    int xs[kLaneNPx], ys[kLaneNPx];
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) {
        xs[j] = x; ys[j] = y;
        if (++x == W) { x = 0; ++y; }
    }
    uint32_t px[kLaneNPx] = {0u, 0u, 0u, 0u};
    for (int base = 0; base < n; base += 64) {
        OutlineRect mine{};
        bool hit = false;
        if (base + lane < n) {
            mine = rects[base + lane];
            hit = outline_hits_run(mine, width, row_first, row_last, col_first, col_last);
        }
        uint64_t m = __ballot(hit);
        while (m != 0) {  // ascending index = list order: a later outline overwrites an earlier one
            const int i = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int x0 = __builtin_amdgcn_readlane(mine.x0, i), y0 = __builtin_amdgcn_readlane(mine.y0, i);
            const int x1 = __builtin_amdgcn_readlane(mine.x1, i), y1 = __builtin_amdgcn_readlane(mine.y1, i);
            const int vlo = __builtin_amdgcn_readlane(mine.vlo, i), vhi = __builtin_amdgcn_readlane(mine.vhi, i);
            const uint32_t ink = (uint32_t)__builtin_amdgcn_readlane((int)mine.rgba, i);
#pragma unroll
            for (int j = 0; j < kLaneNPx; ++j) {
                const int xx = xs[j], yy = ys[j];
                const bool in_x = xx >= x0 && xx <= x1;
                const bool band_y = (yy >= y0 && yy < y0 + width) || (yy > y1 - width && yy <= y1);
                const bool band_x = (xx >= x0 && xx < x0 + width) || (xx > x1 - width && xx <= x1);
                const bool run_y = yy >= vlo && yy <= vhi;
                if ((in_x && band_y) || (band_x && run_y)) px[j] = ink;
            }
        }
    }
    gptr o = (gptr)out;
    if (q0 + kLaneNPx <= n_px) {
        u32x4 v = {px[0], px[1], px[2], px[3]};
        __builtin_nontemporal_store(v, (MIC_GLOBAL u32x4 *)(o + q0));
    } else {
#pragma unroll
        for (int j = 0; j < kLaneNPx; ++j)
            if (q0 + j < n_px) o[q0 + j] = px[j];
    }
}

hipError_t launch_rect_outlines(void *out, int W, int H, const OutlineRect *rects_dev, int n, int width,
                                hipStream_t stream) {
    const size_t n_px = (size_t)W * H;
    if (n_px == 0) return hipSuccess;
    const size_t blocks = (n_px + kPagePx - 1) / kPagePx;
    hipLaunchKernelGGL(rect_outline_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(out), W, H, rects_dev, n, width);
    return hipGetLastError();
}

}  // namespace mic
