// Debug overlay: replaces _save_overlay_debug's drawing (macro_placement_test.py:967-983) --
// ImageDraw.rectangle(box, outline=colour, width) per placement, in list order, on a transparent
// RGBA image.  ImageDraw stores the ink on an RGBA image (no blending), so the result per pixel is
// the ink of the LAST rectangle whose outline covers it: a pure function of (x, y), evaluated in one
// pass that writes every pixel once (4 B/px, HBM-bound like fill_kernel when few outlines cross a row).
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

__global__ __launch_bounds__(256) void rect_outline_kernel(uint32_t *__restrict__ out, int W, int H,
                                                           const OutlineRect *__restrict__ rects, int n, int width) {
    __shared__ OutlineRect tile[256];
    const int64_t n_px = (int64_t)W * H;
    const int64_t q0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kLaneNPx;
    // rows this wave's 256-pixel run touches (wave-uniform culling)
    const int64_t w0 = ((int64_t)blockIdx.x * 256 + (threadIdx.x & ~63)) * kLaneNPx;
    const int row_first = (int)(min(w0, n_px - 1) / W), row_last = (int)(min(w0 + kWavePx - 1, n_px - 1) / W);
    int y = (int)(min(q0, n_px - 1) / W), x = (int)(min(q0, n_px - 1) - (int64_t)y * W);
    int xs[kLaneNPx], ys[kLaneNPx];
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) {
        xs[j] = x; ys[j] = y;
        if (++x == W) { x = 0; ++y; }
    }
    uint32_t px[kLaneNPx] = {0u, 0u, 0u, 0u};
    for (int base = 0; base < n; base += 256) {
        __syncthreads();
        if (base + (int)threadIdx.x < n) tile[threadIdx.x] = rects[base + threadIdx.x];
        __syncthreads();
        const int m = min(256, n - base);
        for (int r = 0; r < m; ++r) {
            const OutlineRect R = tile[r];
            if (R.ymax < row_first || R.ymin > row_last) continue;  // wave-uniform
#pragma unroll
            for (int j = 0; j < kLaneNPx; ++j) {
                const int xx = xs[j], yy = ys[j];
                const bool in_x = xx >= R.x0 && xx <= R.x1;
                const bool band_y = (yy >= R.y0 && yy < R.y0 + width) || (yy > R.y1 - width && yy <= R.y1);
                const bool band_x = (xx >= R.x0 && xx < R.x0 + width) || (xx > R.x1 - width && xx <= R.x1);
                const bool run_y = yy >= R.vlo && yy <= R.vhi;
                if ((in_x && band_y) || (band_x && run_y)) px[j] = R.rgba;
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
