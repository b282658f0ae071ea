// Ordered RGBA alpha-over of resolved layers onto canvases: the hot kernels of the path.
//
// Replaces the per-placement loop of compositor.composite (compositor.py:12-21) and, inside it,
// Pillow's crop + AlphaComposite.c + paste triple pass (Image.alpha_composite(im, dest)).
//
// Mapping (gfx950, wave64).  The canvas is a linear stream of RGBA words cut into 4 KiB pages
// aligned to absolute address.  One workgroup = ONE WAVEFRONT = one page: a lane owns four groups
// of four adjacent pixels, 1 KiB apart, so the page is written by four fully coalesced 1 KiB
// stores.  Workgroups are dealt round-robin over the 8 XCDs, so in dispatch order every XCD keeps
// writing one residue class of pages (mod 8); on MI355X that is what lets a store stream reach
// ~6.7 TB/s (see mic_internal.h and scripts/streambench.hip).
//
// Layer culling happens inside the wave, 64 layers at a time: lane l loads layer record l and tests
// its rectangle against the page's four 256-pixel runs (a handful of integer compares, exact at run
// granularity); four __ballot's give four ordered 64-bit hit masks, and the record of a hit layer is
// broadcast out of the lane that holds it with v_readlane.  No per-tile bin lists are built on the
// host or by a pre-pass kernel (a binning pre-pass was tried: 10% of the time and one more
// dependent memory round trip per wave).
//
// Pixel state stays in registers as 8-bit RGBA between layers, because the reference rounds to
// 8 bits after every object (compositor.py:21) and the result is order dependent.  The canvas is
// written exactly once and every visible cutout pixel is read exactly once: HBM traffic equals
// the algorithmic bytes (4*W*H + 4*visible source pixels).
//
// Blend: Pillow's integer formula, verbatim.  When every source pixel a wave holds for a layer
// has alpha 0 or 255 (the reference's bundles are binary-alpha cutouts) the formula reduces
// exactly to "keep dst" / "take src" (exhaustively checked), and the wave takes that select
// path; any partial alpha in the wave sends it through the full formula.
//
// HBM-bound by construction: MFMA is not used (there is no contraction to feed it).
#include <cstring>

#include "composite_device.h"

namespace mic {

// Instantiations (chosen per job on the host, mic_api.hip):
//   ALIGNED  W % 4 == 0 and a 16-byte aligned canvas: no pixel group straddles a row end;
//   SOLID    solid opaque background: nothing to read, destination alpha stays 255.
// <true, true> is what the reference's pipeline produces (fill_solid canvases,
// background_resizing.py:32).  Launch bounds: hipcc settles for 85 VGPRs (5 waves/SIMD) unless told
// that more waves are wanted; 7 waves = 72 VGPRs fit without spills, 8 would spill.
#ifndef MIC_HOT_WAVES
#define MIC_HOT_WAVES 7
#endif
// MODE (where a wave finds its job and layer records):
//   kFromTables  batches: job blockIdx.y of the device job table, layer records in the device layer table;
//   kJobInArgs   ONE job, handed over BY VALUE in the kernel arguments (scalar loads from the kernarg segment) instead
//                of through the device job table: a wave's first memory round trip -- of the three dependent ones it
//                makes: job, layer records, cutout pixels -- disappears, and so does the table upload.  That is the
//                reference's own call shape: one composite() per call (compositor.py:6-22);
//   kAllInArgs   ... and its layer records too (round 3), when there are at most kPackLayers of them (2 KiB of the
//                4 KiB argument segment): nothing at all is uploaded for such a call -- a one-shot composite() is the
//                launch alone -- and lane l reads record l straight from the argument segment.
enum : int { kFromTables = 0, kJobInArgs = 1, kAllInArgs = 2 };
struct LayerPack {
    Layer l[kPackLayers];
};
template <bool ALIGNED, bool SOLID, int MODE>
__global__ __launch_bounds__(64, (ALIGNED && SOLID) ? MIC_HOT_WAVES : (SOLID ? 6 : 4)) void composite_kernel(
    const Job *__restrict__ jobs, const Layer *__restrict__ layers, const Job one, const LayerPack pack) {
    const Job job = MODE != kFromTables ? one : jobs[blockIdx.y];
    // (page_begin: a band launch of the pipelined LANCZOS path covers pages [page_begin, n_pages) of its canvas; a
    // multiple of 8, so the page <-> XCD pairing holds)
    const int page = (int)blockIdx.x + (MODE != kFromTables ? job.page_begin : 0);
    if (page >= job.n_pages) return;
    const int lane = threadIdx.x;
    const Layer *jl = MODE == kAllInArgs ? pack.l : layers + job.layer_begin;
#include "composite_body.inc"
}

// Jobs arrive sorted by class: [0, n0) aligned+solid, [n0, n1) unaligned+solid, [n1, n2) aligned with
// a background image / translucent colour, [n2, n_jobs) neither.  single != nullptr: the launch's only job,
// passed in the kernel arguments (jobs_dev is not read).
// The argument block of every launch carries a LayerPack (2 KiB copied into the kernarg ring by the runtime: a few
// dozen nanoseconds); only kAllInArgs launches read it.  One per calling thread: contexts are driven concurrently.
static thread_local LayerPack g_pack;

hipError_t launch_composite(const Job *jobs_dev, const Layer *layers_dev, int n_jobs, const int class_end[3],
                            int pitch, const Job *single, const Layer *single_layers_host, hipStream_t stream) {
    if (n_jobs <= 0 || pitch <= 0) return hipSuccess;
    // grid.x (= pitch) is a multiple of 8 so that (linear workgroup id) mod 8 == (page index) mod 8
    // for every job of the launch: the XCD <-> page residue pairing survives the 2-D grid.
    const int b[5] = {0, class_end[0], class_end[1], class_end[2], n_jobs};
    if (single && n_jobs == 1) {
        const dim3 grid((unsigned)pitch, 1u);
        const int cls = b[1] > b[0] ? 0 : b[2] > b[1] ? 1 : b[3] > b[2] ? 2 : 3;
        if (single_layers_host && single->layer_count <= kPackLayers) {
            Job one = *single;
            if (one.layer_count > 0) memcpy(g_pack.l, single_layers_host + one.layer_begin, sizeof(Layer) * (size_t)one.layer_count);
            one.layer_begin = 0;
            switch (cls) {
                case 0: hipLaunchKernelGGL((composite_kernel<true, true, kAllInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                case 1: hipLaunchKernelGGL((composite_kernel<false, true, kAllInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                case 2: hipLaunchKernelGGL((composite_kernel<true, false, kAllInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                default: hipLaunchKernelGGL((composite_kernel<false, false, kAllInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, one, g_pack); break;
            }
            return hipGetLastError();
        }
        switch (cls) {
            case 0: hipLaunchKernelGGL((composite_kernel<true, true, kJobInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, *single, g_pack); break;
            case 1: hipLaunchKernelGGL((composite_kernel<false, true, kJobInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, *single, g_pack); break;
            case 2: hipLaunchKernelGGL((composite_kernel<true, false, kJobInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, *single, g_pack); break;
            default: hipLaunchKernelGGL((composite_kernel<false, false, kJobInArgs>), grid, dim3(64), 0, stream, jobs_dev, layers_dev, *single, g_pack); break;
        }
        return hipGetLastError();
    }
    const Job none{};
    if (b[1] > b[0])
        hipLaunchKernelGGL((composite_kernel<true, true, kFromTables>), dim3((unsigned)pitch, (unsigned)(b[1] - b[0])), dim3(64),
                           0, stream, jobs_dev + b[0], layers_dev, none, g_pack);
    if (b[2] > b[1])
        hipLaunchKernelGGL((composite_kernel<false, true, kFromTables>), dim3((unsigned)pitch, (unsigned)(b[2] - b[1])), dim3(64),
                           0, stream, jobs_dev + b[1], layers_dev, none, g_pack);
    if (b[3] > b[2])
        hipLaunchKernelGGL((composite_kernel<true, false, kFromTables>), dim3((unsigned)pitch, (unsigned)(b[3] - b[2])), dim3(64),
                           0, stream, jobs_dev + b[2], layers_dev, none, g_pack);
    if (b[4] > b[3])
        hipLaunchKernelGGL((composite_kernel<false, false, kFromTables>), dim3((unsigned)pitch, (unsigned)(b[4] - b[3])), dim3(64),
                           0, stream, jobs_dev + b[3], layers_dev, none, g_pack);
    return hipGetLastError();
}

// Image.new("RGBA", size, colour) (background_resizing.py:32): one 4 KiB page per workgroup.
__global__ __launch_bounds__(256) void fill_kernel(uint32_t *__restrict__ out, uint32_t rgba,
                                                   int64_t n_px, int px_shift) {
    const int64_t q0 = (int64_t)blockIdx.x * kPagePx - px_shift + threadIdx.x * kLaneNPx;
    gptr o = (gptr)out;
    if (q0 >= 0 && q0 + kLaneNPx <= n_px) {
        store4(o + q0, (u32x4)(rgba));
    } else {
#pragma unroll
        for (int j = 0; j < kLaneNPx; ++j)
            if (q0 + j >= 0 && q0 + j < n_px) store1(o + (q0 + j), rgba);
    }
}

// fill_gradient's pixel loop (background_resizing.py:80-94): along one axis, position i of n gets
// rgb = (1 - t) * c1 + t * c2 with t = i / max(1, n - 1), evaluated the way NumPy evaluates it there
// -- t in double, (1 - t) and t rounded to float32, two float32 products and one float32 sum, each
// rounded (no FMA contraction) -- then truncated by astype(uint8); alpha 255.
// The colour only depends on the position along the axis: a first tiny launch evaluates the n <= 65535
// colours (one double division each), the second is a fill that looks its colour up (the table stays
// in L1/L2); evaluating the division per pixel made this kernel three times slower than fill_kernel.
__global__ __launch_bounds__(256) void gradient_table_kernel(uint32_t *__restrict__ table, int n, float c1r, float c1g,
                                                             float c1b, float c2r, float c2g, float c2b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double t = (double)i / (double)max(1, n - 1);
    const float a = (float)(1.0 - t), b = (float)t;
    const uint32_t r = (uint32_t)__fadd_rn(__fmul_rn(a, c1r), __fmul_rn(b, c2r));
    const uint32_t g = (uint32_t)__fadd_rn(__fmul_rn(a, c1g), __fmul_rn(b, c2g));
    const uint32_t bl = (uint32_t)__fadd_rn(__fmul_rn(a, c1b), __fmul_rn(b, c2b));
    table[i] = (r & 255u) | ((g & 255u) << 8) | ((bl & 255u) << 16) | 0xFF000000u;
}

__global__ __launch_bounds__(256) void gradient_kernel(uint32_t *__restrict__ out, int W, int H,
                                                       const uint32_t *__restrict__ table, int vertical) {
    const int64_t n_px = (int64_t)W * H;
    const int64_t q0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kLaneNPx;
    if (q0 >= n_px) return;
    gptr o = (gptr)out;
    int y = (int)(q0 / W), x = (int)(q0 - (int64_t)y * W);
    uint32_t px[kLaneNPx];
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) {
        px[j] = table[vertical ? y : x];
        if (++x == W) { x = 0; y = min(y + 1, H - 1); }
    }
    if (q0 + kLaneNPx <= n_px) {
        store4(o + q0, u32x4{px[0], px[1], px[2], px[3]});
    } else {
#pragma unroll
        for (int j = 0; j < kLaneNPx; ++j)
            if (q0 + j < n_px) store1(o + (q0 + j), px[j]);
    }
}

hipError_t launch_gradient(void *out, int W, int H, const uint8_t c1[3], const uint8_t c2[3], int vertical,
                           uint32_t *table_dev, hipStream_t stream) {
    const int64_t n_px = (int64_t)W * H;
    if (n_px <= 0) return hipSuccess;
    const int n = vertical ? H : W;
    hipLaunchKernelGGL(gradient_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, table_dev, n,
                       (float)c1[0], (float)c1[1], (float)c1[2], (float)c2[0], (float)c2[1], (float)c2[2]);
    const int64_t threads = (n_px + kLaneNPx - 1) / kLaneNPx;
    hipLaunchKernelGGL(gradient_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(out), W, H, table_dev, vertical);
    return hipGetLastError();
}

hipError_t launch_fill(void *out, uint32_t rgba, size_t n_px, hipStream_t stream) {
    if (n_px == 0) return hipSuccess;
    const int px_shift = (int)((reinterpret_cast<uint64_t>(out) & 4095u) / 4);
    const size_t pages = (n_px + px_shift + kPagePx - 1) / kPagePx;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)pages), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(out), rgba, (int64_t)n_px, px_shift);
    return hipGetLastError();
}

}  // namespace mic
