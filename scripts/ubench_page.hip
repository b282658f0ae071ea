// Floor of a single-canvas composite launch on MI355X: one wave per 4 KiB page of the canvas (WAVES pages per
// workgroup), four nontemporal 16-byte stores per lane, optionally READ bytes of source per page first (contiguous,
// from a buffer of `src_mb` MB walked page by page -- 16 MB: the C3 atlas) -- no culling, no blending, no layer
// records.  100 back-to-back launches between two HIP events, like scripts/time_single.py.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_page.bin scripts/ubench_page.hip && scripts/ubench_page.bin
#include <hip/hip_runtime.h>

#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(16))) u32x4a;

template <int WAVES, int READ_KB>
__global__ __launch_bounds__(64 * WAVES) void page_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ src, uint32_t n_pages,
                                                         uint32_t src_pages) {
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t page = blockIdx.x * WAVES + wave;
    if (page >= n_pages) return;
    u32x4 v[4] = {u32x4{page, 1u, 2u, 3u}, u32x4{page, 1u, 2u, 3u}, u32x4{page, 1u, 2u, 3u}, u32x4{page, 1u, 2u, 3u}};
    if (READ_KB > 0) {
        const uint32_t *s = src + (size_t)(page % src_pages) * (READ_KB * 256);
#pragma unroll
        for (int r = 0; r < READ_KB; ++r) {
            const u32x4 q = *reinterpret_cast<const u32x4a *>(s + r * 256 + lane * 4);
            v[r & 3] ^= q;
        }
    }
    uint32_t *o = out + (size_t)page * 1024 + lane * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(v[r], reinterpret_cast<u32x4a *>(o + r * 256));
}

template <int WAVES, int READ_KB>
static float run(uint32_t *out, const uint32_t *src, uint32_t n_pages, uint32_t src_pages) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid((n_pages + WAVES - 1) / WAVES), block(64 * WAVES);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((page_kernel<WAVES, READ_KB>), grid, block, 0, 0, out, src, n_pages, src_pages);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((page_kernel<WAVES, READ_KB>), grid, block, 0, 0, out, src, n_pages, src_pages);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 10.0f;  // us per launch
}

int main() {
    const uint32_t sizes[3][2] = {{1920, 1080}, {3840, 2160}, {7680, 4320}};
    uint32_t *out, *src;
    (void)hipMalloc(&out, (size_t)7680 * 4320 * 4 + 4096);
    (void)hipMalloc(&src, 64u << 20);
    (void)hipMemset(src, 7, 64u << 20);
    for (auto &s : sizes) {
        const uint32_t n_pages = (uint32_t)(((size_t)s[0] * s[1] * 4 + 4095) / 4096);
        const float mb = n_pages * 4096 / 1e6f;
        printf("%ux%u (%u pages, %.1f MB written)\n", s[0], s[1], n_pages, mb);
        // reads: 2 KB per page = half the canvas bytes (C3: 16.0 MB of cutouts under a 33 MB canvas), walking a 16 MB buffer
        const uint32_t sp2 = (16u << 20) / 2048, sp4 = (16u << 20) / 4096;
        const float w1 = run<1, 0>(out, src, n_pages, 1), w4 = run<4, 0>(out, src, n_pages, 1), w8 = run<8, 0>(out, src, n_pages, 1);
        printf("  write only            1 / 4 / 8 pages per workgroup: %5.2f %5.2f %5.2f us  (%.2f TB/s at 4)\n", w1, w4, w8, mb / w4 / 1e6 * 1e6 / 1e6);
        const float a1 = run<1, 2>(out, src, n_pages, sp2), a4 = run<4, 2>(out, src, n_pages, sp2), a8 = run<8, 2>(out, src, n_pages, sp2);
        printf("  + 2 KB read per page  1 / 4 / 8 pages per workgroup: %5.2f %5.2f %5.2f us  (%.2f TB/s at 4)\n", a1, a4, a8, mb * 1.5f / a4 / 1e6 * 1e6 / 1e6);
        const float b1 = run<1, 4>(out, src, n_pages, sp4), b4 = run<4, 4>(out, src, n_pages, sp4), b8 = run<8, 4>(out, src, n_pages, sp4);
        printf("  + 4 KB read per page  1 / 4 / 8 pages per workgroup: %5.2f %5.2f %5.2f us  (%.2f TB/s at 4)\n", b1, b4, b8, mb * 2.0f / b4 / 1e6 * 1e6 / 1e6);
        fflush(stdout);
    }
    return 0;
}
