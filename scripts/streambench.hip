// Standalone store/copy pattern microbenchmark for gfx950 (not part of the product).
// hipcc --offload-arch=gfx950 -O3 scripts/streambench.hip -o gpurun_out/streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef u4 __attribute__((aligned(4))) u4a4;
#define G __attribute__((address_space(1)))

// mode 0: tile 256px x ROWS rows per wave (row stride W); mode 1: ROWS KiB contiguous per wave
template <int ROWS, int MODE, bool NT, bool COPY, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(uint32_t *out, const uint32_t *src, int W, int H, int soff) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4 v[ROWS];
    size_t idx[ROWS];
    if (MODE == 0) {
        const int tiles_x = W / 256;
        const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
        const int x = tx * 256 + lane * 4, y0 = (ty * WAVES + wave) * ROWS;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) idx[r] = (size_t)(y0 + r) * W + x;
    } else {
        const size_t p0 = ((size_t)blockIdx.x * WAVES + wave) * (ROWS * 256);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) idx[r] = p0 + r * 256 + lane * 4;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (COPY) {
            const G u4a4 *p = (const G u4a4 *)(src + idx[r] + soff);
            v[r] = NT ? __builtin_nontemporal_load(p) : *p;
        } else v[r] = (u4)(0x01020304u + blockIdx.x);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        G u4a4 *p = (G u4a4 *)(out + idx[r]);
        if (NT) __builtin_nontemporal_store(v[r], p); else *p = v[r];
    }
}

// persistent grid-stride, contiguous 1 KiB per wave per iteration, UNROLL iterations in flight
template <int UNROLL, bool NT, bool COPY>
__global__ __launch_bounds__(256) void kp(uint32_t *out, const uint32_t *src, size_t n_px, int soff) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (; i + (UNROLL - 1) * stride < n_px; i += UNROLL * stride) {
        u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = COPY ? *(const G u4a4 *)(src + i + u * stride + soff) : (u4)(7u);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { G u4a4 *p = (G u4a4 *)(out + i + u * stride); if (NT) __builtin_nontemporal_store(v[u], p); else *p = v[u]; }
    }
    for (; i < n_px; i += stride) { u4 v = COPY ? *(const G u4a4 *)(src + i + soff) : (u4)(7u); *(G u4a4 *)(out + i) = v; }
}

template <int MAP> __global__ __launch_bounds__(256) void k2(uint32_t *out, int param) {
    size_t b;
    if (MAP == 0) b = (size_t)blockIdx.y * param + blockIdx.x;
    else if (MAP == 1) b = (size_t)(blockIdx.x % 8) * param + blockIdx.x / 8;
    else if (MAP == 2) b = blockIdx.x ^ 1;
    else b = (blockIdx.x & ~7u) | ((blockIdx.x + blockIdx.x / 8) & 7u);
    *(G u4a4 *)(out + b * 1024 + threadIdx.x * 4) = (u4)(5u);
}
// block b (XCD b%8): wave w handles pages (b%8) + 8*((b/8)*WAVES*PPW + w*PPW + j), j < PPW = KIB/4
template <int WAVES, int KIB, bool NT, bool COPY> __global__ __launch_bounds__(64 * WAVES) void k3(uint32_t *out, const uint32_t *src) {
    constexpr int PPW = KIB / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4 v[PPW * 4];
    size_t idx[PPW * 4];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const size_t page = (blockIdx.x % 8) + 8 * ((size_t)(blockIdx.x / 8) * WAVES * PPW + wave * PPW + j);
#pragma unroll
        for (int r = 0; r < 4; ++r) idx[j * 4 + r] = page * 1024 + r * 256 + lane * 4;
    }
#pragma unroll
    for (int i = 0; i < PPW * 4; ++i) {
        if (COPY) { const G u4a4 *p = (const G u4a4 *)(src + idx[i]); v[i] = NT ? __builtin_nontemporal_load(p) : *p; }
        else v[i] = (u4)(3u);
    }
#pragma unroll
    for (int i = 0; i < PPW * 4; ++i) { G u4a4 *p = (G u4a4 *)(out + idx[i]); if (NT) __builtin_nontemporal_store(v[i], p); else *p = v[i]; }
}
// write page b; read from a page whose residue mod 8 is scrambled relative to b
template <bool NT> __global__ __launch_bounds__(256) void k4(uint32_t *out, const uint32_t *src) {
    const size_t b = blockIdx.x;
    const size_t sp = (b & ~(size_t)63) + ((b * 37 + 11) & 63);  // permutation within groups of 64 pages
    const G u4a4 *p = (const G u4a4 *)(src + sp * 1024 + threadIdx.x * 4 + 1);
    u4 v = NT ? __builtin_nontemporal_load(p) : *p;
    G u4a4 *o = (G u4a4 *)(out + b * 1024 + threadIdx.x * 4);
    if (NT) __builtin_nontemporal_store(v, o); else *o = v;
}
template <class F> float timeit(F f, int iters = 30) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / iters;
}

int main() {
    const int W = 3840, H = 2160 * 16;
    const size_t n = (size_t)W * H, bytes = n * 4;
    uint32_t *out, *src;
    CK(hipMalloc(&out, bytes + 4096)); CK(hipMalloc(&src, bytes + 4096));
    CK(hipMemset(src, 1, bytes + 4096));
    auto rep = [&](const char *name, float ms, double factor) { printf("%-44s %8.1f us %7.0f GB/s\n", name, ms * 1e3, factor * bytes / ms / 1e6); };
#define RUN(NAME, ROWS, MODE, NT, COPY, WAVES, SOFF) { int blocks = (int)(n / ((size_t)256 * ROWS * WAVES)); \
    float ms = timeit([&] { hipLaunchKernelGGL((k<ROWS, MODE, NT, COPY, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, src, W, H, SOFF); }); rep(NAME, ms, COPY ? 2.0 : 1.0); }
    RUN("copy 1KiBx4 waves, src aligned", 1, 1, false, true, 4, 0)
    RUN("copy 1KiBx4 waves, src +3 pages", 1, 1, false, true, 4, 3 * 1024)
    RUN("copy 1KiBx4 waves, src +1 page", 1, 1, false, true, 4, 1024)
    RUN("copy 1KiBx4 waves, src +3 pages+1dw", 1, 1, false, true, 4, 3 * 1024 + 1)
    RUN("copy 1KiBx4 waves nt, src +3 pages+1dw", 1, 1, true, true, 4, 3 * 1024 + 1)
    RUN("copy 1KiBx4 waves, src +5 pages+100dw", 1, 1, false, true, 4, 5 * 1024 + 100)
    RUN("copy 1KiBx4 waves nt, src +5 pages+100dw", 1, 1, true, true, 4, 5 * 1024 + 100)
    { int blocks = (int)(n / 1024) - 64; float ms;
      ms = timeit([&] { hipLaunchKernelGGL((k4<false>), dim3(blocks), dim3(256), 0, 0, out, src); }); rep("copy 1KiBx4 waves, src page scrambled", ms, 2.0);
      ms = timeit([&] { hipLaunchKernelGGL((k4<true>), dim3(blocks), dim3(256), 0, 0, out, src); }); rep("copy 1KiBx4 waves nt, src page scrambled", ms, 2.0); }
    CK(hipDeviceSynchronize());
    return 0;
}
