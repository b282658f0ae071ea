// Read-only floor of the median kernel's shape on MI355X: N bytes read exactly once per launch by G workgroups of T
// threads, every wave taking 4 KiB trips (four 16-byte loads per lane, issued back to back) G*T/64 trips apart, the
// values OR-ed into a register (one conditional store per wave keeps the loads alive).  Variants: trips in flight per
// wave (1 or 2), and an LDS histogram update per pixel channel (the median's own inner work, 8 replicas) on or off.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_read.bin scripts/ubench_read.hip && scripts/ubench_read.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// FLUSH (with HIST): 1 = every block adds its non-empty bins to one of 8 global copies (agent-scope atomics), waits for
// them; 2 = ... and takes a retirement ticket, the last block reading the copies back (the median's hand-off).
template <int DEPTH, bool HIST, int FLUSH = 0>
__global__ void read_kernel(const uint32_t *__restrict__ px, size_t n_px, uint32_t *__restrict__ out, uint32_t *__restrict__ ghist = nullptr) {
    __shared__ uint32_t lh[HIST ? 3 * 256 * 8 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves = blockDim.x >> 6;
    if (HIST) {
        for (int i = threadIdx.x; i < 3 * 256 * 8; i += blockDim.x) lh[i] = 0;
        __syncthreads();
    }
    const size_t stride = (size_t)gridDim.x * waves * 1024;
    size_t at = ((size_t)blockIdx.x * waves + wave) * 1024;
    const size_t base0 = at;
    uint32_t acc = 0;
    u32x4 t[DEPTH][4];
    auto request = [&](u32x4 (&r)[4], size_t trip) __attribute__((always_inline)) {
        const size_t from = trip + 1024 <= n_px ? trip : base0;
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_memcpy(&r[u], px + from + (size_t)u * 256 + (size_t)lane * 4, 16);
    };
    auto use = [&](const u32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (HIST) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) atomicAdd(&lh[(c * 256 + ((r[u][j] >> (8 * c)) & 255u)) * 8 + (lane & 7)], 1u);
            } else {
                acc |= r[u][0] | r[u][1] | r[u][2] | r[u][3];
            }
        }
    };
    if (at + 1024 > n_px) return;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) request(t[d], at + d * stride);
    for (;;) {
        bool done = false;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            use(t[d]);
            request(t[d], at + DEPTH * stride);
            at += stride;
            if (at + 1024 > n_px) { done = true; break; }
        }
        if (done) break;
    }
    if (HIST && FLUSH == 0) {
        __syncthreads();
        uint32_t s = 0;
        for (int i = threadIdx.x; i < 3 * 256 * 8; i += blockDim.x) s += lh[i];
        acc = s;
    }
    if (HIST && FLUSH >= 1) {
        __shared__ uint32_t is_last;
        __syncthreads();
        uint32_t *copy = ghist + (blockIdx.x & 7) * 1024;
        for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += lh[i * 8 + ((k + threadIdx.x) & 7)];
            if (s) atomicAdd(&copy[i], s);
        }
        if (FLUSH >= 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t ticket = __hip_atomic_fetch_add(ghist + 8 * 1024, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                is_last = ticket == gridDim.x - 1 ? 1u : 0u;
            }
            __syncthreads();
            if (is_last) {
                uint32_t s = 0;
                for (int i = threadIdx.x; i < 3 * 256; i += blockDim.x)
                    for (int g = 0; g < 8; ++g) s += __hip_atomic_load(ghist + g * 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (threadIdx.x == 0) ghist[8 * 1024] = 0;  // ticket back to zero for the next launch
                acc = s;
            }
        }
    }
    if (acc == 0x12345677u) out[blockIdx.x] = acc;
}

template <int DEPTH, bool HIST, int FLUSH = 0>
static float run(const uint32_t *px, size_t n_px, uint32_t *out, int G, int T, int reps, uint32_t *ghist = nullptr) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((read_kernel<DEPTH, HIST, FLUSH>), dim3(G), dim3(T), 0, 0, px, n_px, out, ghist);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<DEPTH, HIST, FLUSH>), dim3(G), dim3(T), 0, 0, px, n_px, out, ghist);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main() {
    const size_t sizes[3] = {(size_t)1920 * 1080, (size_t)3840 * 2160, (size_t)7680 * 4320};
    const char *names[3] = {"1080p", "4k", "8k"};
    uint32_t *px, *out, *ghist;
    hipMalloc(&px, sizes[2] * 4);
    hipMalloc(&ghist, 16 * 1024 * 4);
    hipMemset(ghist, 0, 16 * 1024 * 4);
    hipMalloc(&out, 1 << 20);
    std::vector<uint32_t> h(sizes[2]);
    uint32_t x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x; }
    hipMemcpy(px, h.data(), sizes[2] * 4, hipMemcpyHostToDevice);
    const int shapes[][2] = {{256, 1024}, {512, 1024}, {1024, 256}, {2048, 256}, {4096, 256}, {512, 512}, {1024, 512}};
    for (int s = 0; s < 3; ++s) {
        for (auto &sh : shapes) {
            const float a = run<1, false>(px, sizes[s], out, sh[0], sh[1], 100);
            const float b = run<2, false>(px, sizes[s], out, sh[0], sh[1], 100);
            const float c = run<1, true>(px, sizes[s], out, sh[0], sh[1], 100);
            const float d = run<2, true>(px, sizes[s], out, sh[0], sh[1], 100);
            const float e = run<1, true, 1>(px, sizes[s], out, sh[0], sh[1], 100, ghist);
            const float f = run<1, true, 2>(px, sizes[s], out, sh[0], sh[1], 100, ghist);
            printf("%-6s %5d x %4d   read-only depth1 %6.1f us (%.2f TB/s)  depth2 %6.1f us   + LDS histogram depth1 %6.1f us  depth2 %6.1f us   + flush %6.1f us   + ticket and read-back %6.1f us\n",
                   names[s], sh[0], sh[1], a, sizes[s] * 4 / (a * 1e-6) / 1e12, b, c, d, e, f);
            fflush(stdout);
        }
    }
    return 0;
}
