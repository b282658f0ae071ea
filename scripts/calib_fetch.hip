// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on gfx950 for the access shapes this repo's kernels use:
// every kernel below reads a 256 MiB buffer exactly once (nothing re-read, nothing resident between launches: the
// buffer is larger than L2 and is followed by a 512 MiB flush), so bytes-read is known and the counter's scale for
// that shape follows.   hipcc --offload-arch=gfx950 -O3 -o scripts/calib_fetch.bin scripts/calib_fetch.hip
//   wide1k   : a wave reads 64 lanes x 16 B = 1 KiB contiguous            (composite kernel's cutout rows, fills)
//   seg64    : a wave reads 16 rows x 64 B (4 lanes x 16 B per row), rows 4 KiB apart   (resample band loader)
//   seg128   : a wave reads 8 rows x 128 B, rows 4 KiB apart
//   seg32    : a wave reads 32 rows x 32 B (2 lanes x 16 B per row)
//   dword    : a wave reads 64 lanes x 4 B = 256 B contiguous
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void wide1k(const u32x4 *p, size_t n16, unsigned *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v; }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}
// SEG = bytes per row segment (16 * lanes per row); the buffer is seen as rows of 4096 B; a wave takes 1 KiB per step:
// rows_per_wave = 1024 / SEG rows, all at the same column block; successive steps walk the columns, then the rows.
template <int SEG>
__global__ void seg(const unsigned char *p, size_t bytes, unsigned *sink) {
    constexpr int LPR = SEG / 16, ROWS = 64 / LPR, PITCH = 4096, COLBLKS = PITCH / SEG;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t n_rowgroups = bytes / PITCH / ROWS;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t step = wave; step < n_rowgroups * COLBLKS; step += n_waves) {
        const size_t rg = step / COLBLKS, cb = step % COLBLKS;
        const size_t off = (rg * ROWS + lane / LPR) * PITCH + cb * SEG + (lane % LPR) * 16;
        acc ^= __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + off));
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}
__global__ void dword(const unsigned *p, size_t n4, unsigned *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= __builtin_nontemporal_load(p + i);
    if (acc == 0x12345678u) *sink = 1;
}
__global__ void flush(u32x4 *p, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}
int main() {
    const size_t bytes = (size_t)256 << 20, fbytes = (size_t)512 << 20;
    unsigned char *buf; u32x4 *fl; unsigned *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&fl, fbytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 7, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        flush<<<4096, 256>>>(fl, fbytes / 16);
        wide1k<<<4096, 256>>>((const u32x4 *)buf, bytes / 16, sink);
        flush<<<4096, 256>>>(fl, fbytes / 16);
        seg<64><<<4096, 256>>>(buf, bytes, sink);
        flush<<<4096, 256>>>(fl, fbytes / 16);
        seg<128><<<4096, 256>>>(buf, bytes, sink);
        flush<<<4096, 256>>>(fl, fbytes / 16);
        seg<32><<<4096, 256>>>(buf, bytes, sink);
        flush<<<4096, 256>>>(fl, fbytes / 16);
        dword<<<4096, 256>>>((const unsigned *)buf, bytes / 4, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("each read kernel read %zu bytes exactly once\n", bytes);
    return 0;
}
