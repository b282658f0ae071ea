// Instruction-rate and semantics probes for gfx950 that the resample / composite kernel designs lean on.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_isa.bin scripts/ubench_isa.hip && scripts/ubench_isa.bin
// 1. issue cost (cycles per wave-instruction per SIMD) of the integer / conversion VALU ops the epilogues use, at 1, 2
//    and 4 waves per SIMD, and of v_mfma_i32_16x16x64_i8;
// 2. what v_cvt_pk_u8_f32 does with fractions, negatives and values above 255;
// 3. launch cost of a trivial kernel as a function of workgroup count and size (the single-canvas composite launches
//    8100 one-wave workgroups).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (e);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int kIters = 2048;

// 8 independent chains of OP per iteration; `a` values stay live so nothing is folded.
#define RATE_KERNEL(NAME, ASM)                                                                       \
    __global__ __launch_bounds__(1024) void NAME(uint64_t *cycles, uint32_t *sink) {                \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
                 a6 = a0 + 6, a7 = a0 + 7;                                                          \
        uint32_t b = blockIdx.x | 0x01020304u, c = threadIdx.x * 3u + 0x3f800000u;                 \
        const uint64_t m = __builtin_amdgcn_ballot_w64((threadIdx.x * 7u + blockIdx.x) & 4u);      \
        __syncthreads();                                                                            \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                           \
        _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {                                      \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                    \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                           "+v"(a7)                                                                 \
                         : "v"(b), "v"(c), "s"(m)                                                   \
                         : "vcc");                                                                  \
        }                                                                                           \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                           \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;        \
    }

#define OP_LSHL_ADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 8, %8\n"
#define OP_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define OP_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define OP_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_ASHR_PK(n) "v_ashr_pk_u8_i32 %" #n ", %" #n ", %8, 22\n"
#define OP_CVT_PK_U8(n) "v_cvt_pk_u8_f32 %" #n ", %9, 1, %" #n "\n"
#define OP_MUL_F32(n) "v_mul_f32 %" #n ", %" #n ", %9\n"
#define OP_FMA_F32(n) "v_fma_f32 %" #n ", %" #n ", %9, %9\n"
#define OP_CVT_UB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define OP_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define OP_LSHL_OR(n) "v_lshl_or_b32 %" #n ", %" #n ", 16, %8\n"
#define OP_MED3(n) "v_med3_i32 %" #n ", %" #n ", 0, %8\n"
#define OP_PK_ADD16(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define OP_CVT_U32(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define OP_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define OP_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define OP_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define OP_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"

#define OP_CNDMASK_S(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %10\n"
#define OP_CMP_CNDMASK(n) "v_cmp_lt_u32_e32 vcc, %8, %" #n "\nv_cndmask_b32_e32 %" #n ", %" #n ", %9, vcc\n"
#define OP_CMP_ONLY(n) "v_cmp_lt_u32_e32 vcc, %8, %" #n "\nv_add_u32 %" #n ", %" #n ", %9\n"
#define OP_BFI(n) "v_bfi_b32 %" #n ", %8, %9, %" #n "\n"
#define OP_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define OP_ASHR_BFI(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\nv_bfi_b32 %" #n ", %" #n ", %9, %8\n"
#define OP_MAX(n) "v_max_u32 %" #n ", %" #n ", %8\n"
#define OP_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define OP_SDWA(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
#define OP_MAD64(n) "v_mad_u64_u32 v[20:21], s[20:21], %" #n ", %8, v[22:23]\nv_xor_b32 %" #n ", %" #n ", v20\n"
#define OP_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define OP_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define OP_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\nv_add_u32 %" #n ", %" #n ", %8\n"
RATE_KERNEL(k_mullo, OP_MULLO)
RATE_KERNEL(k_madi24, OP_MADI24)
RATE_KERNEL(k_cndmask_s, OP_CNDMASK_S)
RATE_KERNEL(k_cmp_cndmask, OP_CMP_CNDMASK)
RATE_KERNEL(k_cmp_only, OP_CMP_ONLY)
RATE_KERNEL(k_bfi, OP_BFI)
RATE_KERNEL(k_ashr, OP_ASHR)
RATE_KERNEL(k_ashr_bfi, OP_ASHR_BFI)
RATE_KERNEL(k_max, OP_MAX)
RATE_KERNEL(k_and_or, OP_AND_OR)
RATE_KERNEL(k_mul24, OP_MUL24)
RATE_KERNEL(k_sdwa, OP_SDWA)
RATE_KERNEL(k_lshl_add, OP_LSHL_ADD)
RATE_KERNEL(k_and, OP_AND)
RATE_KERNEL(k_add, OP_ADD)
RATE_KERNEL(k_perm, OP_PERM)
RATE_KERNEL(k_ashr_pk, OP_ASHR_PK)
RATE_KERNEL(k_cvt_pk_u8, OP_CVT_PK_U8)
RATE_KERNEL(k_mul_f32, OP_MUL_F32)
RATE_KERNEL(k_fma_f32, OP_FMA_F32)
RATE_KERNEL(k_cvt_ub, OP_CVT_UB)
RATE_KERNEL(k_mad24, OP_MAD24)
RATE_KERNEL(k_lshl_or, OP_LSHL_OR)
RATE_KERNEL(k_med3, OP_MED3)
RATE_KERNEL(k_pk_add16, OP_PK_ADD16)
RATE_KERNEL(k_cvt_u32, OP_CVT_U32)
RATE_KERNEL(k_mulhi, OP_MULHI)
RATE_KERNEL(k_xor, OP_XOR)
RATE_KERNEL(k_bfe, OP_BFE)
RATE_KERNEL(k_cndmask, OP_CNDMASK)

typedef int v4i __attribute__((ext_vector_type(4)));
// 8 independent accumulators, back to back
__global__ __launch_bounds__(1024) void k_mfma_i8(uint64_t *cycles, uint32_t *sink) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v4i acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = v4i{k, k, k, k};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    int s = 0;
    for (int k = 0; k < 8; ++k) s ^= acc[k][0] ^ acc[k][1] ^ acc[k][2] ^ acc[k][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

// the K = 32 form (gfx940's), 8 independent accumulators
__global__ __launch_bounds__(1024) void k_mfma_i8_k32(uint64_t *cycles, uint32_t *sink) {
    long a = (long)threadIdx.x * 0x0101010101010101l, b = (long)blockIdx.x + 0x0403020104030201l;
    v4i acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = v4i{k, k, k, k};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, acc[k], 0, 0, 0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    int s = 0;
    for (int k = 0; k < 8; ++k) s ^= acc[k][0] ^ acc[k][1] ^ acc[k][2] ^ acc[k][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

// MFMA beside VALU in the same wave: 1 MFMA + V lshl_add per slot
template <int V>
__global__ __launch_bounds__(1024) void k_mfma_mix(uint64_t *cycles, uint32_t *sink) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v4i acc[8];
    uint32_t x[8];
    for (int k = 0; k < 8; ++k) { acc[k] = v4i{k, k, k, k}; x[k] = threadIdx.x + k; }
    const uint32_t y = blockIdx.x;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < kIters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(x[(k + v) & 7]) : "v"(y));
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    int s = 0;
    for (int k = 0; k < 8; ++k) s ^= acc[k][0] ^ acc[k][1] ^ acc[k][2] ^ acc[k][3] ^ (int)x[k];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s;
}

__global__ void k_cvt_semantics(const float *in, uint32_t *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    uint32_t r = 0;
    const float f = in[i];
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(f));
    out[i] = r;
}

__global__ void k_empty(uint32_t *p) {
    if (p && threadIdx.x == 1023 && blockIdx.x == 0x7fffffff) p[0] = 1;
}

typedef void (*rate_fn)(uint64_t *, uint32_t *);

static void run_rate(const char *name, rate_fn fn, uint64_t *cyc_dev, uint32_t *sink_dev) {
    printf("%-22s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD: one block per CU of 4 * wps waves
        const int threads = 256 * wps, blocks = 256;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, nullptr, cyc_dev, sink_dev);
        CHECK(hipDeviceSynchronize());
        static hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (!ev0) { CHECK(hipEventCreate(&ev0)); CHECK(hipEventCreate(&ev1)); }
        CHECK(hipEventRecord(ev0, nullptr));
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, nullptr, cyc_dev, sink_dev);
        CHECK(hipEventRecord(ev1, nullptr));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        std::vector<uint64_t> c((size_t)blocks * threads / 64);
        CHECK(hipMemcpy(c.data(), cyc_dev, c.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (uint64_t v : c) sum += (double)v;
        const double per_wave = sum / c.size();  // s_memtime ticks of one wave's loop
        // per SIMD: wps waves each issued kIters * 8 instructions in that time
        printf("  wps%d: %6.2f tick/instr/SIMD", wps, per_wave / ((double)kIters * 8 * wps));
        if (wps == 4) printf("  [loop %.0f ticks in a %.1f us launch]", per_wave, ms * 1e3);
    }
    printf("\n");
}

int main() {
    uint64_t *cyc_dev;
    uint32_t *sink_dev;
    CHECK(hipMalloc(&cyc_dev, 8 * 256 * 16));
    CHECK(hipMalloc(&sink_dev, 4 * 256 * 1024));
    printf("# s_memtime ticks (shader clock cycles per the microarch guide) per wave-instruction per SIMD\n");
    run_rate("v_lshl_add_u32", k_lshl_add, cyc_dev, sink_dev);
    run_rate("v_and_b32", k_and, cyc_dev, sink_dev);
    run_rate("v_add_u32", k_add, cyc_dev, sink_dev);
    run_rate("v_xor_b32", k_xor, cyc_dev, sink_dev);
    run_rate("v_bfe_u32", k_bfe, cyc_dev, sink_dev);
    run_rate("v_cndmask_b32", k_cndmask, cyc_dev, sink_dev);
    run_rate("v_cndmask_b32 sgpr", k_cndmask_s, cyc_dev, sink_dev);
    run_rate("v_mul_lo_u32", k_mullo, cyc_dev, sink_dev);
    run_rate("v_mad_i32_i24", k_madi24, cyc_dev, sink_dev);
    run_rate("v_cmp+v_cndmask (x2)", k_cmp_cndmask, cyc_dev, sink_dev);
    run_rate("v_cmp+v_add (x2)", k_cmp_only, cyc_dev, sink_dev);
    run_rate("v_bfi_b32", k_bfi, cyc_dev, sink_dev);
    run_rate("v_ashrrev_i32", k_ashr, cyc_dev, sink_dev);
    run_rate("v_ashr+v_bfi (x2)", k_ashr_bfi, cyc_dev, sink_dev);
    run_rate("v_max_u32", k_max, cyc_dev, sink_dev);
    run_rate("v_and_or_b32", k_and_or, cyc_dev, sink_dev);
    run_rate("v_mul_u32_u24", k_mul24, cyc_dev, sink_dev);
    run_rate("v_add_u32 sdwa", k_sdwa, cyc_dev, sink_dev);
    run_rate("v_perm_b32", k_perm, cyc_dev, sink_dev);
    run_rate("v_ashr_pk_u8_i32", k_ashr_pk, cyc_dev, sink_dev);
    run_rate("v_cvt_pk_u8_f32", k_cvt_pk_u8, cyc_dev, sink_dev);
    run_rate("v_mul_f32", k_mul_f32, cyc_dev, sink_dev);
    run_rate("v_fma_f32", k_fma_f32, cyc_dev, sink_dev);
    run_rate("v_cvt_f32_ubyte1", k_cvt_ub, cyc_dev, sink_dev);
    run_rate("v_cvt_u32_f32", k_cvt_u32, cyc_dev, sink_dev);
    run_rate("v_mad_u32_u24", k_mad24, cyc_dev, sink_dev);
    run_rate("v_mul_hi_u32", k_mulhi, cyc_dev, sink_dev);
    run_rate("v_lshl_or_b32", k_lshl_or, cyc_dev, sink_dev);
    run_rate("v_med3_i32", k_med3, cyc_dev, sink_dev);
    run_rate("v_pk_add_u16", k_pk_add16, cyc_dev, sink_dev);
    run_rate("mfma_i32_16x16x64_i8", k_mfma_i8, cyc_dev, sink_dev);
    run_rate("mfma_i32_16x16x32_i8", k_mfma_i8_k32, cyc_dev, sink_dev);
    run_rate("mfma + 2 valu", k_mfma_mix<2>, cyc_dev, sink_dev);
    run_rate("mfma + 4 valu", k_mfma_mix<4>, cyc_dev, sink_dev);
    run_rate("mfma + 8 valu", k_mfma_mix<8>, cyc_dev, sink_dev);

    // ---- v_cvt_pk_u8_f32 semantics
    const float probe[] = {0.0f, 0.25f, 0.5f, 0.75f, 1.0f, 1.5f, 2.5f, 3.5f, 253.996f, 254.5f, 254.999f, 255.0f, 255.4f,
                           255.5f, 256.0f, 300.0f, 1e9f, -0.25f, -0.5f, -1.0f, -300.0f, 126.99999f, 127.5f};
    const int np = (int)(sizeof probe / sizeof probe[0]);
    float *pin;
    uint32_t *pout;
    CHECK(hipMalloc(&pin, sizeof probe));
    CHECK(hipMalloc(&pout, 4 * np));
    CHECK(hipMemcpy(pin, probe, sizeof probe, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt_semantics, dim3(1), dim3(64), 0, nullptr, pin, pout, np);
    std::vector<uint32_t> res(np);
    CHECK(hipMemcpy(res.data(), pout, 4 * np, hipMemcpyDeviceToHost));
    printf("# v_cvt_pk_u8_f32:");
    for (int i = 0; i < np; ++i) printf(" %g->%u", probe[i], res[i]);
    printf("\n");

    // ---- launch cost vs grid shape
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("# trivial kernel, 200 back-to-back launches, us per launch (event time)\n");
    const int shapes[][2] = {{8100, 64}, {4050, 128}, {2025, 256}, {1013, 512}, {16200, 64}, {129600, 64}, {32400, 256},
                             {256, 256}, {1024, 256}, {2048, 64}};
    for (auto &s : shapes) {
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k_empty, dim3(s[0]), dim3(s[1]), 0, nullptr, sink_dev);
        CHECK(hipEventRecord(e0, nullptr));
        for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k_empty, dim3(s[0]), dim3(s[1]), 0, nullptr, sink_dev);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  grid %6d x %4d threads: %7.2f us\n", s[0], s[1], ms * 1000.f / 200.f);
    }
    return 0;
}
