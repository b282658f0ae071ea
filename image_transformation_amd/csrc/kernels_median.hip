// Median background colour: replaces _median_color_nontransparent (background_resizing.py:11-22),
// i.e. np.median over the R, G, B of pixels with alpha > 0 (over all pixels when none has), then
// int() truncation.
//
// An exact median of bytes needs no sort: a 256-bin histogram per channel plus a cumulative scan
// gives the order statistics (n-1)//2 and n//2 whose mean np.median returns.  One streaming read
// of the image (4 B/px): HBM-bound.
//
// ONE launch.  Every pixel goes into one of two histogram sets, keyed by alpha > 0 / alpha == 0.
// The reference's "all pixels" fallback only triggers when no pixel has alpha > 0 -- and then the
// alpha == 0 set IS all pixels -- so the two sets answer both cases from a single pass.  The block
// that retires last (device-scope ticket) runs the 256-bin scan + select and then re-zeroes the
// scratch, so the next call needs neither a memset nor a second kernel.
//
// Shape: at most 512 blocks of 16 waves (two per CU fill every wave slot of the chip), one LDS
// histogram per block (8 bank-interleaved replicas, 48 KiB).  A wave owns 4 KiB of the image per trip and issues its four 16-byte
// loads per lane back to back, unconditionally (a load under a branch makes hipcc wait for it
// before issuing the next); only the single ragged trip at the end of the image goes through a
// guarded per-pixel path.  Few, large blocks also keep the two per-block costs small: the flush of
// non-empty bins (same-address global atomics from every block) and the agent-scope release on the
// ticket.
//
// Backgrounds are mostly flat, so within a wave all 256 pixels of a chunk often hit the same bin;
// instead of letting 256 same-address LDS atomics serialise, the wave tests each channel for that
// case with readfirstlane/__ballot and lets one lane add the population count.  Otherwise (noisy
// regions, mostly distinct bins) every lane issues its own LDS atomics.
#include <algorithm>
#include <cstdlib>

#include "mic_internal.h"

namespace mic {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHistWaves = 16;
constexpr int kChunks = 4;                      // 16-byte loads a lane keeps in flight per trip
constexpr size_t kTripPx = (size_t)256 * kChunks;  // pixels one wave takes per trip (4 KiB)
constexpr unsigned kMaxBlocks = 256;  // one block of 16 waves per CU (round 3 sweep: 512 -> 256: 4K 17.2 -> 16.3 us, 8K 32.2 -> 31.3; 384 / 320 / 192 leave CUs unevenly loaded: 8K 36-41 us)
// LDS histogram replicas, interleaved so the kCopies words of one bin sit in kCopies different banks:
// word = bin_index * kCopies + (lane & (kCopies - 1)).  64 lanes hitting one bin serialise 64/kCopies
// deep instead of 64.
constexpr int kCopies = 8;

// scratch layout of one image's slot (uint32): kGlobalCopies x { [set][channel][256], counts[2] } (set 0 = alpha > 0, set 1 =
// alpha == 0), then the retirement ticket.  Zero on entry (the scratch is a double buffer: a call clears the half the
// previous call used).  Block b flushes into copy b % 8 -- the
// XCD it runs on, as workgroups are dealt round-robin -- so that a bin's same-address atomics (they execute
// one after another at the memory side, ~12 ns each) are spread over eight addresses: a 4K image's 506
// blocks put 63 adds on an address instead of 506.  The last block sums the copies.
constexpr int kSetWords = 3 * 256;
constexpr int kCountOff = 2 * kSetWords;
constexpr int kCopyWords = kCountOff + 16;  // counts[2] + padding to a 64-byte multiple
constexpr int kGlobalCopies = 8;
constexpr int kTicketOff = kGlobalCopies * kCopyWords;
static_assert(kTicketOff < (int)kMedianSlotWords, "median scratch slot too small");
static_assert(kHistWaves == 16, "median_select maps 3 channels x 256 bins onto 1024 threads");

// Small grids (a bundle background is 15 blocks) keep one copy: the contention the copies relieve is not there,
// and the last block would pay 8x the loads and 8x the re-zeroing for nothing.
__device__ __host__ inline int median_copies(unsigned blocks) { return blocks > 32u ? kGlobalCopies : 1; }

__device__ inline uint32_t agent_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One chunk: the 4 pixels each lane of the wave holds (256 consecutive pixels per wave).  ok[] marks
// the pixels inside the image; lane 0's p[0] is always one of them.
//
// Same-address LDS atomics serialise, and backgrounds are mostly flat: left alone, 256 pixels of one
// colour cost 768 serialised adds.  The budget at HBM speed is ~120 VALU instructions and ~120 LDS
// cycles per chunk, so the remedy has to be cheap: per channel, a lane whose 4 pixels all equal the
// leader pixel (lane 0's first) in that channel and alpha class is "flat"; flat lanes are counted
// with one __ballot/popcount and added once by lane 0, and only the other lanes issue LDS atomics.
// A flat chunk costs 3 adds, flat-with-detail a few lanes' worth, noise what the plain loop costs.
template <bool FULL>
__device__ inline void hist_chunk(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, bool ok0, bool ok1, bool ok2,
                                  bool ok3, uint32_t *lh, int lane, uint32_t &n_opaque, uint32_t &n_clear) {
    const uint32_t p[4] = {p0, p1, p2, p3};
    const bool ok[4] = {ok0, ok1, ok2, ok3};
    const uint32_t copy = (uint32_t)lane & (kCopies - 1);
    const uint32_t lp = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);  // the leader pixel
    const uint32_t lset = (lp >> 24) != 0u ? 0u : 1u;
    uint32_t cls[4];  // histogram set of each pixel: 0 = alpha > 0, 1 = alpha == 0
    uint32_t cnt = 0, clr = 0;
    // diff: OR of (pixel ^ leader) over the lane's pixels; bit 24 is replaced by "alpha class differs
    // from the leader's", bits 25-31 are dropped.  A ragged chunk (!FULL) takes no shortcut.
    uint32_t diff = FULL ? 0u : ~0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cls[j] = (p[j] >> 24) != 0u ? 0u : 1u;
        cnt += (FULL || ok[j]) ? 1u : 0u;
        clr += (FULL || ok[j]) ? cls[j] : 0u;
        diff |= ((p[j] ^ lp) & 0x00ffffffu) | ((cls[j] ^ lset) << 24);
    }
    n_opaque += cnt - clr;
    n_clear += clr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sh = 8 * c;
        const bool flat = (diff & ((255u << sh) | (1u << 24))) == 0u;
        const unsigned long long fb = __ballot(flat);
        if (fb != 0 && lane == 0)
            atomicAdd(&lh[((lset * 3 + c) * 256 + ((lp >> sh) & 255u)) * kCopies], 4u * (uint32_t)__popcll(fb));
        if (fb == ~0ull) continue;  // wave-uniform: the whole chunk is flat in this channel
        if (!flat) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (FULL || ok[j])
                    atomicAdd(&lh[((cls[j] * 3 + c) * 256 + ((p[j] >> sh) & 255u)) * kCopies + copy], 1u);
        }
    }
}

// The kernel's work for ONE image (blockIdx.y picks it from the batch).  `stride` = pixels between rows; == w for a
// packed image (the fast path: the image is one linear stream), anything else is a strided VIEW (fill_gradient's
// edge strips of a resident background: 8 px wide, a whole image row apart) and takes the guarded per-pixel path.
struct MedianImage {
    const uint32_t *px;
    uint32_t *hist;      // this image's scratch (kMedianSlotWords, zero on entry)
    uint32_t *out_rgba;  // 4 bytes r, g, b, 255
    uint64_t n_px;
    int32_t w, stride;
    uint32_t blocks;     // workgroups that work on this image (grid.x may be larger: other images of the batch)
    uint32_t copies;     // global histogram copies in use (1 or kGlobalCopies)
};
static_assert(sizeof(MedianImage) == 48, "MedianImage layout");
struct MedianBatch {
    MedianImage img[kMedianMaxBatch];
    uint32_t *zero_ptr;   // scratch the PREVIOUS call used (the other half of the double buffer): re-zeroed here, in the
    uint32_t zero_words;  // shadow of this call's first loads, instead of at the end of that call's critical path
    uint32_t pad;
};

// Last block of an image, 1024 threads: thread t carries bin (t & 255) of channel (t >> 8) (the fourth quarter
// idles).  All global loads are issued up front -- they are agent-scope, a memory round trip each --
// then one 256-bin scan per channel picks the order statistics (n-1)//2 and n//2 and thread 0 writes
// int((lo + hi) / 2): np.median's mean of the two middle values, truncated by int().
__device__ void median_select(const uint32_t *hist, int copies, uint32_t *out_rgba, uint32_t *wave_tot,
                              uint32_t (*res)[2]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = t >> 8, bin = t & 255;
    const bool act = c < 3;
    uint32_t n0 = 0, n1 = 0, v0 = 0, v1 = 0;
#pragma unroll
    for (int g = 0; g < kGlobalCopies; ++g) {  // up to 32 loads issued together: one round trip
        if (g >= copies) break;
        const uint32_t *h = hist + g * kCopyWords;
        n0 += agent_load(h + kCountOff);
        n1 += agent_load(h + kCountOff + 1);
        v0 += act ? agent_load(h + (0 * 3 + c) * 256 + bin) : 0u;
        v1 += act ? agent_load(h + (1 * 3 + c) * 256 + bin) : 0u;
    }
    const uint32_t n = n0 != 0 ? n0 : n1;
    if (n == 0) {  // empty image (block-uniform)
        if (t == 0) out_rgba[0] = 0xff000000u;
        return;
    }
    const uint32_t v = n0 != 0 ? v0 : v1;
    const uint32_t klo = (n - 1) / 2, khi = n / 2;
    uint32_t incl = v;  // inclusive scan within the wave
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (act) {
        for (int w = c * 4; w < wave; ++w) incl += wave_tot[w];
        const uint32_t excl = incl - v;
        if (excl <= klo && klo < incl) res[c][0] = (uint32_t)bin;
        if (excl <= khi && khi < incl) res[c][1] = (uint32_t)bin;
    }
    __syncthreads();
    if (t == 0) {
        const uint32_t r = (res[0][0] + res[0][1]) >> 1;
        const uint32_t g = (res[1][0] + res[1][1]) >> 1;
        const uint32_t b = (res[2][0] + res[2][1]) >> 1;
        out_rgba[0] = r | (g << 8) | (b << 16) | 0xff000000u;
    }
}

// SELECT: true = the one-launch form (the block that retires last selects); false = the histogram half of the
// two-launch form (median_select_kernel follows on the stream).
// STRIDED: some image of the batch is a strided view (the instantiation for packed images carries none of the
// per-pixel index arithmetic).
template <bool SELECT, bool STRIDED>
__global__ __launch_bounds__(64 * kHistWaves) void median_kernel(const MedianBatch B) {
    __shared__ uint32_t lh[2 * kSetWords * kCopies];
    __shared__ uint32_t lcount[2];
    __shared__ uint32_t wave_tot[kHistWaves];
    __shared__ uint32_t res[3][2];
    __shared__ uint32_t is_last;
    const MedianImage &I = B.img[blockIdx.y];
    // The other half of the scratch double buffer -- what the previous call on this context used (stream order: it has
    // finished) -- is cleared by EVERY block of the grid, the ones without pixels of their own included, 16 bytes per
    // store: after a 16-image batch (0.8 MB) a small single-image call would otherwise zero all of it from one block,
    // in front of its first barrier.  Plain stores; the kernel boundary publishes them to the next call.
    auto clear_previous_half = [&]() {
        if (!B.zero_words) return;
        const uint32_t n_thr = gridDim.x * gridDim.y * blockDim.x, n16 = B.zero_words >> 2;  // (slots are multiples of 16 bytes)
        u32x4 *z = reinterpret_cast<u32x4 *>(B.zero_ptr);
        for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += n_thr) z[i] = u32x4{0u, 0u, 0u, 0u};
    };
    if (blockIdx.x >= I.blocks) {  // (block-uniform; such a block takes no ticket)
        clear_previous_half();
        return;
    }
    const uint32_t *__restrict__ px = I.px;
    const size_t n_px = I.n_px;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t n_opaque = 0, n_clear = 0;
    const size_t stride = (size_t)I.blocks * kHistWaves * kTripPx;
    size_t wbase = ((size_t)blockIdx.x * kHistWaves + wave) * kTripPx;
    const bool packed = !STRIDED || I.stride == I.w;  // block-uniform
    // the first trip's loads are issued before the LDS histogram is cleared: the clear (and the barrier
    // behind it) then runs in the shadow of the first memory round trip instead of in front of it
    u32x4 ld[kChunks];
    const bool first_whole = packed && wbase + kTripPx <= n_px;  // wave-uniform
    if (first_whole) {
#pragma unroll
        for (int u = 0; u < kChunks; ++u)
            __builtin_memcpy(&ld[u], px + wbase + (size_t)u * 256 + (size_t)lane * 4, 16);
    }
    for (int i = threadIdx.x; i < 2 * kSetWords * kCopies; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x < 2) lcount[threadIdx.x] = 0;
    clear_previous_half();  // in the shadow of the loads above
    __syncthreads();

    if (packed) {
        while (wbase + kTripPx <= n_px) {  // whole trips: wave-uniform, no guards
#pragma unroll
            for (int u = 0; u < kChunks; ++u) {
                hist_chunk<true>(ld[u][0], ld[u][1], ld[u][2], ld[u][3], true, true, true, true, lh, lane, n_opaque, n_clear);
            }
            wbase += stride;
            if (wbase + kTripPx <= n_px) {
#pragma unroll
                for (int u = 0; u < kChunks; ++u)
                    __builtin_memcpy(&ld[u], px + wbase + (size_t)u * 256 + (size_t)lane * 4, 16);
            }
        }
    }
    for (; wbase < n_px; wbase += stride) {  // packed: the image's ragged last trip (one wave of the grid); a strided view: all of it
        for (int u = 0; u < kChunks; ++u) {
            const size_t cbase = wbase + (size_t)u * 256;
            if (cbase >= n_px) break;
            const size_t i = cbase + (size_t)lane * 4;
            uint32_t p[4];
            bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ok[j] = i + j < n_px;
                size_t at = i + j;
                if (!packed) at = (at / (uint32_t)I.w) * (size_t)I.stride + (at % (uint32_t)I.w);
                p[j] = ok[j] ? px[at] : 0u;
            }
            hist_chunk<false>(p[0], p[1], p[2], p[3], ok[0], ok[1], ok[2], ok[3], lh, lane, n_opaque, n_clear);
        }
        if (packed) break;
    }
    // wave-level count reductions (shuffle), one LDS add per wave and set
    for (int off = 32; off > 0; off >>= 1) {
        n_opaque += __shfl_down(n_opaque, off);
        n_clear += __shfl_down(n_clear, off);
    }
    if (lane == 0) {
        if (n_opaque) atomicAdd(&lcount[0], n_opaque);
        if (n_clear) atomicAdd(&lcount[1], n_clear);
    }
    __syncthreads();

    uint32_t *hist = I.hist;
    uint32_t *hist_copy = hist + (blockIdx.x & (I.copies - 1)) * kCopyWords;
    for (int i = threadIdx.x; i < 2 * kSetWords; i += blockDim.x) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) s += lh[i * kCopies + ((k + threadIdx.x) & (kCopies - 1))];
        if (s) atomicAdd(&hist_copy[i], s);
    }
    if (threadIdx.x < 2 && lcount[threadIdx.x]) atomicAdd(&hist_copy[kCountOff + threadIdx.x], lcount[threadIdx.x]);
    if (!SELECT) return;  // two-launch form: the kernel boundary is the hand-off

    // Retirement ticket, without fences.  Everything a block publishes is an agent-scope ATOMIC (performed at the
    // memory side: there are no plain stores whose dirty L2 lines a release would have to write back), and
    // everything the last block reads back is read with agent-scope (sc1) loads or is the value its own ticket
    // add returned -- the hand-off form MI355X_MICROARCH.md lists as valid without a release/acquire pair, given
    // that (1) every wave waits for its own atomics to be acknowledged (s_waitcnt vmcnt(0)) and (2) the ticket
    // add comes behind a workgroup barrier that all those waves have passed.  The release + acquire fences this
    // replaces (an L2 write-back and an L1 invalidate, ~1.7 us each) were a third of a small image's time.
    // INVARIANT (checked by tests/test_abi.py::test_median_scratch_is_only_touched_by_atomics): between the LDS clear
    // and this point nothing writes I.hist except atomicAdd; a plain store here would need the fences back.
    // -DMIC_MEDIAN_FENCES restores the release/acquire pair (bisecting aid).
#ifdef MIC_MEDIAN_FENCES
    __atomic_thread_fence(__ATOMIC_RELEASE);  // (agent scope is the default for HIP's fences)
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket =
            __hip_atomic_fetch_add(hist + kTicketOff, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == I.blocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
#ifdef MIC_MEDIAN_FENCES
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
    median_select(hist, (int)I.copies, I.out_rgba, wave_tot, res);
    // (the scratch is NOT re-zeroed here: the next call uses the other half of the double buffer and clears this one
    // while its own first loads are in flight)
}

// Second launch of the two-launch form: one block per image.
__global__ __launch_bounds__(64 * kHistWaves) void median_select_kernel(const MedianBatch B) {
    __shared__ uint32_t wave_tot[kHistWaves];
    __shared__ uint32_t res[3][2];
    const MedianImage &I = B.img[blockIdx.x];
    median_select(I.hist, (int)I.copies, I.out_rgba, wave_tot, res);
}

// scratch_dev: 2 * kMedianMaxBatch slots of kMedianSlotWords (a double buffer), zero on entry of the first call; the
// launcher alternates halves with *phase and hands the half used by the previous call to this one for clearing.
hipError_t launch_median_batch(int k, const MedianView *views, uint32_t *const *out_rgba_dev, uint32_t *scratch_dev,
                               MedianState *state, int two_launches, hipStream_t stream) {
    if (k <= 0 || k > kMedianMaxBatch) return hipErrorInvalidValue;
    MedianBatch B{};
    const uint32_t half = (uint32_t)kMedianMaxBatch * (uint32_t)kMedianSlotWords;
    uint32_t *mine = scratch_dev + (size_t)(state->phase & 1) * half;
    unsigned grid_x = 1;
    for (int i = 0; i < k; ++i) {
        const MedianView &v = views[i];
        MedianImage &I = B.img[i];
        I.px = static_cast<const uint32_t *>(v.px);
        I.hist = mine + (size_t)i * kMedianSlotWords;
        I.out_rgba = out_rgba_dev[i];
        I.n_px = (uint64_t)v.w * v.h;
        I.w = v.w;
        I.stride = v.stride_px;
        // one trip (4 KiB) per wave before a block takes a second one, up to one block per CU: a 4K image runs on
        // 256 blocks x 2 trips, a 492 x 492 bundle background on 15 blocks
        const size_t per_block = kTripPx * kHistWaves;  // pixels per block per trip
        size_t blocks = (I.n_px + per_block - 1) / per_block;
        static const size_t max_blocks = [] {  // MIC_MEDIAN_MAX_BLOCKS: tuning knob (default kMaxBlocks)
            const char *e = getenv("MIC_MEDIAN_MAX_BLOCKS");
            const long v = e ? atol(e) : 0;
            return v > 0 && v <= (long)kMaxBlocks ? (size_t)v : (size_t)kMaxBlocks;
        }();
        blocks = std::min<size_t>(std::max<size_t>(blocks, 1), max_blocks);
        I.blocks = (uint32_t)blocks;
        I.copies = (uint32_t)median_copies((unsigned)blocks);
        grid_x = std::max(grid_x, (unsigned)blocks);
    }
    B.zero_ptr = scratch_dev + (size_t)((state->phase + 1) & 1) * half;
    B.zero_words = state->prev_words;
    bool strided = false;
    for (int i = 0; i < k; ++i) strided |= B.img[i].stride != B.img[i].w;
    const dim3 grid(grid_x, (unsigned)k), block(64 * kHistWaves);
    // The two-launch form (kernel boundary + a one-block select launch instead of the ticket + the last block's
    // read-back) is the measured alternative, MIC_MEDIAN_TWO_LAUNCHES=1: with one block per CU it is 0.6 us ahead at 4K
    // (15.7 vs 16.3 us), level at 8K (31.8 vs 31.3), 0.6-1.1 us behind at the bundles' sizes and at 1080p
    // (profiles/r03_median_experiments.txt) -- the one-launch form is the default everywhere.
    if (two_launches == 1) {
        if (strided) hipLaunchKernelGGL((median_kernel<false, true>), grid, block, 0, stream, B);
        else hipLaunchKernelGGL((median_kernel<false, false>), grid, block, 0, stream, B);
        hipLaunchKernelGGL(median_select_kernel, dim3((unsigned)k), block, 0, stream, B);
    } else {
        if (strided) hipLaunchKernelGGL((median_kernel<true, true>), grid, block, 0, stream, B);
        else hipLaunchKernelGGL((median_kernel<true, false>), grid, block, 0, stream, B);
    }
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) {
        state->phase ^= 1u;
        state->prev_words = (uint32_t)k * (uint32_t)kMedianSlotWords;  // slots 0..k-1 of the half just used
    }
    return e;
}

}  // namespace mic
