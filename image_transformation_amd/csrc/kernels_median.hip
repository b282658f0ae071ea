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
// that retires last (device-scope ticket) runs the 256-bin scan + select; the scratch is a double
// buffer that the NEXT call clears, so no call needs a memset or a second kernel.
//
// Shape: at most 256 blocks of 16 waves (one per CU), one LDS histogram per block (16 bank-interleaved
// replicas, 96 KiB).  A wave owns 4 KiB of the image per trip and issues its four 16-byte loads per
// lane back to back, unconditionally (a load under a branch makes hipcc wait for it before issuing
// the next); only the single ragged trip at the end of the image goes through a guarded per-pixel
// path.  Few, large blocks also keep the per-block costs small: the flush of non-empty bins
// (same-address global atomics from every block) and the ticket.
//
// Per pixel the loop is three LDS atomics and their addresses, plus one test per chunk of 256 pixels
// for "all one colour" (then lane 0 adds 256 to three bins).  Rounds 1-3 tested every chunk for
// "flat" lanes per channel (backgrounds are mostly flat, and 64 same-address LDS atomics serialise)
// and kept the pixel counts of both sets in registers: ~190 instructions per chunk, four waves per
// SIMD -- the kernel was bound by issuing them, not by memory or by the LDS
// (profiles/r04_median_experiments.txt: 8K photo-like 40 -> 26 us).  With 16 replicas a chunk of
// near-equal pixels queues 4 lanes deep per word at worst; the counts are the histograms' own totals.
#include <algorithm>
#include <cstdlib>

#include "mic_internal.h"

namespace mic {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHistWaves = 16;
constexpr int kChunks = 4;                      // 16-byte loads a lane keeps in flight per trip
constexpr size_t kTripPx = (size_t)256 * kChunks;  // pixels one wave takes per trip (4 KiB)
constexpr uint64_t kTwoLaunchPx = 6ull << 20;  // images from here up take the two-launch form (see launch_median_batch)
constexpr unsigned kMaxBlocks = 256;  // one block of 16 waves per CU (round 3 sweep: 512 -> 256: 4K 17.2 -> 16.3 us, 8K 32.2 -> 31.3; 384 / 320 / 192 leave CUs unevenly loaded: 8K 36-41 us)
// LDS histogram replicas, interleaved so the kCopies words of one bin sit in kCopies different banks:
// word = bin_index * kCopies + (lane & (kCopies - 1)).  64 lanes hitting one bin serialise 64/kCopies
// deep instead of 64 (round 4: 8 -> 16 replicas, 96 KiB).
constexpr int kCopies = 16;

// scratch layout of one image's slot (uint32): kGlobalCopies x { [set][channel][256] + padding } (set 0 = alpha > 0, set 1 =
// alpha == 0), then the retirement ticket.  (No pixel counts: a set's count is the total of any of its three histograms.)  Zero on entry (the scratch is a double buffer: a call clears the half the
// previous call used).  Block b flushes into copy b % 8 -- the
// XCD it runs on, as workgroups are dealt round-robin -- so that a bin's same-address atomics (they execute
// one after another at the memory side, ~12 ns each) are spread over eight addresses: a 4K image's 506
// blocks put 63 adds on an address instead of 506.  The last block sums the copies.
constexpr int kSetWords = 3 * 256;
constexpr int kCopyWords = 2 * kSetWords + 16;  // (+ padding: the copies start in different L2 channels)
static_assert(kCopyWords % 2 == 0, "bins are flushed in pairs, as one 64-bit atomic");
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
// the pixels inside the image (FULL: all of them).  Three LDS atomics per pixel into the lane's replica.
template <bool FULL>
__device__ inline void hist_chunk(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, bool ok0, bool ok1, bool ok2,
                                  bool ok3, uint32_t *lh, int lane) {
    const uint32_t p[4] = {p0, p1, p2, p3};
    const bool ok[4] = {ok0, ok1, ok2, ok3};
    uint32_t *mine = lh + ((uint32_t)lane & (kCopies - 1));
    if (FULL) {
        // a chunk of ONE colour (flat backgrounds are common): lane 0 adds 256 to its three bins instead of the wave
        // queueing four deep on 16 words.  One compare and one wave-uniform branch per chunk -- not the per-channel,
        // per-lane test of rounds 1-3, which cost more than the conflicts it avoided.
        const uint32_t lp = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0);
        if (__ballot(((p0 ^ lp) | (p1 ^ lp) | (p2 ^ lp) | (p3 ^ lp)) != 0u) == 0ull) {
            if (lane == 0) {
                uint32_t *set = lh + ((lp >> 24) != 0u ? 0 : kSetWords * kCopies);
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(&set[(c * 256 + ((lp >> (8 * c)) & 255u)) * kCopies], 256u);
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!(FULL || ok[j])) continue;
        uint32_t *set = mine + ((p[j] >> 24) != 0u ? 0 : kSetWords * kCopies);  // 0 = alpha > 0, 1 = alpha == 0
#pragma unroll
        for (int c = 0; c < 3; ++c) atomicAdd(&set[(c * 256 + ((p[j] >> (8 * c)) & 255u)) * kCopies], 1u);
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
// idles).  Set 0 (alpha > 0) first: its copies are loaded together -- agent-scope loads, one memory round trip --
// and one 256-bin scan per channel gives both the set's pixel count (the scan's total) and the order statistics
// (n-1)//2 and n//2; only an image without a single pixel of alpha > 0 goes round again for set 1.  Thread 0
// writes int((lo + hi) / 2): np.median's mean of the two middle values, truncated by int().
__device__ void median_select(const uint32_t *hist, int copies, uint32_t *out_rgba, uint32_t *wave_tot,
                              uint32_t (*res)[2]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = t >> 8, bin = t & 255;
    const bool act = c < 3;
    uint32_t v = 0, incl = 0, n = 0;
    for (int set = 0; set < 2; ++set) {
        v = 0;
        if (act) {
            const uint32_t *h = hist + (set * 3 + c) * 256 + bin;
            if (copies == kGlobalCopies) {  // (block-uniform) all eight loads in flight together: no branch between them --
                uint32_t x[kGlobalCopies];   // a loop with an early exit made hipcc wait for every load before the next
#pragma unroll
                for (int g = 0; g < kGlobalCopies; ++g) x[g] = agent_load(h + g * kCopyWords);
                // (hipcc keeps atomic loads in program order and, left alone, sinks the additions between them: load,
                // load, wait, add, ...  The empty asm needs all eight values at once, so the only wait sits behind the eighth.)
                static_assert(kGlobalCopies == 8, "the asm below names eight registers");
                asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#pragma unroll
                for (int g = 0; g < kGlobalCopies; ++g) v += x[g];
            } else {
                for (int g = 0; g < copies; ++g) v += agent_load(h + g * kCopyWords);
            }
        }
        incl = v;  // inclusive scan within the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        n = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];  // channel 0's total = the set's pixel count
        if (n != 0u || set == 1) break;                             // (block-uniform)
        __syncthreads();                                            // wave_tot is written again
    }
    if (n == 0) {  // empty image (block-uniform)
        if (t == 0) out_rgba[0] = 0xff000000u;
        return;
    }
    const uint32_t klo = (n - 1) / 2, khi = n / 2;
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
    __shared__ __attribute__((aligned(16))) uint32_t lh[2 * kSetWords * kCopies];
    __shared__ uint32_t wave_tot[kHistWaves];
    __shared__ uint32_t res[3][2];
    __shared__ uint32_t is_last;
    const MedianImage &I = B.img[blockIdx.y];
    // The other half of the scratch double buffer -- what the previous call on this context used (stream order: it has
    // finished) -- is cleared by EVERY block of the grid, the ones without pixels of their own included, 16 bytes per
    // store: after a 16-image batch (0.8 MB) a small single-image call would otherwise zero all of it from one block,
    // in front of its first barrier.  Plain stores; the kernel boundary publishes them to the next call.
    // (the block size as a constant: blockDim.x is a VECTOR load from the dispatch packet followed by s_waitcnt vmcnt(0),
    // which would also wait for the image loads already in flight)
    constexpr uint32_t kThreads = 64 * kHistWaves;
    auto clear_previous_half = [&]() {
        if (!B.zero_words) return;
        const uint32_t n_thr = gridDim.x * gridDim.y * kThreads, n16 = B.zero_words >> 2;  // (slots are multiples of 16 bytes)
        u32x4 *z = reinterpret_cast<u32x4 *>(B.zero_ptr);
        for (uint32_t i = (blockIdx.y * gridDim.x + blockIdx.x) * kThreads + threadIdx.x; i < n16; i += n_thr) z[i] = u32x4{0u, 0u, 0u, 0u};
    };
    if (blockIdx.x >= I.blocks) {  // (block-uniform; such a block takes no ticket)
        clear_previous_half();
        return;
    }
    const uint32_t *__restrict__ px = I.px;
    const size_t n_px = I.n_px;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: trip bases and loop exits are wave-uniform)
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
    for (int i = threadIdx.x; i < 2 * kSetWords * kCopies / 4; i += kThreads) reinterpret_cast<u32x4 *>(lh)[i] = u32x4{0u, 0u, 0u, 0u};
    clear_previous_half();  // in the shadow of the loads above
    __syncthreads();

    if (packed) {
        while (wbase + kTripPx <= n_px) {  // whole trips: wave-uniform, no guards
#pragma unroll
            for (int u = 0; u < kChunks; ++u) {
                hist_chunk<true>(ld[u][0], ld[u][1], ld[u][2], ld[u][3], true, true, true, true, lh, lane);
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
            hist_chunk<false>(p[0], p[1], p[2], p[3], ok[0], ok[1], ok[2], ok[3], lh, lane);
        }
        if (packed) break;
    }
    __syncthreads();

    // Flush: thread i adds bins 2i and 2i + 1 (of the 1536: [set][channel][256]) to the block's global copy with ONE
    // 64-bit atomic -- the low word cannot carry into the high one, a bin holds at most the image's pixel count -- so a
    // block sends at most 768 atomics instead of 1536 (a 4K noise image: 256 blocks x all bins).
    uint32_t *hist = I.hist;
    uint32_t *hist_copy = hist + (blockIdx.x & (I.copies - 1)) * kCopyWords;
    if (threadIdx.x < kSetWords) {
        // the pair's 32 replica words as eight 16-byte reads, each thread starting at its own chunk: any eight
        // neighbouring lanes then cover all 32 banks once
        static_assert(kCopies == 16, "two bins x 16 replicas = eight 16-byte chunks");
        const u32x4 *row = reinterpret_cast<const u32x4 *>(lh + 2 * threadIdx.x * kCopies);
        uint32_t s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = (k + threadIdx.x) & 7;
            const u32x4 q = row[c];
            const uint32_t sum = q[0] + q[1] + q[2] + q[3];
            s0 += c < 4 ? sum : 0u;
            s1 += c < 4 ? 0u : sum;
        }
        if (s0 | s1)
            atomicAdd(reinterpret_cast<unsigned long long *>(&hist_copy[2 * threadIdx.x]), (unsigned long long)s0 | ((unsigned long long)s1 << 32));
    }
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
    // (profiles/r05_tuning_scaffolding.patch brings the release / acquire pair back behind -DMIC_MEDIAN_FENCES: a bisecting aid)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket =
            __hip_atomic_fetch_add(hist + kTicketOff, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == I.blocks - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
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
    // read-back): ahead once every CU has a block -- 256 blocks' tickets and flushes queue at the memory side -- 4K 10.0
    // against 11.3 us, level at 8K (24.7 / 23.8), behind below that (1080p 9.4 / 7.8, bundle backgrounds 8.6 / 6.4;
    // profiles/r04_median_experiments.txt).  two_launches < 0 picks by size; MIC_MEDIAN_TWO_LAUNCHES=1 / 0 force one form.
    bool two = two_launches == 1;
    if (two_launches < 0) {
        two = true;
        for (int i = 0; i < k; ++i) two &= B.img[i].n_px >= kTwoLaunchPx;
    }
    if (two) {
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
