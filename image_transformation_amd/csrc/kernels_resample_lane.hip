// Pillow-exact LANCZOS / BILINEAR resize of atlas cutouts on the matrix cores, one WAVE per work unit (round 5).
//
// Replaces, for calls big enough to fill the chip, the marching kernel of kernels_resample.hip (Image.resize at
// compositor.py:20).  Same arithmetic -- premultiplied signed-byte planes, taps as three signed-byte digits chained
// through the accumulator of v_mfma_i32_16x16x64_i8, Pillow's 8-bit intermediate between the passes (resample_mfma.h)
// -- in another structure, built around what three rounds of measurements said bounds that kernel: instruction issue
// (217 vector + 119 scalar instructions per wave and 16-row band besides its 24 MFMAs), two workgroup barriers per band
// and a three-deep chain of dependent loads in front of every unit.
//
//   * The source is the cutout's TILED planar copy (planarize_tiled_kernel): four planes of premultiplied signed
//     bytes, each cut into tiles of 16 rows x 16 columns (256 B, row-major inside), tiles ordered band by band.  The
//     64-column window of a band -- the A operand of the horizontal pass: lane (row = l & 15, quarter = l >> 4) holds
//     16 consecutive columns -- is then 1 KiB of CONSECUTIVE bytes: one fully coalesced global_load_dwordx4 per
//     channel, straight into the MFMA operand registers.  No LDS staging, no barrier, no per-lane conditions (the copy
//     is padded to whole tiles, and a tap that does not exist is a zero digit, so what the padding holds never matters).
//   * A wave owns T (1 or 2) adjacent tiles of 16 output columns whose taps fit ONE such window, and a run of tiles
//     of 16 output rows.  Both x-tiles share the window loads; their horizontal taps come from LDS (the four waves of
//     a workgroup work on the same columns, in different rows).
//   * The horizontal pass leaves, per lane, one word = four consecutive rows of one intermediate column (the D layout
//     of the MFMA).  Four such words -- the last four bands -- ARE the A operand of the vertical pass (16 bytes per
//     lane, 64 window rows per wave): the ring of intermediate rows is 16 registers per x-tile, nothing is written to
//     or read from LDS.  Which k position a byte sits at is the business of the host-built tap fragments: band b
//     lives in ring word b & 3 for good, and the vertical fragments are laid out to match (resample_coeffs.cpp:
//     fill_axis_frags_ring).  The band loop is unrolled four times so that the ring word is a compile-time index.
//   * One 128-byte record per wave (scalar loads), from which every other first load is issued at once: the prologue is
//     two dependent round trips, not four.
// Bit-exact with the two-pass kernels of kernels_resample.hip (and so with Pillow): tests/test_gpu_lane.py.
#include <algorithm>
#include <atomic>

#include "mic_internal.h"
#include "resample_mfma.h"

namespace mic {

namespace {

template <int Q>
struct Phase {
    static constexpr int q = Q;
};

// A wave-uniform value the compiler must keep in a scalar register from here on (it would otherwise re-load fields of
// the unit record inside the loops, each time behind an s_waitcnt).
template <class V>
__device__ __forceinline__ V pinned(V v) {
    asm volatile("" : "+s"(v));
    return v;
}

// One piece: march bands band0 .. band_last of the window columns, emit the piece's tiles of output rows.
// hf_lds: this WAVE's own LDS region for the horizontal tap fragments of the piece's x-tiles.
// HD / VD: tap digits of the horizontal / vertical pass -- 3, or 1 for an axis that keeps its size (tile4: the last digit
// alone, no floor shifts; that one MFMA per channel is also the transposition between the passes' operand layouts).
template <int T, int HD, int VD>
__device__ __forceinline__ void lane_run(const RsLaneUnit &U, v4i *hf_lds, const float *recip, const int lane) {
    static_assert((HD == 3 || HD == 1) && (VD == 3 || VD == 1) && HD + VD > 2, "identity forms");
    const int l15 = lane & 15, lh = lane >> 4;
    const uint32_t lane16 = (uint32_t)lane * 16u, l15x4 = (uint32_t)l15 * 4u;
    const int band0 = pinned(U.band0), band_last = pinned(U.band_last), n_vt = pinned(U.n_vtiles);
    const uint32_t band_bytes = pinned(U.band_bytes);
    const uint64_t p0 = pinned(U.src), p1 = pinned(p0 + U.plane_bytes), p2 = pinned(p1 + U.plane_bytes), p3 = pinned(p2 + U.plane_bytes);
    const uint64_t vfrag = pinned(U.vfrag), vbias = pinned(U.vbias);
    // (constant address space: the table entries are wave-uniform and must come through SCALAR loads -- as a vector load
    // the next tile's entry was waited for with vmcnt(0) at the head of every tile, behind the band loads just issued)
    typedef const __attribute__((address_space(4))) int32_t *sciptr;
    sciptr vemit = reinterpret_cast<sciptr>(pinned(U.vemit));
    // every first load of the piece, issued together: the first band, horizontal fragments, biases, the first tile's taps
    uint32_t voff = lane16;  // this lane's 16 bytes of the next band to load (same offset in every plane)
    // ONE set of operand registers: the next band is requested as soon as the horizontal pass has issued its last
    // MFMA on this one, and lands behind the vertical tiles of the step (a second set costs 16 registers: spills)
    v4i A[4];
    auto load_band = [&]() __attribute__((always_inline)) {
        A[0] = *at<v4i>(p0, voff);
        A[1] = *at<v4i>(p1, voff);
        A[2] = *at<v4i>(p2, voff);
        A[3] = *at<v4i>(p3, voff);
        voff += band_bytes;
    };
    load_band();
    v4i hfc[3 * T];
    {
        gv4ptr hfb = at<v4i>(U.hfrag, lane16);
#pragma unroll
        for (int i = 0; i < 3 * T; ++i) hfc[i] = hfb[64 * i];
    }

    // horizontal bias per x-tile (- 128 << 22: the horizontal pass clips to signed bytes, see clip8x4_signed)
    int hb_raw[T];
#pragma unroll
    for (int j = 0; j < T; ++j) hb_raw[j] = *at<int32_t>(U.hbias, (uint32_t)(j * 64) + l15x4);
    // vertical taps of the next tile to emit wait in registers (one set, reloaded in place once its last MFMA is out)
    v4i vf[3];
    int vb = 0;
    auto fetch_taps = [&](int t) __attribute__((always_inline)) {
        vb = *at<int32_t>(vbias, (uint32_t)t * 64u + l15x4);
        gv4ptr base = at<v4i>(vfrag, (uint32_t)t * 3072u + lane16);
        if (VD == 3) { vf[0] = base[0]; vf[1] = base[64]; }
        vf[2] = base[128];
    };
    fetch_taps(0);
    int yt = 0;
    int emit = vemit[0];  // (uniform: scalar load; entries are 4 ints apart: the axis table's meta rows) band after which tile yt can go | ring words it reads << 24

    // the horizontal fragments -> this wave's LDS region (wave-private: LDS operations of one wave execute in order)
#pragma unroll
    for (int i = 0; i < 3 * T; ++i) hf_lds[i * 64 + lane] = hfc[i];

    int hbias1[T];  // (one register per x-tile; the four-register C operand is rebuilt at each use)
#pragma unroll
    for (int j = 0; j < T; ++j) hbias1[j] = hb_raw[j] - (128 << 22);
    v4i ring[T][4];
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) ring[j][c] = v4i{(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
    uint32_t zmask = 0xFu;  // bit q: ring word q holds premultiplied zeros (an all-transparent band, or nothing yet)

    gptr dst = reinterpret_cast<gptr>(pinned(U.dst));
    const int dw = pinned(U.dw), dh = pinned(U.dh), x0 = pinned(U.x0), urow0 = pinned(U.row0);
    const uint32_t lane_idx = (uint32_t)(l15 * dw + x0 + 4 * lh);
    bool x_full[T], ox_ok[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        x_full[j] = x0 + 16 * j + 16 <= dw;       // wave-uniform: every lane's four pixels are inside the row
        ox_ok[j] = x0 + 16 * j + 4 * lh < dw;
    }

    auto step = [&](auto PH, const int b) __attribute__((always_inline)) {
        constexpr int q = decltype(PH)::q;
        if (b < band0 || b > band_last) return;  // (wave-uniform)
        const v4i a[4] = {A[0], A[1], A[2], A[3]};
        // an all-transparent window (the corners around a cutout's shape): premultiplied zeros in, zeros out
        const int k80 = (int)0x80808080u;
        const uint32_t seen = (uint32_t)((a[3][0] ^ k80) | (a[3][1] ^ k80) | (a[3][2] ^ k80) | (a[3][3] ^ k80));
        if (!__any(seen != 0u)) {
            zmask |= 1u << q;
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) ring[j][c][q] = k80;
        } else {
            zmask &= ~(1u << q);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const v4i hf[3] = {hf_lds[(j * 3 + 0) * 64 + lane], hf_lds[(j * 3 + 1) * 64 + lane], hf_lds[(j * 3 + 2) * 64 + lane]};
                uint32_t w[4];
                int hb = hbias1[j];
                asm volatile("" : "+v"(hb));
                tile4<true, HD>([&](int c) { return a[c]; }, hf, v4i{hb, hb, hb, hb}, w);
#pragma unroll
                for (int c = 0; c < 4; ++c) ring[j][c][q] = (int)w[c];
            }
        }
        if (b < band_last) load_band();
        // tiles of output rows whose last tap row is now in the ring
        while (yt < n_vt && (emit & 0xFFFFFF) == b) {
            const uint32_t need = (uint32_t)emit >> 24;
            const bool all_zero = (zmask & need) == need;
            const int t_n = min(yt + 1, n_vt - 1);  // (the last tile re-reads its own entries: always a valid fetch)
            const int emit_n = vemit[4 * t_n];
            const int row0 = urow0 + 16 * yt;
            const bool row_ok = l15 < dh - row0;
            const uint32_t o_idx = (uint32_t)(row0 * dw) + lane_idx;
            u32x4 px[T];
#pragma unroll
            for (int j = 0; j < T; ++j) {
                px[j] = u32x4{0u, 0u, 0u, 0u};
                if (!all_zero) {
                    uint32_t w[4];
                    tile4<false, VD>([&](int c) { return ring[j][c]; }, vf, v4i{vb, vb, vb, vb}, w);
                    if (j == T - 1) fetch_taps(t_n);
                    // alpha bytes all 0 or 255 <=> low 7 bits of every byte equal its top bit
                    const uint32_t top = (w[3] >> 7) & 0x01010101u;
                    const bool soft = (w[3] & 0x7F7F7F7Fu) != (top << 7) - top;
                    px[j] = __any(soft) ? unpremultiply4(w, recip) : interleave4(w);
                } else if (j == T - 1) {
                    fetch_taps(t_n);
                }
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const bool inside = row_ok && ox_ok[j];
                if (x_full[j]) {  // (wave-uniform) one 16-byte store per lane
                    if (inside) *reinterpret_cast<MIC_GLOBAL u32x4_a4 *>(dst + o_idx + 16 * j) = px[j];
                } else if (inside) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x0 + 16 * j + 4 * lh + i < dw) dst[o_idx + 16 * j + i] = px[j][i];
                }
            }
            ++yt;
            emit = emit_n;
        }
    };
    for (int b4 = band0 & ~3; b4 <= band_last; b4 += 4) {
        step(Phase<0>{}, b4);
        step(Phase<1>{}, b4 + 1);
        step(Phase<2>{}, b4 + 2);
        step(Phase<3>{}, b4 + 3);
    }
}

}  // namespace

// One wave = one slot of the launch: it works through its own short chain of pieces (usually one; two where the host's
// equal-cost cut falls across the end of a column strip): record `slot`, then its `next`s.  The four waves of a
// workgroup share the unpremultiply table and nothing else: one barrier, before any memory is touched.
__global__ __launch_bounds__(256, MIC_RS_LANE_WAVES) void resample_lane_kernel(const RsLaneUnit *__restrict__ units) {
    __shared__ v4i hf_lds[4][2 * 3 * 64];  // [wave][x-tile][digit][lane] 16 bytes
    __shared__ float recip[256];           // unpremultiply factors 255 / a: an LDS read per soft pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = (int)blockIdx.x * 4 + wave;
    recip[tid] = (tid == 0 || tid == 255) ? 1.0f : unpremul_factor((uint32_t)tid);
    __syncthreads();
    uint32_t r = (uint32_t)slot;
    do {
        const RsLaneUnit &U = units[r];
        r = U.next;
        if (U.n_vtiles > 0) {
            // (wave-uniform; the forms for layers that keep one axis are code a launch without such layers never fetches)
            if (U.cls == kLaneGeneral) {
                if (U.T == 1)
                    lane_run<1, 3, 3>(U, hf_lds[wave], recip, lane);
                else
                    lane_run<2, 3, 3>(U, hf_lds[wave], recip, lane);
            } else if (U.cls == kLaneKeepsWidth) {
                if (U.T == 1)
                    lane_run<1, 1, 3>(U, hf_lds[wave], recip, lane);
                else
                    lane_run<2, 1, 3>(U, hf_lds[wave], recip, lane);
            } else {
                if (U.T == 1)
                    lane_run<1, 3, 1>(U, hf_lds[wave], recip, lane);
                else
                    lane_run<2, 3, 1>(U, hf_lds[wave], recip, lane);
            }
        }
    } while (r != 0u);
}

hipError_t launch_resample_lane(const RsLaneUnit *units_dev, int n_slots, hipStream_t stream) {
    if (n_slots <= 0) return hipSuccess;  // (n_slots: a multiple of 4; records [0, n_slots) are the slots' first pieces)
    hipLaunchKernelGGL(resample_lane_kernel, dim3((unsigned)(n_slots / 4)), dim3(256), 0, stream, units_dev);
    return hipGetLastError();
}

// ---- tiled planar copy of an atlas cutout -------------------------------------------------------------------------
// Four planes (R, G, B, A premultiplied, stored as signed bytes s - 128) of `bands` x `ct` tiles of 16 rows x 16 columns
// (256 bytes, row-major inside a tile), tiles ordered band by band; columns >= w and rows >= h hold premultiplied
// zeros.  ct = ceil(w / 16) + 3, so that the 4-tile window that starts at any tile of a row stays inside the band.
// One thread = one row of one tile (16 pixels in, 16 bytes per plane out); consecutive threads are consecutive rows of a
// tile, so 16 threads write the tile's 256 bytes of a plane back to back.
__global__ __launch_bounds__(256) void planarize_tiled_kernel(const PlanarJob *__restrict__ jobs) {
    const PlanarJob J = jobs[blockIdx.y];
    const int ct = J.pitch >> 4;           // tiles per band (pitch = 16 ct bytes of columns)
    const int bands = (J.h + 15) >> 4;
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= (int64_t)bands * ct * 16) return;
    const int row = (int)(item & 15), tile = (int)(item >> 4);
    const int band = tile / ct, tx = tile - band * ct;
    const int y = 16 * band + row, x0 = 16 * tx;
    gcptr src = reinterpret_cast<gcptr>(J.src) + (size_t)y * J.w + x0;
    u32x4 out[4];  // [plane] 16 bytes = 16 columns
    const bool whole = y < J.h && x0 + 16 <= J.w;  // all 16 pixels exist: four 16-byte loads (rows are only 4-byte aligned)
    u32x4 in[4] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    if (whole) {
#pragma unroll
        for (int g = 0; g < 4; ++g) in[g] = *reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(src + 4 * g);
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) in[g][j] = (y < J.h && x0 + 4 * g + j < J.w) ? src[4 * g + j] : 0u;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t px[4] = {in[g][0], in[g][1], in[g][2], in[g][3]};
        uint32_t rb[4], ga[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = px[j] >> 24;
            rb[j] = premultiply2_hi(px[j] & 0x00FF00FFu, a);
            ga[j] = premultiply2_hi(byte_perm(px[j], px[j], 0x0c0d0c01u), a);
        }
        const uint32_t rb01 = byte_perm(rb[1], rb[0], 0x07030501u), rb23 = byte_perm(rb[3], rb[2], 0x07030501u);
        const uint32_t ga01 = byte_perm(ga[1], ga[0], 0x07030501u), ga23 = byte_perm(ga[3], ga[2], 0x07030501u);
        out[0][g] = byte_perm(rb23, rb01, 0x05040100u) ^ 0x80808080u;  // R
        out[1][g] = byte_perm(ga23, ga01, 0x05040100u) ^ 0x80808080u;  // G
        out[2][g] = byte_perm(rb23, rb01, 0x07060302u) ^ 0x80808080u;  // B
        out[3][g] = byte_perm(ga23, ga01, 0x07060302u) ^ 0x80808080u;  // A
    }
    const size_t plane = (size_t)bands * ct * 256;  // bytes
    MIC_GLOBAL uint8_t *dst = reinterpret_cast<MIC_GLOBAL uint8_t *>(J.dst) + (size_t)tile * 256 + (size_t)row * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<MIC_GLOBAL u32x4 *>(dst + c * plane) = out[c];
}

hipError_t launch_planarize_tiled(const PlanarJob *jobs_dev, int n_jobs, int64_t max_items, hipStream_t stream) {
    if (n_jobs <= 0 || max_items <= 0) return hipSuccess;
    for (int first = 0; first < n_jobs; first += 65535) {
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(planarize_tiled_kernel, dim3((unsigned)((max_items + 255) / 256), (unsigned)n), dim3(256), 0, stream,
                           jobs_dev + first);
    }
    return hipGetLastError();
}

}  // namespace mic
