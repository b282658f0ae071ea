// Ordered RGBA alpha-over of resolved layers onto canvases: the hot kernels of the path.
//
// Replaces the per-placement loop of compositor.composite (compositor.py:12-21) and, inside it,
// Pillow's crop + AlphaComposite.c + paste triple pass (Image.alpha_composite(im, dest)).
//
// Mapping (gfx950, wave64).  The canvas is a linear stream of RGBA words cut into 4 KiB pages
// aligned to absolute address.  One WAVEFRONT = one page: a lane owns four groups of four adjacent
// pixels, 1 KiB apart, so the page is written by four fully coalesced 1 KiB stores; a workgroup is
// kPagesPerWorkgroup (4) such waves over consecutive pages, sharing nothing.  Workgroups are dealt
// round-robin over the 8 XCDs, so in dispatch order every XCD keeps writing one residue class of
// 16 KiB page groups (mod 8); whole pages per wave in dispatch order are what lets a store stream
// reach ~6.7 TB/s on MI355X (see mic_internal.h and scripts/streambench.hip).
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

#include "mic_internal.h"

namespace mic {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;  // 16-byte access, 4-byte alignment

__device__ __forceinline__ uint32_t div255_shift(uint32_t t) { return ((t >> 8) + t) >> 8; }

// Pillow AlphaComposite.c, one pixel, `s` over `d`; pixels are little-endian RGBA words.
__device__ __forceinline__ uint32_t alpha_over(uint32_t d, uint32_t s) {
    const uint32_t sa = s >> 24;
    const uint32_t da = d >> 24;
    const uint32_t outa255 = sa * 255u + da * (255u - sa);
    // da == 255 (every canvas of the pipeline): outa255 == 255*255 and the quotient is sa*128.
    uint32_t coef1 = sa << 7;
    if (da != 255u && sa != 0u) coef1 = (sa * (255u * 255u * 128u)) / outa255;
    const uint32_t coef2 = 255u * 128u - coef1;
    const uint32_t r = div255_shift((s & 255u) * coef1 + (d & 255u) * coef2 + (0x80u << 7)) >> 7;
    const uint32_t g =
        div255_shift(((s >> 8) & 255u) * coef1 + ((d >> 8) & 255u) * coef2 + (0x80u << 7)) >> 7;
    const uint32_t b =
        div255_shift(((s >> 16) & 255u) * coef1 + ((d >> 16) & 255u) * coef2 + (0x80u << 7)) >> 7;
    const uint32_t a = div255_shift(outa255 + 0x80u);
    const uint32_t o = r | (g << 8) | (b << 16) | (a << 24);
    return sa == 0u ? d : o;
}

// The same formula when the destination alpha is 255 (every canvas of the reference's pipeline):
// it reduces EXACTLY to out.c = div255(s.c*sa + d.c*(255-sa) + 128), out.a = 255 (checked for all
// 2^24 (sa, s.c, d.c) triples, tests/test_blend_identities.py).  R and B ride in the two 16-bit
// halves of one register (255*255 + 128 < 2^16, so the halves never carry into each other).
// {G, 255} ride in a second register the same way: 255*sa + 255*(255-sa) + 128 divides to 255, the
// output alpha.  Each div255 leaves its result in bytes 1 and 3 of t + {t.b1, 0, t.b3, 0}, and one
// v_perm_b32 gathers the four result bytes: 15 instructions per pixel (the scalar form took 23).
__device__ __forceinline__ uint32_t over_opaque_dst(uint32_t d, uint32_t s) {
    const uint32_t sa = s >> 24, na = 255u - sa;
    const uint32_t M = 0x00FF00FFu;
    const uint32_t s_ga = __builtin_amdgcn_perm(s, s, 0x0c0d0c01u), d_ga = __builtin_amdgcn_perm(d, d, 0x0c0d0c01u);
    uint32_t rb = __umul24(d & M, na) + (__umul24(s & M, sa) + 0x00800080u);
    uint32_t ga = __umul24(d_ga, na) + (__umul24(s_ga, sa) + 0x00800080u);
    rb += __builtin_amdgcn_perm(rb, rb, 0x0c030c01u);  // + ((rb >> 8) & M): results in bytes 1, 3
    ga += __builtin_amdgcn_perm(ga, ga, 0x0c030c01u);
    return __builtin_amdgcn_perm(ga, rb, 0x07030501u);  // {rb.b1, ga.b1, rb.b3, ga.b3} = R, G, B, 255
}

// Canvas traffic is touched once: nontemporal hints (background reads, canvas stores).
__device__ __forceinline__ u32x4 load4(gcptr p) {
    return __builtin_nontemporal_load(reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(p));
}
__device__ __forceinline__ uint32_t load1(gcptr p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void store4(gptr p, u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<MIC_GLOBAL u32x4_a4 *>(p));
}
__device__ __forceinline__ void store1(gptr p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// Does layer rect [dx, dx+w) x [dy, dy+h) touch any pixel of the linear run [a, b] (inclusive)
// of a canvas W pixels wide?  (ra, ca) / (rb, cb) are the row/column of a and b.
//
// Written with the sign-bit trick -- a set of conditions "v_k >= 0" holds iff (v_1 | v_2 | ...) >= 0,
// and at least one of several such sets holds iff the AND of their OR-words is >= 0 -- so that the
// whole test is a dozen vector integer ops and ONE compare.  The obvious boolean form compiles to
// lane-mask arithmetic on the scalar unit (one per CU), which was this kernel's busiest resource.
__device__ __forceinline__ bool run_hits(int ra, int ca, int rb, int cb, int W, int dx, int dy, int w,
                                         int h) {
    const int x1 = dx + w - 1, y1 = dy + h - 1;  // inclusive
    if (ra == rb)                                // wave-uniform
        return ((ra - dy) | (y1 - ra) | (cb - dx) | (x1 - ca)) >= 0;
    const int head = (ra - dy) | (y1 - ra) | (x1 - ca) | (W - 1 - dx);  // first row: columns ca..W-1
    const int tail = (rb - dy) | (y1 - rb) | (cb - dx) | x1;            // last row: columns 0..cb
    const int mid = (min(rb - 1, y1) - max(ra + 1, dy)) | x1 | (W - 1 - dx);  // a full row in between
    return (head & tail & mid) >= 0;
}

// ------------------------------------------------------------------------------------------------
// Composite: one wave per page.
// ------------------------------------------------------------------------------------------------

// One group of four pixels of a lane.  Pixels 0..k-1 lie in canvas row y from column x on; when the
// canvas width is not a multiple of 4 a group can straddle a row end: pixels k..3 then start row
// y+1 at column 0 (k == 4: no straddle).
struct Group {
    int x, y, k;
};

// Where (as a pixel offset from L.src) a group segment reads layer L.  A segment that the layer's
// left/right edge cuts through still issues ONE 16-byte load: it starts up to 3 pixels before the
// row or ends up to 3 pixels after it (previous/next row, or the guard band every image the kernels
// read is allocated with) and the stray pixels are masked afterwards.  Lanes the layer does not
// cover read offset 0 (a harmless broadcast) so that the loads of all four groups issue back to back
// with no divergent branch -- conditional loads made hipcc drain vmcnt between groups, one HBM
// round trip after another.
struct Tap {
    int off;  // pixel offset from L.src; the host keeps every layer at <= 2^30 - 8 px, so off * 4 + 16 fits 32 bits
    int sx;   // layer column under pixel 0; -kLaneNPx when this lane does not read the layer
};

__device__ __forceinline__ Tap make_tap(const Layer &L, int x, int y, bool enable) {
    const int sy = y - L.dy, sx = x - L.dx;
    // covered <=> 0 <= sy < h and -4 < sx < w (sign-bit trick, see run_hits)
    const bool covered = enable && ((sy | (L.h - 1 - sy) | (sx + kLaneNPx - 1) | (L.w - 1 - sx)) >= 0);
    Tap t;
    t.off = covered ? sy * L.w + sx : 0;
    t.sx = covered ? sx : -kLaneNPx;
    return t;
}

__device__ __forceinline__ u32x4 load_tap(const Layer &L, const Tap &t) {
    // uniform base (SGPR pair) + unsigned 32-bit lane offset; the 16-byte bias keeps the offset
    // non-negative when the load starts in the guard band before the cutout
    const MIC_GLOBAL char *basep = reinterpret_cast<const MIC_GLOBAL char *>(L.src) - 16;
    // Default cache policy on purpose: the atlas is shared by every canvas of a batch and by
    // neighbouring pages, and lives in L2 / the Infinity Cache between uses.  Nontemporal loads here
    // cost 15% of the kernel (C3 batch: 134.7 -> 114.8 us); nontemporal STORES are worth +5%.
    // (A/B switches for both policies: profiles/r05_tuning_scaffolding.patch, -DMIC_PLAIN_STORES / -DMIC_SRC_NT_LOADS)
    return *reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(basep + (uint32_t)(t.off * 4 + 16));
}

// Keep the loaded pixels whose layer column c = t.sx + j satisfies lo <= c < lo + span, zero
// (transparent) the rest.  For the pixels of a group that lie in the group's own row, lo = 0 and
// span = the layer's width clipped at the canvas' right edge: that one bound also drops the pixels
// of a row-straddling group (W % 4 != 0) that belong to the next row, since their columns are >= W.
__device__ __forceinline__ u32x4 mask_tap(const Tap &t, u32x4 v, int lo, int span) {
    u32x4 s;
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) s[j] = (uint32_t)(t.sx + j - lo) < (uint32_t)span ? v[j] : 0u;
    return s;
}

// First/last page of a canvas that does not start/end on a 4 KiB boundary, and canvases narrower
// than 4 pixels: at most two pages per canvas, so this is written for obviousness, not speed --
// one pixel at a time, every layer tested, Pillow's formula verbatim (it mirrors the oracle).
__device__ __forceinline__ void edge_page(const Job &job, const Layer *jl, int64_t qp, int lane) {
    const uint32_t W = (uint32_t)job.W;
    const int64_t n_px = (int64_t)job.W * job.H;
    gcptr bg = reinterpret_cast<gcptr>(job.bg);
    gptr out = reinterpret_cast<gptr>(job.out);
#pragma unroll 1
    for (int i = 0; i < kGroups * kLaneNPx; ++i) {
        const int64_t q = qp + (i / kLaneNPx) * kWavePx + lane * kLaneNPx + (i % kLaneNPx);
        if (q < 0 || q >= n_px) continue;
        const int y = (int)((uint32_t)q / W);
        const int x = (int)((uint32_t)q - (uint32_t)y * W);
        uint32_t p = job.bg != 0 ? load1(bg + q) : job.bg_rgba;
#pragma unroll 1
        for (int l = 0; l < job.layer_count; ++l) {
            const Layer L = jl[l];
            const int sx = x - L.dx, sy = y - L.dy;
            if ((sx | sy | (L.w - 1 - sx) | (L.h - 1 - sy)) >= 0)
                p = alpha_over(p, load1(reinterpret_cast<gcptr>(L.src) + ((int64_t)sy * L.w + sx)));
        }
        store1(out + q, p);
    }
}

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
// Workgroup = kPagesPerWorkgroup waves, each with a page of its own (nothing is shared between them: no LDS, no
// barrier).  Round 4, A/B in one run against one-wave workgroups (profiles/r04_composite_pages_per_workgroup.txt): four
// pages per workgroup is 3-5 % faster on a single 4K canvas (a quarter of the workgroups to dispatch for the same waves:
// 12.3 -> 11.9 us between events, 10.0 -> 9.5 us per canvas back to back) and 2.4 % on the 16-canvas batch (113.9 ->
// 111.2 us); two are level with one, eight lose 7 %.
template <bool ALIGNED, bool SOLID, int MODE>
__global__ __launch_bounds__(64 * kPagesPerWorkgroup, (ALIGNED && SOLID) ? MIC_HOT_WAVES : (SOLID ? 6 : 4)) void composite_kernel(
    const Job *__restrict__ jobs, const Layer *__restrict__ layers, const Job one, const LayerPack pack) {
    Job job = MODE != kFromTables ? one : jobs[blockIdx.y];
    // (a small single-job launch comes as one-wave workgroups: kJobOnePagePerWorkgroup in the kernel-argument copy of the job)
    const int wpb = (MODE != kFromTables && (one.flags & kJobOnePagePerWorkgroup)) ? 1 : kPagesPerWorkgroup;
    const int page_ = (int)blockIdx.x * wpb + (kPagesPerWorkgroup > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0);
    if (page_ >= job.n_pages) return;
    // A solid colour that lives in device memory (mic_job.bg_rgba_dev: the median kernel's result, consumed without a
    // host round trip): one scalar load, in flight while the layer records are fetched.  The host classes such a job
    // as SOLID (opaque); a colour word that turns out not to be opaque sends the page through the exact per-pixel path
    // just before the store (below).
    const bool colour_word = (job.flags & kJobColourWord) != 0;  // wave-uniform
    if (colour_word) {
        job.bg_rgba = *reinterpret_cast<const MIC_GLOBAL uint32_t *>(job.bg);
        job.bg = 0;
    }
    const int lane = threadIdx.x & 63;
    const int W = job.W;
    const int64_t n_px = (int64_t)job.W * job.H;
    const Layer *jl = MODE == kAllInArgs ? pack.l : layers + job.layer_begin;

    const int64_t qp = (int64_t)page_ * kPagePx - job.px_shift;
    // pages that lie wholly inside the canvas: all but the first/last of a page-misaligned canvas
    if (qp < 0 || W < kLaneNPx) {
        edge_page(job, jl, qp, lane);
        return;
    }
    // The last page of a canvas whose size is not a multiple of 4 KiB runs through the normal path
    // with guarded canvas accesses: its pixels past the end lie in rows >= H, which no store reaches.
    // (It used to take edge_page; being the last workgroup dispatched, that slow loop was the tail of
    // the whole launch -- every canvas whose width is not a multiple of 4 has such a page.)
    const bool tail = qp + kPagePx > n_px;  // wave-uniform
    const int64_t q_lane = qp + lane * kLaneNPx;  // group r starts at q_lane + r * 256

    // The four 256-pixel runs of the page as (row, column) of both ends, and this lane's four
    // pixel groups.  One wave-uniform division gives the row/column of the page's first pixel.
    // (This is scalar-unit work, one unit per CU: kept lean.)
    int ra[kGroups], ca[kGroups], rb[kGroups], cb[kGroups];
    Group G[kGroups];
    const uint32_t uW = (uint32_t)W;
    const int y0 = (int)((uint32_t)qp / uW);
    const uint32_t x0 = (uint32_t)qp - (uint32_t)y0 * uW;
    if (W >= kPagePx) {  // a page spans at most two rows: add/compare only
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            uint32_t xf = x0 + (uint32_t)(r * kWavePx);
            int yf = y0;
            if (xf >= uW) { xf -= uW; yf += 1; }
            uint32_t xl = xf + (uint32_t)(kWavePx - 1);
            int yl = yf;
            if (xl >= uW) { xl -= uW; yl += 1; }
            ra[r] = yf; ca[r] = (int)xf; rb[r] = yl; cb[r] = (int)xl;
            uint32_t x = xf + (uint32_t)(lane * kLaneNPx);
            int y = yf;
            if (x >= uW) { x -= uW; y += 1; }
            G[r].x = (int)x;
            G[r].y = y;
            G[r].k = ALIGNED ? kLaneNPx : min(kLaneNPx, W - (int)x);
        }
    } else {  // narrow canvas: a run spans several rows
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            const uint32_t f = x0 + (uint32_t)(r * kWavePx), l = f + (uint32_t)(kWavePx - 1);
            const uint32_t df = f / uW, dl = l / uW;
            ra[r] = y0 + (int)df; ca[r] = (int)(f - df * uW);
            rb[r] = y0 + (int)dl; cb[r] = (int)(l - dl * uW);
            const uint32_t xq = f + (uint32_t)(lane * kLaneNPx), d = xq / uW;
            G[r].x = (int)(xq - d * uW);
            G[r].y = y0 + (int)d;
            G[r].k = ALIGNED ? kLaneNPx : min(kLaneNPx, W - G[r].x);
        }
    }

    // ---- background ----
    u32x4 px[kGroups];
    bool dst_opaque = true;  // every pixel this wave holds is opaque; then stays so (over 255 gives 255)
    if (SOLID) {
#pragma unroll
        for (int r = 0; r < kGroups; ++r) px[r] = (u32x4)(job.bg_rgba);
    } else if (job.bg == 0) {
#pragma unroll
        for (int r = 0; r < kGroups; ++r) px[r] = (u32x4)(job.bg_rgba);
        dst_opaque = (job.bg_rgba >> 24) == 255u;
    } else {
        gcptr bg = reinterpret_cast<gcptr>(job.bg) + q_lane;
        uint32_t amin = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            if (!tail || q_lane + r * kWavePx + kLaneNPx <= n_px) {
                px[r] = load4(bg + r * kWavePx);
            } else {
#pragma unroll
                for (int j = 0; j < kLaneNPx; ++j)
                    px[r][j] = q_lane + r * kWavePx + j < n_px ? load1(bg + r * kWavePx + j) : 0u;
            }
        }
#pragma unroll
        for (int r = 0; r < kGroups; ++r)
#pragma unroll
            for (int j = 0; j < kLaneNPx; ++j) amin = min(amin, px[r][j]);
        dst_opaque = !__any(amin < 0xFF000000u);
    }

    // W % 4 != 0: which of this lane's groups run over a row end (pixels W - x .. 3 continue at column 0
    // of the next row), and whether any lane of the wave has one -- once per page, not per round.
    uint32_t strad_mask = 0;  // bit r: some lane's group r straddles (kept as one scalar: SGPRs are scarce here)
#pragma unroll
    for (int r = 0; r < kGroups; ++r)
        if (!ALIGNED && __any(G[r].k < kLaneNPx)) strad_mask |= 1u << r;

    // ---- layers ----
    // Order only matters among layers that touch the same pixels, so each of the four groups walks
    // ITS OWN hit mask in list order: one round issues up to four independent 16-byte loads per lane
    // (one per group, possibly from four different layers) and then blends them.  A page touched by
    // three or four side-by-side objects needs one or two rounds, not one HBM round trip per object.
    for (int base = 0; base < job.layer_count; base += 64) {
        // cull 64 layers at once: lane l holds record base + l
        Layer mine{};
        bool hit[kGroups] = {false, false, false, false};
        // The runs' end points are recomputed here from the page origin (a few scalar adds) rather
        // than kept in 16 SGPRs across the rounds below: the unaligned instantiations were spilling
        // scalars.  The empty asm makes the origin opaque so that the recomputation is not hoisted.
        uint32_t ox = x0;
        int oy = y0;
        asm volatile("" : "+s"(ox), "+s"(oy));
        if (base + lane < job.layer_count) {
            mine = jl[base + lane];
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {
                int a_r, a_c, b_r, b_c;
                if (W >= kPagePx) {
                    uint32_t xf = ox + (uint32_t)(r * kWavePx);
                    int yf = oy;
                    if (xf >= uW) { xf -= uW; yf += 1; }
                    uint32_t xl = xf + (uint32_t)(kWavePx - 1);
                    int yl = yf;
                    if (xl >= uW) { xl -= uW; yl += 1; }
                    a_r = yf; a_c = (int)xf; b_r = yl; b_c = (int)xl;
                } else {
                    const uint32_t f = ox + (uint32_t)(r * kWavePx), l = f + (uint32_t)(kWavePx - 1);
                    const uint32_t df = f / uW, dl = l / uW;
                    a_r = oy + (int)df; a_c = (int)(f - df * uW);
                    b_r = oy + (int)dl; b_c = (int)(l - dl * uW);
                }
                hit[r] = run_hits(a_r, a_c, b_r, b_c, W, mine.dx, mine.dy, mine.w, mine.h);
            }
        }
        uint64_t m[kGroups];
#pragma unroll
        for (int r = 0; r < kGroups; ++r) m[r] = __ballot(hit[r]);
        while ((m[0] | m[1] | m[2] | m[3]) != 0) {
            // Groups without a pending layer this round are skipped by WAVE-UNIFORM branches (their masks come from
            // ballots): no tap arithmetic, no load, no masking for them.  The loads of the groups that have one still
            // issue back to back -- nothing between them waits on memory -- and the wave waits once, before the
            // first consume.  (Per-LANE conditions around the loads are what made hipcc drain vmcnt between groups.)
            u32x4 s[kGroups];
            Layer L[kGroups];
            Tap tap[kGroups];
            bool has_layer[kGroups];
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {  // issue: one 16-byte load per lane per group
                const bool has = m[r] != 0;
                has_layer[r] = has;
                s[r] = u32x4{0u, 0u, 0u, 0u};
                if (has) {
                    // `i` comes from a ballot (wave-uniform): v_readlane broadcasts the record
                    const int i = __ffsll((long long)m[r]) - 1;
                    m[r] &= m[r] - 1;
                    L[r].src = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine.src >> 32), i) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine.src, i);
                    L[r].dx = __builtin_amdgcn_readlane(mine.dx, i);
                    L[r].dy = __builtin_amdgcn_readlane(mine.dy, i);
                    L[r].w = __builtin_amdgcn_readlane(mine.w, i);
                    L[r].h = __builtin_amdgcn_readlane(mine.h, i);
                    tap[r] = make_tap(L[r], G[r].x, G[r].y, true);
                    s[r] = load_tap(L[r], tap[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {  // consume
                if (!has_layer[r]) continue;
                // the layer clipped at the canvas' right edge (only matters when groups can straddle)
                const int wclip = ALIGNED ? L[r].w : min(L[r].w, W - L[r].dx);
                s[r] = mask_tap(tap[r], s[r], 0, wclip);
                if (!ALIGNED && ((strad_mask >> r) & 1u)) {
                    // a group that straddles a row end: its pixels W - x .. 3 continue at column 0 of
                    // the next row -- one more masked 16-byte load, only in waves that hold such a
                    // group.  Canvas column = x - W + j >= 0 <=> layer column >= -dx: clip on the left.
                    const Tap t2 = make_tap(L[r], G[r].x - W, G[r].y + 1, G[r].k < kLaneNPx);
                    const int lo = max(0, -L[r].dx);
                    const u32x4 v2 = mask_tap(t2, load_tap(L[r], t2), lo, wclip - lo);
#pragma unroll
                    for (int j = 0; j < kLaneNPx; ++j) s[r][j] |= v2[j];
                }
            }
            // Blend path per 256-pixel group (wave-uniform choices): a group without a layer this round
            // is skipped; a group whose source pixels all have alpha 0 or 255 -- (sa + 1) & 0xFE == 0 --
            // takes the select (what the formula gives, exactly, for any dst); partial alpha anywhere in
            // the group sends it through the arithmetic.  Resampled binary cutouts are soft only along
            // their edges, so most groups of a placements-mode canvas still take the select.
            bool soft_g[kGroups], any_soft = false;
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {
                uint32_t soft = 0;
#pragma unroll
                for (int j = 0; j < kLaneNPx; ++j) soft |= ((s[r][j] >> 24) + 1u) & 0xFEu;
                soft_g[r] = has_layer[r] && __any(soft != 0);
                any_soft = any_soft || soft_g[r];
            }
            if (SOLID || dst_opaque || !any_soft) {
#pragma unroll
                for (int r = 0; r < kGroups; ++r) {
                    if (!has_layer[r]) continue;
                    if (soft_g[r]) {
#pragma unroll
                        for (int j = 0; j < kLaneNPx; ++j) px[r][j] = over_opaque_dst(px[r][j], s[r][j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < kLaneNPx; ++j) px[r][j] = s[r][j] >= 0xFF000000u ? s[r][j] : px[r][j];
                    }
                }
            } else if (!SOLID) {
                // translucent destination (a background with alpha < 255): the verbatim formula, one
                // pixel at a time through ONE copy of the code (registers rotate), so that this rare
                // path does not set the kernel's register budget
#pragma unroll 1
                for (int it = 0; it < kGroups * kLaneNPx; ++it) {
                    const uint32_t o = alpha_over(px[0][0], s[0][0]);
                    px[0] = u32x4{px[0][1], px[0][2], px[0][3], px[1][0]};
                    px[1] = u32x4{px[1][1], px[1][2], px[1][3], px[2][0]};
                    px[2] = u32x4{px[2][1], px[2][2], px[2][3], px[3][0]};
                    px[3] = u32x4{px[3][1], px[3][2], px[3][3], o};
                    s[0] = u32x4{s[0][1], s[0][2], s[0][3], s[1][0]};
                    s[1] = u32x4{s[1][1], s[1][2], s[1][3], s[2][0]};
                    s[2] = u32x4{s[2][1], s[2][2], s[2][3], s[3][0]};
                    s[3] = u32x4{s[3][1], s[3][2], s[3][3], 0u};
                }
            }
        }
    }

    if (SOLID && colour_word && (job.bg_rgba >> 24) != 255u) {  // (wave-uniform, never taken by fill_solid's colours)
        edge_page(job, jl, qp, lane);
        return;
    }
    // ---- the canvas is written exactly once: four coalesced 1 KiB stores per page ----
    gptr out = reinterpret_cast<gptr>(job.out) + q_lane;
#pragma unroll
    for (int r = 0; r < kGroups; ++r) {
        if (!tail || q_lane + r * kWavePx + kLaneNPx <= n_px) {
            store4(out + r * kWavePx, px[r]);
        } else {
#pragma unroll
            for (int j = 0; j < kLaneNPx; ++j)
                if (q_lane + r * kWavePx + j < n_px) store1(out + r * kWavePx + j, px[r][j]);
        }
    }
}

// Jobs arrive sorted by class: [0, n0) aligned+solid, [n0, n1) unaligned+solid, [n1, n2) aligned with
// a background image / translucent colour, [n2, n_jobs) neither.  single != nullptr: the launch's only job,
// passed in the kernel arguments (jobs_dev is not read).
// The argument block of every launch carries a LayerPack (2 KiB copied into the kernarg ring by the runtime: a few
// dozen nanoseconds); only kAllInArgs launches read it.  One per calling thread: contexts are driven concurrently.
static thread_local LayerPack g_pack;

hipError_t launch_composite(const Job *jobs_dev, const Layer *layers_dev, int n_jobs, const int class_end[3],
                            int pitch, const Job *single, const Layer *single_layers_host, hipStream_t stream,
                            uint64_t *launched_workgroups) {
    // every launch below covers `pitch` pages per job with workgroups of kPagesPerWorkgroup pages; what is reported
    // is computed from the very grid that is launched (mic_stats.composite_blocks, tests/test_gpu_parity.py)
    if (launched_workgroups) *launched_workgroups = 0;
    if (n_jobs <= 0 || pitch <= 0) return hipSuccess;
    const unsigned wgs = (unsigned)pitch / kPagesPerWorkgroup;
    if (launched_workgroups) *launched_workgroups = (uint64_t)wgs * (uint64_t)n_jobs;
    // grid.x (= pitch) is a multiple of 8 so that (linear workgroup id) mod 8 == (page index) mod 8
    // for every job of the launch: the XCD <-> page residue pairing survives the 2-D grid.
    const int b[5] = {0, class_end[0], class_end[1], class_end[2], n_jobs};
    if (single && n_jobs == 1) {
        const bool small = single->n_pages <= kSmallCanvasPages;
        const dim3 grid(small ? (unsigned)pitch : wgs, 1u);
        const dim3 block(small ? 64u : 64u * kPagesPerWorkgroup);
        if (small && launched_workgroups) *launched_workgroups = (uint64_t)pitch;
        const int cls = b[1] > b[0] ? 0 : b[2] > b[1] ? 1 : b[3] > b[2] ? 2 : 3;
        Job one = *single;
        if (small) one.flags |= kJobOnePagePerWorkgroup;
        if (single_layers_host && single->layer_count <= kPackLayers) {
            if (one.layer_count > 0) memcpy(g_pack.l, single_layers_host + one.layer_begin, sizeof(Layer) * (size_t)one.layer_count);
            one.layer_begin = 0;
            switch (cls) {
                case 0: hipLaunchKernelGGL((composite_kernel<true, true, kAllInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                case 1: hipLaunchKernelGGL((composite_kernel<false, true, kAllInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                case 2: hipLaunchKernelGGL((composite_kernel<true, false, kAllInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
                default: hipLaunchKernelGGL((composite_kernel<false, false, kAllInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
            }
            return hipGetLastError();
        }
        switch (cls) {
            case 0: hipLaunchKernelGGL((composite_kernel<true, true, kJobInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
            case 1: hipLaunchKernelGGL((composite_kernel<false, true, kJobInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
            case 2: hipLaunchKernelGGL((composite_kernel<true, false, kJobInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
            default: hipLaunchKernelGGL((composite_kernel<false, false, kJobInArgs>), grid, block, 0, stream, jobs_dev, layers_dev, one, g_pack); break;
        }
        return hipGetLastError();
    }
    const Job none{};
    if (b[1] > b[0])
        hipLaunchKernelGGL((composite_kernel<true, true, kFromTables>), dim3(wgs, (unsigned)(b[1] - b[0])), dim3(64 * kPagesPerWorkgroup),
                           0, stream, jobs_dev + b[0], layers_dev, none, g_pack);
    if (b[2] > b[1])
        hipLaunchKernelGGL((composite_kernel<false, true, kFromTables>), dim3(wgs, (unsigned)(b[2] - b[1])), dim3(64 * kPagesPerWorkgroup),
                           0, stream, jobs_dev + b[1], layers_dev, none, g_pack);
    if (b[3] > b[2])
        hipLaunchKernelGGL((composite_kernel<true, false, kFromTables>), dim3(wgs, (unsigned)(b[3] - b[2])), dim3(64 * kPagesPerWorkgroup),
                           0, stream, jobs_dev + b[2], layers_dev, none, g_pack);
    if (b[4] > b[3])
        hipLaunchKernelGGL((composite_kernel<false, false, kFromTables>), dim3(wgs, (unsigned)(b[4] - b[3])), dim3(64 * kPagesPerWorkgroup),
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
