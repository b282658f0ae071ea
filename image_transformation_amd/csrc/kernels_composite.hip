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
__device__ __forceinline__ uint32_t over_opaque_dst(uint32_t d, uint32_t s) {
    const uint32_t sa = s >> 24, na = 255u - sa;
    const uint32_t M = 0x00FF00FFu;
    uint32_t rb = (s & M) * sa + (d & M) * na + 0x00800080u;
    uint32_t g = ((s >> 8) & 0xFFu) * sa + ((d >> 8) & 0xFFu) * na + 0x80u;
    rb = ((((rb >> 8) & M) + rb) >> 8) & M;
    g = ((g >> 8) + g) >> 8;
    return rb | (g << 8) | 0xFF000000u;
}

// Streaming accesses: every byte is touched once, so nontemporal hints on both sides.
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

// One group of four pixels of a lane: linear indices q .. q+3, of which [lo, hi) lie in the canvas.
// ALIGNED jobs (W % 4 == 0 and a 16-byte aligned canvas): a group is either wholly inside one
// canvas row or wholly outside the canvas, so `regular` is the whole story and lo/hi are unused.
struct Group {
    int x, y;      // canvas column/row of the group's first in-canvas pixel (pixel `lo`)
    int lo, hi;    // 0, 4 except at the canvas ends of a page-misaligned canvas
    bool regular;  // all four pixels inside the canvas and in the same row
};

// Where (as a pixel offset from L.src) one group reads layer L, and whether it reads at all.
// A group that the layer's left/right edge cuts through still issues ONE 16-byte load: it starts
// up to 3 pixels before the row or ends up to 3 pixels after it (previous/next row, or the guard
// band every image the kernels read is allocated with) and the stray pixels are masked afterwards.
// Lanes the layer does not cover read offset 0 (a harmless broadcast) so that the loads of all
// four groups issue back to back with no divergent branch -- conditional loads made hipcc drain
// vmcnt between groups, one HBM round trip after another.
struct Tap {
    int off;  // pixel offset from L.src (< 2^31: the host rejects layers of 2^31 pixels or more)
    int sx;   // layer column under pixel 0; -kLaneNPx when this lane does not read the layer
};

__device__ __forceinline__ Tap tap_regular(const Layer &L, const Group &G, bool has_layer) {
    const int sy = G.y - L.dy, sx = G.x - L.dx;
    // covered <=> 0 <= sy < h and -4 < sx < w (sign-bit trick, see run_hits)
    const bool covered = has_layer && G.regular && ((sy | (L.h - 1 - sy) | (sx + kLaneNPx - 1) | (L.w - 1 - sx)) >= 0);
    Tap t;
    t.off = covered ? sy * L.w + sx : 0;
    t.sx = covered ? sx : -kLaneNPx;
    return t;
}

// Keep the loaded pixels that lie inside the layer row, zero (transparent) the rest.
__device__ __forceinline__ u32x4 mask_tap(const Tap &t, const Layer &L, u32x4 v) {
    u32x4 s;
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) s[j] = (uint32_t)(t.sx + j) < (uint32_t)L.w ? v[j] : 0u;
    return s;
}

// Row-straddling group (W % 4 != 0) or ragged canvas end: walk the pixels, no division (general
// kernel only; such groups are one lane per canvas row).
__device__ __forceinline__ u32x4 fetch_straddler(const Layer &L, const Group &G, int W) {
    u32x4 s = (u32x4)(0u);
    gcptr src = reinterpret_cast<gcptr>(L.src);
    int x = G.x, y = G.y;
#pragma unroll
    for (int j = 0; j < kLaneNPx; ++j) {
        if (j >= G.lo && j < G.hi) {
            while (x >= W) {  // once when W >= 4
                x -= W;
                ++y;
            }
            const int sy = y - L.dy, sx = x - L.dx;
            if (sy >= 0 && sy < L.h && sx >= 0 && sx < L.w) s[j] = load1(src + ((int64_t)sy * L.w + sx));
            ++x;
        }
    }
    return s;
}

// (launch bounds: hipcc settles for 85 VGPRs = 5 waves/SIMD unless told that 7 are wanted; 70 VGPRs,
// no spills.  8 waves would spill.)
// HOT = jobs with W % 4 == 0, a 16-byte aligned canvas and a solid opaque background (what the
// reference's pipeline produces: fill_solid canvases, background_resizing.py:32); every other job
// (odd widths, background images, translucent colours) takes the general instantiation.
template <bool HOT>
__global__ __launch_bounds__(64, HOT ? 7 : 1) void composite_kernel(const Job *__restrict__ jobs,
                                                       const Layer *__restrict__ layers) {
    const Job job = jobs[blockIdx.y];
    if ((int)blockIdx.x >= job.n_pages) return;
    const int lane = threadIdx.x;
    const int W = job.W;
    const int64_t n_px = (int64_t)job.W * job.H;
    const bool wide = W >= kPagePx;  // a page then spans at most two rows
    constexpr bool ALIGNED = HOT;

    const int64_t qp = (int64_t)blockIdx.x * kPagePx - job.px_shift;
    // pages that lie wholly inside the canvas: all but the first/last of a page-misaligned canvas
    const bool interior = qp >= 0 && qp + kPagePx <= n_px;
    const int64_t q_lane = qp + lane * kLaneNPx;  // group r starts at q_lane + r * 256
    // the four 256-pixel runs of the page, clipped to the canvas, as (row, column) of both ends,
    // and this lane's four pixel groups
    int ra[kGroups], ca[kGroups], rb[kGroups], cb[kGroups];
    bool live[kGroups];
    Group G[kGroups];
    if (interior && wide) {
        // The common case, kept lean (this is scalar-unit work, one unit per CU): the page lies
        // inside a canvas at least one page wide, so it spans at most two rows.  One wave-uniform
        // division gives the row/column of its first pixel; everything else is add/compare.
        const uint32_t qa = (uint32_t)qp;
        const int y0 = (int)(qa / (uint32_t)W);
        const uint32_t x0 = qa - (uint32_t)y0 * (uint32_t)W;
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            uint32_t xf = x0 + (uint32_t)(r * kWavePx);
            int yf = y0;
            if (xf >= (uint32_t)W) { xf -= (uint32_t)W; yf += 1; }
            uint32_t xl = xf + (uint32_t)(kWavePx - 1);
            int yl = yf;
            if (xl >= (uint32_t)W) { xl -= (uint32_t)W; yl += 1; }
            ra[r] = yf; ca[r] = (int)xf; rb[r] = yl; cb[r] = (int)xl;
            live[r] = true;
            uint32_t x = xf + (uint32_t)(lane * kLaneNPx);
            int y = yf;
            if (x >= (uint32_t)W) { x -= (uint32_t)W; y += 1; }
            G[r].x = (int)x;
            G[r].y = y;
            G[r].lo = 0;
            G[r].hi = kLaneNPx;
            G[r].regular = ALIGNED || (int)x + kLaneNPx <= W;
        }
    } else {
        // first/last page of a page-misaligned canvas, or a canvas narrower than a page
        // first page only: pixels of the page that precede the canvas (0 elsewhere)
        const int lead = qp < 0 ? (int)(-qp) : 0;
        // row/column of the page's first in-canvas pixel: the one wave-uniform division of the page
        struct { int y0, x0; } pd;
        {
            const uint32_t qa = (uint32_t)max(qp, (int64_t)0);
            pd.y0 = (int)(qa / (uint32_t)W);
            pd.x0 = (int)(qa - (uint32_t)pd.y0 * (uint32_t)W);
        }
        // the four 256-pixel runs of the page, clipped to the canvas, as (row, column) of both ends
    #pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            const int first = max(r * kWavePx - lead, 0);  // offsets from the page's first in-canvas pixel
            const int64_t last64 = min((int64_t)(r * kWavePx + kWavePx - 1 - lead), n_px - 1 - max(qp, (int64_t)0));
            live[r] = last64 >= first && r * kWavePx + kWavePx > lead;
            const int last = live[r] ? (int)last64 : first;
            uint32_t xf = (uint32_t)pd.x0 + (uint32_t)first, xl = (uint32_t)pd.x0 + (uint32_t)last;
            int yf = pd.y0, yl = pd.y0;
            if (wide) {
                if (xf >= (uint32_t)W) { xf -= (uint32_t)W; yf += 1; }
                if (xl >= (uint32_t)W) { xl -= (uint32_t)W; yl += 1; }
            } else {
                const uint32_t df = xf / (uint32_t)W, dl = xl / (uint32_t)W;
                yf += (int)df; xf -= df * (uint32_t)W;
                yl += (int)dl; xl -= dl * (uint32_t)W;
            }
            ra[r] = yf; ca[r] = (int)xf; rb[r] = yl; cb[r] = (int)xl;
        }

    #pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            const int in_page = r * kWavePx + lane * kLaneNPx;  // offset of pixel 0 inside the page
            G[r].lo = 0;
            G[r].hi = kLaneNPx;
            if (!interior) {
                const int64_t q = q_lane + r * kWavePx;
                G[r].lo = (int)min(max(-q, (int64_t)0), (int64_t)kLaneNPx);
                G[r].hi = (int)min(max(n_px - q, (int64_t)0), (int64_t)kLaneNPx);
            }
            // offset of the first in-canvas pixel from the page's first in-canvas pixel (pd.x0, pd.y0)
            uint32_t x = (uint32_t)pd.x0 + (uint32_t)max(in_page - lead, 0);
            int y = pd.y0;
            if (wide) {
                if (x >= (uint32_t)W) { x -= (uint32_t)W; y += 1; }
            } else {
                const uint32_t d = x / (uint32_t)W;
                y += (int)d;
                x -= d * (uint32_t)W;
            }
            G[r].x = (int)x;
            G[r].y = y;
            G[r].regular = G[r].lo == 0 && G[r].hi == kLaneNPx && (ALIGNED || (int)x + kLaneNPx <= W);
        }
    }

    // ---- background ----
    u32x4 px[kGroups];
    if (HOT || job.bg == 0) {
#pragma unroll
        for (int r = 0; r < kGroups; ++r) px[r] = (u32x4)(job.bg_rgba);
    } else if (interior) {
        gcptr bg = reinterpret_cast<gcptr>(job.bg) + q_lane;
#pragma unroll
        for (int r = 0; r < kGroups; ++r) px[r] = load4(bg + r * kWavePx);
    } else {
        gcptr bg = reinterpret_cast<gcptr>(job.bg) + q_lane;
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            px[r] = (u32x4)(0u);
            if (ALIGNED) {
                if (G[r].regular) px[r] = load4(bg + r * kWavePx);
            } else {
#pragma unroll
                for (int j = 0; j < kLaneNPx; ++j)
                    if (j >= G[r].lo && j < G[r].hi) px[r][j] = load1(bg + (r * kWavePx + j));
            }
        }
    }

    // Is every pixel this wave holds opaque?  It then stays so: alpha-over onto alpha 255 gives 255.
    bool dst_opaque = true;
    if (!HOT) {
        dst_opaque = (job.bg_rgba >> 24) == 255u;
        if (job.bg != 0) {
            uint32_t amin = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < kGroups; ++r)
#pragma unroll
                for (int j = 0; j < kLaneNPx; ++j) amin = min(amin, px[r][j]);
            dst_opaque = interior && !__any(amin < 0xFF000000u);
        }
    }

    // ---- layers ----
    // Order only matters among layers that touch the same pixels, so each of the four groups walks
    // ITS OWN hit mask in list order: one round issues up to four independent 16-byte loads per lane
    // (one per group, possibly from four different layers) and then blends them.  A page touched by
    // three or four side-by-side objects needs one or two rounds, not one HBM round trip per object.
    const Layer *jl = layers + job.layer_begin;
    for (int base = 0; base < job.layer_count; base += 64) {
        // cull 64 layers at once: lane l holds record base + l
        Layer mine{};
        bool hit[kGroups] = {false, false, false, false};
        if (base + lane < job.layer_count) {
            mine = jl[base + lane];
#pragma unroll
            for (int r = 0; r < kGroups; ++r)
                hit[r] = live[r] && run_hits(ra[r], ca[r], rb[r], cb[r], W, mine.dx, mine.dy, mine.w, mine.h);
        }
        uint64_t m[kGroups];
#pragma unroll
        for (int r = 0; r < kGroups; ++r) m[r] = __ballot(hit[r]);
        while ((m[0] | m[1] | m[2] | m[3]) != 0) {
            // Straight-line on purpose (no per-group branches): groups without a pending layer
            // replay a record that some other group hit and mask everything away, so that hipcc
            // issues the four loads back to back and waits once.
            const uint64_t m_any = m[0] | m[1] | m[2] | m[3];
            const int i_any = __ffsll((long long)m_any) - 1;
            u32x4 s[kGroups];
            Layer L[kGroups];
            Tap tap[kGroups];
            bool has_layer[kGroups];
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {  // issue: one 16-byte load per lane per group
                const bool has = m[r] != 0;
                // `i` comes from a ballot (wave-uniform): v_readlane broadcasts the record
                const int i = has ? __ffsll((long long)m[r]) - 1 : i_any;
                m[r] &= m[r] - 1;
                L[r].src = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine.src >> 32), i) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine.src, i);
                L[r].dx = __builtin_amdgcn_readlane(mine.dx, i);
                L[r].dy = __builtin_amdgcn_readlane(mine.dy, i);
                L[r].w = __builtin_amdgcn_readlane(mine.w, i);
                L[r].h = __builtin_amdgcn_readlane(mine.h, i);
                has_layer[r] = has;
                tap[r] = tap_regular(L[r], G[r], has);
                // uniform base (SGPR pair) + unsigned 32-bit lane offset -> the `saddr` form of
                // global_load: no 64-bit address in vector registers.  The 16-byte bias keeps the
                // offset non-negative when the load starts in the guard band before the cutout.
                const MIC_GLOBAL char *basep = reinterpret_cast<const MIC_GLOBAL char *>(L[r].src) - 16;
                s[r] = __builtin_nontemporal_load(
                    reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(basep + (uint32_t)(tap[r].off * 4 + 16)));
            }
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {  // consume
                s[r] = mask_tap(tap[r], L[r], s[r]);
                if (!ALIGNED) {
                    if (__any(!G[r].regular) && L[r].w > 0) {
                        const u32x4 t = fetch_straddler(L[r], G[r], W);
                        if (!G[r].regular && has_layer[r]) s[r] = t;
                    }
                }
            }
            // partial alpha anywhere in the wave?  (sa + 1) & 0xFE == 0  <=>  sa in {0, 255}
            uint32_t soft = 0;
#pragma unroll
            for (int r = 0; r < kGroups; ++r) {
                {
#pragma unroll
                    for (int j = 0; j < kLaneNPx; ++j) soft |= ((s[r][j] >> 24) + 1u) & 0xFEu;
                }
            }
            if (!__any(soft != 0)) {
                // alpha 0 keeps dst, alpha 255 takes src: what the formula gives, exactly, for any dst
#pragma unroll
                for (int r = 0; r < kGroups; ++r) {
                    {
#pragma unroll
                        for (int j = 0; j < kLaneNPx; ++j)
                            px[r][j] = s[r][j] >= 0xFF000000u ? s[r][j] : px[r][j];
                    }
                }
            } else if (HOT || dst_opaque) {
#pragma unroll
                for (int r = 0; r < kGroups; ++r) {
                    {
#pragma unroll
                        for (int j = 0; j < kLaneNPx; ++j) px[r][j] = over_opaque_dst(px[r][j], s[r][j]);
                    }
                }
            } else if (!HOT) {
                // translucent destination (a background image with alpha < 255): the verbatim formula,
                // one pixel at a time through ONE copy of the code (registers rotate), so that this
                // rare path does not set the kernel's register budget
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

    // ---- the canvas is written exactly once: four coalesced 1 KiB stores per page ----
    gptr out = reinterpret_cast<gptr>(job.out) + q_lane;
    if (interior) {
#pragma unroll
        for (int r = 0; r < kGroups; ++r) store4(out + r * kWavePx, px[r]);
    } else {
#pragma unroll
        for (int r = 0; r < kGroups; ++r) {
            if (ALIGNED) {
                if (G[r].regular) store4(out + r * kWavePx, px[r]);
            } else {
#pragma unroll
                for (int j = 0; j < kLaneNPx; ++j)
                    if (j >= G[r].lo && j < G[r].hi) store1(out + (r * kWavePx + j), px[r][j]);
            }
        }
    }
}

hipError_t launch_composite(const Job *jobs_dev, const Layer *layers_dev, int n_jobs, int n_hot, int pitch,
                            hipStream_t stream) {
    if (n_jobs <= 0 || pitch <= 0) return hipSuccess;
    // grid.x (= pitch) is a multiple of 8 so that (linear workgroup id) mod 8 == (page index) mod 8
    // for every job of the launch: the XCD <-> page residue pairing survives the 2-D grid.
    // Jobs [0, n_hot) take the lean instantiation, the rest the general one (see composite_kernel).
    if (n_hot > 0)
        hipLaunchKernelGGL(composite_kernel<true>, dim3((unsigned)pitch, (unsigned)n_hot, 1), dim3(64), 0,
                           stream, jobs_dev, layers_dev);
    if (n_jobs > n_hot)
        hipLaunchKernelGGL(composite_kernel<false>, dim3((unsigned)pitch, (unsigned)(n_jobs - n_hot), 1),
                           dim3(64), 0, stream, jobs_dev + n_hot, layers_dev);
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

hipError_t launch_fill(void *out, uint32_t rgba, size_t n_px, hipStream_t stream) {
    if (n_px == 0) return hipSuccess;
    const int px_shift = (int)((reinterpret_cast<uint64_t>(out) & 4095u) / 4);
    const size_t pages = (n_px + px_shift + kPagePx - 1) / kPagePx;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)pages), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(out), rgba, (int64_t)n_px, px_shift);
    return hipGetLastError();
}

}  // namespace mic
