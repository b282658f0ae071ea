// Device-side helpers of the composite kernels (kernels_composite.hip, kernels_fused.hip): Pillow's alpha-over
// arithmetic, the page / layer geometry tests, the 16-byte tap loads.  See kernels_composite.hip for the mapping.
#pragma once
#include "mic_internal.h"

namespace mic {

#ifndef MIC_HAVE_U32X4
#define MIC_HAVE_U32X4
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#endif
#ifndef MIC_HAVE_U32X4_A4
#define MIC_HAVE_U32X4_A4
typedef u32x4 __attribute__((aligned(4))) u32x4_a4;  // 16-byte access, 4-byte alignment
#endif

#ifndef MIC_HAVE_DIV255
#define MIC_HAVE_DIV255
__device__ __forceinline__ uint32_t div255_shift(uint32_t t) { return ((t >> 8) + t) >> 8; }
#endif

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
#ifdef MIC_PLAIN_STORES
    *reinterpret_cast<MIC_GLOBAL u32x4_a4 *>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<MIC_GLOBAL u32x4_a4 *>(p));
#endif
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
#ifdef MIC_SRC_NT_LOADS
    return __builtin_nontemporal_load(
        reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(basep + (uint32_t)(t.off * 4 + 16)));
#else
    return *reinterpret_cast<const MIC_GLOBAL u32x4_a4 *>(basep + (uint32_t)(t.off * 4 + 16));
#endif
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

}  // namespace mic
