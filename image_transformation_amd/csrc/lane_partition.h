// Host side of the lane resample kernel (kernels_resample_lane.hip): the cost model of a piece and the cut of a launch's
// strips x tiles of output rows into equal-cost chunks, one per wave slot, dealt to the slots XCD by XCD.  Plain C++ (no
// HIP): mic_api.hip builds the strips from its plans and tables; tests/native/lane_partition_main.cpp runs the cut under
// AddressSanitizer / UBSan on random layers and checks that every tile of every strip is emitted exactly once.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <tuple>
#include <vector>

#include "lane_unit.h"

namespace mic {

// Cost of a lane piece in shader cycles of one wave among four per SIMD (fitted on the C3 placements call with a
// cycle-stamp build, scripts/ubench_lane.hip): prologue + bands x (window loads + T horizontal tile passes) + tiles of
// output rows x (tap fetch, stores + T vertical tile passes and epilogues).  Only the ratios matter: the pieces of a
// launch are cut so that every wave slot gets the same sum.
constexpr double kLaneC0 = 7000, kLaneCL = 200, kLaneCH = 1200, kLaneCS = 3500, kLaneCV = 400;
// the same for a whole layer, from its sizes alone (two x-tiles per strip assumed): what decides whether a call is big
// enough for the lane kernel before any table exists
inline double lane_layer_cost(int sh, int dw, int dh) {
    const double tx = (dw + 15) / 16, ty = (dh + 15) / 16, bands = (sh + 15) / 16;
    return tx * (kLaneCH * bands + kLaneCV * ty) + tx / 2 * (kLaneCL * bands + kLaneCS * ty);
}

struct LaneStrip {   // one column strip of a layer: T x-tiles from t0 (window start ws), every tile of output rows
    int t0, T, ws;
    int sh, dw, dh;              // source rows, output size of the layer
    int tiled_ct;                // tiles per band of the cutout's tiled planar copy
    uint64_t tiled_src, dst;     // device addresses: plane 0 of the tiled copy, the layer's pixels
    uint64_t hfrag, hbias;       // the horizontal axis' lane tables (kFragsLaneH): fragments, bias
    uint64_t vfrag, vbias, vmeta;  // the vertical axis' (kFragsLaneV): fragments, bias, meta rows (device)
    const int32_t *vm;           // ... and the meta rows on the host: [ty][4] = {first tap row, last band | ring words << 24, chunk, end}
    int ty;                      // tiles of 16 output rows
    int cls;                     // kLaneGeneral / kLaneKeepsWidth / kLaneKeepsHeight (lane_unit.h)
};

struct LaneCut {
    std::vector<RsLaneUnit> records;  // [0, slots): the slots' first pieces (n_vtiles == 0: nothing to do); chained pieces behind
    int slots = 0, chunks = 0, passes = 0;
};

inline double lane_piece_cost(const int32_t *vm, int T, int y0, int y1) {
    const int nb = (vm[4 * (y1 - 1) + 1] & 0xFFFFFF) - (vm[4 * y0] >> 4) + 1;
    return kLaneC0 + nb * (kLaneCL + kLaneCH * T) + (y1 - y0) * (kLaneCS + kLaneCV * T);
}

// Cut the 1-D sequence of strips x tiles of output rows into at most max_slots chunks of equal cost (a chunk = the
// pieces of one wave: one piece, or two where the cut falls across the end of a strip), and deal the chunks to the
// wave slots XCD by XCD: workgroup w runs on XCD w mod 8 (round-robin dispatch, observed), so run k of the chunk
// sequence goes to workgroups k, k + 8, k + 16, ... -- strips that share source columns, a layer's vertical taps and
// neighbouring output rows then share an L2 (dealt in launch order the same launch moved 2.4x the bytes over the
// fabric and took 43 us instead of 35: profiles/r05_lane_kernel.txt).
inline void lane_partition(const std::vector<LaneStrip> &strips, double chunk_cost, int max_slots, LaneCut *out) {
    std::vector<RsLaneUnit> &lane = out->records;
    static const bool trace = getenv("MIC_LANE_TRACE") != nullptr;  // one line per launch on stderr (tuning)
    const auto t_begin = std::chrono::steady_clock::now();
    int attempts = 0;
    double total = 0;
    for (const LaneStrip &s : strips) total += lane_piece_cost(s.vm, s.T, 0, s.ty);
    // How many wave slots: small calls are cut fine (chunk_cost, ~2 tiles of output rows: a piece's prologue is half of
    // that, but such a launch is over in 10-20 us and only parallelism shortens it); once that would exceed the 4096
    // waves the chip holds at this kernel's occupancy (256 CUs x 4 SIMDs x 4) the launch is WHOLE rounds of 4096 slots
    // of ~55 K cycles each -- a partial last round is a tail with three quarters of the chip idle (C3 placements canvas:
    // 38.9 us at 4096 slots, 43-46 at 3 700, 5 500 or 8 800), and within a round neighbouring chunks (which share
    // source columns and taps through the XCD's L2) run together.  profiles/r05_lane_kernel.txt.
    constexpr int kRound = 4096;
    double n = total / chunk_cost;
    const bool whole_rounds = n > kRound;
    if (n > kRound) n = kRound * std::max(1.0, std::floor(total / (kRound * 55000.0) + 0.5));
    int n_slots = (int)std::min<double>(max_slots, std::max(32.0, n));
    // (whole workgroups, eight at a time; rounded UP: a call of 56 slots of work cut into 32 chunks gave every strip a chunk and a
    // half -- two pieces in series per wave, 11.9 us against 9.4 for its neighbours in size)
    n_slots = std::max(32, (n_slots + 31) / 32 * 32);
    // every cut re-does up to three bands at the top of the next piece: ~ half a prologue + 2 bands per slot
    // (+ 3 %: what the greedy cut loses at chunk ends; with it the first pass nearly always fits -- a pass is ~100 us of host time)
    double target = 1.03 * (total + n_slots * (kLaneC0 + 2 * (kLaneCL + 2 * kLaneCH))) / n_slots;
    std::vector<uint32_t> first;
    lane.reserve((size_t)n_slots + n_slots / 4 + strips.size());
    for (int attempt = 0; attempt < 40; ++attempt, target *= 1.03) {
        ++attempts;
        lane.clear();
        first.assign(1, 0u);
        double acc = 0;
        const double tgt = target;
        for (const LaneStrip &s : strips) {
            const int32_t *vm = s.vm;
            const int ty = s.ty;
            int y0 = 0;
            while (y0 < ty) {
                // the longest piece that still fits the chunk: cost grows with y1, nearly linearly -- start from the
                // tile count the strip's average cost per tile row predicts and walk (a step or two) to the exact answer
                int lo = y0;  // [y0, lo) fits (lo == y0: nothing yet)
                if (acc + lane_piece_cost(vm, s.T, y0, ty) <= tgt) {
                    lo = ty;
                } else {
                    const double per_tile = (kLaneCS + kLaneCV * s.T) + (kLaneCL + kLaneCH * s.T) * ((s.sh + 15) / 16) / (double)ty;
                    const double room = tgt - acc - kLaneC0 - 2 * (kLaneCL + kLaneCH * s.T);
                    lo = std::min(ty - 1, std::max(y0, y0 + (int)(room / per_tile)));
                    while (lo > y0 && acc + lane_piece_cost(vm, s.T, y0, lo) > tgt) --lo;
                    while (lo < ty - 1 && acc + lane_piece_cost(vm, s.T, y0, lo + 1) <= tgt) ++lo;
                }
                int y1 = lo;
                if (y1 - y0 < std::min(2, ty - y0)) {
                    if (acc > 0) {  // does not fit: close the chunk
                        first.push_back((uint32_t)lane.size());
                        acc = 0;
                        continue;
                    }
                    y1 = std::min(ty, y0 + 2);  // (an empty chunk takes at least two tiles)
                }
                RsLaneUnit u{};
                u.T = s.T; u.n_vtiles = y1 - y0; u.cls = s.cls;
                u.band0 = vm[4 * y0] >> 4;
                u.band_last = vm[4 * (y1 - 1) + 1] & 0xFFFFFF;
                u.plane_bytes = (uint32_t)((size_t)((s.sh + 15) / 16) * s.tiled_ct * 256);
                u.band_bytes = (uint32_t)(s.tiled_ct * 256);
                u.src = s.tiled_src + ((uint64_t)u.band0 * s.tiled_ct + s.ws / 16) * 256;
                u.dst = s.dst;
                u.hfrag = s.hfrag + (uint64_t)s.t0 * 3072;
                u.hbias = s.hbias + (uint64_t)s.t0 * 64;
                u.vfrag = s.vfrag + (uint64_t)y0 * 3072;
                u.vbias = s.vbias + (uint64_t)y0 * 64;
                u.vemit = s.vmeta + ((uint64_t)4 * y0 + 1) * 4;
                u.x0 = 16 * s.t0; u.row0 = 16 * y0; u.dw = s.dw; u.dh = s.dh;
                acc += lane_piece_cost(vm, s.T, y0, y1);
                lane.push_back(u);
                y0 = y1;
                if (acc >= 0.97 * tgt) {
                    first.push_back((uint32_t)lane.size());
                    acc = 0;
                }
            }
        }
        if (first.back() != lane.size()) first.push_back((uint32_t)lane.size());
        if ((int)first.size() - 1 <= n_slots) break;
    }
    const int chunks = (int)first.size() - 1;
    const int slots = std::max(32, (chunks + 31) / 32 * 32);  // whole workgroups, eight at a time
    first.resize((size_t)slots + 1, (uint32_t)lane.size());
    out->slots = slots;
    // records [0, slots): the slots' first pieces (a wave finds its work with ONE scalar load); further pieces of a
    // chunk follow behind, chained through `next`
    std::vector<RsLaneUnit> dealt((size_t)slots);
    dealt.reserve((size_t)slots + lane.size() - (size_t)chunks + 8);
    const int n_wg = slots / 4, per = n_wg / 8;
    static const bool in_order = [] { const char *e = getenv("MIC_RS_LANE_XCDMAP"); return e && atoi(e) == 0; }();  // (A/B: chunk c -> slot c)
    // Launches of whole rounds deal their chunks in the order (layer, rows, columns) rather than the order of the cut (layer,
    // columns, rows): pieces side by side in x become neighbours in the sequence.  The fabric bytes do not move (98.1 -> 97.4
    // MB: the shared halves of the windows were L2 hits already); binary-alpha cutouts gain 3.7 % (35.0 -> 33.7 us: their
    // all-transparent top and bottom pieces, which skip the arithmetic, no longer share a workgroup with full ones), soft
    // alpha and the batch are level; calls below a round lose 4 % and keep the cut's order.  MIC_LANE_ORDER=0: always the cut's.
    static const bool by_rows = [] { const char *e = getenv("MIC_LANE_ORDER"); return !e || atoi(e) != 0; }();
    std::vector<int> order((size_t)slots);
    for (int c = 0; c < slots; ++c) order[(size_t)c] = c;
    if (by_rows && whole_rounds) {
        auto key = [&](int c) {
            const uint32_t b0 = first[(size_t)c];
            if (b0 == first[(size_t)c + 1]) return std::make_tuple(~uint64_t(0), 0, 0);  // (empty slots last)
            const RsLaneUnit &u = lane[b0];
            return std::make_tuple(u.dst, u.row0, u.x0);
        };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
    }
    for (int ci = 0; ci < slots; ++ci) {
        const int c = order[(size_t)ci];
        const int k = ci / (4 * per), j = ci % (4 * per);
        const size_t slot = in_order ? (size_t)ci : (size_t)(4 * (8 * (j / 4) + k) + (j % 4));
        const uint32_t b0 = first[(size_t)c], b1 = first[(size_t)c + 1];
        if (b0 == b1) continue;  // (an empty slot: the zero record, n_vtiles == 0)
        dealt[slot] = lane[b0];
        size_t prev = slot;
        for (uint32_t r = b0 + 1; r < b1; ++r) {
            dealt[prev].next = (uint32_t)dealt.size();
            prev = dealt.size();
            dealt.push_back(lane[r]);
        }
    }
    lane.swap(dealt);
    out->chunks = chunks;
    out->passes = attempts;
    if (trace)
        fprintf(stderr, "lane_partition: %zu strips, model cost %.0f, %d slots asked, %d chunks, %zu records, %d pass(es), %.0f us\n",
                strips.size(), total, n_slots, chunks, lane.size(), attempts,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
}

}  // namespace mic
