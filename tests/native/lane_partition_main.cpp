// csrc/lane_partition.h (the host-side cut of a lane-kernel launch) under AddressSanitizer / UBSan: random layers through
// the real axis tables (resample_coeffs.cpp, lane forms), random chunk costs and slot caps; every tile of output rows of
// every strip must be emitted exactly once, every record must be reachable from exactly one wave slot, and what a record
// says about its bands must be what the vertical table says about its tiles.
#include <cstdio>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <vector>

#include "lane_partition.h"
#include "resample_coeffs.h"

int main() {
    std::mt19937 rng(20251005);
    auto uni = [&](int lo, int hi) { return (int)(lo + rng() % (unsigned)(hi - lo + 1)); };
    long launches = 0, pieces_total = 0, chained = 0;
    for (int trial = 0; trial < 60; ++trial) {
        const int n_layers = trial % 7 == 0 ? 1 : uni(1, 40);
        std::vector<mic::LaneStrip> strips;
        std::vector<std::unique_ptr<mic::AxisFrags>> keep;
        uint64_t next_addr = 0x100000;
        for (int l = 0; l < n_layers; ++l) {
            const int sw = uni(20, 1400), sh = uni(20, 900);
            const double sc = 0.5 + (rng() % 1000) / 1000.0 * (trial % 5 == 0 ? 7.5 : 1.2);
            int dw = std::max(1, (int)(sw * sc)), dh = std::max(1, (int)(sh * sc));
            if (l % 5 == 1) dw = sw;  // a layer that keeps its width ...
            if (l % 5 == 3) dh = sh;  // ... its height (lane_unit.h: piece classes)
            const int filter = (int)(rng() % 2);
            const mic::AxisTable th = sw == dw ? mic::identity_axis_table(sw) : mic::build_axis_table(sw, dw, filter);
            const mic::AxisTable tv = sh == dh ? mic::identity_axis_table(sh) : mic::build_axis_table(sh, dh, filter);
            keep.emplace_back(new mic::AxisFrags(mic::build_axis_frags(th, mic::kFragsLaneH)));
            const mic::AxisFrags &fh = *keep.back();
            keep.emplace_back(new mic::AxisFrags(mic::build_axis_frags(tv, mic::kFragsLaneV)));
            const mic::AxisFrags &fv = *keep.back();
            if (fh.max_chunks != 1 || fv.max_chunks != 1) continue;  // (the layer would go to the tile kernel)
            mic::LaneStrip st{};
            st.sh = sh; st.dw = dw; st.dh = dh;
            st.cls = sw == dw ? mic::kLaneKeepsWidth : sh == dh ? mic::kLaneKeepsHeight : mic::kLaneGeneral;
            st.tiled_ct = (sw + 15) / 16 + 3;
            st.tiled_src = next_addr; next_addr += (uint64_t)4 * ((sh + 15) / 16) * st.tiled_ct * 256 + 4096;
            st.dst = next_addr; next_addr += (uint64_t)dw * dh * 4 + 4096;
            st.hfrag = next_addr; next_addr += (uint64_t)fh.tiles * 3072;
            st.hbias = next_addr; next_addr += (uint64_t)fh.tiles * 64;
            st.vfrag = next_addr; next_addr += (uint64_t)fv.tiles * 3072;
            st.vbias = next_addr; next_addr += (uint64_t)fv.tiles * 64;
            st.vmeta = next_addr; next_addr += (uint64_t)fv.tiles * 16;
            st.vm = fv.meta.data(); st.ty = fv.tiles;
            for (int t = 0; t < fh.tiles; ++t) {
                if (fh.meta[4 * t + 1] == 0) continue;
                st.t0 = t; st.T = fh.meta[4 * t + 1]; st.ws = fh.meta[4 * t];
                strips.push_back(st);
            }
        }
        if (strips.empty()) continue;
        const double chunk = (double[]){5000, 15000, 15000, 60000, 250000}[trial % 5];
        const int cap = (int[]){32, 4096, 1 << 20, 256, 8192}[trial % 5];
        mic::LaneCut cut;
        mic::lane_partition(strips, chunk, cap, &cut);
        ++launches;
        const std::vector<mic::RsLaneUnit> &R = cut.records;
        if (cut.slots < 32 || cut.slots % 32 != 0 || (int)R.size() < cut.slots) { fprintf(stderr, "slots\n"); return 2; }
        std::map<std::pair<uint64_t, int>, const mic::LaneStrip *> by_key;  // (dst, x0) -> strip
        for (const mic::LaneStrip &s : strips) by_key[{s.dst, 16 * s.t0}] = &s;
        std::map<std::pair<uint64_t, int>, std::vector<char>> covered;
        std::vector<char> reached(R.size(), 0);
        for (int slot = 0; slot < cut.slots; ++slot) {
            uint32_t r = (uint32_t)slot;
            int guard = 0;
            do {
                if (r >= R.size() || reached[r] || ++guard > 1000) { fprintf(stderr, "chain\n"); return 3; }
                reached[r] = 1;
                const mic::RsLaneUnit &u = R[r];
                if (r != (uint32_t)slot) ++chained;
                if (u.n_vtiles > 0) {
                    ++pieces_total;
                    auto it = by_key.find({u.dst, u.x0});
                    if (it == by_key.end()) { fprintf(stderr, "a record of no strip\n"); return 4; }
                    const mic::LaneStrip &s = *it->second;
                    const int y0 = u.row0 / 16, y1 = y0 + u.n_vtiles;
                    if (u.row0 % 16 || y0 < 0 || y1 > s.ty || u.T != s.T || u.dw != s.dw || u.dh != s.dh || u.cls != s.cls) { fprintf(stderr, "geometry\n"); return 5; }
                    if (u.band0 != (s.vm[4 * y0] >> 4) || u.band_last != (s.vm[4 * (y1 - 1) + 1] & 0xFFFFFF) || u.band0 > u.band_last ||
                        u.band_last >= (s.sh + 15) / 16) { fprintf(stderr, "bands\n"); return 6; }
                    if (u.vfrag != s.vfrag + (uint64_t)y0 * 3072 || u.vbias != s.vbias + (uint64_t)y0 * 64 || u.vemit != s.vmeta + ((uint64_t)4 * y0 + 1) * 4 ||
                        u.hfrag != s.hfrag + (uint64_t)s.t0 * 3072 || u.hbias != s.hbias + (uint64_t)s.t0 * 64) { fprintf(stderr, "tables\n"); return 7; }
                    // the four window tiles of every band the piece marches through lie inside the plane
                    const uint64_t off = u.src - s.tiled_src, last = off + (uint64_t)(u.band_last - u.band0) * u.band_bytes + 1024;
                    if (u.src < s.tiled_src || last > u.plane_bytes || u.band_bytes != (uint32_t)s.tiled_ct * 256) { fprintf(stderr, "window\n"); return 8; }
                    std::vector<char> &cv = covered[it->first];
                    cv.resize((size_t)s.ty, 0);
                    for (int y = y0; y < y1; ++y) {
                        if (cv[(size_t)y]) { fprintf(stderr, "a tile emitted twice\n"); return 9; }
                        cv[(size_t)y] = 1;
                    }
                } else if (r >= (uint32_t)cut.slots) { fprintf(stderr, "an empty chained record\n"); return 10; }
                r = u.next;
                if (r != 0 && r < (uint32_t)cut.slots) { fprintf(stderr, "a chain into the slot heads\n"); return 11; }
            } while (r != 0);
        }
        for (size_t r = 0; r < R.size(); ++r)
            if (!reached[r]) { fprintf(stderr, "an unreachable record\n"); return 12; }
        for (const auto &kv : by_key) {
            const std::vector<char> &cv = covered[kv.first];
            if ((int)cv.size() != kv.second->ty) { fprintf(stderr, "a strip without pieces\n"); return 13; }
            for (char c : cv)
                if (!c) { fprintf(stderr, "a tile never emitted\n"); return 14; }
        }
    }
    printf("launches=%ld pieces=%ld chained=%ld ok\n", launches, pieces_total, chained);
    return 0;
}
