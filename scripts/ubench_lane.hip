// Timing harness of the lane resample kernel (csrc/kernels_resample_lane.hip) BEFORE its host plumbing existed: the
// 32 layers of the C3 placements workload (sizes from synthetic.placements_workload(3840, 2160, 32, 3)), real unit
// geometry (x-groups of <= 2 tiles per 64-column window, tiles of output rows emitted after their last band, the
// equal-cost cut into one chunk of pieces per wave slot), random source bytes and tap digits -- the arithmetic does not
// depend on the values, only the all-transparent-band shortcut does (never taken on random alpha).
//   (the ablation / probe builds need the scaffolding: git apply profiles/r05_lane_scaffolding.patch)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I image_transformation_amd/csrc -I include \
//         -o build/ubench_lane.bin scripts/ubench_lane.hip
//   build/ubench_lane.bin [slots = 4096] [launches = 50] [binary alpha 0/1] [C0 CL CH CS CV: cost model] [min tiles = 2]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels_resample_lane.hip"

using namespace mic;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static const int kLayers[32][4] = {
    {1130, 308, 898, 245}, {594, 297, 684, 342}, {750, 671, 1112, 995}, {1147, 332, 807, 234}, {1158, 348, 887, 266},
    {642, 690, 333, 358}, {516, 454, 439, 386}, {851, 685, 852, 685}, {568, 379, 293, 195}, {825, 354, 661, 283},
    {801, 649, 1180, 956}, {1109, 590, 1272, 676}, {697, 672, 423, 408}, {890, 592, 787, 524}, {870, 546, 506, 318},
    {1080, 368, 546, 186}, {601, 353, 435, 256}, {748, 670, 924, 828}, {1248, 336, 916, 247}, {938, 568, 1195, 724},
    {598, 530, 576, 511}, {486, 473, 442, 430}, {1134, 524, 600, 277}, {1181, 395, 1327, 444}, {561, 586, 764, 798},
    {509, 397, 520, 406}, {1022, 343, 1037, 348}, {802, 547, 1179, 804}, {836, 619, 1148, 850}, {833, 711, 632, 539},
    {978, 717, 802, 588}, {497, 479, 721, 695}};

struct Axis {  // Resample.c precompute_coeffs' bounds (LANCZOS), per tile of 16 outputs: first / one past last input
    std::vector<int> lo, hi;
};
static Axis axis(int in, int out) {
    Axis a;
    const double scale = (double)in / out, fs = scale < 1 ? 1 : scale, support = 3.0 * fs;
    const int tiles = (out + 15) / 16;
    a.lo.assign(tiles, 1 << 30);
    a.hi.assign(tiles, 0);
    for (int o = 0; o < out; ++o) {
        const double c = (o + 0.5) * scale;
        int first = (int)(c - support + 0.5), last = (int)(c + support + 0.5);
        if (first < 0) first = 0;
        if (last > in) last = in;
        a.lo[o / 16] = std::min(a.lo[o / 16], first);
        a.hi[o / 16] = std::max(a.hi[o / 16], last);
    }
    return a;
}

int main(int argc, char **argv) {
    const int n_slots = argc > 1 ? atoi(argv[1]) : 4096, launches = argc > 2 ? atoi(argv[2]) : 50, binary = argc > 3 ? atoi(argv[3]) : 0;
    // cost model in cycles, fitted by the probe build: piece ~ C0 + bands * (CL + CH T) + tiles * (CS + CV T)
    const double C0 = argc > 4 ? atof(argv[4]) : 7400, CL = argc > 5 ? atof(argv[5]) : 240, CH = argc > 6 ? atof(argv[6]) : 1257,
                 CS = argc > 7 ? atof(argv[7]) : 3570, CV = argc > 8 ? atof(argv[8]) : 470;
    const int MINT = argc > 9 ? atoi(argv[9]) : 2;
    double QW[4] = {1, 1, 1, 1};  // relative chunk cost by quarter of the grid (dispatch order): older workgroups can take more
    for (int k = 0; k < 4; ++k) if (argc > 10 + k) QW[k] = atof(argv[10 + k]);
    const int seg = n_slots;
    std::vector<RsLaneUnit> units;
    size_t src_bytes = 0, dst_bytes = 0, frag_bytes = 0;
    struct L {
        size_t src, dst, hfrag, vfrag, vemit;
        int ct, bands;
    };
    std::vector<L> lay(32);
    std::vector<int32_t> vemit_host;
    long h_passes = 0, v_passes = 0, skipped = 0;
    // layout pass
    std::vector<std::vector<int>> groups(32);  // first x-tile of each group (T = next - this)
    std::vector<Axis> hax(32), vax(32);
    for (int i = 0; i < 32; ++i) {
        const int sw = kLayers[i][0], sh = kLayers[i][1], dw = kLayers[i][2], dh = kLayers[i][3];
        hax[i] = axis(sw, dw);
        vax[i] = axis(sh, dh);
        const int ct = (sw + 15) / 16 + 3, bands = (sh + 15) / 16;
        lay[i].ct = ct; lay[i].bands = bands;
        lay[i].src = src_bytes; src_bytes += (size_t)4 * bands * ct * 256;
        lay[i].dst = dst_bytes; dst_bytes += (size_t)dw * dh * 4;
        const int tx = (dw + 15) / 16, ty = (dh + 15) / 16;
        for (int t = 0; t < tx;) {
            groups[i].push_back(t);
            const int ws = hax[i].lo[t] & ~15;
            if (hax[i].hi[t] - ws > 64) { fprintf(stderr, "layer %d: window > 64\n", i); return 1; }
            if (t + 1 < tx && hax[i].hi[t + 1] - ws <= 64) t += 2; else t += 1;
        }
        groups[i].push_back(tx);
        lay[i].hfrag = frag_bytes; frag_bytes += (size_t)tx * 3072;
        lay[i].vfrag = frag_bytes; frag_bytes += (size_t)ty * 3072;
        lay[i].vemit = vemit_host.size();
        for (int t = 0; t < ty; ++t) {
            const int e = (vax[i].hi[t] - 1) >> 4, f = vax[i].lo[t] >> 4;
            if (e - f > 3) { fprintf(stderr, "layer %d: vertical window > 4 bands\n", i); return 1; }
            int need = 0;
            for (int b = f; b <= e; ++b) need |= 1 << (b & 3);
            vemit_host.push_back(e | (need << 24));  // (entries 4 ints apart, as in the library's axis tables: meta rows)
            vemit_host.push_back(0); vemit_host.push_back(0); vemit_host.push_back(0);
        }
    }
    uint8_t *src, *frag; uint32_t *dst; int32_t *bias, *vemit;
    CK(hipMalloc(&src, src_bytes + 4096)); CK(hipMalloc(&dst, dst_bytes + 4096)); CK(hipMalloc(&frag, frag_bytes + 4096));
    CK(hipMalloc(&bias, 1 << 20)); CK(hipMalloc(&vemit, vemit_host.size() * 4 + 64));
    {
        std::vector<uint8_t> h(std::max(src_bytes, frag_bytes));
        uint32_t s = 12345;
        for (size_t k = 0; k < h.size(); ++k) { s = s * 1664525u + 1013904223u; h[k] = (uint8_t)(s >> 24); }
        if (binary) {  // alpha plane: all 0 or 255 -> signed 0x80 / 0x7f
            for (int i = 0; i < 32; ++i) {
                const size_t plane = (size_t)lay[i].bands * lay[i].ct * 256;
                for (size_t k = 0; k < plane; ++k) h[lay[i].src + 3 * plane + k] = 0x7f;
            }
        }
        CK(hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice));
        for (size_t k = 0; k < frag_bytes; ++k) h[k] = (uint8_t)((h[k] & 7) - 3);  // small digits: sums stay in range
        CK(hipMemcpy(frag, h.data(), frag_bytes, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, 1 << 20));
        CK(hipMemcpy(vemit, vemit_host.data(), vemit_host.size() * 4, hipMemcpyHostToDevice));
    }
    // equal-cost cut of the 1-D sequence of column strips (layer, x-group) x tiles of output rows into n_slots chunks;
    // a chunk = 1 piece, or 2+ where the cut falls across the end of a strip.  Cost in tile passes (model constants
    // from the command line: prologue C0, per band CL + T, per tile CS + T).
    auto piece_cost = [&](int i, int T, int y0, int y1) {
        const int nb = ((vax[i].hi[y1 - 1] - 1) >> 4) - (vax[i].lo[y0] >> 4) + 1;
        return C0 + nb * (CL + CH * T) + (y1 - y0) * (CS + CV * T);
    };
    std::vector<uint32_t> first;
    double target = 0;
    {
        double total = 0;
        for (int i = 0; i < 32; ++i)
            for (size_t g = 0; g + 1 < groups[i].size(); ++g)
                total += piece_cost(i, groups[i][g + 1] - groups[i][g], 0, (kLayers[i][3] + 15) / 16);
        target = total / n_slots;
    }
    for (int attempt = 0; attempt < 60; ++attempt, target *= 1.01) {
        units.clear(); first.clear(); h_passes = v_passes = 0;
        double acc = 0;
        first.push_back(0);
        const double base_target = target;
        auto slot_target = [&]() { const int qtr = std::min(3, (int)(4 * (first.size() - 1) / std::max(1, n_slots))); return base_target * QW[qtr]; };
        double target = slot_target();
        for (int i = 0; i < 32; ++i) {
            const int dw = kLayers[i][2], dh = kLayers[i][3], ty = (dh + 15) / 16;
            for (size_t g = 0; g + 1 < groups[i].size(); ++g) {
                const int t0 = groups[i][g], T = groups[i][g + 1] - t0;
                const int ws = hax[i].lo[t0] & ~15;
                int y0 = 0;
                while (y0 < ty) {
                    int y1 = y0;  // the longest piece that still fits the chunk
                    while (y1 < ty && acc + piece_cost(i, T, y0, y1 + 1) <= target) ++y1;
                    if (y1 - y0 < std::min(MINT, ty - y0)) {
                        if (acc > 0) {  // does not fit: close the chunk
                            first.push_back((uint32_t)units.size());
                            acc = 0;
                            target = slot_target();
                            continue;
                        }
                        y1 = std::min(ty, y0 + MINT);  // (an empty chunk takes at least MINT tiles)
                    }
                    RsLaneUnit u{};
                    u.T = T; u.n_vtiles = y1 - y0;
                    u.hfrag = (uint64_t)(frag + lay[i].hfrag + (size_t)t0 * 3072);
                    u.hbias = (uint64_t)bias;
                    u.dst = (uint64_t)((uint8_t *)dst + lay[i].dst);
                    u.dw = dw; u.dh = dh; u.x0 = 16 * t0; u.row0 = 16 * y0;
                    u.plane_bytes = (uint32_t)((size_t)lay[i].bands * lay[i].ct * 256);
                    u.band_bytes = (uint32_t)(lay[i].ct * 256);
                    u.band0 = vax[i].lo[y0] >> 4;
                    u.band_last = (vax[i].hi[y1 - 1] - 1) >> 4;
                    u.vfrag = (uint64_t)(frag + lay[i].vfrag + (size_t)y0 * 3072);
                    u.vbias = (uint64_t)bias;
                    u.vemit = (uint64_t)(vemit + lay[i].vemit + 4 * y0);
                    u.src = (uint64_t)(src + lay[i].src + ((size_t)u.band0 * lay[i].ct + ws / 16) * 256);
                    h_passes += (long)(u.band_last - u.band0 + 1) * T;
                    v_passes += (long)u.n_vtiles * T;
                    acc += piece_cost(i, T, y0, y1);
                    units.push_back(u);
                    y0 = y1;
                    if (acc >= 0.97 * target) {
                        first.push_back((uint32_t)units.size());
                        acc = 0;
                        target = slot_target();
                    }
                }
            }
        }
        if (first.back() != units.size()) first.push_back((uint32_t)units.size());
        if ((int)first.size() - 1 <= n_slots) break;
    }
    const int chunks = (int)first.size() - 1;
    while ((int)first.size() - 1 < (chunks + 31) / 32 * 32) first.push_back((uint32_t)units.size());  // whole workgroups, 8 at a time
    const int slots_used = (int)first.size() - 1;
    // chunk -> slot: XCD-contiguous (workgroup w runs on XCD w mod 8: the chunk sequence is cut into 8 runs, run k goes
    // to the workgroups k, k + 8, k + 16, ...) or in order (XCDMAP=0)
    const bool xcdmap = getenv("XCDMAP") ? atoi(getenv("XCDMAP")) != 0 : true;
    // records [0, slots): the slots' first pieces; further pieces of a chunk chained behind through `next` (as the library does)
    std::vector<RsLaneUnit> dealt((size_t)slots_used);
    {
        const int n_wg = slots_used / 4, per = (n_wg + 7) / 8;  // workgroups per XCD
        for (int c = 0; c < slots_used; ++c) {
            int slot = c;
            if (xcdmap) {
                const int k = c / (4 * per), j = c % (4 * per);
                const int wg = 8 * (j / 4) + k;
                slot = wg < n_wg ? 4 * wg + (j % 4) : -1;
            }
            if (slot < 0) { fprintf(stderr, "slot map overflow\n"); return 1; }
            if (first[c] == first[c + 1]) continue;
            dealt[slot] = units[first[c]];
            size_t prev = (size_t)slot;
            for (uint32_t r = first[c] + 1; r < first[c + 1]; ++r) {
                dealt[prev].next = (uint32_t)dealt.size();
                prev = dealt.size();
                dealt.push_back(units[r]);
            }
        }
    }
    units.swap(dealt);
    uint64_t *probe_dev = nullptr;
    CK(hipMalloc(&probe_dev, units.size() * 32));
    CK(hipMemset(probe_dev, 0, units.size() * 32));
    for (size_t k = 0; k < units.size(); ++k) {
        const uint64_t a = (uint64_t)(probe_dev + 4 * k);
        memcpy(&units[k].pad[0], &a, 8);
    }
    RsLaneUnit *units_dev;
    CK(hipMalloc(&units_dev, units.size() * sizeof(RsLaneUnit)));
    CK(hipMemcpy(units_dev, units.data(), units.size() * sizeof(RsLaneUnit), hipMemcpyHostToDevice));
    printf("slots %d (asked %d): %zu pieces in %d chunks, target cost %.1f; horizontal tile passes %ld, vertical %ld; src %.1f MB, dst %.1f MB\n",
           slots_used, n_slots, units.size(), chunks, target, h_passes, v_passes, src_bytes / 1e6, dst_bytes / 1e6);
    (void)skipped;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 5; ++k) CK(launch_resample_lane(units_dev, slots_used, 0));
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int k = 0; k < launches; ++k) {
        CK(hipEventRecord(e0, 0));
        CK(launch_resample_lane(units_dev, slots_used, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms); sum += ms;
    }
    // back to back (what a stream of calls sees)
    CK(hipEventRecord(e0, 0));
    for (int k = 0; k < launches; ++k) CK(launch_resample_lane(units_dev, slots_used, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("seg %d alpha %s: events around one launch mean %.2f us, min %.2f us; back to back %.2f us per launch\n", seg,
           binary ? "binary" : "soft", sum / launches * 1e3, best * 1e3, ms / launches * 1e3);
#ifdef MIC_LANE_PROBE
    {
        std::vector<uint64_t> pr(units.size() * 4);
        CK(hipMemcpy(pr.data(), probe_dev, pr.size() * 8, hipMemcpyDeviceToHost));
        // per chunk (= wave): total cycles; least squares of piece cycles on (1, bands, bands*T, tiles, tiles*T)
        double sum = 0, mx = 0, mn = 1e18, wsum = 0, psum = 0, rsum = 0;
        uint64_t t_first = ~0ull, t_last = 0;
        std::vector<double> chunk_cyc;
        for (size_t c = 0; c + 1 < first.size(); ++c) {
            double cyc = 0;
            for (uint32_t r = first[c]; r < first[c + 1]; ++r) { cyc += (double)pr[4 * r]; wsum += (double)pr[4 * r + 1]; psum += (double)(pr[4 * r + 2] & 0xffffffffu); rsum += (double)(pr[4 * r + 2] >> 32);
                t_first = std::min(t_first, pr[4 * r + 3]); t_last = std::max(t_last, pr[4 * r + 3]); }
            if (first[c + 1] > first[c]) { chunk_cyc.push_back(cyc); sum += cyc; mx = std::max(mx, cyc); mn = std::min(mn, cyc); }
        }
        {   // by dispatch order: mean cycles (and mean end time after the first end) of the waves of each eighth of the grid
            const size_t nc = first.size() - 1;
            printf("probe: by eighth of the grid (workgroup index): cycles / end us:");
            for (int e = 0; e < 8; ++e) {
                double cs = 0, es = 0; int k = 0;
                for (size_t c = nc * e / 8; c < nc * (e + 1) / 8; ++c) {
                    if (first[c + 1] == first[c]) continue;
                    double cyc = 0; uint64_t end = 0;
                    for (uint32_t r = first[c]; r < first[c + 1]; ++r) { cyc += (double)pr[4 * r]; end = std::max(end, pr[4 * r + 3]); }
                    cs += cyc; es += (double)(end - t_first) / 100.0; ++k;
                }
                printf(" %.0f / %.1f", cs / std::max(k, 1), es / std::max(k, 1));
            }
            printf("\n");
        }
        std::sort(chunk_cyc.begin(), chunk_cyc.end());
        const size_t n = chunk_cyc.size();
        printf("probe: shader clock over the pieces' lives: %.0f MHz (s_memtime cycles / s_memrealtime 10 ns ticks)\n", sum / rsum * 100.0);
        printf("probe: %zu waves: cycles per wave mean %.0f, min %.0f, p10 %.0f, p50 %.0f, p90 %.0f, p99 %.0f, max %.0f; waiting for the band %.1f %%, prologue %.1f %% of wave cycles; "
               "first-to-last piece end %.2f us (100 MHz realtime)\n", n, sum / n, mn, chunk_cyc[n / 10], chunk_cyc[n / 2], chunk_cyc[n * 9 / 10], chunk_cyc[n * 99 / 100], mx,
               100 * wsum / sum, 100 * psum / sum, (double)(t_last - t_first) / 100.0);
        // normal equations, 5 unknowns
        double M[5][6] = {};
        for (size_t k = 0; k < units.size(); ++k) {
            const RsLaneUnit &u = units[k];
            const double nb = u.band_last - u.band0 + 1, nt = u.n_vtiles;
            const double x[5] = {1, nb, nb * u.T, nt, nt * u.T}, y = (double)pr[4 * k];
            for (int a = 0; a < 5; ++a) { for (int b = 0; b < 5; ++b) M[a][b] += x[a] * x[b]; M[a][5] += x[a] * y; }
        }
        for (int a = 0; a < 5; ++a) {
            int piv = a; for (int r = a + 1; r < 5; ++r) if (fabs(M[r][a]) > fabs(M[piv][a])) piv = r;
            for (int c = 0; c < 6; ++c) std::swap(M[a][c], M[piv][c]);
            for (int r = 0; r < 5; ++r) if (r != a) { const double f = M[r][a] / M[a][a]; for (int c = a; c < 6; ++c) M[r][c] -= f * M[a][c]; }
        }
        printf("probe: piece cycles ~ %.0f + bands * (%.0f + T * %.0f) + tiles * (%.0f + T * %.0f)\n", M[0][5] / M[0][0], M[1][5] / M[1][1], M[2][5] / M[2][2], M[3][5] / M[3][3], M[4][5] / M[4][4]);
    }
#endif
    std::vector<uint32_t> out(1024);
    CK(hipMemcpy(out.data(), dst, 4096, hipMemcpyDeviceToHost));
    uint32_t x = 0; for (uint32_t v : out) x ^= v;
    printf("checksum %08x\n", x);
    return 0;
}
