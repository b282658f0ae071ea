// Pillow-exact separable resampling of RGBA cutouts (Image.resize(size, LANCZOS), call site
// compositor.py:20; thumbnails macro_placement_test.py:194).
//
// Restates Pillow's 8 bpc path: premultiply (Convert.c rgbA2rgba) -> horizontal pass -> 8-bit
// intermediate -> vertical pass -> unpremultiply (Convert.c rgba2rgbA).  Each output sample is
//   clip8((2^21 + sum_k in[first + k] * K[k]) >> 22)
// with int32 fixed-point coefficients K built on the host in double precision exactly as
// Resample.c precompute_coeffs / normalize_coeffs_8bpc do (resample_coeffs.cpp), so the device
// passes are pure integer arithmetic and bit-exact.
//
// One launch processes every resampled layer of a composite call (blockIdx.z = layer).
#include <algorithm>
#include <atomic>

#include "mic_internal.h"
#include "resample_mfma.h"

namespace mic {

__device__ __forceinline__ uint32_t div255_shift(uint32_t t) { return ((t >> 8) + t) >> 8; }

__device__ __forceinline__ uint32_t premultiply(uint32_t p) {
    const uint32_t a = p >> 24;
    const uint32_t r = div255_shift((p & 255u) * a + 128u);
    const uint32_t g = div255_shift(((p >> 8) & 255u) * a + 128u);
    const uint32_t b = div255_shift(((p >> 16) & 255u) * a + 128u);
    return r | (g << 8) | (b << 16) | (a << 24);
}

// Convert.c rgba2rgbA: c = min(255, 255*c' / a) for 0 < a < 255.  The integer division is replaced
// by one v_mul_hi_u32 with R[a] = ceil(255 * 2^24 / a): floor(c' * R[a] / 2^24) == floor(255*c'/a)
// exactly for c' in 0..255 (the excess is < 2^-16, the quotient's fractional part is a multiple
// of 1/a <= 1 - 1/254); checked exhaustively in tests/test_blend_identities.py.
struct UnpremulTable {
    uint32_t r[256];
    constexpr UnpremulTable() : r{} {
        for (uint32_t a = 1; a < 256; ++a) r[a] = (uint32_t)((((uint64_t)255 << 24) + a - 1) / a);
    }
};
__device__ __constant__ UnpremulTable kUnpremul{};

__device__ __forceinline__ uint32_t unpremultiply(uint32_t p) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const uint32_t R = kUnpremul.r[a];
    const uint32_t r = min(255u, __umulhi((p & 255u) << 8, R));
    const uint32_t g = min(255u, __umulhi(p & 0xFF00u, R));
    const uint32_t b = min(255u, __umulhi((p >> 8) & 0xFF00u, R));
    return r | (g << 8) | (b << 16) | (a << 24);
}

__device__ __forceinline__ uint32_t unpremultiply_with(uint32_t p, const float *table) {
    const uint32_t a = p >> 24;
    if (a == 0u || a == 255u) return p;
    const float F = table[a];
    const uint32_t r = min(255u, (uint32_t)((float)(p & 255u) * F));
    const uint32_t g = min(255u, (uint32_t)((float)((p >> 8) & 255u) * F));
    const uint32_t b = min(255u, (uint32_t)((float)((p >> 16) & 255u) * F));
    return r | (g << 8) | (b << 16) | (a << 24);
}

__device__ __forceinline__ uint32_t clip8(int32_t v) {
    v >>= kPrecisionBits;  // arithmetic shift, like Pillow's clip8 lookup index
    // hipcc (ROCm 7.2, gfx950) fuses "shift, clamp to 0..255, pack" into v_ashr_pk_u8_i32 and then
    // ORs the 16-bit result as if the destination's upper half were zero; it is not (the
    // instruction only writes D[15:0]), which corrupted blue/alpha on the MI355X.  The empty asm
    // keeps the shift and the clamp apart so the clamp lowers to v_med3_i32.
    asm volatile("" : "+v"(v));
    return (uint32_t)min(255, max(0, v));
}

// acc += channel * tap for the four channels of one pixel.  Both factors fit 24 bits (bytes; taps
// are 22-bit fixed point, |k| < 2^23), so v_mad_i32_i24 is exact -- and is what must be asked for:
// a plain int32 multiply-add made hipcc emit 64-bit v_mad_u64_u32, several times slower.
__device__ __forceinline__ void mac4(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, uint32_t p, int32_t k) {
    s0 += __mul24((int)(p & 255u), k);
    s1 += __mul24((int)((p >> 8) & 255u), k);
    s2 += __mul24((int)((p >> 16) & 255u), k);
    s3 += __mul24((int)(p >> 24), k);
}

__device__ __forceinline__ uint32_t pack_clip(int32_t s0, int32_t s1, int32_t s2, int32_t s3) {
    return clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16) | (clip8(s3) << 24);
}

// Horizontal pass: one thread per output pixel (x', y).  Coefficients are stored TRANSPOSED
// ([ksize][out_w]) so that the 64 lanes of a wave read consecutive words for each tap.
__global__ __launch_bounds__(256) void resample_h_kernel(const RsJob *__restrict__ jobs) {
    const RsJob J = jobs[blockIdx.z];
    const int xx = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (y >= J.in_h || xx >= J.out_w) return;
    gciptr bounds = reinterpret_cast<gciptr>(J.bounds);
    gciptr kt = reinterpret_cast<gciptr>(J.coeffs) + xx;
    const int first = bounds[2 * xx], n = bounds[2 * xx + 1];
    gcptr row = reinterpret_cast<gcptr>(J.src) + (size_t)y * J.in_w + first;
    const bool pre = (J.flags & kRsPremultiplyOnLoad) != 0;
    int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < n; ++i) {
        uint32_t p = row[i];
        if (pre) p = premultiply(p);
        const int32_t k = kt[(size_t)i * J.out_w];
        mac4(s0, s1, s2, s3, p, k);
    }
    uint32_t o = pack_clip(s0, s1, s2, s3);
    if (J.flags & kRsUnpremultiplyOnStore) o = unpremultiply(o);
    reinterpret_cast<gptr>(J.dst)[(size_t)y * J.out_w + xx] = o;
}

// Vertical pass: one thread per output pixel (x, y'); taps walk down a column, lanes are
// adjacent columns (coalesced).  Coefficients [out_h][ksize] are wave-uniform per row.
__global__ __launch_bounds__(256) void resample_v_kernel(const RsJob *__restrict__ jobs) {
    const RsJob J = jobs[blockIdx.z];
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yy = blockIdx.y;
    if (yy >= J.out_h || x >= J.out_w) return;
    gciptr bounds = reinterpret_cast<gciptr>(J.bounds);
    gciptr k = reinterpret_cast<gciptr>(J.coeffs) + (size_t)yy * J.ksize;
    const int first = bounds[2 * yy], n = bounds[2 * yy + 1];
    gcptr col = reinterpret_cast<gcptr>(J.src) + (size_t)first * J.in_w + x;
    const bool pre = (J.flags & kRsPremultiplyOnLoad) != 0;
    int32_t s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    for (int i = 0; i < n; ++i) {
        uint32_t p = col[(size_t)i * J.in_w];
        if (pre) p = premultiply(p);
        const int32_t c = k[i];
        mac4(s0, s1, s2, s3, p, c);
    }
    uint32_t o = pack_clip(s0, s1, s2, s3);
    if (J.flags & kRsUnpremultiplyOnStore) o = unpremultiply(o);
    reinterpret_cast<gptr>(J.dst)[(size_t)yy * J.out_w + x] = o;
}

// Fused resize on the matrix cores, marching down column strips.
//
// A separable resample is a banded matrix product per axis -- out = in x K^T with K[x][k] the taps
// of output sample x -- and it is exact integer arithmetic, so it maps onto v_mfma_i32_16x16x64_i8
// without touching the result: the 8-bit samples are stored as signed bytes (s - 128, the constant
// 128 * sum(taps) goes into the accumulator's initial value together with Pillow's 2^21 rounding
// term) and each 22-bit tap is split into three signed-byte digits, c = d0 + 256 d1 + 65536 d2.
// The three digit products are not summed afterwards; they are CHAINED through the accumulator:
//     acc = mfma(data, d0, bias);  acc >>= 8;  acc = mfma(data, d1, acc);  acc >>= 8;
//     acc = mfma(data, d2, acc);   out = sat8(acc >> 6)
// which is Pillow's clip8((bias + sum) >> 22) exactly, because floor((x + floor(y / n)) / m) ==
// floor((x + y / n) / m) for integers x, y and positive n, m (arithmetic shifts are floor divisions).
// Two plain shifts per value replace the shift-adds of a digit recombination, and a channel needs one
// 4-register accumulator instead of three -- registers are what decides this kernel: it is bound by
// vector issue (profiles/r02_ubench_isa.txt: one wave alone issues a VALU instruction every ~8 cycles,
// four waves per SIMD are needed to approach the 2-cycle rate, the matrix pipe takes ~10), so the
// design goal is many resident waves, few instructions per value.
//
// Work unit = one workgroup (4 waves) = a column strip of 64 output columns x `seg_tiles` tiles of 16
// output rows of one layer; wave w owns the strip's x-tile w for both passes.  The unit marches down
// the source in bands of 16 rows:
//   1. the band's rows x the strip's column window come from the cutout's planar premultiplied copy
//      (planarize_kernel) into LDS, 16 bytes per lane, prefetched one band ahead in registers;
//   2. horizontal pass (wave w: 16 rows x 16 outputs of x-tile w, taps resident in registers) -> clip ->
//      the 8-bit intermediate Pillow keeps between its passes, into the wave's PRIVATE ring of
//      intermediate rows ([channel][x][ring row], 16-row slots), so the only workgroup barriers are the
//      two around the shared source band;
//   3. every tile of 16 output rows whose last tap row is now in the ring: vertical pass with the ring
//      as the A operand (M = x), so a lane ends up with 4 horizontally adjacent pixels of one output row
//      -> unpremultiply -> one 16-byte store.
// Neither the source rows nor the horizontal pass are redone for vertical neighbours inside a unit (the
// 64 x 64-tile version of round 1 re-read 2.1x the source and redid 25-40% of the horizontal pass).
// Bands whose window holds no pixel of alpha > 0 (the corners around a cutout's shape) skip the
// horizontal pass, and output tiles that only see such bands are stored as transparent black.
// The k index of both operands is defined by the same (lane >> 4, byte) -> window position map, so
// the result does not depend on the hardware's internal k order; C/D follow the documented
// col = lane & 15, row = 4 (lane >> 4) + reg map.  Bit-exact with the two-pass kernels above.
// ---- resident planar copy of an atlas ------------------------------------------------------------
// Premultiplying and planarising a cutout is a pure function of the cutout, and the atlas stays
// resident across composites / refine iterations / batches: the first resample that touches an atlas
// builds, once, a planar copy of every cutout -- four planes [row][column] of premultiplied samples
// stored as signed bytes (s - 128), row pitch = width rounded up to 16, padding = premultiplied zero --
// and phase 1 of the MFMA kernel becomes a 16-byte copy per lane instead of ~25 instructions per pixel
// (times the 1.7x tile halo).  mic_resize's arbitrary source pointers keep the interleaved loader.
__global__ __launch_bounds__(256) void planarize_kernel(const PlanarJob *__restrict__ jobs) {
    const PlanarJob J = jobs[blockIdx.y];
    const int groups = J.pitch >> 2;  // groups of 4 columns per row (pitch is a multiple of 16)
    const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (item >= (int64_t)groups * J.h) return;
    const int y = (int)(item / groups), x = 4 * (int)(item - (int64_t)y * groups);
    gcptr src = reinterpret_cast<gcptr>(J.src) + (size_t)y * J.w + x;
    uint32_t px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) px[j] = x + j < J.w ? src[j] : 0u;
    uint32_t rb[4], ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t a = px[j] >> 24;
        rb[j] = premultiply2_hi(px[j] & 0x00FF00FFu, a);
        ga[j] = premultiply2_hi(byte_perm(px[j], px[j], 0x0c0d0c01u), a);
    }
    const uint32_t rb01 = byte_perm(rb[1], rb[0], 0x07030501u), rb23 = byte_perm(rb[3], rb[2], 0x07030501u);
    const uint32_t ga01 = byte_perm(ga[1], ga[0], 0x07030501u), ga23 = byte_perm(ga[3], ga[2], 0x07030501u);
    MIC_GLOBAL uint32_t *dst = reinterpret_cast<MIC_GLOBAL uint32_t *>(J.dst + (size_t)y * J.pitch + x);
    const size_t plane = (size_t)J.h * J.pitch / 4;  // words
    dst[0 * plane] = byte_perm(rb23, rb01, 0x05040100u) ^ 0x80808080u;  // R
    dst[1 * plane] = byte_perm(ga23, ga01, 0x05040100u) ^ 0x80808080u;  // G
    dst[2 * plane] = byte_perm(rb23, rb01, 0x07060302u) ^ 0x80808080u;  // B
    dst[3 * plane] = byte_perm(ga23, ga01, 0x07060302u) ^ 0x80808080u;  // A
}

hipError_t launch_planarize(const PlanarJob *jobs_dev, int n_jobs, int64_t max_items, hipStream_t stream) {
    if (n_jobs <= 0 || max_items <= 0) return hipSuccess;
    for (int first = 0; first < n_jobs; first += 65535) {
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(planarize_kernel, dim3((unsigned)((max_items + 255) / 256), (unsigned)n), dim3(256), 0, stream,
                           jobs_dev + first);
    }
    return hipGetLastError();
}

// Every tile of both axes of a layer that comes here has its taps inside ONE 64-sample window (any scale down to
// ~1/3: the host checks it per layer; deeper shrinks, single images and small calls take the tile kernel of
// kernels_resample_tile.hip) -- no chunk loops, their registers or their branches.
__global__ __launch_bounds__(256, MIC_RS_WAVES) void resample_march_kernel(const RsMarch *__restrict__ jobs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds8[];
    __shared__ float recip[256];            // unpremultiply factors 255/a: an LDS read per pixel
    __shared__ v4i vm_lds[kRsMaxSegTiles];  // the unit's vertical tile table
    __shared__ uint32_t band_alpha;         // does the band in LDS hold a pixel of alpha > 0
    const RsMarch J = jobs[blockIdx.y];
    // XCD-aware unit order (see RsMarch): blockIdx.x & 7 is the XCD this workgroup lands on
    const int unit = (((int)blockIdx.x + J.xcd_rot) & 7) * (4 * J.n_entries) + 4 * J.entry + ((int)blockIdx.x >> 3);
    if (unit >= J.strips * J.segs) return;
    const int seg = unit / J.strips, strip = unit - seg * J.strips;
    const int xt0 = strip * 4, n_xt = min(4, J.tiles_x - xt0);
    const int yt0 = seg * J.seg_tiles, n_yt = min(J.seg_tiles, J.tiles_y - yt0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lh = lane >> 4;

    gv4ptr hmeta = reinterpret_cast<gv4ptr>(J.hmeta), vmeta = reinterpret_cast<gv4ptr>(J.vmeta);
    recip[tid] = (tid == 0 || tid == 255) ? 1.0f : unpremul_factor((uint32_t)tid);
    if (tid < n_yt) vm_lds[tid] = vmeta[yt0 + tid];
    const int c_lo = hmeta[xt0][0];  // the strip's first source column (a multiple of 16)
    // 16-byte column chunks of a band: what the strip's tiles can touch, inside the cutout's padded rows
    const int n16 = min(J.pitch_c >> 4, (J.planar_pitch - c_lo) >> 4);
    const int plane_s = 16 * J.pitch_c;
    uint8_t *srcP = lds8;                 // [4][16][pitch_c]     the source band
    // intermediate rows: [x 64][ring16 slots of 16 rows][channel 4][16 rows] -- the four channels of a slot lie 16
    // bytes apart, so one address register serves all four (instruction offsets); a column is pitch_r =
    // 64 ring16 + 16 bytes (the 16 spread the columns over the banks)
    uint8_t *ring = lds8 + 4 * plane_s;
    const int rmask = J.ring16 - 1;
    const int band0 = vmeta[yt0][0] >> 4;                  // window starts are multiples of 16
    const int band_last = (vmeta[yt0 + n_yt - 1][3] - 1) >> 4;

    // ---- this wave's x-tile: horizontal taps stay in registers for the whole unit
    const bool active = wave < n_xt;  // wave-uniform
    const int xt = xt0 + (active ? wave : 0);
    const v4i hm_v = hmeta[xt];  // the same for every lane: kept in scalar registers
    const int hm[3] = {__builtin_amdgcn_readfirstlane(hm_v[0]), __builtin_amdgcn_readfirstlane(hm_v[1]),
                       __builtin_amdgcn_readfirstlane(hm_v[2])};
    const uint32_t lane16 = (uint32_t)lane * 16u, l15x4 = (uint32_t)l15 * 4u;
    // (- 128 << 22: the horizontal pass clips to signed bytes, see clip8x4_signed)
    const int hb = *at<int32_t>(J.hbias + (uint64_t)xt * 64, l15x4) - (128 << 22);
    const v4i hbias = {hb, hb, hb, hb};
    // (uniform 64-bit base + 32-bit lane offset: the loads take the scalar-base addressing form, no 64-bit
    // vector address arithmetic)
    const uint64_t hfb = J.hfrag + (uint64_t)hm[2] * 3072;
    gv4ptr hfbase = at<v4i>(hfb, lane16);
    const v4i hf[3] = {hfbase[0], hfbase[64], hfbase[128]};
    const uint8_t *a0 = srcP + l15 * J.pitch_c + (hm[0] - c_lo) + 16 * lh;
    uint8_t *m0 = ring + (wave * 16 + l15) * J.pitch_r + 4 * lh;  // + 64 slot + 16 c
    const uint8_t *r0 = ring + (wave * 16 + l15) * J.pitch_r;     // + 64 slot + 16 c
    gptr dst = reinterpret_cast<gptr>(J.dst);
    const int ox = (xt0 + wave) * 16 + 4 * lh;
    const bool x_full = (xt0 + wave) * 16 + 16 <= J.dw;   // wave-uniform: every lane's 4 pixels are inside the row
    const uint32_t lane_idx = (uint32_t)(l15 * J.dw + ox);  // pixel index of this lane inside a tile of output rows
    const bool ox_ok = ox < J.dw;

    // ---- band loader: wave = plane, lane = (row, chunk mod 4); chunks cl, cl + 4, ... < n16
    const int lrow = lane >> 2, cl = lane & 3;
    const uint64_t gplane = J.src + (uint64_t)wave * ((uint64_t)J.planar_pitch * J.sh) + c_lo;  // uniform
    uint8_t *lds_dst = srcP + wave * plane_s + lrow * J.pitch_c + 16 * cl;
    v4i pre[2];
    // byte offset of this lane's first chunk of band b inside the plane (a plane is < 2^31 bytes)
    auto band_off = [&](int b) { return (uint32_t)min(16 * b + lrow, J.sh - 1) * (uint32_t)J.planar_pitch + 16u * cl; };
    auto prefetch = [&](int b) {
        const uint32_t o = band_off(b);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (cl + 4 * k < n16) pre[k] = *at<v4i>(gplane, o + 64u * k);
    };
    prefetch(band0);

    int yt = 0;           // next tile of output rows (relative to yt0) to emit ...
    int v_ws = 0, v_hi = 0;  // ... and its window (first source row, one past the last)
    // Its taps and bias wait in REGISTERS: they are fetched as soon as the tile before it has issued its last MFMA,
    // ahead of that tile's epilogue and stores -- a tile's tap loads sit most of a tile (often a whole band) ahead of
    // their use, and the in-order vmcnt wait for them does not cover the stores issued after them.  (One register
    // set, reloaded in place: two alternating sets made hipcc merge the two copies of the tile code again and copy
    // one set into the other behind a vmcnt(0).)
    v4i vf[3];
    int vb = 0;
    auto fetch_taps = [&](int frag, int row0) __attribute__((always_inline)) {
        vb = *at<int32_t>(J.vbias + (uint64_t)row0 * 4, l15x4);
        gv4ptr vfbase = at<v4i>(J.vfrag + (uint64_t)frag * 3072, lane16);
        vf[0] = vfbase[0]; vf[1] = vfbase[64]; vf[2] = vfbase[128];
    };
    {
        const v4i vm = vmeta[yt0];  // (the first entry straight from memory: vm_lds is not visible before the first barrier)
        v_ws = __builtin_amdgcn_readfirstlane(vm[0]); v_hi = __builtin_amdgcn_readfirstlane(vm[3]);
        fetch_taps(__builtin_amdgcn_readfirstlane(vm[2]), yt0 * 16);
    }
    uint32_t zmask = 0;   // bit s: ring slot s holds an all-zero band
    const uint32_t ring_bits = (uint32_t)((1ull << J.ring16) - 1ull);
    for (int b = band0; b <= band_last; ++b) {
        __syncthreads();  // every wave is done reading the previous band
        {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (cl + 4 * k < n16) *reinterpret_cast<v4i *>(lds_dst + 64 * k) = pre[k];
            uint32_t seen = 0;  // alpha plane (wave 3): OR of (alpha ^ 0x80) bytes, zero iff every alpha is 0
            const int k80 = (int)0x80808080u;
            if (wave == 3) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (cl + 4 * k < n16)
                        seen |= (uint32_t)((pre[k][0] ^ k80) | (pre[k][1] ^ k80) | (pre[k][2] ^ k80) | (pre[k][3] ^ k80));
            }
            if (n16 > 8) {  // wide strips (shrinks below ~1/1.3): the rest of the band, not prefetched
                const uint32_t o = band_off(b);
                for (int k = 2; cl + 4 * k < n16; ++k) {
                    const v4i v = *at<v4i>(gplane, o + 64u * k);
                    *reinterpret_cast<v4i *>(lds_dst + 64 * k) = v;
                    seen |= (uint32_t)((v[0] ^ k80) | (v[1] ^ k80) | (v[2] ^ k80) | (v[3] ^ k80));
                }
            }
            if (wave == 3) {
                const bool any = __any(seen != 0u);
                if (lane == 0) band_alpha = any ? 1u : 0u;
            }
        }
        __syncthreads();  // the band (and its alpha flag) is in LDS
        if (b < band_last) prefetch(b + 1);
        const bool zero_band = __builtin_amdgcn_readfirstlane((int)band_alpha) == 0;
        const int slot = b & rmask;
        if (active) {
            uint8_t *m = m0 + 64 * slot;
            if (zero_band) {
                // premultiplied zeros in, clip8(2^21 >> 22) = 0 out: the intermediate rows are zero
                zmask |= 1u << slot;
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<uint32_t *>(m + 16 * c) = 0x80808080u;
            } else {
                zmask &= ~(1u << slot);
                // D[row = 4 lh + reg (band row)][col = l15 (x)]: 4 consecutive rows of one column
                uint32_t w[4];
                auto load = [&](int c) { return *reinterpret_cast<const v4i *>(a0 + c * plane_s); };
                tile4<true>(load, hf, hbias, w);
#pragma unroll
                for (int c = 0; c < 4; ++c) *reinterpret_cast<uint32_t *>(m + 16 * c) = w[c];
            }
        }
        // ---- tiles of output rows whose last tap row is now in the ring (the next tile's table entry waits in
        // scalar registers: a band that completes no tile costs one compare)
        while (yt < n_yt && v_hi <= 16 * (b + 1)) {
            // the table entry of the tile after this one (the last tile re-reads its own: always a valid fetch)
            const int t_n = min(yt + 1, n_yt - 1);
            const v4i vm_n = vm_lds[t_n];
            const int n_ws = __builtin_amdgcn_readfirstlane(vm_n[0]), n_frag = __builtin_amdgcn_readfirstlane(vm_n[2]);
            const int n_hi = __builtin_amdgcn_readfirstlane(vm_n[3]);
            if (active) {
                const int row0 = (yt0 + yt) * 16;  // first output row of the tile (scalar)
                const bool inside = ox_ok && l15 < J.dh - row0;
                const uint32_t o_idx = (uint32_t)(row0 * J.dw) + lane_idx;  // < 2^30 px
                // every 16-row slot of the tile's window holds zeros? (window slots as a bit mask, rotated into the ring)
                const int s_first = (v_ws >> 4) & rmask, n_slots = ((v_hi - 1) >> 4) - (v_ws >> 4) + 1;
                // (64-bit: a ring of 32 slots makes these shifts reach 32 bits and beyond)
                const uint64_t span = ((1ull << n_slots) - 1ull) << s_first;
                const uint32_t need = (uint32_t)(span | (span >> J.ring16)) & ring_bits;
                const bool all_zero = (zmask & need) == need;
                u32x4 px = {0u, 0u, 0u, 0u};
                if (!all_zero) {
                    const int base16 = (v_ws >> 4) + lh;
                    uint32_t w[4];
                    // A[m = l15 (x)][k = 16 lh + j (window row)]; D[row = 4 lh + reg (x)][col = l15 (output row)]
                    auto load = [&](int c) { return *reinterpret_cast<const v4i *>(r0 + ((base16 & rmask) << 6) + 16 * c); };
                    tile4<false>(load, vf, v4i{vb, vb, vb, vb}, w);
                    fetch_taps(n_frag, (yt0 + t_n) * 16);
                    // alpha bytes all 0 or 255 <=> low 7 bits of every byte equal its top bit
                    const uint32_t top = (w[3] >> 7) & 0x01010101u;
                    const bool soft = (w[3] & 0x7F7F7F7Fu) != (top << 7) - top;
                    px = __any(soft) ? unpremultiply4(w, recip) : interleave4(w);
                } else {
                    fetch_taps(n_frag, (yt0 + t_n) * 16);
                }
                if (x_full) {  // (wave-uniform) one 16-byte store per lane
                    if (inside) *reinterpret_cast<MIC_GLOBAL u32x4_a4 *>(dst + o_idx) = px;
                } else if (inside) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ox + j < J.dw) dst[o_idx + j] = px[j];
                }
            }
            ++yt;
            v_ws = n_ws; v_hi = n_hi;
        }
    }
}

hipError_t launch_resample_march(const RsMarch *jobs_dev, int n_jobs, size_t lds_bytes, hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    // opt the kernel in for more than 64 KB of dynamic LDS, once per device of this process
    // (atomic flags: two threads racing here both set the same attribute, which is harmless)
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(resample_march_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRsMarchMaxLds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    for (int first = 0; first < n_jobs; first += 65535) {  // grid.y limit
        const int n = std::min(65535, n_jobs - first);
        hipLaunchKernelGGL(resample_march_kernel, dim3((unsigned)kRsUnitsPerEntry, (unsigned)n), dim3(256), lds_bytes, stream,
                           jobs_dev + first);
    }
    return hipGetLastError();
}

// ---- known-answer canary ------------------------------------------------------------------------------------------
// hipcc 7.2 miscompiled "shift, clamp to 0..255, pack" (it ORs v_ashr_pk_u8_i32's 16-bit result as if the destination's
// upper half were zero); clip8's empty asm and the explicit builtin + 16-bit truncation in clip8x4 work around it.
// Nothing but the parity tests would notice a ROCm update that changes the lowering again, so the library carries a
// tiny kernel that pushes known values through exactly these helpers -- every lane a different mix of negative,
// in-range, saturating and boundary sums -- and mic_selftest compares with plain host arithmetic.
__global__ void selftest_clip_kernel(uint32_t *__restrict__ out) {
    const int l = threadIdx.x;
    // four sums per lane around the interesting boundaries of clip8((bias + sum) >> 22) and sat8(v >> 6)
    const int32_t base[8] = {INT32_MIN, -(1 << 22) - 1, -1, 0, (1 << 22) - 1, 255 << 22, (256 << 22) + 5, INT32_MAX};
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)  // (unsigned arithmetic: the sums wrap by definition)
        v[j] = (int32_t)((uint32_t)base[(l + 3 * j) & 7] +
                         (uint32_t)((int32_t)((uint32_t)l * 0x9E3779B1u >> (j + 3)) * ((l + j) & 1 ? 1 : -1) / 7));
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));  // values the compiler cannot fold
    out[l * 4 + 0] = pack_clip(v[0], v[1], v[2], v[3]);
    const v4i q = {v[0] >> 16, v[1] >> 16, v[2] >> 16, v[3] >> 16};  // the MFMA epilogue's range: sat8(q >> 6)
    out[l * 4 + 1] = clip8x4(q);
    out[l * 4 + 2] = clip8x4_signed(q);
    out[l * 4 + 3] = premultiply((uint32_t)v[0]) ^ unpremultiply((uint32_t)v[1]);
}

static int32_t selftest_value(int l, int j) {
    const int32_t base[8] = {INT32_MIN, -(1 << 22) - 1, -1, 0, (1 << 22) - 1, 255 << 22, (256 << 22) + 5, INT32_MAX};
    return (int32_t)((uint32_t)base[(l + 3 * j) & 7] +
                     (uint32_t)((int32_t)((uint32_t)l * 0x9E3779B1u >> (j + 3)) * ((l + j) & 1 ? 1 : -1) / 7));
}

// Runs the canary on `stream`, waits for it, and returns the number of mismatching words (0 = the helpers compute
// what Pillow's arithmetic says); *first_bad (optional) names the first one.
hipError_t run_selftest_clip(hipStream_t stream, int *mismatches, int *first_bad) {
    uint32_t *dev = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dev), 64 * 4 * sizeof(uint32_t));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(selftest_clip_kernel, dim3(1), dim3(64), 0, stream, dev);
    uint32_t got[256];
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(got, dev, sizeof got, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return e;
    auto c8 = [](int64_t v) { return (uint32_t)std::min<int64_t>(255, std::max<int64_t>(0, v)); };
    auto c8s = [](int64_t v) { return (uint32_t)(std::min<int64_t>(127, std::max<int64_t>(-128, v)) & 255); };
    auto d255 = [](uint32_t t) { return ((t >> 8) + t) >> 8; };
    auto premul = [&](uint32_t p) {
        const uint32_t a = p >> 24;
        return d255((p & 255u) * a + 128u) | (d255(((p >> 8) & 255u) * a + 128u) << 8) | (d255(((p >> 16) & 255u) * a + 128u) << 16) | (a << 24);
    };
    auto unpremul = [](uint32_t p) {
        const uint32_t a = p >> 24;
        if (a == 0u || a == 255u) return p;
        auto ch = [&](uint32_t c) { return std::min<uint32_t>(255u, 255u * c / a); };
        return ch(p & 255u) | (ch((p >> 8) & 255u) << 8) | (ch((p >> 16) & 255u) << 16) | (a << 24);
    };
    int bad = 0, first = -1;
    for (int l = 0; l < 64; ++l) {
        int32_t v[4];
        for (int j = 0; j < 4; ++j) v[j] = selftest_value(l, j);
        uint32_t want[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            want[0] |= c8((int64_t)v[j] >> 22) << (8 * j);
            want[1] |= c8(((int64_t)v[j] >> 16) >> 6) << (8 * j);
            want[2] |= c8s(((int64_t)v[j] >> 16) >> 6) << (8 * j);
        }
        want[3] = premul((uint32_t)v[0]) ^ unpremul((uint32_t)v[1]);
        for (int k = 0; k < 4; ++k)
            if (got[l * 4 + k] != want[k]) {
                if (first < 0) first = l * 4 + k;
                ++bad;
            }
    }
    *mismatches = bad;
    if (first_bad) *first_bad = first;
    return hipSuccess;
}

hipError_t launch_resample_h(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_rows,
                             hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_out_w + 255) / 256), (unsigned)max_rows, (unsigned)n_jobs);
    hipLaunchKernelGGL(resample_h_kernel, grid, dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

hipError_t launch_resample_v(const RsJob *jobs_dev, int n_jobs, int max_out_w, int max_out_h,
                             hipStream_t stream) {
    if (n_jobs <= 0) return hipSuccess;
    dim3 grid((unsigned)((max_out_w + 255) / 256), (unsigned)max_out_h, (unsigned)n_jobs);
    hipLaunchKernelGGL(resample_v_kernel, grid, dim3(256), 0, stream, jobs_dev);
    return hipGetLastError();
}

}  // namespace mic
