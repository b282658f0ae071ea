// Host runtime + C ABI of libmic.so (see include/mic.h for the contract).
//
// What lives here: the context (pinned/params staging ring, scratch arena for resampled layers,
// device-resident coefficient tables), the atlas (packed cutouts), and the resolution of
// placements into device layer tables for the kernels.  No pixel arithmetic runs on the host.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <array>
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <new>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/mic.h"  // declares mic_plan, mic_ctx, mic_atlas
#include "mic_internal.h"
#include "lane_partition.h"
#include "flex_place.h"
#include "host_pool.h"
#include "png_decode.h"
#include "png_encode.h"
#include "resample_coeffs.h"

using namespace mic;

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? MIC_ERR_NOMEM : MIC_ERR_HIP, "%s: %s",   \
                        #expr, hipGetErrorString(e_));                                       \
    } while (0)

extern "C" const char *mic_last_error(void) { return g_err; }
extern "C" int mic_version(void) { return (1 << 16) | 9; }  // 1.9: + mic_plan_invalidate, mic_layer_cache_clear, mic_stats.cached_layers (resident resampled layers), mic_job.bg_rgba_dev, mic_render_job, mic_render_batch, mic_png_info / _decode(_rows, _many, _counts); 1.8: + mic_png_write_async / mic_png_wait; 1.7: + mic_median_rgb_batch, mic_host_rows_solid, mic_download(_wait); 1.6: + mic_png_* (1.5: + mic_stats.marched_layers; 1.4: + mic_contact_sheet(_size); thread-safe contexts)

// ------------------------------------------------------------------------------------ blob layout
namespace {

constexpr uint32_t kBlobMagic = 0x4143494du;  // "MICA"
constexpr uint32_t kBlobVersion = 2;  // v2: >= 16 guard bytes around every cutout
constexpr size_t kPixelAlign = 256;
// The composite kernel reads a layer row with one 16-byte load per lane even where the layer edge
// cuts through the lane's four pixels, i.e. up to 12 bytes before a cutout's first / after its last
// pixel.  Every image the kernels read (atlas cutouts, resampled layers) therefore sits between
// guard bands of at least kGuard readable bytes.
constexpr size_t kGuard = 16;
constexpr int64_t kMaxDim = 65535;
// The composite kernel addresses a layer with an unsigned 32-bit BYTE offset (pixel offset * 4 + the
// 16-byte guard bias, kernels_composite.hip: load_tap), so an image it reads holds at most this many
// pixels -- atlas cutouts (also in blobs received from elsewhere) and resampled layers alike.
constexpr int64_t kMaxLayerPx = ((int64_t)1 << 30) - 8;

struct BlobHeader {
    uint32_t magic, version, n, reserved;
    uint64_t total_bytes, pixels_offset;
};
struct BlobEntry {
    int32_t id, w, h, pad;
    uint64_t offset, reserved;
};
static_assert(sizeof(BlobHeader) == 32 && sizeof(BlobEntry) == 32, "blob layout");

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline size_t header_bytes(int n) { return sizeof(BlobHeader) + sizeof(BlobEntry) * (size_t)n; }

int blob_layout(int n, const int32_t *ids, const int32_t *w, const int32_t *h,
                std::vector<BlobEntry> *entries, size_t *total) {
    if (n < 0) return fail(MIC_ERR_INVALID, "atlas: negative object count");
    size_t off = align_up(header_bytes(n) + kGuard, kPixelAlign);
    const size_t pixels_offset = off;
    if (entries) entries->clear();
    for (int i = 0; i < n; ++i) {
        if (w[i] <= 0 || h[i] <= 0 || w[i] > kMaxDim || h[i] > kMaxDim || (int64_t)w[i] * h[i] > kMaxLayerPx)
            return fail(MIC_ERR_INVALID, "atlas: object %d has invalid size %dx%d", i, w[i], h[i]);
        if (entries) entries->push_back(BlobEntry{ids ? ids[i] : i, w[i], h[i], 0, off, 0});
        off = align_up(off + (size_t)w[i] * h[i] * 4 + kGuard, kPixelAlign);
    }
    (void)pixels_offset;
    *total = off;
    return MIC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ context
namespace {
constexpr int kSlots = 8;

struct Slot {
    void *host = nullptr;
    void *dev = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct CoefKey {
    int in, out, filter, transposed;
    bool operator<(const CoefKey &o) const {
        return std::tie(in, out, filter, transposed) < std::tie(o.in, o.out, o.filter, o.transposed);
    }
};
struct CoefEntry {
    int ksize = 0;
    int32_t *bounds = nullptr;  // device
    int32_t *coeffs = nullptr;  // device
};
// One axis in the MFMA kernel's fragment form (resample_coeffs.h AxisFrags), one device allocation.
// Device buffer of one axis' fragment tables.  Ref-counted: the context's cache holds one reference,
// every plan whose tables point into the buffer another, so the cache can drop old entries (a service
// that keeps meeting new sizes) without pulling memory from under a persistent plan.
// One device allocation that many axis tables are carved from (bump allocation, freed when its last table dies): a new
// box size used to cost a hipMalloc + three synchronous hipMemcpy per axis -- ~0.25 ms each, ~2 ms for the first LANCZOS
// call on a bundle -- now its tables are a slice of the current slab, filled by ONE asynchronous copy out of a pinned ring.
struct TableSlab {
    void *dev = nullptr;
    size_t cap = 0, used = 0;
    int device = 0;
    ~TableSlab() {
        if (!dev) return;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipFree(dev);
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
};
struct FragBuffer {
    void *dev = nullptr;
    size_t bytes = 0;
    int device = 0;  // the last reference may die on a thread whose current device is another one
    std::shared_ptr<TableSlab> slab;  // set: `dev` is a slice of this slab (kept alive, not freed here)
    ~FragBuffer() {
        if (!dev || slab) return;
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipFree(dev);  // waits for the device: in-flight kernels of a dead transient plan finish first
        if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    }
};
struct FragEntry {
    int tiles = 0;
    int max_chunks = 0;  // most 64-sample chunks any tile's window has
    std::shared_ptr<FragBuffer> buf;  // meta | bias | frags
    uint64_t meta = 0, bias = 0, frags = 0;
    std::shared_ptr<std::vector<int32_t>> meta_host;  // [tiles][4], for sizing the LDS windows
};
constexpr size_t kFragCacheBytes = (size_t)256 << 20;
}  // namespace

struct mic_ctx {
    // Every entry point that takes the context (or a plan / atlas of it) holds this lock for the whole
    // call: the staging ring, the arena, the table caches and last_stream are shared state.  Recursive
    // because mic_render calls mic_composite_batch.  (The Pillow calls this library replaces are
    // thread-safe, and the reference's Streamlit app runs every session on its own thread.)
    mutable std::recursive_mutex mu;
    int device = 0;
    Slot slots[kSlots];
    int next_slot = 0;
    void *arena = nullptr;
    size_t arena_cap = 0;
    std::map<CoefKey, CoefEntry> coefs;
    std::map<CoefKey, FragEntry> frags;  // key.transposed unused (0)
    std::deque<CoefKey> frag_order;      // insertion order, for eviction
    // one pinned buffer for the bulk upload of the axis tables a call builds ahead (prebuild_axis_frags)
    void *bulk_host = nullptr;
    size_t bulk_cap = 0;
    hipEvent_t bulk_ev = nullptr;
    bool bulk_pending = false;
    std::shared_ptr<TableSlab> table_slab;  // the slab new axis tables are carved from
    struct TableStage {                  // pinned ring the tables are uploaded through (asynchronously, on the call's stream)
        void *host = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool pending = false;
    };
    static constexpr int kTableStages = 4;
    TableStage table_stage[kTableStages];
    int next_table_stage = 0;
    size_t frag_bytes = 0;
    size_t frag_cache_cap = kFragCacheBytes;  // MIC_FRAG_CACHE_MB at mic_create (tests shrink it)
    // Work units a call's resampled layers must add up to before they take the marching kernel (two workgroups
    // per CU); MIC_RS_MARCH_MIN_UNITS at mic_create (tests set 0 to run every qualifying layer through it).
    int64_t march_min_units = 512;
    // The lane kernel (kernels_resample_lane.hip, round 5) takes the layers that qualify for it once the call's
    // resampling adds up to at least this many wave slots of work; MIC_RS_LANE=0 leaves them to the marching / tile
    // kernels (A/B runs, the marching kernel's own tests), MIC_RS_LANE_MIN_SLOTS moves the threshold (0: every
    // qualifying layer), MIC_RS_LANE_SLOTS caps the slots of a launch (default 4096 = 256 CUs x 4 SIMDs x 4 waves).
    bool lane_on = true;
    int lane_min_slots = 256, lane_max_slots = 1 << 20;  // (the reference-sized call, 4 cutouts of ~0.02 Mpx: 67 us of wall through the tile kernel, 80 through this one)
    double lane_chunk = 15000;  // MIC_RS_LANE_CHUNK: least cost (shader cycles of the model in lane_partition) of one slot's pieces
    double lane_split_slots = 4096, lane_split_chunk = 9000;  // MIC_RS_LANE_SPLIT=<slots>,<chunk>: calls below <slots> slots of work (one round) run one x-tile per piece, cut at <chunk> or more (0,0: never)
    uint32_t *median_scratch = nullptr;  // device: two sets of histogram slots (a double buffer) + kMedianMaxBatch result words
    MedianState median_state;            // which half the next call works in, what the previous one left to clear
    bool layer_args = true;              // MIC_LAYER_ARGS=0: single-canvas launches read their layer records from the device table
    bool tile_args = true;               // MIC_RS_TILE_ARGS=0: the tile kernel always reads its entries from the device table
    int64_t tile_small_px = 1 << 20;     // MIC_RS_TILE_SMALL_PX: calls that resize fewer output pixels than this take 32 x 32 tiles first (0: never)
    bool lane_keeps = true;              // MIC_RS_LANE_KEEPS=0: layers that keep one axis run the general lane kernel
    int median_two_launches = -1;        // MIC_MEDIAN_TWO_LAUNCHES=1 / 0: force the two-launch / one-launch median (-1: by image size)
    uint32_t *gradient_table = nullptr;  // device: fill_gradient's per-position colours (allocated on first use)
    uint32_t *median_host = nullptr;     // pinned
    hipStream_t last_stream = nullptr;
    // mic_download: a ring of events, one per copy in flight (ticket = ring index); an entry that comes round again
    // while nobody has waited for it is waited for by the enqueuer before it is re-recorded
    static constexpr int kDownloads = 16;
    hipEvent_t dl_event[kDownloads] = {};
    bool dl_pending[kDownloads] = {};
    uint32_t dl_gen[kDownloads] = {};  // how often the entry has been handed out: a ticket is (generation << 8) | entry
    int dl_next = 0;
    mic_stats stats{};
    uint64_t next_atlas_uid = 1;
    // ---- resampled layers stay resident (round 4): a transient call (mic_composite_batch, mic_render, mic_contact_sheet)
    // writes its resampled layers into this cache instead of the arena, keyed (atlas, cutout, box size, filter); a later
    // call that places the same cutout at the same size -- a refine iteration that moves a box without resizing it
    // (macro_placement_test.py:1679-1697), the contact sheet's thumbnails -- finds the pixels and skips the resample.
    // Bump-allocated regions, everything dropped at once when full (entries are re-made on demand); MIC_LAYER_CACHE_MB
    // (default 2048, 0 = off).  Persistent plans keep their resampled layers in their own scratch (mic_plan: resident).
    struct LayerKey {
        uint64_t atlas_uid;
        int32_t entry, w, h, filter;
        bool operator<(const LayerKey &o) const {
            return std::tie(atlas_uid, entry, w, h, filter) < std::tie(o.atlas_uid, o.entry, o.w, o.h, o.filter);
        }
    };
    struct LayerRegion {
        char *dev = nullptr;
        size_t cap = 0, used = 0;
    };
    std::map<LayerKey, uint64_t> layer_cache;  // -> device address of the resampled pixels
    std::vector<LayerRegion> layer_regions;
    size_t layer_cache_cap = (size_t)2048 << 20, layer_cache_total = 0;
    // optional event brackets around the kernels (mic_profile_begin/end)
    std::vector<hipEvent_t> prof_events;  // 3 per call: before resample, before composite, after
    int prof_calls = 0, prof_max = 0, prof_every = 1, prof_seen = 0;
    bool profiling = false;
};

struct mic_atlas {
    mic_ctx *ctx = nullptr;
    int device = 0;  // copy of ctx->device: mic_atlas_destroy must work after mic_destroy(ctx)
    void *blob = nullptr;
    size_t bytes = 0;
    bool owns = false;
    uint64_t uid = 0;
    std::vector<BlobEntry> entries;
    std::unordered_map<int32_t, int> index;
    // resident planar premultiplied copy of every cutout (built by the first resample that needs it)
    // Resident planar premultiplied copies of the cutouts that have been resampled (a cache: filled
    // through const atlases, cutout by cutout, by the first call that resamples it).  Ref-counted like
    // the fragment tables: a persistent plan whose pass tables point into the buffer keeps it alive past
    // mic_atlas_destroy.
    mutable std::shared_ptr<FragBuffer> planar;
    mutable std::vector<uint64_t> planar_off;   // per entry, bytes from `planar`
    mutable std::vector<int32_t> planar_pitch;
    mutable std::vector<uint8_t> planar_built;  // per entry
    // The same for the lane kernel: TILED planar copies (planes of 16 x 16 tiles, band by band; tiles per band =
    // ceil(w / 16) + 3, so that the 4-tile window that starts at any tile of a row stays inside its band).
    mutable std::shared_ptr<FragBuffer> tiled;
    mutable std::vector<uint64_t> tiled_off;
    mutable std::vector<int32_t> tiled_ct;      // tiles per band
    mutable std::vector<uint8_t> tiled_built;
};

// Lock the context for the rest of the calling function and make its device current.
#define CTX_ENTER(ctx)                                                   \
    if (!(ctx)) return fail(MIC_ERR_INVALID, "null context");            \
    std::lock_guard<std::recursive_mutex> ctx_lock_((ctx)->mu);          \
    HIP_TRY(hipSetDevice((ctx)->device))

extern "C" int mic_create(int device, mic_ctx **out) {
    if (!out) return fail(MIC_ERR_INVALID, "mic_create: null out");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(MIC_ERR_NODEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count)
        return fail(MIC_ERR_INVALID, "device %d out of range (have %d)", device, count);
    const bool trace = getenv("MIC_CREATE_TRACE") != nullptr;  // stage times of this call on stderr (bench.py: cold_start)
    auto t_prev = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mic_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    HIP_TRY(hipSetDevice(device));
    tick("hipGetDeviceCount+SetDevice");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    tick("hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MIC_ERR_NODEVICE, "device %d is %s; libmic is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    mic_ctx *ctx = new (std::nothrow) mic_ctx();
    if (!ctx) return fail(MIC_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    if (const char *mb = getenv("MIC_FRAG_CACHE_MB"))
        if (atoi(mb) > 0) ctx->frag_cache_cap = (size_t)atoi(mb) << 20;
    if (const char *mu = getenv("MIC_RS_MARCH_MIN_UNITS")) ctx->march_min_units = std::max<long long>(0, atoll(mu));
    if (const char *ln = getenv("MIC_RS_LANE")) ctx->lane_on = atoi(ln) != 0;
    if (const char *ln = getenv("MIC_RS_LANE_MIN_SLOTS")) ctx->lane_min_slots = std::max(0, atoi(ln));
    if (const char *ln = getenv("MIC_RS_LANE_SLOTS")) ctx->lane_max_slots = std::min(1 << 20, std::max(32, atoi(ln) / 32 * 32));
    if (const char *ln = getenv("MIC_RS_LANE_CHUNK")) ctx->lane_chunk = std::max(5000.0, atof(ln));
    if (const char *ls = getenv("MIC_RS_LANE_SPLIT")) {
        double a = 0, b = 0;
        if (sscanf(ls, "%lf,%lf", &a, &b) == 2) {
            ctx->lane_split_slots = std::max(0.0, a);
            ctx->lane_split_chunk = std::max(5000.0, b);
        }
    }
    for (auto &s : ctx->slots) {
        e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            mic_destroy(ctx);
            return fail(MIC_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    tick("8 events");
    if (const char *la = getenv("MIC_LAYER_ARGS")) ctx->layer_args = atoi(la) != 0;
    if (const char *ta = getenv("MIC_RS_TILE_ARGS")) ctx->tile_args = atoi(ta) != 0;
    if (const char *sp = getenv("MIC_RS_TILE_SMALL_PX")) ctx->tile_small_px = std::max(0ll, atoll(sp));
    if (const char *lk = getenv("MIC_RS_LANE_KEEPS")) ctx->lane_keeps = atoi(lk) != 0;
    if (const char *v = getenv("MIC_LAYER_CACHE_MB")) ctx->layer_cache_cap = (size_t)std::max(0ll, atoll(v)) << 20;
    if (const char *tl = getenv("MIC_MEDIAN_TWO_LAUNCHES")) ctx->median_two_launches = atoi(tl) != 0 ? 1 : 0;
    e = hipMalloc(&ctx->median_scratch, (kMedianScratchWords + 64) * sizeof(uint32_t));
    tick("hipMalloc median scratch");
    // zeroed once: every median call clears the half of the double buffer the call before it used
    // (with the library's own fill kernel, on the null stream, not waited for: the first hipMemset of a process loads the
    // runtime's blit kernels -- 85 - 134 ms measured here, most of what a fresh process waited for in mic_create.  Every
    // entry point orders its stream behind ctx->last_stream (adopt_stream), which starts as the null stream.)
    if (e == hipSuccess) e = launch_fill(ctx->median_scratch, 0u, kMedianScratchWords + 64, nullptr);
    tick("zero the median scratch");
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ctx->median_host), 64, 0);
    tick("hipHostMalloc 64 B");
    if (e != hipSuccess) {
        mic_destroy(ctx);
        return fail(MIC_ERR_HIP, "context allocation: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return MIC_OK;
}

extern "C" int mic_destroy(mic_ctx *ctx) {
    if (!ctx) return MIC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &s : ctx->slots) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.dev) (void)hipFree(s.dev);
        if (s.ev) (void)hipEventDestroy(s.ev);
    }
    for (auto &kv : ctx->coefs) {
        if (kv.second.bounds) (void)hipFree(kv.second.bounds);
        if (kv.second.coeffs) (void)hipFree(kv.second.coeffs);
    }
    ctx->frags.clear();  // buffers free themselves (plans still alive keep theirs)
    for (hipEvent_t ev : ctx->prof_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->dl_event)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &t : ctx->table_stage) {
        if (t.host) (void)hipHostFree(t.host);
        if (t.ev) (void)hipEventDestroy(t.ev);
    }
    ctx->table_slab.reset();
    if (ctx->bulk_host) (void)hipHostFree(ctx->bulk_host);
    if (ctx->bulk_ev) (void)hipEventDestroy(ctx->bulk_ev);
    for (auto &r : ctx->layer_regions)
        if (r.dev) (void)hipFree(r.dev);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->median_scratch) (void)hipFree(ctx->median_scratch);
    if (ctx->gradient_table) (void)hipFree(ctx->gradient_table);
    if (ctx->median_host) (void)hipHostFree(ctx->median_host);
    delete ctx;
    return MIC_OK;
}

extern "C" int mic_sync(mic_ctx *ctx, void *stream) {
    if (!ctx) return fail(MIC_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(ctx->device));  // no context state is touched: other threads are not kept waiting
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return MIC_OK;
}

// A context is driven from one stream at a time; if the caller switches streams, order the new
// one after the old one's work (the arena and the staging ring are shared).
static int adopt_stream(mic_ctx *ctx, hipStream_t stream) {
    if (ctx->last_stream != stream) {
        HIP_TRY(hipStreamSynchronize(ctx->last_stream));
        ctx->last_stream = stream;
    }
    return MIC_OK;
}

static int acquire_slot(mic_ctx *ctx, size_t bytes, Slot **out) {
    Slot &s = ctx->slots[ctx->next_slot];
    ctx->next_slot = (ctx->next_slot + 1) % kSlots;
    if (s.pending) {  // the copy out of this slot's pinned buffer must have finished
        HIP_TRY(hipEventSynchronize(s.ev));
        s.pending = false;
    }
    if (s.cap < bytes) {
        // The device half may still be read by an earlier launch of the same stream.
        HIP_TRY(hipStreamSynchronize(ctx->last_stream));
        if (s.host) HIP_TRY(hipHostFree(s.host));
        if (s.dev) HIP_TRY(hipFree(s.dev));
        s.host = s.dev = nullptr;
        s.cap = 0;
        const size_t cap = align_up(std::max(bytes, (size_t)64 << 10), 4096);
        HIP_TRY(hipHostMalloc(&s.host, cap, 0));
        HIP_TRY(hipMalloc(&s.dev, cap));
        s.cap = cap;
    }
    *out = &s;
    return MIC_OK;
}

static int ensure_arena(mic_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->arena_cap) return MIC_OK;
    HIP_TRY(hipStreamSynchronize(ctx->last_stream));
    if (ctx->arena) HIP_TRY(hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_cap = 0;
    const size_t cap = align_up(std::max(bytes + bytes / 4, (size_t)16 << 20), (size_t)1 << 20);
    HIP_TRY(hipMalloc(&ctx->arena, cap));
    ctx->arena_cap = cap;
    return MIC_OK;
}

static int get_coefs(mic_ctx *ctx, int in, int out, int filter, bool transposed, CoefEntry *res) {
    const CoefKey key{in, out, filter, transposed ? 1 : 0};
    auto it = ctx->coefs.find(key);
    if (it != ctx->coefs.end()) {
        *res = it->second;
        return MIC_OK;
    }
    AxisTable t = build_axis_table(in, out, filter);
    CoefEntry e;
    e.ksize = t.ksize;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e.bounds), t.bounds.size() * sizeof(int32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e.coeffs), t.coeffs.size() * sizeof(int32_t)));
    // Pageable source: the runtime stages it before returning, the vectors may die afterwards.
    HIP_TRY(hipMemcpy(e.bounds, t.bounds.data(), t.bounds.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (transposed) {
        std::vector<int32_t> tr = transpose_coeffs(t);
        HIP_TRY(hipMemcpy(e.coeffs, tr.data(), tr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemcpy(e.coeffs, t.coeffs.data(), t.coeffs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    ctx->coefs[key] = e;
    *res = e;
    return MIC_OK;
}

// in == out: the identity table (Pillow skips that pass; one tap of weight 1.0 gives the same bytes).
// A table met for the first time is carved from the context's current slab and uploaded by one asynchronous copy on
// `stream` (the stream of the call that needs it: its kernels follow on the same stream; a later call on another stream
// is ordered behind it by adopt_stream()).
static int get_frags(mic_ctx *ctx, int in, int out, int filter, hipStream_t stream, FragEntry *res, int form = kFragsTile) {
    const CoefKey key{in, out, in == out ? -1 : filter, form};  // (.transposed carries the fragment form: resample_coeffs.h)
    auto it = ctx->frags.find(key);
    if (it != ctx->frags.end()) {
        *res = it->second;
        return MIC_OK;
    }
    const AxisFrags f = build_axis_frags(in == out ? identity_axis_table(in) : build_axis_table(in, out, filter), form);
    FragEntry e;
    e.tiles = f.tiles;
    e.max_chunks = f.max_chunks;
    const size_t meta_b = align_up(f.meta.size() * sizeof(int32_t), 64);
    const size_t bias_b = align_up(f.bias.size() * sizeof(int32_t), 64);
    e.buf = std::make_shared<FragBuffer>();
    e.buf->device = ctx->device;
    e.buf->bytes = align_up(meta_b + bias_b + f.frags.size(), 256);
    // bounded cache, oldest first (entries a live plan still points into stay allocated until it dies)
    while (!ctx->frag_order.empty() && ctx->frag_bytes + e.buf->bytes > ctx->frag_cache_cap) {
        auto old = ctx->frags.find(ctx->frag_order.front());
        ctx->frag_order.pop_front();
        if (old != ctx->frags.end()) {
            ctx->frag_bytes -= old->second.buf->bytes;
            ctx->frags.erase(old);
        }
    }
    // a slice of the current slab (a new slab when it is full: 4 MiB, or the table's own size)
    if (!ctx->table_slab || ctx->table_slab->used + e.buf->bytes > ctx->table_slab->cap) {
        auto slab = std::make_shared<TableSlab>();
        slab->device = ctx->device;
        slab->cap = std::max(e.buf->bytes, (size_t)4 << 20);
        HIP_TRY(hipMalloc(&slab->dev, slab->cap));
        ctx->table_slab = std::move(slab);
    }
    e.buf->slab = ctx->table_slab;
    e.buf->dev = static_cast<char *>(ctx->table_slab->dev) + ctx->table_slab->used;
    ctx->table_slab->used += e.buf->bytes;
    e.meta = reinterpret_cast<uint64_t>(e.buf->dev);
    e.bias = e.meta + meta_b;
    e.frags = e.bias + bias_b;
    // one pinned stage of the ring, one asynchronous copy
    mic_ctx::TableStage &st = ctx->table_stage[ctx->next_table_stage];
    ctx->next_table_stage = (ctx->next_table_stage + 1) % mic_ctx::kTableStages;
    if (!st.ev) HIP_TRY(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    if (st.pending) {
        HIP_TRY(hipEventSynchronize(st.ev));
        st.pending = false;
    }
    if (st.cap < e.buf->bytes) {
        if (st.host) HIP_TRY(hipHostFree(st.host));
        st.host = nullptr;
        st.cap = 0;
        const size_t cap = align_up(std::max(e.buf->bytes, (size_t)512 << 10), 4096);
        HIP_TRY(hipHostMalloc(&st.host, cap, 0));
        st.cap = cap;
    }
    char *hp = static_cast<char *>(st.host);
    memcpy(hp, f.meta.data(), f.meta.size() * sizeof(int32_t));
    memcpy(hp + meta_b, f.bias.data(), f.bias.size() * sizeof(int32_t));
    memcpy(hp + meta_b + bias_b, f.frags.data(), f.frags.size());
    HIP_TRY(hipMemcpyAsync(e.buf->dev, st.host, meta_b + bias_b + f.frags.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(st.ev, stream));
    st.pending = true;
    e.meta_host = std::make_shared<std::vector<int32_t>>(f.meta);
    ctx->frags[key] = e;
    ctx->frag_order.push_back(key);
    ctx->frag_bytes += e.buf->bytes;
    *res = e;
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ atlas
extern "C" int mic_atlas_blob_size(int n, const int32_t *widths, const int32_t *heights, size_t *bytes) {
    if (!bytes || (n > 0 && (!widths || !heights))) return fail(MIC_ERR_INVALID, "mic_atlas_blob_size: null argument");
    return blob_layout(n, nullptr, widths, heights, nullptr, bytes);
}

extern "C" int mic_atlas_blob_layout(int n, const int32_t *ids, const int32_t *widths, const int32_t *heights,
                                     void *blob_host, size_t bytes, uint64_t *pixel_offsets) {
    if (!blob_host || (n > 0 && (!ids || !widths || !heights || !pixel_offsets)))
        return fail(MIC_ERR_INVALID, "mic_atlas_blob_layout: null argument");
    std::vector<BlobEntry> entries;
    size_t total = 0;
    if (int rc = blob_layout(n, ids, widths, heights, &entries, &total)) return rc;
    if (bytes < header_bytes(n)) return fail(MIC_ERR_INVALID, "mic_atlas_blob_layout: buffer too small");
    BlobHeader h{kBlobMagic, kBlobVersion, (uint32_t)n, 0, total, align_up(header_bytes(n) + kGuard, kPixelAlign)};
    memcpy(blob_host, &h, sizeof h);
    if (n) memcpy(static_cast<char *>(blob_host) + sizeof h, entries.data(), sizeof(BlobEntry) * n);
    for (int i = 0; i < n; ++i) pixel_offsets[i] = entries[i].offset;
    return MIC_OK;
}

static int atlas_finish(mic_atlas *a) {
    for (size_t i = 0; i < a->entries.size(); ++i) {
        // first occurrence wins for duplicate ids, like successive dict writes would not: the Python
        // binding never passes duplicates (dict keys), so this is only a defined behaviour for C callers
        a->index.emplace(a->entries[i].id, (int)i);
    }
    a->uid = a->ctx->next_atlas_uid++;
    a->device = a->ctx->device;
    return MIC_OK;
}

extern "C" int mic_atlas_create(mic_ctx *ctx, int n, const int32_t *ids, const int32_t *widths,
                                const int32_t *heights, const uint8_t *const *rgba_host, mic_atlas **out) {
    CTX_ENTER(ctx);
    if (!out || (n > 0 && (!ids || !widths || !heights || !rgba_host)))
        return fail(MIC_ERR_INVALID, "mic_atlas_create: null argument");
    *out = nullptr;
    std::vector<BlobEntry> entries;
    size_t total = 0;
    if (int rc = blob_layout(n, ids, widths, heights, &entries, &total)) return rc;
    std::vector<uint8_t> host(total, 0);
    std::vector<uint64_t> offs((size_t)std::max(n, 1));
    if (int rc = mic_atlas_blob_layout(n, ids, widths, heights, host.data(), total, offs.data())) return rc;
    for (int i = 0; i < n; ++i) {
        if (!rgba_host[i]) return fail(MIC_ERR_INVALID, "mic_atlas_create: object %d has no pixels", i);
        memcpy(host.data() + offs[i], rgba_host[i], (size_t)widths[i] * heights[i] * 4);
    }
    mic_atlas *a = new (std::nothrow) mic_atlas();
    if (!a) return fail(MIC_ERR_NOMEM, "out of host memory");
    a->ctx = ctx;
    a->bytes = total;
    a->owns = true;
    hipError_t e = hipMalloc(&a->blob, total);
    if (e == hipSuccess) e = hipMemcpy(a->blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (a->blob) (void)hipFree(a->blob);
        delete a;
        return fail(e == hipErrorOutOfMemory ? MIC_ERR_NOMEM : MIC_ERR_HIP, "atlas upload: %s", hipGetErrorString(e));
    }
    a->entries = std::move(entries);
    atlas_finish(a);
    *out = a;
    return MIC_OK;
}

extern "C" int mic_atlas_from_device_blob(mic_ctx *ctx, const void *blob_dev, size_t bytes,
                                          const void *header_host, mic_atlas **out) {
    CTX_ENTER(ctx);
    if (!blob_dev || !out) return fail(MIC_ERR_INVALID, "mic_atlas_from_device_blob: null argument");
    *out = nullptr;
    if (bytes < sizeof(BlobHeader)) return fail(MIC_ERR_FORMAT, "atlas blob shorter than its header");
    BlobHeader h;
    if (header_host) memcpy(&h, header_host, sizeof h);
    else HIP_TRY(hipMemcpy(&h, blob_dev, sizeof h, hipMemcpyDeviceToHost));
    if (h.magic != kBlobMagic || h.version != kBlobVersion)
        return fail(MIC_ERR_FORMAT, "atlas blob: bad magic/version %08x/%u", h.magic, h.version);
    if (h.total_bytes > bytes || header_bytes((int)h.n) > bytes)
        return fail(MIC_ERR_FORMAT, "atlas blob: declared %llu bytes, buffer has %zu",
                    (unsigned long long)h.total_bytes, bytes);
    std::vector<BlobEntry> entries(h.n);
    if (h.n) {
        if (header_host) memcpy(entries.data(), static_cast<const char *>(header_host) + sizeof h, sizeof(BlobEntry) * h.n);
        else HIP_TRY(hipMemcpy(entries.data(), static_cast<const char *>(blob_dev) + sizeof h,
                               sizeof(BlobEntry) * h.n, hipMemcpyDeviceToHost));
    }
    for (const BlobEntry &e : entries) {
        if (e.w <= 0 || e.h <= 0 || e.w > kMaxDim || e.h > kMaxDim || (int64_t)e.w * e.h > kMaxLayerPx ||
            e.offset % 4 != 0 || e.offset < kGuard ||
            e.offset + (uint64_t)e.w * e.h * 4 + kGuard > h.total_bytes)
            return fail(MIC_ERR_FORMAT, "atlas blob: entry for id %d is out of bounds", e.id);
    }
    mic_atlas *a = new (std::nothrow) mic_atlas();
    if (!a) return fail(MIC_ERR_NOMEM, "out of host memory");
    a->ctx = ctx;
    a->blob = const_cast<void *>(blob_dev);
    a->bytes = bytes;
    a->owns = false;
    a->entries = std::move(entries);
    atlas_finish(a);
    *out = a;
    return MIC_OK;
}

extern "C" int mic_atlas_device_blob(const mic_atlas *atlas, const void **blob_dev, size_t *bytes) {
    if (!atlas || !blob_dev || !bytes) return fail(MIC_ERR_INVALID, "mic_atlas_device_blob: null argument");
    *blob_dev = atlas->blob;
    *bytes = atlas->bytes;
    return MIC_OK;
}

extern "C" int mic_atlas_count(const mic_atlas *atlas) { return atlas ? (int)atlas->entries.size() : 0; }

extern "C" int mic_atlas_lookup(const mic_atlas *atlas, int32_t id, int32_t *width, int32_t *height,
                                const void **rgba_dev) {
    if (!atlas) return fail(MIC_ERR_INVALID, "mic_atlas_lookup: null atlas");
    auto it = atlas->index.find(id);
    if (it == atlas->index.end()) return fail(MIC_ERR_INVALID, "object id %d is not in the atlas", id);
    const BlobEntry &e = atlas->entries[it->second];
    if (width) *width = e.w;
    if (height) *height = e.h;
    if (rgba_dev) *rgba_dev = static_cast<const char *>(atlas->blob) + e.offset;
    return MIC_OK;
}

extern "C" int mic_atlas_destroy(mic_atlas *atlas) {
    if (!atlas) return MIC_OK;
    if (atlas->owns && atlas->blob) {
        (void)hipSetDevice(atlas->device);
        (void)hipDeviceSynchronize();
        (void)hipFree(atlas->blob);
    }
    delete atlas;  // the planar copy goes with its last reference (plans may still hold one)
    return MIC_OK;
}

// Planar premultiplied copies (kernels_resample.hip: planarize_kernel) of the cutouts in `need`, the form
// the marching resample kernel reads.  Premultiplying and planarising is a pure function of the cutout
// and the atlas outlives every composite / refine iteration / batch, so each cutout is converted once, by
// the first call that resamples it (+4 B/px of HBM for those cutouts; the buffer is sized for the whole
// atlas the first time).  Runs on the stream of the calling entry point (job table through the staging
// ring): the context is driven from one stream at a time and adopt_stream() orders a later stream behind
// this one, so every later launch sees the copies without a device-wide synchronisation here.
// Split in two: the buffer is allocated (and every cutout's address in it fixed) when a plan is BUILT, the conversion
// kernels are enqueued when a plan is first RUN, on that run's stream -- mic_plan_create has no stream of its own, and
// the stream the caller uploaded the blob on is the one it will run on (or has ordered before it).
static int atlas_planar_alloc(const mic_atlas *A) {
    mic_ctx *ctx = A->ctx;
    const size_t n = A->entries.size();
    if (!A->planar) {
        std::vector<uint64_t> off(n);
        std::vector<int32_t> pitches(n);
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            const BlobEntry &e = A->entries[i];
            const int pitch = (e.w + 15) / 16 * 16;
            off[i] = total;
            pitches[i] = pitch;
            total = align_up(total + (size_t)4 * e.h * pitch, 256);
        }
        auto buf = std::make_shared<FragBuffer>();
        buf->device = ctx->device;
        buf->bytes = total;
        HIP_TRY(hipMalloc(&buf->dev, total));
        A->planar_off = std::move(off);
        A->planar_pitch = std::move(pitches);
        A->planar_built.assign(n, 0);
        A->planar = std::move(buf);
    }
    return MIC_OK;
}

static int atlas_planar_build(const mic_atlas *A, const std::vector<int> &need, hipStream_t stream) {
    mic_ctx *ctx = A->ctx;
    if (int rc = atlas_planar_alloc(A)) return rc;
    std::vector<PlanarJob> jobs;
    std::vector<int> building;  // marked built only once their kernel has been enqueued
    int64_t max_items = 0;
    for (int i : need) {
        if (A->planar_built[(size_t)i] || std::find(building.begin(), building.end(), i) != building.end()) continue;
        building.push_back(i);
        const BlobEntry &e = A->entries[(size_t)i];
        PlanarJob j{};
        j.src = reinterpret_cast<uint64_t>(A->blob) + e.offset;
        j.dst = reinterpret_cast<uint64_t>(A->planar->dev) + A->planar_off[(size_t)i];
        j.w = e.w; j.h = e.h; j.pitch = A->planar_pitch[(size_t)i];
        jobs.push_back(j);
        max_items = std::max<int64_t>(max_items, (int64_t)(j.pitch / 4) * e.h);
    }
    if (jobs.empty()) return MIC_OK;
    Slot *slot = nullptr;
    if (int rc = acquire_slot(ctx, sizeof(PlanarJob) * jobs.size(), &slot)) return rc;
    memcpy(slot->host, jobs.data(), sizeof(PlanarJob) * jobs.size());
    HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, sizeof(PlanarJob) * jobs.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(slot->ev, stream));
    slot->pending = true;
    HIP_TRY(launch_planarize(static_cast<const PlanarJob *>(slot->dev), (int)jobs.size(), max_items, stream));
    for (int i : building) A->planar_built[(size_t)i] = 1;
    return MIC_OK;
}

// The TILED planar copies the lane kernel reads (kernels_resample_lane.hip: planarize_tiled_kernel): same life cycle as
// the row-major copies above -- addresses fixed when a plan is built, pixels converted by the first run that needs them.
static int atlas_tiled_alloc(const mic_atlas *A) {
    mic_ctx *ctx = A->ctx;
    const size_t n = A->entries.size();
    if (!A->tiled) {
        std::vector<uint64_t> off(n);
        std::vector<int32_t> cts(n);
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            const BlobEntry &e = A->entries[i];
            const int ct = (e.w + 15) / 16 + 3, bands = (e.h + 15) / 16;
            off[i] = total;
            cts[i] = ct;
            total = align_up(total + (size_t)4 * bands * ct * 256, 256);
        }
        auto buf = std::make_shared<FragBuffer>();
        buf->device = ctx->device;
        buf->bytes = total + 4096;  // (slack: nothing reads past a plane, the margin is for good measure)
        HIP_TRY(hipMalloc(&buf->dev, buf->bytes));
        A->tiled_off = std::move(off);
        A->tiled_ct = std::move(cts);
        A->tiled_built.assign(n, 0);
        A->tiled = std::move(buf);
    }
    return MIC_OK;
}

static int atlas_tiled_build(const mic_atlas *A, const std::vector<int> &need, hipStream_t stream) {
    mic_ctx *ctx = A->ctx;
    if (int rc = atlas_tiled_alloc(A)) return rc;
    std::vector<PlanarJob> jobs;
    std::vector<int> building;
    int64_t max_items = 0;
    for (int i : need) {
        if (A->tiled_built[(size_t)i] || std::find(building.begin(), building.end(), i) != building.end()) continue;
        building.push_back(i);
        const BlobEntry &e = A->entries[(size_t)i];
        PlanarJob j{};
        j.src = reinterpret_cast<uint64_t>(A->blob) + e.offset;
        j.dst = reinterpret_cast<uint64_t>(A->tiled->dev) + A->tiled_off[(size_t)i];
        j.w = e.w; j.h = e.h; j.pitch = 16 * A->tiled_ct[(size_t)i];
        jobs.push_back(j);
        max_items = std::max<int64_t>(max_items, (int64_t)(j.pitch / 16) * ((e.h + 15) / 16 * 16));  // rows of tiles
    }
    if (jobs.empty()) return MIC_OK;
    Slot *slot = nullptr;
    if (int rc = acquire_slot(ctx, sizeof(PlanarJob) * jobs.size(), &slot)) return rc;
    memcpy(slot->host, jobs.data(), sizeof(PlanarJob) * jobs.size());
    HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, sizeof(PlanarJob) * jobs.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(slot->ev, stream));
    slot->pending = true;
    HIP_TRY(launch_planarize_tiled(static_cast<const PlanarJob *>(slot->dev), (int)jobs.size(), max_items, stream));
    for (int i : building) A->tiled_built[(size_t)i] = 1;
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ resample planning
namespace {

// One layer that needs Image.resize.  Three device paths: the marching MFMA kernel (layers of atlases, one 64-sample
// chunk per tile, calls big enough to fill the chip), the tile MFMA kernel (single images, deep shrinks, small
// calls), the two-pass VALU kernels (windows no LDS holds).
struct ResizePlan {
    uint64_t src;
    int sw, sh, dw, dh;
    size_t tmp_off = 0;   // scratch offset of the horizontal pass output (two-pass fallback, both axes)
    size_t dst_off = 0;   // scratch offset of the final image (unused when dst_ptr is set)
    uint64_t dst_ptr = 0; // caller-provided destination (mic_resize)
    // marching kernel (source band + ring of intermediate rows in LDS)
    bool march_ok = false;   // the layer qualifies (one chunk per tile on both axes, ring and band fit LDS)
    bool march = false;      // ... and the call routes it there
    int pitch_c = 0, ring16 = 0, pitch_r = 0;
    // tile kernel (source planes + 8-bit intermediate in LDS); tx16 == 0: two-pass fallback
    int tx16 = 0, ty16 = 0, t_pitch_c = 0, t_pitch_r = 0, rows16 = 0;
    uint64_t planar_src = 0;  // the cutout in its atlas' planar premultiplied copy (0: none)
    int planar_pitch = 0;
    int atlas = -1, entry = -1;  // where the source is a cutout of an atlas
    bool cached = false;  // the pixels are in the context's resident layer cache already (dst_ptr): no pass is emitted
    // lane kernel (one wave per piece, tiled planar source)
    bool lane_ok = false;  // the layer qualifies (a cutout of an atlas; both axes have lane tables: max_chunks == 1)
    bool lane = false;     // ... and the call routes it there
    uint64_t tiled_src = 0;
    int tiled_ct = 0;
};

struct PassTables {
    std::vector<RsMarch> fused;   // marching kernel entries
    std::vector<RsTile> tiles;    // tile kernel entries: whole-window ones first, banded ones after
    int tiles_whole = 0;
    size_t tiles_lds = 0;
    std::vector<std::shared_ptr<FragBuffer>> frag_refs;  // keeps the tables the entries point into alive
    int fused_layers = 0;
    size_t lds_march = 0;
    std::vector<RsJob> h, v;
    int max_h_out_w = 0, max_h_rows = 0, max_v_out_w = 0, max_v_out_h = 0;
    // lane kernel: pieces, and per wave slot the [begin, end) of its pieces (slots: a multiple of 32)
    std::vector<RsLaneUnit> lane;  // records [0, lane_slots): first piece per slot; chained pieces behind
    int lane_slots = 0, lane_layers = 0;
};

// Over all groups of `per` consecutive 16-sample tiles along one axis: the largest window extent a
// group's tiles may touch (first tile's window start .. end of the last 64-sample chunk of any tile).
int window_touched(const std::vector<int32_t> &meta, int tiles, int per) {
    int touched = 0;
    for (int t0 = 0; t0 < tiles; t0 += per) {
        const int t1 = std::min(tiles, t0 + per);
        const int lo = meta[4 * t0];
        int end = lo;
        for (int t = t0; t < t1; ++t) end = std::max(end, meta[4 * t] + 64 * meta[4 * t + 1]);
        touched = std::max(touched, end - lo);
    }
    return touched;
}

int round16(int v) { return (v + 15) / 16 * 16; }

// Over all workgroup tiles of `per` 16-sample tiles along one axis: the largest span of samples a tile needs
// (its window start .. one past its last tap).
int window_needed(const std::vector<int32_t> &meta, int tiles, int per) {
    int needed = 0;
    for (int t0 = 0; t0 < tiles; t0 += per) {
        const int t1 = std::min(tiles, t0 + per);
        needed = std::max(needed, meta[4 * (t1 - 1) + 3] - meta[4 * t0]);
    }
    return needed;
}

// Pick the workgroup tile of the tile kernel for one layer: the biggest of a short list whose source planes +
// intermediate planes fit LDS, preferring sizes that let three workgroups share a CU.  Leaves tx16 == 0 when
// nothing fits (extreme shrinks: the two-pass kernels take those).
// small_call: the whole call resizes less than ~1 Mpx, i.e. has fewer tiles of 64 x 64 than the chip has CUs -- then tiles
// of 32 x 32 first: four times the workgroups, each with a quarter of the passes between its barriers (a small launch is
// as long as its longest workgroup: 4 layers of 100 - 250 px 10.4 - 11.3 -> 7.9 - 9.3 us, the reference-sized call 61.5 ->
// 58.9 us of wall; from 1 Mpx up the bigger tile's shared halo wins again.  profiles/r05_small_calls.txt table 6).
int choose_tile(mic_ctx *ctx, ResizePlan *p, int filter, hipStream_t stream, bool small_call = false) {
    FragEntry fh, fv;
    if (int rc = get_frags(ctx, p->sw, p->dw, filter, stream, &fh)) return rc;
    if (int rc = get_frags(ctx, p->sh, p->dh, filter, stream, &fv)) return rc;
    static const int kBig[][2] = {{4, 4}, {4, 2}, {2, 2}, {2, 1}, {1, 1}}, kSmall[][2] = {{2, 2}, {2, 1}, {1, 1}, {4, 2}, {4, 4}};
    const int (*kTiles)[2] = small_call ? kSmall : kBig;
    p->tx16 = 0;
    if ((int64_t)p->sw * p->sh < 4) return MIC_OK;  // the kernel's 16-byte loads need 4 pixels to clamp into
    if ((int64_t)p->sw * p->sh >= ((int64_t)1 << 30)) return MIC_OK;  // its 32-bit pixel index steps past the end
    // Whole window resident first (preferred LDS size, then anything that fits); only windows too tall for that
    // (deep shrinks) get source planes that hold one band of rows at a time.  Those have few output tiles, so the
    // banded candidates go from the smallest tile up (more workgroups), with enough row tiles per band to keep
    // the four waves busy.
    for (const bool banded : {false, true}) {
        for (const size_t cap : {kRsTilePreferredLds, kRsTileMaxLds}) {
            for (int ti = 0; ti < 5; ++ti) {
                const auto &t = banded ? kBig[4 - ti] : kTiles[ti];
                // Pitches cover what a tile needs, not what its 64-sample chunks touch: a read past the end of a
                // row lands in the next row (or in the slack after the last one) and meets zero tap digits.
                const int pitch_c = round16(window_needed(*fh.meta_host, fh.tiles, t[0]));
                const int pitch_r = round16(window_needed(*fv.meta_host, fv.tiles, t[1]));
                const int rows16 = banded ? std::min(pitch_r, std::max(16, 64 / t[0])) : pitch_r;
                if (banded && rows16 == pitch_r) continue;  // same as the unbanded candidate
                if (rs_tile_lds_bytes(rows16, pitch_c, t[0], pitch_r) <= cap) {
                    p->tx16 = t[0]; p->ty16 = t[1]; p->t_pitch_c = pitch_c; p->t_pitch_r = pitch_r; p->rows16 = rows16;
                    return MIC_OK;
                }
            }
        }
    }
    return MIC_OK;
}

// Does the layer qualify for the marching kernel, and with what LDS: the source band covers what the 4 x-tiles of
// a strip can touch, the ring holds the 16-row slots between the first and the last tap row of any tile of 16
// output rows (a tile is emitted as soon as its last band is in).
int choose_march(mic_ctx *ctx, ResizePlan *p, int filter, hipStream_t stream) {
    FragEntry fh, fv;
    if (int rc = get_frags(ctx, p->sw, p->dw, filter, stream, &fh)) return rc;
    if (int rc = get_frags(ctx, p->sh, p->dh, filter, stream, &fv)) return rc;
    p->march_ok = false;
    if (fh.max_chunks != 1 || fv.max_chunks != 1) return MIC_OK;  // the kernel has no chunk loops
    int pitch_c = round16(window_touched(*fh.meta_host, fh.tiles, 4));
    if ((pitch_c / 16) % 2 == 0) pitch_c += 16;  // an odd number of 16-byte units per row spreads the rows over the banks
    int slots = 1;
    const std::vector<int32_t> &vm = *fv.meta_host;
    for (int t = 0; t < fv.tiles; ++t) slots = std::max(slots, (vm[4 * t + 3] - 1) / 16 - vm[4 * t] / 16 + 1);
    int ring16 = 4;
    while (ring16 < slots) ring16 *= 2;
    const int pitch_r = 64 * ring16 + 16;  // per column: ring16 slots x 4 channels x 16 rows, + 16 to spread the banks
    if (rs_march_lds_bytes(pitch_c, pitch_r) > kRsMarchMaxLds) return MIC_OK;
    p->march_ok = true; p->pitch_c = pitch_c; p->ring16 = ring16; p->pitch_r = pitch_r;
    return MIC_OK;
}

// Does the layer qualify for the lane kernel: a cutout of an atlas (it reads the atlas' tiled planar copy) whose axes
// both have lane tables -- every group of x-tiles inside one 64-column window, every tile of 16 output rows inside
// four 16-row bands (any scale down to ~1/2.1; deeper shrinks stay with the tile kernel).
int choose_lane(mic_ctx *ctx, ResizePlan *p, int filter, hipStream_t stream) {
    p->lane_ok = false;
    if (!ctx->lane_on || p->atlas < 0) return MIC_OK;
    FragEntry fh, fv;
    if (int rc = get_frags(ctx, p->sw, p->dw, filter, stream, &fh, kFragsLaneH)) return rc;
    if (fh.max_chunks != 1) return MIC_OK;
    if (int rc = get_frags(ctx, p->sh, p->dh, filter, stream, &fv, kFragsLaneV)) return rc;
    p->lane_ok = fv.max_chunks == 1;
    return MIC_OK;
}

// Work units the marching kernel would cut a layer into (strips of 4 x-tiles x segments of seg tiles)
int march_units(const ResizePlan &p, int64_t unit_px, int *seg_tiles_out) {
    const int tiles_x = (p.dw + 15) / 16, tiles_y = (p.dh + 15) / 16;
    // segments of equal height: about unit_px / 64 rows per unit, counted on whichever side has more of them (a
    // unit's time goes with the source bands it marches through as much as with the output tiles it emits)
    const double rows_per_tile = 16.0 * std::max(1.0, (double)p.sh / p.dh);
    const int64_t want = std::max<int64_t>(1, (int64_t)((double)unit_px / 64.0 / rows_per_tile + 0.5));
    const int seg_cap = (int)std::min<int64_t>(want, kRsMaxSegTiles);
    int segs = (tiles_y + seg_cap - 1) / seg_cap;
    const int seg_tiles = (tiles_y + segs - 1) / segs;
    segs = (tiles_y + seg_tiles - 1) / seg_tiles;
    if (seg_tiles_out) *seg_tiles_out = seg_tiles;
    return ((tiles_x + 3) / 4) * segs;
}

// Pixels one work unit of the marching kernel should produce: enough units to fill every CU several
// workgroups deep, few enough that the rows a unit re-does at its top (the vertical taps' reach) stay a
// small share.  MIC_RS_UNIT_PX overrides (tuning).
int64_t march_unit_px(int64_t total_px) {
    static const int64_t forced = [] { const char *e = getenv("MIC_RS_UNIT_PX"); return e ? atoll(e) : 0ll; }();
    if (forced > 0) return forced;
    const int64_t target_units = 256 * 5 * 2;  // CUs x resident workgroups x two rounds
    // (upper end measured on 16-canvas calls, 230 Mpx: 8 K 44.8 us per canvas, 16 K 42.7, 32 K 43.7, 64 K 46.3)
    return std::max<int64_t>(64 * 48, std::min<int64_t>(total_px / target_units, 16 * 1024));
}

int plan_passes(mic_ctx *ctx, const std::vector<ResizePlan> &plans, int filter, void *scratch, hipStream_t stream, PassTables *pt) {
    const uint64_t arena = reinterpret_cast<uint64_t>(scratch);
    int64_t march_px = 0;
    for (const ResizePlan &p : plans)
        if (p.march) march_px += (int64_t)p.dw * p.dh;
    const int64_t unit_px = march_unit_px(march_px);
    std::vector<LaneStrip> strips;
    std::vector<std::shared_ptr<std::vector<int32_t>>> lane_meta;
    for (size_t pi = 0; pi < plans.size(); ++pi) {
        const ResizePlan &p = plans[pi];
        if (p.cached) continue;  // the pixels are in the resident layer cache: no pass
        const bool need_h = p.dw != p.sw, need_v = p.dh != p.sh;
        const uint64_t dst = p.dst_ptr ? p.dst_ptr : arena + p.dst_off;
        if (p.lane) {
            FragEntry fh, fv;
            if (int rc = get_frags(ctx, p.sw, p.dw, filter, stream, &fh, kFragsLaneH)) return rc;
            if (int rc = get_frags(ctx, p.sh, p.dh, filter, stream, &fv, kFragsLaneV)) return rc;
            pt->frag_refs.push_back(fh.buf);
            pt->frag_refs.push_back(fv.buf);
            lane_meta.push_back(fv.meta_host);  // (the strips point into it until the cut below)
            LaneStrip st{};
            st.sh = p.sh; st.dw = p.dw; st.dh = p.dh;
            st.tiled_ct = p.tiled_ct; st.tiled_src = p.tiled_src; st.dst = dst;
            st.hfrag = fh.frags; st.hbias = fh.bias;
            st.vfrag = fv.frags; st.vbias = fv.bias; st.vmeta = fv.meta;
            st.vm = fv.meta_host->data(); st.ty = fv.tiles;
            const std::vector<int32_t> &hm = *fh.meta_host;
            // a layer that keeps its width or its height: the pass Pillow skips (Resample.c need_horizontal / need_vertical)
            st.cls = !ctx->lane_keeps ? kLaneGeneral : !need_h ? kLaneKeepsWidth : !need_v ? kLaneKeepsHeight : kLaneGeneral;
            for (int t = 0; t < fh.tiles; ++t) {
                if (hm[4 * t + 1] == 0) continue;  // (the second tile of a group)
                st.t0 = t; st.T = hm[4 * t + 1]; st.ws = hm[4 * t];
                strips.push_back(st);
            }
            ++pt->lane_layers;
            continue;
        }
        if (p.march) {
            FragEntry fh, fv;
            if (int rc = get_frags(ctx, p.sw, p.dw, filter, stream, &fh)) return rc;
            if (int rc = get_frags(ctx, p.sh, p.dh, filter, stream, &fv)) return rc;
            pt->frag_refs.push_back(fh.buf);
            pt->frag_refs.push_back(fv.buf);
            RsMarch f{};
            f.src = p.planar_src; f.dst = dst;
            f.planar_pitch = p.planar_pitch;
            f.hmeta = fh.meta; f.hbias = fh.bias; f.hfrag = fh.frags;
            f.vmeta = fv.meta; f.vbias = fv.bias; f.vfrag = fv.frags;
            f.sw = p.sw; f.sh = p.sh; f.dw = p.dw; f.dh = p.dh;
            f.tiles_x = fh.tiles; f.tiles_y = fv.tiles;
            f.strips = (fh.tiles + 3) / 4;
            const int n_units = march_units(p, unit_px, &f.seg_tiles);
            f.segs = n_units / f.strips;
            f.pitch_c = p.pitch_c; f.ring16 = p.ring16; f.pitch_r = p.pitch_r;
            f.n_entries = (n_units + kRsUnitsPerEntry - 1) / kRsUnitsPerEntry;
            f.xcd_rot = pt->fused_layers++ & 7;  // the XCD that gets a layer's short last run rotates
            for (f.entry = 0; f.entry < f.n_entries; ++f.entry) pt->fused.push_back(f);
            pt->lds_march = std::max(pt->lds_march, rs_march_lds_bytes(f.pitch_c, f.pitch_r));
            continue;
        }
        if (p.tx16 > 0) {
            FragEntry fh, fv;
            if (int rc = get_frags(ctx, p.sw, p.dw, filter, stream, &fh)) return rc;
            if (int rc = get_frags(ctx, p.sh, p.dh, filter, stream, &fv)) return rc;
            pt->frag_refs.push_back(fh.buf);
            pt->frag_refs.push_back(fv.buf);
            RsTile f{};
            f.src = p.planar_src ? p.planar_src : p.src; f.dst = dst;
            f.planar_pitch = p.planar_src ? p.planar_pitch : 0;
            f.hmeta = fh.meta; f.hbias = fh.bias; f.hfrag = fh.frags;
            f.vmeta = fv.meta; f.vbias = fv.bias; f.vfrag = fv.frags;
            f.sw = p.sw; f.sh = p.sh; f.dw = p.dw; f.dh = p.dh;
            f.tx16 = p.tx16; f.ty16 = p.ty16;
            f.tiles_x = (fh.tiles + p.tx16 - 1) / p.tx16; f.tiles_y = (fv.tiles + p.ty16 - 1) / p.ty16;
            f.pitch_c = p.t_pitch_c; f.pitch_r = p.t_pitch_r; f.rows16 = p.rows16;
            const int n_tiles = f.tiles_x * f.tiles_y;
            f.n_entries = (n_tiles + kRsTilesPerEntry - 1) / kRsTilesPerEntry;
            f.xcd_rot = pt->fused_layers++ & 7;  // the XCD that gets a layer's short last band rotates
            for (f.entry = 0; f.entry < f.n_entries; ++f.entry) pt->tiles.push_back(f);
            pt->tiles_lds = std::max(pt->tiles_lds, rs_tile_lds_bytes(f.rows16, f.pitch_c, f.tx16, f.pitch_r));
            continue;
        }
        uint64_t v_src = p.src;
        uint32_t v_flags = kRsUnpremultiplyOnStore | kRsPremultiplyOnLoad;
        if (need_h) {
            CoefEntry ce;
            if (int rc = get_coefs(ctx, p.sw, p.dw, filter, true, &ce)) return rc;
            RsJob j{};
            j.src = p.src;
            j.dst = need_v ? arena + p.tmp_off : dst;
            j.bounds = reinterpret_cast<uint64_t>(ce.bounds);
            j.coeffs = reinterpret_cast<uint64_t>(ce.coeffs);
            j.in_w = p.sw; j.in_h = p.sh; j.out_w = p.dw; j.out_h = p.sh;
            j.ksize = ce.ksize;
            j.flags = kRsPremultiplyOnLoad | (need_v ? 0u : (uint32_t)kRsUnpremultiplyOnStore);
            pt->h.push_back(j);
            pt->max_h_out_w = std::max(pt->max_h_out_w, p.dw);
            pt->max_h_rows = std::max(pt->max_h_rows, p.sh);
            v_src = j.dst;
            v_flags = kRsUnpremultiplyOnStore;
        }
        if (need_v) {
            CoefEntry ce;
            if (int rc = get_coefs(ctx, p.sh, p.dh, filter, false, &ce)) return rc;
            RsJob j{};
            j.src = v_src;
            j.dst = dst;
            j.bounds = reinterpret_cast<uint64_t>(ce.bounds);
            j.coeffs = reinterpret_cast<uint64_t>(ce.coeffs);
            j.in_w = p.dw; j.in_h = p.sh; j.out_w = p.dw; j.out_h = p.dh;
            j.ksize = ce.ksize;
            j.flags = v_flags;
            pt->v.push_back(j);
            pt->max_v_out_w = std::max(pt->max_v_out_w, p.dw);
            pt->max_v_out_h = std::max(pt->max_v_out_h, p.dh);
        }
    }
    if (!strips.empty()) {
        double chunk = ctx->lane_chunk;
        // A call below one round of wave slots runs every x-tile as a piece of its own (T = 1: twice the waves, each with
        // half the arithmetic per band): a lone wave per SIMD runs the digit chains at half speed
        // (profiles/r05_lane_stage_probe.txt), and with so few waves the shared window loads are not what costs.  4 layers of
        // 250 - 500 px: 9.7 -> 7.6, 11.0 -> 9.7, 13.2 -> 11.9 us; level from ~1500 slots up (the dispatch ramp of twice the
        // waves), never worse below a round (profiles/r05_small_calls.txt).  The tables are the same: tile j of a group
        // uses the group's window.  Such pieces are cut at 9 K cycles -- longer for upscales, where one band of source rows
        // feeds several tiles of output rows and a cut inside that run repeats the band's horizontal pass (C5's x8
        // upscales: 19.2 -> 17.5 us at 40 K).
        double total = 0, bands = 0, rows = 0;
        for (const LaneStrip &st : strips) {
            total += lane_piece_cost(st.vm, st.T, 0, st.ty);
            bands += (st.sh + 15) / 16;
            rows += st.ty;
        }
        if (total < ctx->lane_split_slots * ctx->lane_chunk) {
            std::vector<LaneStrip> one;
            one.reserve(2 * strips.size());
            for (LaneStrip st : strips) {
                const int T = st.T;
                st.T = 1;
                for (int j = 0; j < T; ++j, ++st.t0) one.push_back(st);
            }
            strips.swap(one);
            chunk = std::min(40000.0, std::max(ctx->lane_split_chunk, 5000.0 * rows / std::max(1.0, bands)));
        }
        LaneCut cut;
        lane_partition(strips, chunk, ctx->lane_max_slots, &cut);
        pt->lane.swap(cut.records);
        pt->lane_slots = cut.slots;
    }
    // tile kernel: whole-window entries first, banded ones after (two instantiations, launch_resample_tile)
    auto whole = [](const RsTile &f) { return f.rows16 >= f.pitch_r; };
    pt->tiles_whole = (int)(std::stable_partition(pt->tiles.begin(), pt->tiles.end(), whole) - pt->tiles.begin());
    return MIC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------ composite
// A plan = the resolved form of a batch of jobs: device layer records, resample pass tables and
// scratch for resampled layers.  mic_composite_batch builds a transient one on the context's
// arena/staging ring; mic_plan_create builds a persistent one that owns its device memory, so
// that re-running it (mic_plan_run) only uploads the small job table (output pointers) and
// launches.  Pixel work is redone on every run -- nothing is cached but addresses.
struct mic_plan {
    mic_ctx *ctx = nullptr;
    int device = 0;  // copy of ctx->device (mic_plan_destroy does not touch the context)
    int filter = 0;
    bool persistent = false;
    std::vector<Job> jobs;     // caller order; out / px_shift / n_pages are set per run
    std::vector<Job> ordered;  // hot jobs first, rebuilt per run
    std::vector<Layer> layers;
    PassTables pt;
    // cutouts whose planar premultiplied copy the marching kernel reads: converted by the first run that finds them
    // unconverted, on that run's stream (the atlases outlive the runs, mic.h)
    std::vector<std::pair<const mic_atlas *, std::vector<int>>> planar_todo;
    std::vector<std::pair<const mic_atlas *, std::vector<int>>> tiled_todo;  // the same for the lane kernel's tiled copies
    void *scratch = nullptr;   // resampled layers (persistent plans own it; transient ones borrow ctx->arena)
    size_t scratch_bytes = 0;
    void *tables_dev = nullptr;  // persistent plans: jobs | layers | h passes | v passes
    size_t off_layers = 0, off_f = 0, off_t = 0, off_h = 0, off_v = 0, off_lane = 0, total = 0;
    mic_stats stats{};
    // Persistent plans: the job table only depends on the output pointers, so the device copies for
    // the last few sets of outputs are kept (callers rotate over a handful of output sets); a run onto
    // a known set uploads nothing -- it is one kernel launch.
    struct JobTable {
        std::vector<uint64_t> outs;  // key: the canvases of that run, caller order
        Job *dev = nullptr;
        int class_end[3] = {0, 0, 0};
        int pitch = 0;
        uint64_t last_use = 0;

    };
    static constexpr int kJobTables = 4;
    JobTable job_tables[kJobTables];
    uint64_t run_counter = 0;
    // Persistent plans: the resampled layers of the plan's placements live in its own scratch and are a pure function
    // of the (immutable) atlases, so a later run finds them there and only composites -- a refine iteration, a batch
    // re-rendered onto other canvases.  mic_plan_invalidate() makes the next run resample again (bench.py's cold legs).
    bool resampled_valid = false;
};

static void plan_offsets(mic_plan *P) {
    P->off_layers = align_up(sizeof(Job) * P->jobs.size(), 64);
    P->off_f = align_up(P->off_layers + sizeof(Layer) * P->layers.size(), 64);
    P->off_t = align_up(P->off_f + sizeof(RsMarch) * P->pt.fused.size(), 64);
    P->off_h = align_up(P->off_t + sizeof(RsTile) * P->pt.tiles.size(), 64);
    P->off_v = align_up(P->off_h + sizeof(RsJob) * P->pt.h.size(), 64);
    P->off_lane = align_up(P->off_v + sizeof(RsJob) * P->pt.v.size(), 128);
    P->total = align_up(P->off_lane + sizeof(RsLaneUnit) * P->pt.lane.size(), 64);
}

// Build the axis tables a call is about to need that the context has not seen -- on several host threads, straight
// into ONE pinned buffer, uploaded by ONE asynchronous copy into one slice of table memory -- and enter them into the
// cache.  A table is Pillow's precompute_coeffs in double precision: two libm sin() per tap, ~50 us per 1000-sample axis
// on the GPU box's host, and a composite() whose 32 boxes all have new sizes needs 64 of them: 6 ms of host time in front
// of 68 us of GPU work when built and uploaded one after another (scripts/time_new_sizes.py).  The tables are
// independent; each thread runs the very same scalar code on its share (bit-identical: no vector maths, no reordering
// inside a table).  Anything that goes wrong here is not an error: get_frags builds what is still missing.
static void prebuild_axis_frags(mic_ctx *ctx, const std::vector<CoefKey> &keys, hipStream_t stream) {
    const size_t n = keys.size();
    if (n < 4) return;  // (waking the pool is worth it for a few tables at least)
    std::vector<AxisTable> tables(n);
    std::vector<AxisFrags> lay(n);
    std::vector<size_t> chunks(n), off(n), meta_b(n), bias_b(n), bytes(n);
    HostPool &pool = HostPool::get();
    try {
        if (!pool.run((int)n, [&](int i) {  // phase 1: the coefficient tables (the sin() calls) and the fragment layouts
                const CoefKey &k = keys[(size_t)i];
                tables[(size_t)i] = k.filter == -1 ? identity_axis_table(k.in) : build_axis_table(k.in, k.out, k.filter);
                chunks[(size_t)i] = axis_frags_layout(tables[(size_t)i], &lay[(size_t)i], k.transposed);
            }))
            return;
        size_t total = 0;
        for (size_t i = 0; i < n; ++i) {
            meta_b[i] = align_up(lay[i].meta.size() * sizeof(int32_t), 64);
            bias_b[i] = align_up((size_t)lay[i].tiles * 16 * sizeof(int32_t), 64);
            bytes[i] = align_up(meta_b[i] + bias_b[i] + chunks[i] * 3072, 256);
            off[i] = total;
            total += bytes[i];
        }
        // the pinned buffer (grown geometrically; the previous bulk upload out of it must have landed)
        if (ctx->bulk_pending) {
            if (hipEventSynchronize(ctx->bulk_ev) != hipSuccess) return;
            ctx->bulk_pending = false;
        }
        if (ctx->bulk_cap < total) {
            if (ctx->bulk_host) (void)hipHostFree(ctx->bulk_host);
            ctx->bulk_host = nullptr;
            ctx->bulk_cap = 0;
            const size_t cap = align_up(std::max(total + total / 2, (size_t)4 << 20), 4096);
            if (hipHostMalloc(&ctx->bulk_host, cap, 0) != hipSuccess) { (void)hipGetLastError(); ctx->bulk_host = nullptr; return; }
            ctx->bulk_cap = cap;
        }
        if (!ctx->bulk_ev && hipEventCreateWithFlags(&ctx->bulk_ev, hipEventDisableTiming) != hipSuccess) return;
        // one slice of table memory for all of them (a slab of its own when the current one has no room)
        if (!ctx->table_slab || ctx->table_slab->used + total > ctx->table_slab->cap) {
            auto slab = std::make_shared<TableSlab>();
            slab->device = ctx->device;
            slab->cap = std::max(total, (size_t)4 << 20);
            if (hipMalloc(&slab->dev, slab->cap) != hipSuccess) { (void)hipGetLastError(); return; }
            ctx->table_slab = std::move(slab);
        }
        char *dev = static_cast<char *>(ctx->table_slab->dev) + ctx->table_slab->used;
        char *hp = static_cast<char *>(ctx->bulk_host);
        if (!pool.run((int)n, [&](int ii) {  // phase 2: meta | bias | fragments of every table, written in place
                const size_t i = (size_t)ii;
                char *at = hp + off[i];
                memcpy(at, lay[i].meta.data(), lay[i].meta.size() * sizeof(int32_t));
                fill_axis_frags(tables[i], lay[i], reinterpret_cast<int32_t *>(at + meta_b[i]),
                                reinterpret_cast<int8_t *>(at + meta_b[i] + bias_b[i]), chunks[i], keys[i].transposed);
            }))
            return;
        if (hipMemcpyAsync(dev, hp, total, hipMemcpyHostToDevice, stream) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipEventRecord(ctx->bulk_ev, stream) != hipSuccess) return;
        ctx->bulk_pending = true;
        ctx->table_slab->used += total;
        for (size_t i = 0; i < n; ++i) {
            FragEntry e;
            e.tiles = lay[i].tiles;
            e.max_chunks = lay[i].max_chunks;
            e.buf = std::make_shared<FragBuffer>();
            e.buf->device = ctx->device;
            e.buf->bytes = bytes[i];
            e.buf->slab = ctx->table_slab;
            e.buf->dev = dev + off[i];
            e.meta = reinterpret_cast<uint64_t>(e.buf->dev);
            e.bias = e.meta + meta_b[i];
            e.frags = e.bias + bias_b[i];
            e.meta_host = std::make_shared<std::vector<int32_t>>(std::move(lay[i].meta));
            while (!ctx->frag_order.empty() && ctx->frag_bytes + e.buf->bytes > ctx->frag_cache_cap) {
                auto old = ctx->frags.find(ctx->frag_order.front());
                ctx->frag_order.pop_front();
                if (old != ctx->frags.end()) {
                    ctx->frag_bytes -= old->second.buf->bytes;
                    ctx->frags.erase(old);
                }
            }
            ctx->frags[keys[i]] = e;
            ctx->frag_order.push_back(keys[i]);
            ctx->frag_bytes += e.buf->bytes;
        }
    } catch (const std::exception &) {  // (bad_alloc, system_error: the tables are then built one by one, as needed)
    }
}

// ---- resident layer cache (see mic_ctx) -----------------------------------------------------------------------------
static size_t layer_bytes(const ResizePlan &rp) { return align_up((size_t)rp.dw * rp.dh * 4 + kGuard, kPixelAlign); }

static void layer_cache_drop(mic_ctx *ctx) {  // forget every entry (the regions stay allocated and are refilled)
    ctx->layer_cache.clear();
    for (auto &r : ctx->layer_regions) r.used = kPixelAlign;  // (leading guard band)
}

static uint64_t layer_cache_alloc(mic_ctx *ctx, size_t bytes) {
    for (auto &r : ctx->layer_regions)
        if (r.used + bytes <= r.cap) {
            const uint64_t at = reinterpret_cast<uint64_t>(r.dev) + r.used;
            r.used += bytes;
            return at;
        }
    // a new region: big enough for this layer; otherwise 32 MiB for the first one (a bundle's thumbnails and a few
    // resized cutouts: the first LANCZOS call of a process should not wait for a 128 MiB allocation), doubling up to
    // 128 MiB; inside the cap
    size_t want = std::max(bytes + kPixelAlign, std::min<size_t>((size_t)128 << 20, std::max<size_t>((size_t)32 << 20, 2 * ctx->layer_cache_total)));
    if (ctx->layer_cache_total + want > ctx->layer_cache_cap) want = ctx->layer_cache_cap - std::min(ctx->layer_cache_cap, ctx->layer_cache_total);
    if (want < bytes + kPixelAlign) return 0;
    mic_ctx::LayerRegion r;
    if (hipMalloc(reinterpret_cast<void **>(&r.dev), want) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    r.cap = want;
    r.used = kPixelAlign + bytes;
    ctx->layer_cache_total += want;
    ctx->layer_regions.push_back(r);
    return reinterpret_cast<uint64_t>(r.dev) + kPixelAlign;
}

// Give every resampled layer of a transient call its place in the resident cache: a hit (same cutout, same box size,
// same filter, placed by an earlier call) needs no pass at all; a miss is resampled into the cache instead of the arena.
// false: the cache is off or cannot hold this call's layers at once -- the call uses the arena as before.
static bool layer_cache_assign(mic_ctx *ctx, mic_atlas *const *atlases, std::vector<ResizePlan> &plans, int filter,
                               uint64_t *hits) {
    if (ctx->layer_cache_cap == 0 || plans.empty()) return false;
    size_t total = kPixelAlign;
    for (const ResizePlan &rp : plans) total += layer_bytes(rp);
    if (total > ctx->layer_cache_cap) return false;
    auto key_of = [&](const ResizePlan &rp) {
        return mic_ctx::LayerKey{atlases[rp.atlas]->uid, rp.entry, rp.dw, rp.dh, filter};
    };
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool ok = true;
        *hits = 0;
        std::vector<std::pair<mic_ctx::LayerKey, uint64_t>> fresh;
        for (ResizePlan &rp : plans) {
            const auto key = key_of(rp);
            auto it = ctx->layer_cache.find(key);
            if (it != ctx->layer_cache.end()) {
                rp.dst_ptr = it->second;
                rp.cached = true;
                ++*hits;
                continue;
            }
            rp.cached = false;
            rp.dst_ptr = layer_cache_alloc(ctx, layer_bytes(rp));
            if (!rp.dst_ptr) {
                ok = false;
                break;
            }
            fresh.push_back({key, rp.dst_ptr});
        }
        if (ok) {
            for (const auto &kv : fresh) ctx->layer_cache[kv.first] = kv.second;
            return true;
        }
        // no room beside what is cached: drop everything and place the whole call again (its layers fit an empty
        // cache unless the regions are fragmented -- then the regions go too, and ONE region of the right size comes)
        layer_cache_drop(ctx);
        if (attempt == 0) {
            bool fits = false;
            for (const auto &r : ctx->layer_regions) fits |= r.cap >= total;
            if (!fits && ctx->layer_cache_total + total > ctx->layer_cache_cap) {
                if (hipStreamSynchronize(ctx->last_stream) != hipSuccess) break;
                for (auto &r : ctx->layer_regions) (void)hipFree(r.dev);
                ctx->layer_regions.clear();
                ctx->layer_cache_total = 0;
            }
        }
        for (ResizePlan &rp : plans) { rp.dst_ptr = 0; rp.cached = false; }
    }
    for (ResizePlan &rp : plans) { rp.dst_ptr = 0; rp.cached = false; }
    layer_cache_drop(ctx);
    return false;
}

static int plan_build(mic_ctx *ctx, int n_atlases, mic_atlas *const *atlases, int n_jobs, const mic_job *jobs,
                      int filter, bool persistent, hipStream_t stream, mic_plan *P) {
    if (n_jobs < 0 || n_atlases < 0 || (n_jobs > 0 && !jobs) || (n_atlases > 0 && !atlases))
        return fail(MIC_ERR_INVALID, "composite: bad arguments");
    if (filter != MIC_FILTER_LANCZOS && filter != MIC_FILTER_BILINEAR)
        return fail(MIC_ERR_INVALID, "unknown filter %d", filter);
    if (n_jobs > 65535) return fail(MIC_ERR_INVALID, "at most 65535 jobs per call");
    for (int a = 0; a < n_atlases; ++a)
        if (!atlases[a] || atlases[a]->ctx != ctx)
            return fail(MIC_ERR_INVALID, "atlas %d is null or belongs to another context", a);
    P->ctx = ctx;
    P->device = ctx->device;
    P->filter = filter;
    P->persistent = persistent;
    P->jobs.assign((size_t)n_jobs, Job{});
    mic_stats st{};

    std::vector<ResizePlan> plans;
    struct Pending { size_t layer; size_t plan; };
    std::vector<Pending> pending;  // layers whose src is a scratch offset, patched once scratch exists
    std::map<std::tuple<uint64_t, int, int, int>, size_t> dedup;  // (atlas uid, entry, w, h) -> plan
    size_t scratch_need = kPixelAlign;  // leading guard band
    // per atlas: (entry, plan) of the cutouts this call runs through the marching resample kernel
    std::vector<std::vector<std::pair<int, size_t>>> planar_need((size_t)std::max(n_atlases, 1));

    // Is this a call for the lane kernel?  Decided up front on the boxes alone (the same estimate the routing below
    // confirms with the tables in hand), because it says which FORM of the axis tables to build: the lane forms for such
    // a call, the tile forms otherwise -- never both for the same never-seen box size (a table is ~60 us of sin()).
    bool lane_call = false, small_call = false;
    {
        double cost = 0;
        int64_t out_px = 0;
        for (int ji = 0; ji < n_jobs; ++ji) {
            const mic_job &J = jobs[ji];
            if (J.n_placements <= 0 || !J.placements) continue;
            for (int pi = 0; pi < J.n_placements; ++pi) {
                const mic_placement &Pl = J.placements[pi];
                if (Pl.atlas < 0 || Pl.atlas >= n_atlases) continue;
                const mic_atlas *A = atlases[Pl.atlas];
                auto it = A->index.find(Pl.object_id);
                if (it == A->index.end()) continue;
                const BlobEntry &E = A->entries[it->second];
                const int64_t w = std::max<int64_t>(1, (int64_t)Pl.box[2] - Pl.box[0]), h = std::max<int64_t>(1, (int64_t)Pl.box[3] - Pl.box[1]);
                if ((w == E.w && h == E.h) || w > kMaxDim || h > kMaxDim) continue;
                cost += lane_layer_cost(E.h, (int)w, (int)h);  // (a box placed twice counts twice: an estimate)
                out_px += w * h;
            }
        }
        lane_call = ctx->lane_on && cost >= ctx->lane_chunk * std::max(1, ctx->lane_min_slots);
        small_call = out_px < ctx->tile_small_px;
    }
    {   // the axis tables of this call's resized boxes that the context has not seen yet, built ahead on several threads
        std::vector<CoefKey> need;
        std::map<CoefKey, int> listed;
        for (int ji = 0; ji < n_jobs; ++ji) {
            const mic_job &J = jobs[ji];
            if (J.n_placements <= 0 || !J.placements) continue;
            for (int pi = 0; pi < J.n_placements; ++pi) {
                const mic_placement &Pl = J.placements[pi];
                if (Pl.atlas < 0 || Pl.atlas >= n_atlases) continue;  // (reported by the loop below)
                const mic_atlas *A = atlases[Pl.atlas];
                auto it = A->index.find(Pl.object_id);
                if (it == A->index.end()) continue;
                const BlobEntry &E = A->entries[it->second];
                const int64_t w = std::max<int64_t>(1, (int64_t)Pl.box[2] - Pl.box[0]), h = std::max<int64_t>(1, (int64_t)Pl.box[3] - Pl.box[1]);
                if ((w == E.w && h == E.h) || w > kMaxDim || h > kMaxDim) continue;
                const CoefKey kx{E.w, (int)w, E.w == (int)w ? -1 : filter, lane_call ? (int)kFragsLaneH : (int)kFragsTile},
                    ky{E.h, (int)h, E.h == (int)h ? -1 : filter, lane_call ? (int)kFragsLaneV : (int)kFragsTile};
                for (const CoefKey &k : {kx, ky})
                    if (!ctx->frags.count(k) && listed.emplace(k, 1).second) need.push_back(k);
            }
        }
        prebuild_axis_frags(ctx, need, stream);
    }
    for (int ji = 0; ji < n_jobs; ++ji) {
        const mic_job &J = jobs[ji];
        if (J.width <= 0 || J.height <= 0 || J.width > kMaxDim || J.height > kMaxDim)
            return fail(MIC_ERR_INVALID, "job %d: invalid canvas size %dx%d", ji, J.width, J.height);
        if (J.n_placements < 0 || (J.n_placements > 0 && !J.placements))
            return fail(MIC_ERR_INVALID, "job %d: bad placement list", ji);
        Job d{};
        d.out = reinterpret_cast<uint64_t>(J.out_dev);
        d.bg = reinterpret_cast<uint64_t>(J.bg_dev);
        d.bg_rgba = (uint32_t)J.bg_rgba[0] | ((uint32_t)J.bg_rgba[1] << 8) | ((uint32_t)J.bg_rgba[2] << 16) |
                    ((uint32_t)J.bg_rgba[3] << 24);
        if (J.bg_rgba_dev) {  // the solid colour is a word in device memory (mic_median_rgb_dev's result)
            if (J.bg_dev) return fail(MIC_ERR_INVALID, "job %d: both a background image and a device colour word", ji);
            d.bg = reinterpret_cast<uint64_t>(J.bg_rgba_dev);
            d.bg_rgba = 0xff000000u;  // (never read by the kernels; the class below says opaque solid)
            d.flags = kJobColourWord;
        }
        d.W = J.width;
        d.H = J.height;
        d.layer_begin = (int32_t)P->layers.size();
        if (d.bg % 4 != 0) return fail(MIC_ERR_INVALID, "job %d: canvas pointers must be 4-byte aligned", ji);
        st.canvas_pixels += (uint64_t)J.width * J.height;

        for (int pi = 0; pi < J.n_placements; ++pi) {
            const mic_placement &Pl = J.placements[pi];
            if (Pl.atlas < 0 || Pl.atlas >= n_atlases)
                return fail(MIC_ERR_INVALID, "job %d placement %d: atlas index %d out of range", ji, pi, Pl.atlas);
            const mic_atlas *A = atlases[Pl.atlas];
            auto it = A->index.find(Pl.object_id);
            if (it == A->index.end()) {  // compositor.py:14-15
                ++st.skipped_placements;
                continue;
            }
            const BlobEntry &E = A->entries[it->second];
            const int64_t x1 = Pl.box[0], y1 = Pl.box[1];
            const int64_t w = std::max<int64_t>(1, (int64_t)Pl.box[2] - x1);
            const int64_t h = std::max<int64_t>(1, (int64_t)Pl.box[3] - y1);
            // in-canvas part of the layer; layers that miss the canvas have no effect at all
            const int64_t vx0 = std::max<int64_t>(x1, 0), vx1 = std::min<int64_t>(x1 + w, J.width);
            const int64_t vy0 = std::max<int64_t>(y1, 0), vy1 = std::min<int64_t>(y1 + h, J.height);
            if (vx0 >= vx1 || vy0 >= vy1) continue;
            if (w > kMaxDim || h > kMaxDim || w * h > kMaxLayerPx)
                return fail(MIC_ERR_INVALID, "job %d placement %d: box %lldx%lld is too large (side <= %lld, "
                            "area <= 2^30 - 8 pixels)", ji, pi, (long long)w, (long long)h, (long long)kMaxDim);
            Layer L{};
            L.dx = (int32_t)x1; L.dy = (int32_t)y1; L.w = (int32_t)w; L.h = (int32_t)h;
            st.layer_pixels += (uint64_t)(vx1 - vx0) * (vy1 - vy0);
            st.source_pixels += (uint64_t)E.w * E.h;
            if (w == E.w && h == E.h) {
                L.src = reinterpret_cast<uint64_t>(A->blob) + E.offset;
                ++st.identity_layers;
            } else {
                ++st.resampled_layers;
                auto key = std::make_tuple(A->uid, it->second, (int)w, (int)h);
                auto dit = dedup.find(key);
                size_t plan_idx;
                if (dit != dedup.end()) {
                    plan_idx = dit->second;
                } else {
                    ResizePlan rp{};
                    rp.src = reinterpret_cast<uint64_t>(A->blob) + E.offset;
                    rp.sw = E.w; rp.sh = E.h; rp.dw = (int)w; rp.dh = (int)h;
                    rp.atlas = Pl.atlas; rp.entry = it->second;
                    if (lane_call)
                        if (int rc = choose_lane(ctx, &rp, filter, stream)) return rc;
                    if (!rp.lane_ok) {  // (a layer the lane kernel does not take, or a small call: the tile forms)
                        if (int rc = choose_march(ctx, &rp, filter, stream)) return rc;
                        if (int rc = choose_tile(ctx, &rp, filter, stream, small_call)) return rc;
                    }
                    plan_idx = plans.size();
                    plans.push_back(rp);
                    dedup.emplace(key, plan_idx);
                }
                pending.push_back({P->layers.size(), plan_idx});
            }
            P->layers.push_back(L);
        }
        d.layer_count = (int32_t)P->layers.size() - d.layer_begin;
        P->jobs[(size_t)ji] = d;
    }
    // transient calls: resampled layers go to (or are found in) the context's resident cache
    if (!persistent) {
        uint64_t hits = 0;
        if (layer_cache_assign(ctx, atlases, plans, filter, &hits)) st.cached_layers = hits;
    }
    // scratch: the two-pass fallback's intermediates, and every layer that has no place of its own
    for (ResizePlan &rp : plans) {
        if (rp.cached) continue;
        if (rp.tx16 == 0 && rp.dw != rp.sw && rp.dh != rp.sh) {
            rp.tmp_off = scratch_need;
            scratch_need = align_up(scratch_need + (size_t)rp.dw * rp.sh * 4 + kGuard, kPixelAlign);
        }
        if (!rp.dst_ptr) {
            rp.dst_off = scratch_need;
            scratch_need = align_up(scratch_need + (size_t)rp.dw * rp.dh * 4 + kGuard, kPixelAlign);
        }
    }
    P->stats = st;

    if (scratch_need > ((size_t)64 << 30))
        return fail(MIC_ERR_NOMEM, "resampled layers of this call need %zu bytes of scratch", scratch_need);
    void *scratch = nullptr;
    if (plans.empty() || scratch_need <= kPixelAlign) {  // nothing to resample, or every layer has its place in the cache
        scratch_need = 0;
    } else if (persistent) {
        HIP_TRY(hipMalloc(&P->scratch, scratch_need));
        P->scratch_bytes = scratch_need;
        scratch = P->scratch;
    } else {
        if (int rc = ensure_arena(ctx, scratch_need)) return rc;
        scratch = ctx->arena;
    }
    for (const Pending &pd : pending)
        P->layers[pd.layer].src = plans[pd.plan].dst_ptr ? plans[pd.plan].dst_ptr
                                                          : reinterpret_cast<uint64_t>(scratch) + plans[pd.plan].dst_off;
    // Route the layers that qualify for the marching kernel: it wins once the call has enough work units to fill
    // the chip (its units are long chains of dependent bands; a few of them are a serial tail) and pays for the
    // cutouts' planar copies, which a persistent plan or a bundle's later calls amortise.  Everything else goes to
    // the tile kernel (reading a planar copy where the atlas already has one) or the two-pass fallback.
    std::vector<std::vector<std::pair<int, size_t>>> tiled_need((size_t)n_atlases);
    {   // the lane kernel first: every qualifying layer of a call that was sized for it above (what is left of such a
        // call once the resident layers are taken out goes the same way: its tables exist in the lane forms only)
        for (size_t i = 0; i < plans.size(); ++i) {
            ResizePlan &rp = plans[i];
            rp.lane = rp.lane_ok && lane_call && !rp.cached;
            if (!rp.lane) continue;
            rp.march_ok = false;  // (not a candidate for the marching kernel any more)
            ++P->stats.marched_layers;
            tiled_need[(size_t)rp.atlas].push_back({rp.entry, i});
        }
    }
    for (int a = 0; a < n_atlases; ++a) {
        const mic_atlas *A = atlases[a];
        if (tiled_need[(size_t)a].empty()) continue;
        std::vector<int> entries;
        for (const auto &ne : tiled_need[(size_t)a]) entries.push_back(ne.first);
        if (int rc = atlas_tiled_alloc(A)) return rc;
        P->tiled_todo.push_back({A, std::move(entries)});
        for (const auto &ne : tiled_need[(size_t)a]) {
            ResizePlan &rp = plans[ne.second];
            rp.tiled_src = reinterpret_cast<uint64_t>(A->tiled->dev) + A->tiled_off[(size_t)rp.entry];
            rp.tiled_ct = A->tiled_ct[(size_t)rp.entry];
        }
        P->pt.frag_refs.push_back(A->tiled);  // the pieces point into it
    }
    {
        int64_t px = 0;
        for (const ResizePlan &rp : plans)
            if (rp.march_ok && !rp.cached) px += (int64_t)rp.dw * rp.dh;
        const int64_t unit_px = march_unit_px(px);
        int64_t units = 0;
        for (const ResizePlan &rp : plans)
            if (rp.march_ok && !rp.cached) units += march_units(rp, unit_px, nullptr);
        const bool use_march = units >= ctx->march_min_units;
        for (size_t i = 0; i < plans.size(); ++i) {
            ResizePlan &rp = plans[i];
            rp.march = rp.march_ok && use_march && !rp.cached;
            if (rp.march) ++P->stats.marched_layers;
            if (rp.march) planar_need[(size_t)rp.atlas].push_back({rp.entry, i});
        }
    }
    for (int a = 0; a < n_atlases; ++a) {
        const mic_atlas *A = atlases[a];
        if (!planar_need[(size_t)a].empty()) {
            std::vector<int> entries;
            for (const auto &ne : planar_need[(size_t)a]) entries.push_back(ne.first);
            if (int rc = atlas_planar_alloc(A)) return rc;
            P->planar_todo.push_back({A, std::move(entries)});
        }
        if (!A->planar) continue;
        bool used = false;
        for (ResizePlan &rp : plans) {  // marching layers need the copy (converted by the first run, plan_submit);
            // tile-kernel layers take it where an earlier call has already converted it
            if (rp.atlas != a || !(rp.march || (rp.tx16 > 0 && A->planar_built[(size_t)rp.entry]))) continue;
            rp.planar_src = reinterpret_cast<uint64_t>(A->planar->dev) + A->planar_off[(size_t)rp.entry];
            rp.planar_pitch = A->planar_pitch[(size_t)rp.entry];
            used = true;
        }
        if (used) P->pt.frag_refs.push_back(A->planar);  // the pass tables point into it
    }
    if (int rc = plan_passes(ctx, plans, filter, scratch, stream, &P->pt)) return rc;
    plan_offsets(P);
    if (persistent && P->total > 0) {
        HIP_TRY(hipMalloc(&P->tables_dev, P->total));
        if (!P->layers.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_layers, P->layers.data(),
                              sizeof(Layer) * P->layers.size(), hipMemcpyHostToDevice));
        if (!P->pt.fused.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_f, P->pt.fused.data(),
                              sizeof(RsMarch) * P->pt.fused.size(), hipMemcpyHostToDevice));
        if (!P->pt.tiles.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_t, P->pt.tiles.data(),
                              sizeof(RsTile) * P->pt.tiles.size(), hipMemcpyHostToDevice));
        if (!P->pt.h.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_h, P->pt.h.data(),
                              sizeof(RsJob) * P->pt.h.size(), hipMemcpyHostToDevice));
        if (!P->pt.v.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_v, P->pt.v.data(),
                              sizeof(RsJob) * P->pt.v.size(), hipMemcpyHostToDevice));
        if (!P->pt.lane.empty())
            HIP_TRY(hipMemcpy(static_cast<char *>(P->tables_dev) + P->off_lane, P->pt.lane.data(),
                              sizeof(RsLaneUnit) * P->pt.lane.size(), hipMemcpyHostToDevice));
    }
    return MIC_OK;
}

static int plan_submit(mic_plan *P, void *const *outs, hipStream_t stream) {
    mic_ctx *ctx = P->ctx;
    const int n_jobs = (int)P->jobs.size();
    if (n_jobs == 0) {
        ctx->stats = P->stats;
        return MIC_OK;
    }
    if (int rc = adopt_stream(ctx, stream)) return rc;
    // ---- persistent plan onto a set of outputs it has seen: no table work, no upload ----
    // A single job travels in the kernel arguments (launch_composite): no device job table, no upload, no
    // table cache -- a persistent single-canvas plan's run is the launch alone.
    const bool one = n_jobs == 1;
    const bool pack_layers = one && ctx->layer_args && P->layers.size() <= (size_t)kPackLayers;
    mic_plan::JobTable *slot_tab = nullptr;
    if (P->persistent && !one) {
        ++P->run_counter;
        std::vector<uint64_t> key((size_t)n_jobs);
        for (int ji = 0; ji < n_jobs; ++ji)
            key[(size_t)ji] = outs ? reinterpret_cast<uint64_t>(outs[ji]) : P->jobs[(size_t)ji].out;
        mic_plan::JobTable *lru = &P->job_tables[0];
        for (auto &t : P->job_tables) {
            if (t.dev && t.outs == key) {
                slot_tab = &t;
                break;
            }
            if (t.last_use < lru->last_use) lru = &t;
        }
        if (!slot_tab) {
            if (!lru->dev) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&lru->dev), sizeof(Job) * (size_t)n_jobs));
            lru->outs.clear();  // filled in below once the table is valid
            slot_tab = lru;
            slot_tab->pitch = 0;
        }
        slot_tab->last_use = P->run_counter;
    }
    // A cached table was validated (null / alignment / overlap checks below) when this exact set of
    // output pointers was first seen; backgrounds and sizes are fixed at plan creation, so the same
    // pointers give the same verdict and the checks are not repeated.
    const bool cached = slot_tab && !slot_tab->outs.empty();

    int class_end[3] = {0, 0, 0};
    int pitch = 0;
    char *dp = nullptr;
    bool tiles_in_args = false;
    if (cached) {
        memcpy(class_end, slot_tab->class_end, sizeof class_end);
        pitch = slot_tab->pitch;
        dp = static_cast<char *>(P->tables_dev);
    } else {
        int max_pages = 0;
        P->ordered = P->jobs;  // this run's copy: the plan keeps the caller's own out_dev untouched
        for (int ji = 0; ji < n_jobs; ++ji) {
            Job &d = P->ordered[(size_t)ji];
            if (outs) d.out = reinterpret_cast<uint64_t>(outs[ji]);
            if (!d.out) return fail(MIC_ERR_INVALID, "job %d: null output canvas", ji);
            const uint64_t bytes = (uint64_t)d.W * d.H * 4;
            // compositor.py:11 copies the background: a canvas that overlaps it anywhere would be read
            // by one wave after another wave has written it
            const uint64_t bg_bytes = (d.flags & kJobColourWord) ? 4 : bytes;
            if (d.bg && d.out < d.bg + bg_bytes && d.bg < d.out + bytes)
                return fail(MIC_ERR_INVALID, "job %d: output overlaps the background", ji);
            if (d.out % 4 != 0) return fail(MIC_ERR_INVALID, "job %d: canvas pointers must be 4-byte aligned", ji);
            // 4 KiB pages aligned to absolute address: one workgroup per page (see mic_internal.h)
            d.px_shift = (int32_t)((d.out & 4095u) / 4);
            d.n_pages = (int32_t)(((uint64_t)d.W * d.H + d.px_shift + kPagePx - 1) / kPagePx);
            max_pages = std::max(max_pages, d.n_pages);
        }
        if (n_jobs > 1) {  // two canvases of one launch must not overlap (their pages are written concurrently)
            std::vector<std::pair<uint64_t, uint64_t>> spans((size_t)n_jobs);
            for (int ji = 0; ji < n_jobs; ++ji) {
                const Job &d = P->ordered[(size_t)ji];
                spans[(size_t)ji] = {d.out, d.out + (uint64_t)d.W * d.H * 4};
            }
            std::sort(spans.begin(), spans.end());
            for (int ji = 1; ji < n_jobs; ++ji)
                if (spans[(size_t)ji].first < spans[(size_t)ji - 1].second)
                    return fail(MIC_ERR_INVALID, "two output canvases of the batch overlap");
        }
        // (a multiple of 8 workgroups of kPagesPerWorkgroup pages: (linear workgroup id) mod 8 == (workgroup's index
        // inside its canvas) mod 8 for every job of the launch)
        pitch = (max_pages + 8 * kPagesPerWorkgroup - 1) / (8 * kPagesPerWorkgroup) * (8 * kPagesPerWorkgroup);
        // sort the job table by kernel class (see launch_composite): 0 = aligned + solid opaque
        // background (the pipeline's own canvases), 1 = unaligned + solid, 2 = aligned + other, 3 = rest
        auto job_class = [](const Job &d) {
            const bool aligned = d.W % 4 == 0 && d.out % 16 == 0;
            const bool solid = (d.bg == 0 && (d.bg_rgba >> 24) == 255u) || (d.flags & kJobColourWord);
            return (solid ? 0 : 2) + (aligned ? 0 : 1);
        };
        std::vector<uint64_t> key;
        if (slot_tab) {
            key.resize((size_t)n_jobs);
            for (int ji = 0; ji < n_jobs; ++ji) key[(size_t)ji] = P->ordered[(size_t)ji].out;
        }
        std::stable_sort(P->ordered.begin(), P->ordered.end(),
                         [&](const Job &a, const Job &b) { return job_class(a) < job_class(b); });
        for (const Job &d : P->ordered)
            for (int c = job_class(d); c < 3; ++c) ++class_end[c];

        // persistent plans upload the job table into the chosen cache slot; transient ones upload
        // everything into the staging slot's device buffer
        const size_t upload = P->persistent ? sizeof(Job) * P->ordered.size() : P->total;
        Slot *slot = nullptr;
        if (P->persistent && one) {
            dp = static_cast<char *>(P->tables_dev);
        } else if (pack_layers && P->pt.fused.empty() && P->pt.h.empty() && P->pt.v.empty() && P->pt.lane.empty() &&
                   (P->pt.tiles.empty() || (ctx->tile_args && rs_tile_in_args((int)P->pt.tiles.size(), P->pt.tiles_whole)))) {
            // one canvas, <= 64 layers, identity-scale (the reference's own call from the Flex pipeline,
            // compositor.py:6-22) or a handful resized by the tile kernel (its own LANCZOS call, compositor.py:18-21):
            // job, layer records AND tile entries ride in the kernel arguments -- nothing is staged, nothing uploaded
            tiles_in_args = !P->pt.tiles.empty();
            static char nothing[64];
            dp = nothing;  // (no table is read through it: every launcher below sees a count of 0)
        } else {
        if (int rc = acquire_slot(ctx, upload, &slot)) return rc;
        char *hp = static_cast<char *>(slot->host);
        memcpy(hp, P->ordered.data(), sizeof(Job) * P->ordered.size());
        void *upload_dst;
        if (P->persistent) {
            dp = static_cast<char *>(P->tables_dev);
            upload_dst = slot_tab->dev;
        } else {
            dp = static_cast<char *>(slot->dev);
            upload_dst = slot->dev;
            if (!P->layers.empty()) memcpy(hp + P->off_layers, P->layers.data(), sizeof(Layer) * P->layers.size());
            if (!P->pt.fused.empty()) memcpy(hp + P->off_f, P->pt.fused.data(), sizeof(RsMarch) * P->pt.fused.size());
            if (!P->pt.tiles.empty()) memcpy(hp + P->off_t, P->pt.tiles.data(), sizeof(RsTile) * P->pt.tiles.size());
            if (!P->pt.h.empty()) memcpy(hp + P->off_h, P->pt.h.data(), sizeof(RsJob) * P->pt.h.size());
            if (!P->pt.v.empty()) memcpy(hp + P->off_v, P->pt.v.data(), sizeof(RsJob) * P->pt.v.size());
            if (!P->pt.lane.empty()) memcpy(hp + P->off_lane, P->pt.lane.data(), sizeof(RsLaneUnit) * P->pt.lane.size());
        }
        HIP_TRY(hipMemcpyAsync(upload_dst, slot->host, upload, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(slot->ev, stream));
        slot->pending = true;
        }
        if (slot_tab) {
            slot_tab->outs = std::move(key);
            memcpy(slot_tab->class_end, class_end, sizeof class_end);
            slot_tab->pitch = pitch;
        }
    }
    const Job *jobs_dev = one ? nullptr : (P->persistent ? slot_tab->dev : reinterpret_cast<const Job *>(dp));

    for (const auto &todo : P->planar_todo) {
        bool missing = false;
        for (int e : todo.second) missing |= !todo.first->planar_built[(size_t)e];
        if (missing)
            if (int rc = atlas_planar_build(todo.first, todo.second, stream)) return rc;
    }
    for (const auto &todo : P->tiled_todo) {
        bool missing = false;
        for (int e : todo.second) missing |= !todo.first->tiled_built[(size_t)e];
        if (missing)
            if (int rc = atlas_tiled_build(todo.first, todo.second, stream)) return rc;
    }
    const bool prof = ctx->profiling && ctx->prof_calls < ctx->prof_max && (ctx->prof_seen++ % ctx->prof_every) == 0;
    hipEvent_t *pe = prof ? &ctx->prof_events[(size_t)ctx->prof_calls * 3] : nullptr;
    if (prof) HIP_TRY(hipEventRecord(pe[0], stream));
    const Layer *layers_dev = reinterpret_cast<const Layer *>(dp + P->off_layers);
    const RsMarch *fused_dev = reinterpret_cast<const RsMarch *>(dp + P->off_f);
    // a persistent plan that has run before finds its resampled layers in its own scratch: composite only
    const bool resident = P->persistent && P->resampled_valid;
    if (!resident) {
        if (P->pt.lane_slots > 0 && !P->pt.lane.empty())
            HIP_TRY(launch_resample_lane(reinterpret_cast<const RsLaneUnit *>(dp + P->off_lane), P->pt.lane_slots, stream));
        HIP_TRY(launch_resample_march(fused_dev, (int)P->pt.fused.size(), P->pt.lds_march, stream));
        HIP_TRY(launch_resample_tile(reinterpret_cast<const RsTile *>(dp + P->off_t), (int)P->pt.tiles.size(),
                                      P->pt.tiles_whole, P->pt.tiles_lds, stream, tiles_in_args ? P->pt.tiles.data() : nullptr));
        HIP_TRY(launch_resample_h(reinterpret_cast<const RsJob *>(dp + P->off_h), (int)P->pt.h.size(),
                                  P->pt.max_h_out_w, P->pt.max_h_rows, stream));
        HIP_TRY(launch_resample_v(reinterpret_cast<const RsJob *>(dp + P->off_v), (int)P->pt.v.size(),
                                  P->pt.max_v_out_w, P->pt.max_v_out_h, stream));
    }
    if (prof) HIP_TRY(hipEventRecord(pe[1], stream));
    // (Overlapping the issue-bound resample with the memory-bound composite was built twice in round 4 -- resample groups
    // on side streams with the composite of a canvas band / a chunk of canvases behind each group's event, and FUSED
    // launches with the two kernels as roles of one -- bit-exact both, slower both: a cross-stream edge costs 15-50 us on
    // this runtime, and the composite starves inside the resample kernel's register / LDS budget.
    // profiles/r04_pipeline_streams.txt, r04_fused_launches.txt, r04_overlap_experiments.patch.)
    HIP_TRY(launch_composite(jobs_dev, layers_dev, n_jobs, class_end, pitch, one ? &P->ordered[0] : nullptr,
                             pack_layers ? P->layers.data() : nullptr, stream, &P->stats.composite_blocks));
    if (prof) {
        HIP_TRY(hipEventRecord(pe[2], stream));
        ++ctx->prof_calls;
    }
    if (P->persistent) P->resampled_valid = true;
    ctx->stats = P->stats;
    return MIC_OK;
}

extern "C" int mic_composite_batch(mic_ctx *ctx, int n_atlases, mic_atlas *const *atlases, int n_jobs,
                                   const mic_job *jobs, int filter, void *stream_v) {
    CTX_ENTER(ctx);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = adopt_stream(ctx, stream)) return rc;  // before the transient plan may regrow the arena
    mic_plan P;
    int rc = plan_build(ctx, n_atlases, atlases, n_jobs, jobs, filter, /*persistent=*/false, stream, &P);
    if (rc == MIC_OK) rc = plan_submit(&P, nullptr, stream);
    // a failed call may have entered layers into the resident cache whose resample never ran: forget them all
    if (rc != MIC_OK && !ctx->layer_cache.empty()) layer_cache_drop(ctx);
    return rc;
}

extern "C" int mic_layer_cache_clear(mic_ctx *ctx) {
    CTX_ENTER(ctx);
    layer_cache_drop(ctx);
    return MIC_OK;
}

extern "C" int mic_plan_invalidate(mic_plan *plan) {
    if (!plan) return fail(MIC_ERR_INVALID, "mic_plan_invalidate: null plan");
    CTX_ENTER(plan->ctx);
    plan->resampled_valid = false;
    return MIC_OK;
}

extern "C" int mic_render(mic_ctx *ctx, mic_atlas *atlas, const char *layout_json, size_t len, int32_t width,
                          int32_t height, const void *bg_dev, const uint8_t bg_rgba[4], int filter, void *out_dev,
                          void *stream_v, int32_t *n_placed) {
    if (!bg_dev && !bg_rgba) return fail(MIC_ERR_INVALID, "mic_render: bad arguments");
    mic_job canvas{};
    canvas.width = width; canvas.height = height;
    canvas.bg_dev = bg_dev;
    if (bg_rgba) memcpy(canvas.bg_rgba, bg_rgba, 4);
    canvas.out_dev = out_dev;
    return mic_render_job(ctx, atlas, layout_json, len, &canvas, filter, stream_v, n_placed);
}

// The sizes the placer sees of an atlas: one entry per id, the first occurrence (what a dict would hold).
static void placer_table(const mic_atlas *atlas, std::vector<int32_t> *ids, std::vector<int32_t> *ws, std::vector<int32_t> *hs) {
    const size_t n = atlas->index.size();
    ids->reserve(n); ws->reserve(n); hs->reserve(n);
    for (size_t i = 0; i < atlas->entries.size(); ++i) {
        const BlobEntry &e = atlas->entries[i];
        if (atlas->index.at(e.id) != (int)i) continue;
        ids->push_back(e.id); ws->push_back(e.w); hs->push_back(e.h);
    }
}

extern "C" int mic_render_job(mic_ctx *ctx, mic_atlas *atlas, const char *layout_json, size_t len, const mic_job *canvas,
                              int filter, void *stream_v, int32_t *n_placed) {
    const char *layouts[1] = {layout_json};
    const size_t lens[1] = {len};
    return mic_render_batch(ctx, atlas, 1, layouts, lens, canvas, filter, stream_v, n_placed);
}

extern "C" int mic_render_batch(mic_ctx *ctx, mic_atlas *atlas, int32_t n, const char *const *layouts, const size_t *lens,
                                const mic_job *canvases, int filter, void *stream_v, int32_t *n_placed) {
    CTX_ENTER(ctx);
    if (n < 0 || (n > 0 && (!layouts || !lens || !canvases)) || !atlas) return fail(MIC_ERR_INVALID, "mic_render: bad arguments");
    if (atlas->ctx != ctx) return fail(MIC_ERR_INVALID, "mic_render: atlas belongs to another context");
    std::vector<int32_t> ids, ws, hs;
    placer_table(atlas, &ids, &ws, &hs);
    // every tree is placed before anything is launched: one that needs the Python mirror declines the whole call
    std::vector<std::vector<mic_placement>> pls((size_t)n);
    std::vector<mic_job> jobs((size_t)n);
    std::vector<int32_t> oi, ob;
    std::string err;
    for (int32_t k = 0; k < n; ++k) {
        const mic_job &cv = canvases[k];
        if (!layouts[k] || !cv.out_dev || cv.width <= 0 || cv.height <= 0) return fail(MIC_ERR_INVALID, "mic_render: layout %d: bad arguments", k);
        oi.clear();
        ob.clear();
        const int frc = flex_place(layouts[k], lens[k], (int)ids.size(), ids.data(), ws.data(), hs.data(), cv.width, cv.height, &oi,
                                   &ob, &err);
        if (frc == kFlexMalformed) return fail(MIC_ERR_FORMAT, "mic_render: layout %d: %s", k, err.c_str());
        if (frc == kFlexUnsupported)
            return fail(MIC_ERR_UNSUPPORTED, "mic_render: layout %d uses features only the Python placer mirrors", k);
        if (n_placed) n_placed[k] = (int32_t)oi.size();
        std::vector<mic_placement> &pl = pls[(size_t)k];
        pl.resize(oi.size());
        for (size_t i = 0; i < oi.size(); ++i) {
            pl[i].atlas = 0;
            pl[i].object_id = oi[i];
            for (int c = 0; c < 4; ++c) pl[i].box[c] = ob[4 * i + c];
        }
        jobs[(size_t)k] = cv;
        jobs[(size_t)k].n_placements = (int32_t)pl.size();
        jobs[(size_t)k].placements = pl.data();
    }
    mic_atlas *atl[1] = {atlas};
    return mic_composite_batch(ctx, 1, atl, n, jobs.data(), filter, stream_v);
}

extern "C" int mic_contact_sheet_size(int32_t n, int32_t thumb_w, int32_t thumb_h, int32_t cols, int32_t label_h,
                                      int32_t *sheet_w, int32_t *sheet_h) {
    if (!sheet_w || !sheet_h || n < 0 || thumb_w <= 0 || thumb_h <= 0 || cols <= 0 || label_h < 0)
        return fail(MIC_ERR_INVALID, "mic_contact_sheet_size: bad arguments");
    const int64_t cell_w = thumb_w, cell_h = (int64_t)thumb_h + label_h;
    // macro_placement_test.py:198-207: an empty list gives one blank cell
    const int64_t w = n == 0 ? cell_w : (int64_t)cols * cell_w;
    const int64_t h = n == 0 ? cell_h : (int64_t)((n + cols - 1) / cols) * cell_h;
    if (w > kMaxDim || h > kMaxDim) return fail(MIC_ERR_INVALID, "contact sheet of %lldx%lld is too large", (long long)w, (long long)h);
    *sheet_w = (int32_t)w;
    *sheet_h = (int32_t)h;
    return MIC_OK;
}

extern "C" int mic_contact_sheet(mic_ctx *ctx, mic_atlas *atlas, int32_t n, const int32_t *object_ids, int32_t thumb_w,
                                 int32_t thumb_h, int32_t cols, int32_t label_h, int32_t n_strips,
                                 const mic_label_strip *strips, void *out_dev, void *stream_v) {
    CTX_ENTER(ctx);
    int32_t W = 0, H = 0;
    if (int rc = mic_contact_sheet_size(n, thumb_w, thumb_h, cols, label_h, &W, &H)) return rc;
    if (!atlas || !out_dev || (n > 0 && !object_ids) || n_strips < 0 || (n_strips > 0 && !strips))
        return fail(MIC_ERR_INVALID, "mic_contact_sheet: bad arguments");
    if (atlas->ctx != ctx) return fail(MIC_ERR_INVALID, "mic_contact_sheet: atlas belongs to another context");
    const int cell_w = thumb_w, cell_h = thumb_h + label_h;
    // the labels become a throw-away atlas of black RGBA strips whose alpha is the coverage mask: blending one
    // with the alpha-over kernel is ImageDraw.text's own mask blend on an opaque sheet, div255(dst * (255 - m) + 128).
    // The strips travel through a slot of the staging ring (pinned host half -> device half, one asynchronous copy) and
    // are wrapped by a non-owning atlas on this stack frame: no allocation, no device-wide wait (round 4; it used to be
    // an atlas of its own: hipMalloc + a synchronous copy + hipDeviceSynchronize + hipFree per sheet).
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = adopt_stream(ctx, stream)) return rc;
    std::vector<int32_t> sid((size_t)n_strips), sw((size_t)n_strips), sh((size_t)n_strips);
    for (int k = 0; k < n_strips; ++k) {
        const mic_label_strip &S = strips[k];
        if (S.w <= 0 || S.h <= 0 || !S.coverage_host || S.cell < 0 || S.cell >= n)
            return fail(MIC_ERR_INVALID, "mic_contact_sheet: label strip %d is malformed", k);
        if ((int64_t)S.w * S.h > kMaxLayerPx) return fail(MIC_ERR_INVALID, "mic_contact_sheet: label strip %d is too large", k);
        sid[(size_t)k] = k; sw[(size_t)k] = S.w; sh[(size_t)k] = S.h;
    }
    mic_atlas labels_obj;
    mic_atlas *labels = nullptr;
    if (n_strips > 0) {
        size_t total = 0;
        if (int rc = blob_layout(n_strips, sid.data(), sw.data(), sh.data(), &labels_obj.entries, &total)) return rc;
        Slot *slot = nullptr;
        if (int rc = acquire_slot(ctx, total, &slot)) return rc;
        uint8_t *hp = static_cast<uint8_t *>(slot->host);
        memset(hp, 0, total);  // (guard bands, gaps and the strips' r, g, b)
        for (int k = 0; k < n_strips; ++k) {
            const mic_label_strip &S = strips[k];
            uint8_t *px = hp + labels_obj.entries[(size_t)k].offset;
            for (size_t i = 0; i < (size_t)S.w * S.h; ++i) px[4 * i + 3] = S.coverage_host[i];
        }
        HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, total, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(slot->ev, stream));
        slot->pending = true;
        labels_obj.ctx = ctx;
        labels_obj.blob = slot->dev;  // (read by this call's composite launch, which follows on the same stream)
        labels_obj.bytes = total;
        labels_obj.owns = false;
        atlas_finish(&labels_obj);
        labels = &labels_obj;
    }
    std::vector<mic_placement> pl;
    for (int i = 0; i < n; ++i) {
        auto it = atlas->index.find(object_ids[i]);
        if (it == atlas->index.end())
            return fail(MIC_ERR_INVALID, "mic_contact_sheet: object id %d is not in the atlas", object_ids[i]);
        const BlobEntry &E = atlas->entries[(size_t)it->second];
        int tw = 0, th = 0;
        thumbnail_size(E.w, E.h, thumb_w, thumb_h, &tw, &th);
        const int r = i / cols, c = i % cols;
        // :216-217, Python floor division (tw <= thumb_w and th <= thumb_h: never negative)
        const int x = c * cell_w + (cell_w - tw) / 2, y = r * cell_h + (thumb_h - th) / 2;
        pl.push_back(mic_placement{0, object_ids[i], {x, y, x + tw, y + th}});
        for (int k = 0; k < n_strips; ++k)
            if (strips[k].cell == i)
                pl.push_back(mic_placement{1, k, {strips[k].x, strips[k].y, strips[k].x + strips[k].w, strips[k].y + strips[k].h}});
    }
    mic_job job{};
    job.width = W; job.height = H;
    job.bg_rgba[0] = job.bg_rgba[1] = job.bg_rgba[2] = job.bg_rgba[3] = 255;  // :207 white, opaque
    job.n_placements = (int32_t)pl.size();
    job.placements = pl.data();
    job.out_dev = out_dev;
    mic_atlas *atl[2] = {atlas, labels};
    return mic_composite_batch(ctx, labels ? 2 : 1, atl, 1, &job, MIC_FILTER_LANCZOS, stream_v);
}

extern "C" int mic_plan_create(mic_ctx *ctx, int n_atlases, mic_atlas *const *atlases, int n_jobs,
                               const mic_job *jobs, int filter, mic_plan **out) {
    CTX_ENTER(ctx);
    if (!out) return fail(MIC_ERR_INVALID, "mic_plan_create: null out");
    *out = nullptr;
    mic_plan *P = new (std::nothrow) mic_plan();
    if (!P) return fail(MIC_ERR_NOMEM, "out of host memory");
    if (int rc = plan_build(ctx, n_atlases, atlases, n_jobs, jobs, filter, /*persistent=*/true, ctx->last_stream, P)) {
        mic_plan_destroy(P);
        return rc;
    }
    *out = P;
    return MIC_OK;
}

extern "C" int mic_plan_run(mic_plan *plan, void *const *outs, void *stream) {
    if (!plan) return fail(MIC_ERR_INVALID, "mic_plan_run: null plan");
    CTX_ENTER(plan->ctx);
    return plan_submit(plan, outs, static_cast<hipStream_t>(stream));
}

extern "C" int mic_plan_destroy(mic_plan *plan) {
    if (!plan) return MIC_OK;
    if (plan->ctx) {
        (void)hipSetDevice(plan->device);
        if (plan->scratch || plan->tables_dev) (void)hipDeviceSynchronize();
    }
    if (plan->scratch) (void)hipFree(plan->scratch);
    if (plan->tables_dev) (void)hipFree(plan->tables_dev);
    for (auto &t : plan->job_tables)
        if (t.dev) (void)hipFree(t.dev);
    delete plan;
    return MIC_OK;
}

extern "C" int mic_plan_stats(const mic_plan *plan, mic_stats *out) {
    if (!plan || !out) return fail(MIC_ERR_INVALID, "mic_plan_stats: null argument");
    *out = plan->stats;
    return MIC_OK;
}

extern "C" int mic_last_stats(const mic_ctx *ctx, mic_stats *out) {
    if (!ctx || !out) return fail(MIC_ERR_INVALID, "mic_last_stats: null argument");
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    *out = ctx->stats;
    return MIC_OK;
}

extern "C" int mic_profile_begin(mic_ctx *ctx, int max_calls) { return mic_profile_begin_sampled(ctx, max_calls, 1); }

extern "C" int mic_profile_begin_sampled(mic_ctx *ctx, int max_calls, int every) {
    CTX_ENTER(ctx);
    if (max_calls <= 0 || max_calls > (1 << 20) || every <= 0)
        return fail(MIC_ERR_INVALID, "mic_profile_begin: bad max_calls / every");
    ctx->prof_every = every;
    ctx->prof_seen = 0;
    while ((int)ctx->prof_events.size() < max_calls * 3) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        ctx->prof_events.push_back(ev);
    }
    ctx->prof_max = max_calls;
    ctx->prof_calls = 0;
    ctx->profiling = true;
    return MIC_OK;
}

extern "C" int mic_profile_end(mic_ctx *ctx, void *stream, int *n_calls, double *composite_ms, double *resample_ms) {
    CTX_ENTER(ctx);
    if (!n_calls || !composite_ms || !resample_ms) return fail(MIC_ERR_INVALID, "mic_profile_end: null argument");
    ctx->profiling = false;
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    double c = 0.0, r = 0.0;
    for (int i = 0; i < ctx->prof_calls; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_events[(size_t)i * 3], ctx->prof_events[(size_t)i * 3 + 1]));
        r += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_events[(size_t)i * 3 + 1], ctx->prof_events[(size_t)i * 3 + 2]));
        c += ms;
    }
    *n_calls = ctx->prof_calls;
    *composite_ms = c;
    *resample_ms = r;
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ resize
extern "C" int mic_resize(mic_ctx *ctx, const void *src_dev, int32_t src_w, int32_t src_h, void *dst_dev,
                          int32_t dst_w, int32_t dst_h, int filter, void *stream_v) {
    CTX_ENTER(ctx);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (!src_dev || !dst_dev) return fail(MIC_ERR_INVALID, "mic_resize: null image");
    if (src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0 || src_w > kMaxDim || src_h > kMaxDim ||
        dst_w > kMaxDim || dst_h > kMaxDim)
        return fail(MIC_ERR_INVALID, "mic_resize: invalid size %dx%d -> %dx%d", src_w, src_h, dst_w, dst_h);
    // the kernels index pixels with 32 bits (also across the two-pass fallback's dst_w x src_h intermediate)
    if ((int64_t)src_w * src_h >= ((int64_t)1 << 31) || (int64_t)dst_w * dst_h >= ((int64_t)1 << 31) ||
        (int64_t)dst_w * src_h >= ((int64_t)1 << 31))
        return fail(MIC_ERR_INVALID, "mic_resize: images of 2^31 pixels or more are not supported (%dx%d -> %dx%d)",
                    src_w, src_h, dst_w, dst_h);
    if (filter != MIC_FILTER_LANCZOS && filter != MIC_FILTER_BILINEAR)
        return fail(MIC_ERR_INVALID, "unknown filter %d", filter);
    if (int rc = adopt_stream(ctx, stream)) return rc;
    if (src_w == dst_w && src_h == dst_h) {  // Image.resize returns a copy
        HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, (size_t)src_w * src_h * 4, hipMemcpyDeviceToDevice, stream));
        return MIC_OK;
    }
    ResizePlan rp{};
    rp.src = reinterpret_cast<uint64_t>(src_dev);
    rp.sw = src_w; rp.sh = src_h; rp.dw = dst_w; rp.dh = dst_h;
    rp.dst_ptr = reinterpret_cast<uint64_t>(dst_dev);
    // a single image: the tile kernel (it premultiplies and planarises the window while loading it; the marching
    // kernel would first need a planar copy of the whole source, and one image rarely fills the chip with its units)
    if (int rc = choose_tile(ctx, &rp, filter, stream, (int64_t)dst_w * dst_h < ctx->tile_small_px)) return rc;
    size_t need = 0;
    if (rp.tx16 == 0 && dst_w != src_w && dst_h != src_h) need = (size_t)dst_w * src_h * 4 + kGuard;
    if (int rc = ensure_arena(ctx, need)) return rc;
    PassTables pt;
    std::vector<ResizePlan> plans{rp};
    if (int rc = plan_passes(ctx, plans, filter, ctx->arena, stream, &pt)) return rc;
    if (ctx->tile_args && pt.h.empty() && pt.v.empty() && rs_tile_in_args((int)pt.tiles.size(), pt.tiles_whole)) {
        // (a small image: its tile entries ride in the kernel arguments, nothing is staged or uploaded)
        HIP_TRY(launch_resample_tile(nullptr, (int)pt.tiles.size(), pt.tiles_whole, pt.tiles_lds, stream, pt.tiles.data()));
        return MIC_OK;
    }
    const size_t off_v = 64, off_t = 128;
    const size_t total = off_t + sizeof(RsTile) * std::max<size_t>(1, pt.tiles.size());
    Slot *slot = nullptr;
    if (int rc = acquire_slot(ctx, total, &slot)) return rc;
    char *hp = static_cast<char *>(slot->host);
    if (!pt.h.empty()) memcpy(hp, pt.h.data(), sizeof(RsJob));
    if (!pt.v.empty()) memcpy(hp + off_v, pt.v.data(), sizeof(RsJob));
    if (!pt.tiles.empty()) memcpy(hp + off_t, pt.tiles.data(), sizeof(RsTile) * pt.tiles.size());
    HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, total, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(slot->ev, stream));
    slot->pending = true;
    char *dp = static_cast<char *>(slot->dev);
    HIP_TRY(launch_resample_tile(reinterpret_cast<const RsTile *>(dp + off_t), (int)pt.tiles.size(), pt.tiles_whole,
                                  pt.tiles_lds, stream));
    HIP_TRY(launch_resample_h(reinterpret_cast<const RsJob *>(dp), (int)pt.h.size(), pt.max_h_out_w,
                              pt.max_h_rows, stream));
    HIP_TRY(launch_resample_v(reinterpret_cast<const RsJob *>(dp + off_v), (int)pt.v.size(), pt.max_v_out_w,
                              pt.max_v_out_h, stream));
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ background
static int median_view(const mic_image_view &v, int i, MedianView *out) {
    if (!v.rgba_dev || v.width <= 0 || v.height <= 0 || v.width > kMaxDim || v.height > kMaxDim)
        return fail(MIC_ERR_INVALID, "mic_median_rgb: image %d: bad pointer or size", i);
    const int64_t stride = v.stride_bytes == 0 ? (int64_t)v.width * 4 : v.stride_bytes;
    if (stride % 4 != 0 || stride < (int64_t)v.width * 4 || stride / 4 > INT32_MAX ||
        reinterpret_cast<uintptr_t>(v.rgba_dev) % 4 != 0)
        return fail(MIC_ERR_INVALID, "mic_median_rgb: image %d: stride must be a multiple of 4 and >= width * 4, pixels 4-byte aligned", i);
    out->px = v.rgba_dev;
    out->w = v.width;
    out->h = v.height;
    // a one-row view, or rows that follow each other without a gap, is a packed image whatever its stride says
    out->stride_px = (v.height == 1 || stride == (int64_t)v.width * 4) ? v.width : (int32_t)(stride / 4);
    return MIC_OK;
}

// results of call chunk: device words res_dev[0..k)
static int median_launch(mic_ctx *ctx, int k, const mic_image_view *views, uint32_t *const *outs, hipStream_t stream) {
    MedianView mv[kMedianMaxBatch];
    for (int i = 0; i < k; ++i)
        if (int rc = median_view(views[i], i, &mv[i])) return rc;
    HIP_TRY(launch_median_batch(k, mv, outs, ctx->median_scratch, &ctx->median_state, ctx->median_two_launches, stream));
    return MIC_OK;
}

extern "C" int mic_median_rgb_dev(mic_ctx *ctx, const void *rgba_dev, int32_t width, int32_t height,
                                  void *rgba_out_dev, void *stream_v) {
    CTX_ENTER(ctx);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (!rgba_dev || !rgba_out_dev || width <= 0 || height <= 0)
        return fail(MIC_ERR_INVALID, "mic_median_rgb: bad arguments");
    if (int rc = adopt_stream(ctx, stream)) return rc;
    const mic_image_view v{rgba_dev, width, height, 0};
    uint32_t *out = static_cast<uint32_t *>(rgba_out_dev);
    return median_launch(ctx, 1, &v, &out, stream);
}

extern "C" int mic_median_rgb(mic_ctx *ctx, const void *rgba_dev, int32_t width, int32_t height,
                              uint8_t out_rgb[3], void *stream_v) {
    const mic_image_view v{rgba_dev, width, height, 0};
    return mic_median_rgb_batch(ctx, 1, &v, out_rgb, stream_v);
}

extern "C" int mic_median_rgb_batch(mic_ctx *ctx, int32_t n, const mic_image_view *views, uint8_t *out_rgb,
                                    void *stream_v) {
    if (n < 0 || (n > 0 && (!views || !out_rgb))) return fail(MIC_ERR_INVALID, "mic_median_rgb_batch: bad arguments");
    CTX_ENTER(ctx);  // held across the copy-back: median_host is context state
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = adopt_stream(ctx, stream)) return rc;
    uint32_t *res_dev = ctx->median_scratch + kMedianScratchWords;
    for (int32_t at = 0; at < n; at += kMedianMaxBatch) {  // one launch, one copy-back, one wait per 16 images
        const int k = std::min<int32_t>(kMedianMaxBatch, n - at);
        uint32_t *outs[kMedianMaxBatch];
        for (int i = 0; i < k; ++i) outs[i] = res_dev + i;
        if (int rc = median_launch(ctx, k, views + at, outs, stream)) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->median_host, res_dev, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int i = 0; i < k; ++i) {
            const uint32_t v = ctx->median_host[i];
            out_rgb[3 * (size_t)(at + i) + 0] = (uint8_t)(v & 255u);
            out_rgb[3 * (size_t)(at + i) + 1] = (uint8_t)((v >> 8) & 255u);
            out_rgb[3 * (size_t)(at + i) + 2] = (uint8_t)((v >> 16) & 255u);
        }
    }
    return MIC_OK;
}

extern "C" int mic_fill_solid(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height,
                              const uint8_t rgba[4], void *stream_v) {
    CTX_ENTER(ctx);
    if (!out_dev || !rgba || width <= 0 || height <= 0) return fail(MIC_ERR_INVALID, "mic_fill_solid: bad arguments");
    const uint32_t c = (uint32_t)rgba[0] | ((uint32_t)rgba[1] << 8) | ((uint32_t)rgba[2] << 16) | ((uint32_t)rgba[3] << 24);
    HIP_TRY(launch_fill(out_dev, c, (size_t)width * height, static_cast<hipStream_t>(stream_v)));
    return MIC_OK;
}

extern "C" int mic_fill_gradient(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height, const uint8_t c1[3],
                                 const uint8_t c2[3], int vertical, void *stream_v) {
    CTX_ENTER(ctx);
    if (!out_dev || !c1 || !c2 || width <= 0 || height <= 0 || width > kMaxDim || height > kMaxDim)
        return fail(MIC_ERR_INVALID, "mic_fill_gradient: bad arguments");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = adopt_stream(ctx, stream)) return rc;  // the table is context scratch: one stream at a time
    if (!ctx->gradient_table)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->gradient_table), kGradientTableWords * sizeof(uint32_t)));
    HIP_TRY(launch_gradient(out_dev, width, height, c1, c2, vertical ? 1 : 0, ctx->gradient_table, stream));
    return MIC_OK;
}

extern "C" int mic_draw_rect_outlines(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height, int32_t n,
                                      const int32_t *boxes, const uint8_t *colours, int32_t outline_width,
                                      void *stream_v) {
    CTX_ENTER(ctx);
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (!out_dev || width <= 0 || height <= 0 || width > kMaxDim || height > kMaxDim || n < 0 ||
        (n > 0 && (!boxes || !colours)) || outline_width < 0 || outline_width > kMaxDim)
        return fail(MIC_ERR_INVALID, "mic_draw_rect_outlines: bad arguments");
    if (int rc = adopt_stream(ctx, stream)) return rc;
    const int w = outline_width == 0 ? 1 : outline_width;  // Draw.c: width 0 draws width 1
    std::vector<OutlineRect> rects((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int64_t x0 = boxes[4 * i], y0 = boxes[4 * i + 1], x1 = boxes[4 * i + 2], y1 = boxes[4 * i + 3];
        if (x1 < x0 || y1 < y0)
            return fail(MIC_ERR_INVALID, "box %d: %s must be greater than or equal to %s", i, x1 < x0 ? "x1" : "y1",
                        x1 < x0 ? "x0" : "y0");
        auto clampi = [](int64_t v) { return (int32_t)std::max<int64_t>(-(1 << 30), std::min<int64_t>(v, 1 << 30)); };
        // the vertical lines start at ya and take |yb - ya| steps towards yb without reaching it
        const int64_t ya = y0 + w, yb = y1 - w + 1;
        const int64_t vlo = yb > ya ? ya : yb + 1, vhi = yb > ya ? yb - 1 : ya;
        OutlineRect r{};
        r.x0 = clampi(x0); r.y0 = clampi(y0); r.x1 = clampi(x1); r.y1 = clampi(y1);
        r.vlo = clampi(vlo); r.vhi = clampi(vhi);
        r.ymin = clampi(std::min(std::min(y0, y1 - w + 1), vlo));
        r.ymax = clampi(std::max(std::max(y1, y0 + w - 1), vhi));
        r.rgba = (uint32_t)colours[4 * i] | ((uint32_t)colours[4 * i + 1] << 8) | ((uint32_t)colours[4 * i + 2] << 16) |
                 ((uint32_t)colours[4 * i + 3] << 24);
        rects[(size_t)i] = r;
    }
    const size_t bytes = std::max<size_t>(sizeof(OutlineRect), sizeof(OutlineRect) * rects.size());
    Slot *slot = nullptr;
    if (int rc = acquire_slot(ctx, bytes, &slot)) return rc;
    if (n > 0) memcpy(slot->host, rects.data(), sizeof(OutlineRect) * rects.size());
    HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(slot->ev, stream));
    slot->pending = true;
    HIP_TRY(launch_rect_outlines(out_dev, width, height, static_cast<const OutlineRect *>(slot->dev), n, w, stream));
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ self test
extern "C" int mic_selftest(mic_ctx *ctx, void *stream_v) {
    CTX_ENTER(ctx);
    int bad = 0, first = -1;
    HIP_TRY(run_selftest_clip(static_cast<hipStream_t>(stream_v), &bad, &first));
    if (bad)
        return fail(MIC_ERR_HIP, "mic_selftest: %d of 256 known answers of the clip / pack helpers differ (first: word %d); the "
                    "compiler lowers v_ashr_pk_u8_i32 or its neighbours differently from the build this library was "
                    "validated with (kernels_resample.hip: clip8, clip8x4)", bad, first);
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ PIL-level drop-in helpers
extern "C" int mic_host_rows_solid(const void *const *rows_host, int32_t width, int32_t y0, int32_t y1,
                                   const uint8_t rgba[4], int *is_solid) {
    if (!rows_host || !rgba || !is_solid || width <= 0 || y0 < 0 || y1 < y0)
        return fail(MIC_ERR_INVALID, "mic_host_rows_solid: bad arguments");
    uint32_t c;
    memcpy(&c, rgba, 4);
    const uint64_t cc = (uint64_t)c | ((uint64_t)c << 32);
    *is_solid = 0;
    for (int32_t y = y0; y < y1; ++y) {
        const uint8_t *row = static_cast<const uint8_t *>(rows_host[y]);
        if (!row) return fail(MIC_ERR_INVALID, "mic_host_rows_solid: row %d is null", y);
        int32_t x = 0;
        uint64_t diff = 0;  // OR of (pixel pair ^ colour): branch-free inner loop, one test per 64 pixels
        for (; x + 64 <= width; x += 64) {
            for (int k = 0; k < 32; ++k) {
                uint64_t v;
                memcpy(&v, row + (size_t)(x + 2 * k) * 4, 8);
                diff |= v ^ cc;
            }
            if (diff) return MIC_OK;
        }
        for (; x < width; ++x) {
            uint32_t v;
            memcpy(&v, row + (size_t)x * 4, 4);
            diff |= v ^ c;
        }
        if (diff) return MIC_OK;
    }
    *is_solid = 1;
    return MIC_OK;
}

extern "C" int mic_download(mic_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes, void *stream_v,
                            int32_t *ticket) {
    CTX_ENTER(ctx);
    if (!ticket || (bytes > 0 && (!src_dev || !dst_host))) return fail(MIC_ERR_INVALID, "mic_download: null pointer");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    const int t = ctx->dl_next;
    ctx->dl_next = (t + 1) % mic_ctx::kDownloads;
    if (!ctx->dl_event[t]) HIP_TRY(hipEventCreateWithFlags(&ctx->dl_event[t], hipEventDisableTiming));
    if (ctx->dl_pending[t]) HIP_TRY(hipEventSynchronize(ctx->dl_event[t]));  // 16 copies later and still not waited for
    if (bytes > 0) HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ctx->dl_event[t], stream));
    ctx->dl_pending[t] = true;
    ctx->dl_gen[t] = (ctx->dl_gen[t] + 1) & 0x7fffffu;
    *ticket = (int32_t)((ctx->dl_gen[t] << 8) | (uint32_t)t);
    return MIC_OK;
}

extern "C" int mic_download_wait(mic_ctx *ctx, int32_t ticket) {
    const int slot = ticket & 0xff;
    const uint32_t gen = (uint32_t)ticket >> 8;
    if (!ctx || ticket < 0 || slot >= mic_ctx::kDownloads) return fail(MIC_ERR_INVALID, "mic_download_wait: bad arguments");
    hipEvent_t ev;
    {
        std::lock_guard<std::recursive_mutex> lock(ctx->mu);
        // A ticket of an earlier generation: the entry has come round again, and mic_download waited for the old record
        // before it re-recorded the event -- that copy has landed, and the entry now belongs to someone else.
        if (ctx->dl_gen[slot] != gen || !ctx->dl_pending[slot]) return MIC_OK;
        ev = ctx->dl_event[slot];
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ev));  // (not under the lock: other threads keep enqueueing)
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (ctx->dl_gen[slot] == gen) ctx->dl_pending[slot] = false;  // (a newer holder's flag is not this call's to clear)
    return MIC_OK;
}

// ------------------------------------------------------------------------------------ PNG writer (host only)
static int png_pieces(const void *const *rows, const void *base, int32_t w, int32_t h, size_t stride, int level, int threads,
                      mic::PngPieces *pieces) {
    if (w <= 0 || h <= 0 || w > kMaxDim || h > kMaxDim || (!rows && !base) || (!rows && stride < (size_t)w * 4))
        return fail(MIC_ERR_INVALID, "mic_png: bad arguments");
    std::vector<const uint8_t *> table;
    if (!rows) {
        table.resize((size_t)h);
        for (int32_t y = 0; y < h; ++y) table[(size_t)y] = static_cast<const uint8_t *>(base) + (size_t)y * stride;
    } else {
        for (int32_t y = 0; y < h; ++y)
            if (!rows[y]) return fail(MIC_ERR_INVALID, "mic_png: row %d is null", y);
    }
    std::string err;
    const uint8_t *const *r = rows ? reinterpret_cast<const uint8_t *const *>(rows) : table.data();
    if (int rc = mic::png_encode_rows(r, w, h, level, threads, pieces, &err))
        return fail(rc == -3 ? MIC_ERR_NOMEM : MIC_ERR_INVALID, "%s", err.c_str());
    return MIC_OK;
}

static int png_to_memory(const mic::PngPieces &pieces, void *out, size_t capacity, size_t *out_bytes) {
    const size_t total = pieces.total();
    if (out_bytes) *out_bytes = total;
    if (!out || capacity < total) return fail(MIC_ERR_INVALID, "mic_png_encode: %zu bytes needed, capacity %zu", total, capacity);
    uint8_t *o = static_cast<uint8_t *>(out);
    for (const auto &p : pieces.pieces) {
        memcpy(o, p.data, p.size);
        o += p.size;
    }
    return MIC_OK;
}

static int png_to_file(const mic::PngPieces &pieces, const char *path) {
    std::string err;
    if (mic::png_write_file(pieces, path, &err) != 0) return fail(MIC_ERR_INVALID, "%s", err.c_str());
    return MIC_OK;
}

extern "C" int mic_png_write_async(const char *path, const void *const *rows_host, int32_t width, int32_t height, int level,
                                   int threads, int64_t *job) {
    if (!job || !rows_host || !path || width <= 0 || height <= 0 || width > kMaxDim || height > kMaxDim)
        return fail(MIC_ERR_INVALID, "mic_png_write_async: bad arguments");
    for (int32_t y = 0; y < height; ++y)
        if (!rows_host[y]) return fail(MIC_ERR_INVALID, "mic_png_write_async: row %d is null", y);
    std::string err;
    const int64_t id = mic::png_write_async(path, reinterpret_cast<const uint8_t *const *>(rows_host), width, height, level, threads, &err);
    if (id < 0) return fail(MIC_ERR_NOMEM, "%s", err.c_str());
    *job = id;
    return MIC_OK;
}

extern "C" int mic_png_wait(int64_t job) {
    std::string err;
    const int rc = mic::png_wait(job, &err);
    if (rc != 0) return fail(rc == -3 ? MIC_ERR_NOMEM : MIC_ERR_INVALID, "%s", err.c_str());
    return MIC_OK;
}

extern "C" size_t mic_png_bound(int32_t width, int32_t height) { return mic::png_bound(width, height); }

extern "C" int mic_png_encode(const void *rgba_host, int32_t width, int32_t height, size_t stride_bytes, int level,
                              int threads, void *out, size_t capacity, size_t *out_bytes) {
    mic::PngPieces pieces;
    if (int rc = png_pieces(nullptr, rgba_host, width, height, stride_bytes, level, threads, &pieces)) return rc;
    return png_to_memory(pieces, out, capacity, out_bytes);
}

extern "C" int mic_png_encode_rows(const void *const *rows_host, int32_t width, int32_t height, int level, int threads,
                                   void *out, size_t capacity, size_t *out_bytes) {
    if (!rows_host) return fail(MIC_ERR_INVALID, "mic_png_encode_rows: null row table");
    mic::PngPieces pieces;
    if (int rc = png_pieces(rows_host, nullptr, width, height, 0, level, threads, &pieces)) return rc;
    return png_to_memory(pieces, out, capacity, out_bytes);
}

extern "C" int mic_png_write(const char *path, const void *rgba_host, int32_t width, int32_t height, size_t stride_bytes,
                             int level, int threads) {
    mic::PngPieces pieces;
    if (int rc = png_pieces(nullptr, rgba_host, width, height, stride_bytes, level, threads, &pieces)) return rc;
    return png_to_file(pieces, path);
}

extern "C" int mic_png_write_rows(const char *path, const void *const *rows_host, int32_t width, int32_t height, int level,
                                  int threads) {
    if (!rows_host) return fail(MIC_ERR_INVALID, "mic_png_write_rows: null row table");
    mic::PngPieces pieces;
    if (int rc = png_pieces(rows_host, nullptr, width, height, 0, level, threads, &pieces)) return rc;
    return png_to_file(pieces, path);
}

// ------------------------------------------------------------------------------------ PNG reader (host only)
static std::atomic<uint64_t> g_png_decoded{0}, g_png_declined{0};

static int png_status(int rc, const std::string &err) {
    if (rc == 0) {
        g_png_decoded.fetch_add(1, std::memory_order_relaxed);
        return MIC_OK;
    }
    g_png_declined.fetch_add(1, std::memory_order_relaxed);
    return fail(rc == mic::kPngUnsupported ? MIC_ERR_UNSUPPORTED : rc == mic::kPngNoMem ? MIC_ERR_NOMEM : MIC_ERR_FORMAT, "%s", err.c_str());
}

extern "C" int mic_png_info(const void *png, size_t bytes, int32_t *width, int32_t *height) {
    if (!png || !width || !height) return fail(MIC_ERR_INVALID, "mic_png_info: null argument");
    std::string err;
    const int rc = mic::png_decode_info(static_cast<const uint8_t *>(png), bytes, width, height, &err);
    if (rc == 0) return MIC_OK;
    return fail(rc == mic::kPngUnsupported ? MIC_ERR_UNSUPPORTED : MIC_ERR_FORMAT, "%s", err.c_str());
}

extern "C" int mic_png_decode_rows(const void *png, size_t bytes, void *const *rows_host, int32_t width, int32_t height) {
    if (!png || !rows_host || width <= 0 || height <= 0) return fail(MIC_ERR_INVALID, "mic_png_decode: bad arguments");
    for (int32_t y = 0; y < height; ++y)
        if (!rows_host[y]) return fail(MIC_ERR_INVALID, "mic_png_decode: row %d is null", y);
    std::string err;
    const int rc = mic::png_decode_rows(static_cast<const uint8_t *>(png), bytes, reinterpret_cast<uint8_t *const *>(rows_host), width,
                                        height, true, &err);
    return png_status(rc, err);
}

extern "C" int mic_png_decode(const void *png, size_t bytes, void *rgba_out, size_t stride_bytes, int32_t width, int32_t height) {
    if (!png || !rgba_out || width <= 0 || height <= 0 || stride_bytes < (size_t)width * 4)
        return fail(MIC_ERR_INVALID, "mic_png_decode: bad arguments");
    std::vector<void *> rows((size_t)height);
    for (int32_t y = 0; y < height; ++y) rows[(size_t)y] = static_cast<uint8_t *>(rgba_out) + (size_t)y * stride_bytes;
    return mic_png_decode_rows(png, bytes, rows.data(), width, height);
}

extern "C" int mic_png_decode_many(int32_t n, const void *const *pngs, const size_t *bytes, void *const *const *rows_host,
                                   const int32_t *widths, const int32_t *heights, int threads, int32_t *status) {
    if (n < 0 || (n > 0 && (!pngs || !bytes || !rows_host || !widths || !heights || !status)))
        return fail(MIC_ERR_INVALID, "mic_png_decode_many: bad arguments");
    for (int32_t i = 0; i < n; ++i) {
        if (!pngs[i] || !rows_host[i] || widths[i] <= 0 || heights[i] <= 0) return fail(MIC_ERR_INVALID, "mic_png_decode_many: file %d: bad arguments", i);
        for (int32_t y = 0; y < heights[i]; ++y)
            if (!rows_host[i][y]) return fail(MIC_ERR_INVALID, "mic_png_decode_many: file %d: row %d is null", i, y);
    }
    std::string err;
    std::vector<int> st((size_t)std::max(n, 1), 0);
    const int rc = mic::png_decode_many(n, reinterpret_cast<const uint8_t *const *>(pngs), bytes,
                                        reinterpret_cast<uint8_t *const *const *>(rows_host), widths, heights, threads, st.data(), &err);
    uint64_t ok = 0;
    for (int32_t i = 0; i < n; ++i) {
        ok += st[(size_t)i] == 0;
        status[i] = st[(size_t)i] == 0 ? MIC_OK : st[(size_t)i] == mic::kPngUnsupported ? MIC_ERR_UNSUPPORTED
                                         : st[(size_t)i] == mic::kPngNoMem ? MIC_ERR_NOMEM : MIC_ERR_FORMAT;
    }
    g_png_decoded.fetch_add(ok, std::memory_order_relaxed);
    g_png_declined.fetch_add((uint64_t)n - ok, std::memory_order_relaxed);
    if (rc == 0) return MIC_OK;
    return fail(rc == mic::kPngUnsupported ? MIC_ERR_UNSUPPORTED : rc == mic::kPngNoMem ? MIC_ERR_NOMEM : MIC_ERR_FORMAT, "%s", err.c_str());
}

extern "C" int mic_png_decode_counts(uint64_t *decoded, uint64_t *declined) {
    if (decoded) *decoded = g_png_decoded.load(std::memory_order_relaxed);
    if (declined) *declined = g_png_declined.load(std::memory_order_relaxed);
    return MIC_OK;
}

extern "C" int mic_flex_place(const char *layout_json, size_t len, int n_objects, const int32_t *ids,
                              const int32_t *widths, const int32_t *heights, int32_t canvas_w, int32_t canvas_h,
                              int32_t capacity, int32_t *out_ids, int32_t *out_boxes, int32_t *out_count) {
    if (!layout_json || !out_count || n_objects < 0 || (n_objects > 0 && (!ids || !widths || !heights)) ||
        capacity < 0 || (capacity > 0 && (!out_ids || !out_boxes)) || canvas_w <= 0 || canvas_h <= 0)
        return fail(MIC_ERR_INVALID, "mic_flex_place: bad arguments");
    std::vector<int32_t> oi, ob;
    std::string err;
    const int rc = flex_place(layout_json, len, n_objects, ids, widths, heights, canvas_w, canvas_h, &oi, &ob, &err);
    if (rc == kFlexMalformed) return fail(MIC_ERR_FORMAT, "mic_flex_place: %s", err.c_str());
    if (rc == kFlexUnsupported)
        return fail(MIC_ERR_UNSUPPORTED, "mic_flex_place: layout uses features only the Python placer mirrors");
    *out_count = (int32_t)oi.size();
    if ((int32_t)oi.size() > capacity)
        return fail(MIC_ERR_INVALID, "mic_flex_place: %zu placements, capacity %d", oi.size(), capacity);
    if (!oi.empty()) {
        memcpy(out_ids, oi.data(), oi.size() * sizeof(int32_t));
        memcpy(out_boxes, ob.data(), ob.size() * sizeof(int32_t));
    }
    return MIC_OK;
}

extern "C" int mic_thumbnail_size(int32_t w, int32_t h, int32_t req_w, int32_t req_h, int32_t *out_w,
                                  int32_t *out_h) {
    if (!out_w || !out_h || w <= 0 || h <= 0 || req_w <= 0 || req_h <= 0)
        return fail(MIC_ERR_INVALID, "mic_thumbnail_size: bad arguments");
    int ow = 0, oh = 0;
    thumbnail_size(w, h, req_w, req_h, &ow, &oh);
    *out_w = ow;
    *out_h = oh;
    return MIC_OK;
}
