/*
 * mic.h -- C ABI of the MI355X-native compositor (libmic.so).
 *
 * The reference (FelixMul/image_transformation) has no FFI: its pixel path is three
 * importable Python callables that delegate the arithmetic to Pillow / NumPy.  This
 * header is the boundary a binding for that path would attach to; each entry point
 * names the reference interface it replaces (file:line under /root/reference).
 * The Python package image_transformation_amd binds it with ctypes (see
 * image_transformation_amd/_native.py) and keeps the reference's call surface.
 *
 * Conventions
 *   - every function returns 0 on success, a negative mic_status on failure;
 *     mic_last_error() returns a thread-local message for the last failure;
 *   - no exceptions cross the boundary; no torch types appear in any signature;
 *   - pixel buffers are RGBA8, interleaved, row-major, non-premultiplied, top-left origin
 *     (Pillow mode "RGBA"), row stride = width * 4 bytes;
 *   - "dev" pointers are device memory of the context's GPU (e.g. a torch tensor's
 *     data_ptr()); "host" pointers are ordinary memory owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued asynchronously on it; outputs are complete once the stream has drained.
 *     A context must be driven from one stream at a time.
 *   - there is NO CPU fallback: without a usable HIP device mic_create fails.
 *
 * Threads.  Every entry point that takes a context, or a plan / an atlas of it, may be called from
 * any thread at any time: the context carries a mutex and each call holds it from entry to return
 * (the staging ring, the scratch arena, the table caches and the current stream are shared state),
 * so concurrent calls on one context are serialised, calls on different contexts run in parallel.
 * That matches the Pillow calls this library replaces, which are thread-safe (the reference's
 * Streamlit app runs every session on its own thread).  Work is still ENQUEUED asynchronously:
 * two threads that pass different streams make the context switch streams, which synchronises the
 * old one (correct, slow); pass the same stream, or give each thread a context of its own.
 * mic_destroy, mic_atlas_destroy and mic_plan_destroy must not race with calls that use the
 * object being destroyed.  Lifetimes: atlases must outlive the plans made from them being RUN, and
 * a context must outlive mic_plan_run / mic_composite_batch calls on it; the destroy calls
 * themselves may come in any order (plans and atlases keep what they need to free their memory,
 * and a plan keeps the tables it points into alive).
 */
#ifndef MIC_H
#define MIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mic_ctx mic_ctx;     /* per-process, per-device state (tables, scratch arena) */
typedef struct mic_atlas mic_atlas; /* device-resident packed cutouts + id/size table        */
typedef struct mic_plan mic_plan;   /* a resolved batch of composite jobs, re-runnable            */

enum mic_status {
    MIC_OK = 0,
    MIC_ERR_INVALID = -1,   /* bad argument                                    */
    MIC_ERR_HIP = -2,       /* a HIP runtime call failed (message has details) */
    MIC_ERR_NOMEM = -3,     /* host or device allocation failed                */
    MIC_ERR_NODEVICE = -4,  /* no usable gfx950 device                         */
    MIC_ERR_FORMAT = -5,    /* malformed atlas blob / layout JSON              */
    MIC_ERR_UNSUPPORTED = -6 /* valid input this native path leaves to the caller */
};

enum mic_filter {
    MIC_FILTER_LANCZOS = 0, /* Pillow Image.LANCZOS (a = 3): what compositor.py:20 asks for */
    MIC_FILTER_BILINEAR = 1 /* Pillow Image.BILINEAR, same fixed-point machinery            */
};

const char *mic_last_error(void);
/* Library/ABI version: major << 16 | minor. */
int mic_version(void);

/* ---- context --------------------------------------------------------------------------- */
int mic_create(int device, mic_ctx **out);
int mic_destroy(mic_ctx *ctx);
/* Known-answer canary: a tiny kernel pushes boundary values through the shift / saturate / pack helpers the resample
 * kernels are built from (hipcc 7.2 miscompiled one such pattern -- v_ashr_pk_u8_i32 -- and the sources work around
 * it) and the results are compared with host arithmetic.  MIC_ERR_HIP with the details if a ROCm update changes the
 * lowering.  Synchronises `stream`.  Run by __graft_entry__.smoke() and the GPU tests.                          */
int mic_selftest(mic_ctx *ctx, void *stream);
/* Block until everything enqueued through this context on `stream` has finished. */
int mic_sync(mic_ctx *ctx, void *stream);

/* ---- atlas: replaces the dict returned by compositor.load_object_images -----------------
 * (compositor.py:25-35: {int object_id: RGBA Image}).  The atlas is uploaded once per
 * bundle and stays resident across composites / refine iterations / batches.            */

/* Upload n cutouts from host memory into one device blob owned by the atlas. */
int mic_atlas_create(mic_ctx *ctx, int n, const int32_t *ids, const int32_t *widths,
                     const int32_t *heights, const uint8_t *const *rgba_host, mic_atlas **out);
/* Bytes a blob for these cutouts needs (so a caller can allocate it as a torch tensor). */
int mic_atlas_blob_size(int n, const int32_t *widths, const int32_t *heights, size_t *bytes);
/* Write the blob header/table into host memory `blob_host` (bytes from mic_atlas_blob_size) and
 * return per-object pixel offsets; the caller copies pixels to blob+offset[i] and uploads.  */
int mic_atlas_blob_layout(int n, const int32_t *ids, const int32_t *widths, const int32_t *heights,
                          void *blob_host, size_t bytes, uint64_t *pixel_offsets);
/* Wrap a device blob (layout above; e.g. received by an RCCL broadcast).  The atlas does not
 * own the memory: the caller keeps it alive until mic_atlas_destroy.  `header_host` may pass
 * the first 32 + 32*n bytes of the blob (header + table) if the caller has them on the host,
 * otherwise NULL and the header is read back from the device.
 * The blob is IMMUTABLE while the atlas exists (as the cutouts of load_object_images are between the reference's
 * iterations): libmic derives and keeps data from it -- planar premultiplied copies of cutouts that get resampled,
 * resampled layers in persistent plans and in the context's layer cache (1.9) -- and cannot see a caller rewriting
 * the bytes in place.  A caller that must refresh the pixels destroys the atlas and wraps the blob again (a new
 * atlas is a new cache identity; plans built on the old one are rebuilt), or, for pixels that only identity-scale
 * placements read, calls mic_plan_invalidate / mic_layer_cache_clear.                                              */
int mic_atlas_from_device_blob(mic_ctx *ctx, const void *blob_dev, size_t bytes,
                               const void *header_host, mic_atlas **out);
int mic_atlas_device_blob(const mic_atlas *atlas, const void **blob_dev, size_t *bytes);
int mic_atlas_count(const mic_atlas *atlas);
/* Size / device address of one cutout; MIC_ERR_INVALID if the id is unknown. */
int mic_atlas_lookup(const mic_atlas *atlas, int32_t id, int32_t *width, int32_t *height,
                     const void **rgba_dev);
int mic_atlas_destroy(mic_atlas *atlas);

/* ---- composite: replaces compositor.composite (compositor.py:6-22) -------------------------
 * One job = one canvas: start from the background (a device image, or a solid colour that is
 * synthesised in-kernel: background_resizing.py:32), then for each placement IN LIST ORDER:
 * w = max(1, x2-x1), h = max(1, y2-y1); the cutout is resized to (w,h) with `filter` exactly
 * as Pillow does (identity size = no resampling) and alpha-composited at (x1,y1), clipped to
 * the canvas, with 8-bit rounding after every layer (Pillow AlphaComposite.c).  Placements whose
 * object id is not in their atlas are skipped (compositor.py:14-15).  Box coordinates are
 * already integers here: int() coercion of JSON values is the Python binding's job.        */
typedef struct mic_placement {
    int32_t atlas;     /* index into the `atlases` array of the call */
    int32_t object_id; /* id inside that atlas                       */
    int32_t box[4];    /* x1, y1, x2, y2 on the canvas               */
} mic_placement;

typedef struct mic_job {
    int32_t width, height;           /* canvas size                                       */
    const void *bg_dev;              /* device RGBA background image, or NULL for solid    */
    uint8_t bg_rgba[4];              /* solid background colour when bg_dev == NULL       */
    int32_t n_placements;
    const mic_placement *placements; /* host array                                        */
    void *out_dev;                   /* device RGBA canvas, width*height*4 bytes; must not
                                        alias bg_dev (compositor.py:11 copies the bg)      */
    const void *bg_rgba_dev;         /* (1.9) NULL, or the solid colour as 4 bytes r,g,b,a in DEVICE memory, read by
                                        the kernel when it runs -- e.g. what mic_median_rgb_dev wrote earlier on the
                                        same stream: fill_solid()'s colour (background_resizing.py:25-33) reaches the
                                        composite that consumes it without a host round trip.  bg_dev must be NULL
                                        and bg_rgba is ignored then.  Opaque colours (a = 255, what mic_median_rgb_dev
                                        writes) take the fast kernels; any other alpha is composited exactly, slowly. */
} mic_job;

int mic_composite_batch(mic_ctx *ctx, int n_atlases, mic_atlas *const *atlases, int n_jobs,
                        const mic_job *jobs, int filter, void *stream);

/* The same batch, resolved once: mic_plan_create does what mic_composite_batch does before its
 * launches (placement -> device layer records, resample tables, scratch for resampled layers,
 * all owned by the plan) and mic_plan_run re-executes ALL the pixel work -- only addresses are
 * kept, never pixels -- writing canvas i to outs[i] (or to the jobs' own out_dev if outs is NULL;
 * out_dev may be NULL at creation when outs will be given).  This is what the refine loop of
 * run_macro_only (macro_placement_test.py:1523-1703) or a batch renderer calls per iteration:
 * per run the host cost is one small job-table upload and the launches.  The atlases must outlive
 * the plan.                                                                                    */
int mic_plan_create(mic_ctx *ctx, int n_atlases, mic_atlas *const *atlases, int n_jobs,
                    const mic_job *jobs, int filter, mic_plan **out);
int mic_plan_run(mic_plan *plan, void *const *outs, void *stream);
int mic_plan_destroy(mic_plan *plan);
/* Resampled layers stay resident (1.9).  A cutout resized to a box's size (compositor.py:20) is a pure function of
 * (cutout, box size, filter) and the atlas is immutable while it exists, so the pixels are kept where they were
 * written: a persistent plan keeps them in its own scratch -- the first mic_plan_run resamples, later runs only
 * composite (the refine loop moving boxes onto other canvases, macro_placement_test.py:1679-1697) -- and transient
 * calls (mic_composite_batch, mic_render, mic_contact_sheet) keep theirs in a per-context cache keyed (atlas, object,
 * box size, filter), MIC_LAYER_CACHE_MB (default 2048, 0 = off).  Results are bit-identical either way.
 * mic_plan_invalidate: the plan's next run resamples again; mic_layer_cache_clear: the context forgets its cached
 * layers (bench.py's cold legs call these; a caller that overwrites an atlas blob it owns in place must, too).       */
int mic_plan_invalidate(mic_plan *plan);
int mic_layer_cache_clear(mic_ctx *ctx);

/* Image.resize((out_w,out_h), filter) of one device RGBA image (compositor.py:20 call shape;
 * also the thumbnail resample of macro_placement_test.py:194).                             */
int mic_resize(mic_ctx *ctx, const void *src_dev, int32_t src_w, int32_t src_h, void *dst_dev,
               int32_t dst_w, int32_t dst_h, int filter, void *stream);

/* ---- background synthesis: replaces background_resizing.py:11-33 ------------------------- */
/* _median_color_nontransparent: per-channel median over pixels with alpha > 0 (all pixels if
 * none), int() truncation.  Result written to host out_rgb after an internal stream sync.   */
int mic_median_rgb(mic_ctx *ctx, const void *rgba_dev, int32_t width, int32_t height,
                   uint8_t out_rgb[3], void *stream);
/* Same, result left on the device as 4 bytes r,g,b,255 at rgba_out_dev (no host sync).      */
int mic_median_rgb_dev(mic_ctx *ctx, const void *rgba_dev, int32_t width, int32_t height,
                       void *rgba_out_dev, void *stream);
/* The same for n images in ONE launch, one copy-back and one stream wait per 16 images: fill_gradient's four
 * 8-pixel edge strips (background_resizing.py:36-57) are four VIEWS of the resident background -- the strided form
 * (stride_bytes = the parent image's row pitch; 0 = packed, width * 4) needs no .contiguous() copy.  out_rgb: n x 3
 * bytes.  Pixels 4-byte aligned, stride a multiple of 4 and >= width * 4.                                     */
typedef struct mic_image_view {
    const void *rgba_dev;
    int32_t width, height;
    int64_t stride_bytes;
} mic_image_view;
int mic_median_rgb_batch(mic_ctx *ctx, int32_t n, const mic_image_view *views, uint8_t *out_rgb, void *stream);
/* Image.new("RGBA", (w,h), colour) on the device (background_resizing.py:32).               */
int mic_fill_solid(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height,
                   const uint8_t rgba[4], void *stream);

/* fill_gradient's pixel loop (background_resizing.py:80-94, uncalled in the reference today): a
 * linear gradient c1 -> c2 along x (vertical = 0) or y (vertical = 1), float32 arithmetic and uint8
 * truncation exactly as NumPy evaluates it there; alpha 255.                                  */
int mic_fill_gradient(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height, const uint8_t c1[3],
                      const uint8_t c2[3], int vertical, void *stream);

/* _save_overlay_debug's drawing (macro_placement_test.py:967-983): a transparent RGBA image of
 * width x height on which ImageDraw.rectangle(box, outline=colour, width=outline_width) is applied
 * per box, in order (later outlines overwrite earlier ones; nothing is blended).
 * boxes_xyxy: n x 4 int32 (x1, y1, x2, y2 as in a placement's "box", inclusive corners, x2 >= x1 and
 * y2 >= y1 or MIC_ERR_INVALID -- ImageDraw raises ValueError there); colours_rgba: n x 4 bytes.    */
int mic_draw_rect_outlines(mic_ctx *ctx, void *out_dev, int32_t width, int32_t height, int32_t n,
                           const int32_t *boxes_xyxy, const uint8_t *colours_rgba, int32_t outline_width,
                           void *stream);

/* ---- contact sheet: replaces _build_labeled_contact_sheet's pixel work ----------------------
 * (macro_placement_test.py:162-242).  The n cutouts `object_ids` of `atlas`, in sheet order (the
 * reference sorts the items by object_id, :173), each reduced by Image.thumbnail((thumb_w, thumb_h),
 * LANCZOS) (:194; a cutout that already fits keeps its size) and alpha-composited centred in the top
 * thumb_h rows of its cell (:216-218) of an opaque white sheet of cols cells per row, each cell
 * thumb_w x (thumb_h + label_h) (:201-207); n == 0 gives one blank cell (:198-199).  Text stays on the
 * host (FreeType): every label arrives as an 8-bit coverage mask, `coverage_host` (w x h bytes, row-major),
 * with its position on the sheet, and is blended as ImageDraw.text(fill=(0,0,0,255)) blends it onto an
 * opaque sheet (:240); a strip is drawn right after the thumbnail of cell `cell` (the reference's order:
 * thumbnail, label, next cell).  out_dev: the sheet, RGBA, mic_contact_sheet_size() pixels.
 * An object id that is not in the atlas is MIC_ERR_INVALID (the reference fails opening its file).   */
typedef struct mic_label_strip {
    int32_t cell;                 /* index into object_ids of the thumbnail this label follows */
    int32_t x, y, w, h;           /* position (may overhang the sheet: clipped) and size of the mask */
    const uint8_t *coverage_host; /* w*h bytes */
} mic_label_strip;
int mic_contact_sheet_size(int32_t n, int32_t thumb_w, int32_t thumb_h, int32_t cols, int32_t label_h,
                           int32_t *sheet_w, int32_t *sheet_h);
int mic_contact_sheet(mic_ctx *ctx, mic_atlas *atlas, int32_t n, const int32_t *object_ids, int32_t thumb_w,
                      int32_t thumb_h, int32_t cols, int32_t label_h, int32_t n_strips,
                      const mic_label_strip *strips, void *out_dev, void *stream);

/* ---- host-side helpers of the PIL-level drop-in (compositor.py:6-22 called with PIL images) ---------------
 * mic_host_rows_solid: is every pixel of rows [y0, y1) of a host RGBA image equal to rgba?  rows_host: one pointer
 * per row (Pillow's own row table), width pixels each.  *is_solid receives 1 or 0.  Exact: every byte is compared.
 * The reference's backgrounds are fill_solid() canvases re-opened from canvas.png every iteration
 * (macro_placement_test.py:1510); a one-colour background is synthesised in-kernel instead of uploaded.      */
int mic_host_rows_solid(const void *const *rows_host, int32_t width, int32_t y0, int32_t y1, const uint8_t rgba[4],
                        int *is_solid);
/* mic_download: enqueue a device -> pinned-host copy on `stream` and mark its end with an event of its own;
 * *ticket names it.  mic_download_wait(ticket) blocks the calling thread until THAT copy has landed (an event wait,
 * not a stream drain; other threads keep enqueueing meanwhile).  A ticket is waited for at most once.
 * bytes == 0: no copy, only the mark -- for a kernel that wrote its canvas straight into pinned host memory (out_dev
 * of a job may be the device-visible address of such a buffer: small canvases skip the copy engine that way).    */
int mic_download(mic_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes, void *stream, int32_t *ticket);
int mic_download_wait(mic_ctx *ctx, int32_t ticket);

/* ---- PNG writer: replaces PIL's encoder behind the reference's artifact saves ---------------------
 * (`canvas_img.save(canvas_path)` macro_placement_test.py:1428-1430, `draft.save(...)` :1513 / :1699, the overlay
 * :1514 / :1700).  Host-only (no context, no device): RGBA8 rows in host memory -> an 8-bit RGBA, non-interlaced
 * PNG.  The bytes differ from Pillow's (other deflate), the decoded pixels are identical.  The image is cut into
 * stripes that are filtered (Sub / Up per row), deflated (LZ77 + dynamic Huffman, stored where that is smaller)
 * and CRC'd on `threads` worker threads (<= 0: one per ~384 KiB of pixels, at most min(cores, 16)); level 0 writes
 * stored blocks only.  stride_bytes: distance between rows (>= width * 4).  The *_rows forms take one pointer per
 * row (a PIL image keeps large images in several blocks).                                                     */
size_t mic_png_bound(int32_t width, int32_t height);
int mic_png_encode(const void *rgba_host, int32_t width, int32_t height, size_t stride_bytes, int level,
                   int threads, void *out, size_t capacity, size_t *out_bytes);
int mic_png_encode_rows(const void *const *rows_host, int32_t width, int32_t height, int level, int threads,
                        void *out, size_t capacity, size_t *out_bytes);
int mic_png_write(const char *path, const void *rgba_host, int32_t width, int32_t height, size_t stride_bytes,
                  int level, int threads);
int mic_png_write_rows(const char *path, const void *const *rows_host, int32_t width, int32_t height, int level,
                       int threads);
/* The same on worker threads of the library's own: returns at once with *job; the PIXELS behind rows_host (the table
 * itself is copied) must stay valid until mic_png_wait(job) has returned the job's status.  A caller that saves several
 * artifacts while it goes on composing (run_macro_only's drafts, overlays, sheet: macro_placement_test.py:1428-1430,
 * 1513-1514, 1699-1700) queues them and waits once at the end; no interpreter thread is involved.            */
int mic_png_write_async(const char *path, const void *const *rows_host, int32_t width, int32_t height, int level,
                        int threads, int64_t *job);
int mic_png_wait(int64_t job);

/* ---- PNG reader: replaces Image.open(path).convert("RGBA") behind load_object_images (compositor.py:25-35, re-run
 * every iteration at macro_placement_test.py:1493, 1679) and the background loader (background_resizing.py:6-8).
 * Host-only.  Takes the PNG kinds this path meets -- 8-bit RGBA / RGB / grey / grey + alpha, palette images of 1-8 bits
 * with or without tRNS, non-interlaced -- and produces exactly the bytes Pillow's convert("RGBA") holds.  Anything else
 * is DECLINED, never guessed: MIC_ERR_UNSUPPORTED for a valid PNG of another kind (16-bit samples, Adam7, a tRNS colour
 * key, APNG), MIC_ERR_FORMAT for anything irregular (bad signature / CRC / Adler-32 / Huffman code, truncated or
 * trailing data); the Python binding hands such a file to Pillow, which decodes it or raises its own error.
 * mic_png_info: size of the image, and whether the decoder takes the file.  mic_png_decode / _rows: into caller memory
 * (stride_bytes >= width * 4; one pointer per row).  mic_png_decode_many: n files on up to `threads` library threads
 * (<= 0: min(n, 8)); status[i] receives each file's result, the call returns the first failure (0: all decoded).
 * mic_png_decode_counts: files this process has decoded / declined so far (a caller's fallback meter).            */
int mic_png_info(const void *png, size_t bytes, int32_t *width, int32_t *height);
int mic_png_decode(const void *png, size_t bytes, void *rgba_out, size_t stride_bytes, int32_t width, int32_t height);
int mic_png_decode_rows(const void *png, size_t bytes, void *const *rows_host, int32_t width, int32_t height);
int mic_png_decode_many(int32_t n, const void *const *pngs, const size_t *bytes, void *const *const *rows_host,
                        const int32_t *widths, const int32_t *heights, int threads, int32_t *status);
int mic_png_decode_counts(uint64_t *decoded, uint64_t *declined);

/* ---- layout: the integer half of render() -------------------------------------------------
 * Flex-DSL JSON text ({"root": {...}}) + cutout sizes + canvas size -> object ids and clamped boxes
 * in depth-first order: _measure_flex_node / _place_flex_container / _clamp_boxes_to_canvas
 * (macro_placement_test.py:637-964) for the well-formed subset of the DSL.  Host-only, no device.
 * MIC_ERR_UNSUPPORTED: the tree uses something whose behaviour hangs on Python's type rules or on
 * the object-level validators (pin / offset_px / stick_to, non-integer numbers, ...); the Python
 * binding then runs its own mirror (flex.py), which returns the result or raises the reference's
 * error.  MIC_ERR_FORMAT: not JSON.  On success *out_count placements were written (capacity is the
 * room in out_ids / out_boxes (4 per placement); too small -> MIC_ERR_INVALID with *out_count set). */
int mic_flex_place(const char *layout_json, size_t len, int n_objects, const int32_t *ids,
                   const int32_t *widths, const int32_t *heights, int32_t canvas_w, int32_t canvas_h,
                   int32_t capacity, int32_t *out_ids, int32_t *out_boxes_xyxy, int32_t *out_count);

/* render(layout_json, objects, canvas) in one call (north_star; macro_placement_test.py:1495-1498 +
 * :1511): the Flex tree is placed from (0,0) over the whole canvas with the atlas' cutout sizes,
 * clamped, and composited -- mic_flex_place + mic_composite_batch without a round trip through the
 * caller.  bg_dev: device RGBA background of width x height, or NULL for the solid colour bg_rgba.
 * Same status codes as mic_flex_place (MIC_ERR_UNSUPPORTED: place it with the Python mirror and call
 * mic_composite_batch); *n_placed (optional) receives the number of placements the tree produced. */
int mic_render(mic_ctx *ctx, mic_atlas *atlas, const char *layout_json, size_t len, int32_t width,
               int32_t height, const void *bg_dev, const uint8_t bg_rgba[4], int filter, void *out_dev,
               void *stream, int32_t *n_placed);
/* (1.9) The same with the canvas described by a mic_job: width, height, bg_dev / bg_rgba / bg_rgba_dev and out_dev are
 * taken from *canvas, its placement fields are ignored (the Flex tree provides the placements).                   */
int mic_render_job(mic_ctx *ctx, mic_atlas *atlas, const char *layout_json, size_t len, const mic_job *canvas,
                   int filter, void *stream, int32_t *n_placed);
/* (1.9) n Flex trees onto n canvases in ONE call and one composite launch per kernel class -- a batch of aspect-ratio
 * variants of one bundle (BASELINE configs[3]; the reference runs run_macro_only once per ratio): layouts[i] (lens[i]
 * bytes of JSON text) is placed over canvases[i] and composited into its out_dev.  Every tree is placed before anything
 * is launched: MIC_ERR_UNSUPPORTED / MIC_ERR_FORMAT name the first tree the native placer leaves to the caller or cannot
 * parse, and nothing has been enqueued then.  n_placed (optional): n counts.                                         */
int mic_render_batch(mic_ctx *ctx, mic_atlas *atlas, int32_t n, const char *const *layouts, const size_t *lens,
                     const mic_job *canvases, int filter, void *stream, int32_t *n_placed);

/* ---- helpers ----------------------------------------------------------------------------- */
/* Pillow Image.thumbnail size rule (macro_placement_test.py:194). */
int mic_thumbnail_size(int32_t w, int32_t h, int32_t req_w, int32_t req_h, int32_t *out_w,
                       int32_t *out_h);
/* Counters of the last mic_composite_batch call on this context (for bench/roofline):
 * algorithmic bytes = 4*W*H per canvas + 4*visible source pixels read.                     */
typedef struct mic_stats {
    uint64_t canvas_pixels;       /* sum of W*H over jobs                                */
    uint64_t layer_pixels;        /* sum of in-canvas layer pixels (alpha-over operations)  */
    uint64_t source_pixels;       /* sum of cutout pixels referenced (each cutout counted per use) */
    uint64_t resampled_layers;    /* layers that needed a resize                           */
    uint64_t identity_layers;
    uint64_t skipped_placements;  /* unknown ids                                           */
    uint64_t composite_blocks;    /* workgroups launched by the composite kernel           */
    uint64_t marched_layers;      /* of resampled_layers: those run by the big-call MFMA resample kernels -- the lane
                                   * kernel (round 5), or the marching kernel under MIC_RS_LANE=0 -- rather than by the
                                   * tile kernel / two-pass fallback (1.5)                                            */
    uint64_t cached_layers;       /* of the call's distinct resampled layers: found in the resident cache (1.9) */
} mic_stats;
int mic_last_stats(const mic_ctx *ctx, mic_stats *out);
int mic_plan_stats(const mic_plan *plan, mic_stats *out);

/* Kernel timing with HIP events recorded on the launch stream, for bench.py's roofline: between
 * mic_profile_begin and mic_profile_end every mic_composite_batch call brackets its composite
 * kernel (and, separately, its resample passes) with an event pair.  mic_profile_end waits for
 * the stream, then reports the number of bracketed calls and the summed durations in ms.       */
int mic_profile_begin(mic_ctx *ctx, int max_calls);
/* Same, bracketing only every `every`-th call (an event pair between two back-to-back kernels costs
 * a few microseconds of idle GPU; sampling keeps the timed loop representative).                 */
int mic_profile_begin_sampled(mic_ctx *ctx, int max_calls, int every);
int mic_profile_end(mic_ctx *ctx, void *stream, int *n_calls, double *composite_ms, double *resample_ms);

#ifdef __cplusplus
}
#endif
#endif /* MIC_H */
