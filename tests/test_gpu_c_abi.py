"""The C ABI used directly (no image_transformation_amd.compositor in between), the way INTEGRATION.md
section 2 shows a foreign binding would: mic_create -> mic_atlas_create (host pointers, atlas-owned
device blob) -> mic_composite_batch / mic_plan_* / mic_resize / mic_median_rgb / mic_fill_solid ->
compare with the oracle.  Also the error contract: negative status + mic_last_error, no crash."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
import oracle  # noqa: E402

P = ctypes.c_void_p
U8P = ctypes.POINTER(ctypes.c_uint8)


@pytest.fixture(scope="module")
def abi():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X")
    from image_transformation_amd import _native
    lib = _native.lib()
    ctx = P()
    assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(ctx)) == 0, lib.mic_last_error()
    yield lib, ctx, _native
    assert lib.mic_destroy(ctx) == 0


def _stream():
    import torch
    return P(torch.cuda.current_stream().cuda_stream)


def _make_atlas(lib, ctx, objs):
    ids = (ctypes.c_int32 * len(objs))(*objs.keys())
    ws = (ctypes.c_int32 * len(objs))(*[a.shape[1] for a in objs.values()])
    hs = (ctypes.c_int32 * len(objs))(*[a.shape[0] for a in objs.values()])
    keep = [np.ascontiguousarray(a) for a in objs.values()]
    ptrs = (P * len(objs))(*[a.ctypes.data for a in keep])  # const uint8_t *const *rgba_host
    atlas = P()
    assert lib.mic_atlas_create(ctx, len(objs), ids, ws, hs, ptrs, ctypes.byref(atlas)) == 0, lib.mic_last_error()
    return atlas


def test_raw_abi_composite_plan_resize_median(abi):
    import torch
    lib, ctx, nat = abi
    rng = np.random.default_rng(4242)
    objs = {7: cases.synthetic.make_cutout(rng, 61, 45, "soft"), 3: cases.synthetic.make_cutout(rng, 33, 80, "binary")}
    atlas = _make_atlas(lib, ctx, objs)
    assert lib.mic_atlas_count(atlas) == 2
    w, h, ptr = ctypes.c_int32(), ctypes.c_int32(), P()
    assert lib.mic_atlas_lookup(atlas, 3, ctypes.byref(w), ctypes.byref(h), ctypes.byref(ptr)) == 0
    assert (w.value, h.value) == (33, 80) and ptr.value
    assert lib.mic_atlas_lookup(atlas, 99, None, None, None) < 0 and b"not in the atlas" in lib.mic_last_error()
    blob, nbytes = P(), ctypes.c_size_t()
    assert lib.mic_atlas_device_blob(atlas, ctypes.byref(blob), ctypes.byref(nbytes)) == 0 and nbytes.value > 61 * 45 * 4

    W, H = 333, 207  # W % 4 != 0: the unaligned kernel class
    placements = [{"object_id": 7, "box": [-10, 5, 51, 50]}, {"object_id": 3, "box": [300, 150, 333 + 20, 150 + 100]},
                  {"object_id": 7, "box": [100, 100, 222, 190]}, {"object_id": 5, "box": [0, 0, 9, 9]}]
    pl = (nat.Placement * len(placements))()
    for i, p in enumerate(placements):
        pl[i].atlas, pl[i].object_id = 0, p["object_id"]
        for k in range(4):
            pl[i].box[k] = p["box"][k]
    out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    job = nat.Job()
    job.width, job.height, job.bg_dev = W, H, None
    for k, v in enumerate((9, 200, 33, 255)):
        job.bg_rgba[k] = v
    job.n_placements, job.placements, job.out_dev = len(placements), pl, out.data_ptr()
    atl = (P * 1)(atlas)
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 0, _stream()) == 0, lib.mic_last_error()
    bg = oracle.fill_solid((W, H), (9, 200, 33, 255))
    want = oracle.composite(bg, objs, placements)
    assert np.array_equal(out.cpu().numpy(), want)
    st = nat.Stats()
    assert lib.mic_last_stats(ctx, ctypes.byref(st)) == 0
    assert st.skipped_placements == 1 and st.resampled_layers == 2 and st.identity_layers == 1

    # the same through a persistent plan, run twice into different canvases
    plan = P()
    job.out_dev = None
    assert lib.mic_plan_create(ctx, 1, atl, 1, ctypes.byref(job), 0, ctypes.byref(plan)) == 0, lib.mic_last_error()
    for _ in range(2):
        o2 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        outs = (P * 1)(o2.data_ptr())
        assert lib.mic_plan_run(plan, outs, _stream()) == 0, lib.mic_last_error()
        assert np.array_equal(o2.cpu().numpy(), want)
    assert lib.mic_plan_run(plan, None, _stream()) < 0 and b"null output" in lib.mic_last_error()
    assert lib.mic_plan_destroy(plan) == 0

    # resize, median, fill
    src = torch.from_numpy(objs[7]).cuda()
    dst = torch.empty((20, 90, 4), dtype=torch.uint8, device="cuda")
    assert lib.mic_resize(ctx, P(src.data_ptr()), 61, 45, P(dst.data_ptr()), 90, 20, 0, _stream()) == 0
    assert np.array_equal(dst.cpu().numpy(), oracle.resize(objs[7], (90, 20)))
    rgb = (ctypes.c_uint8 * 3)()
    assert lib.mic_median_rgb(ctx, P(out.data_ptr()), W, H, rgb, _stream()) == 0
    assert tuple(rgb) == oracle.median_rgb(want)
    col = (ctypes.c_uint8 * 4)(1, 2, 3, 4)
    assert lib.mic_fill_solid(ctx, P(out.data_ptr()), W, H, col, _stream()) == 0
    assert (out.cpu().numpy() == np.array([1, 2, 3, 4], np.uint8)).all()
    assert lib.mic_sync(ctx, _stream()) == 0
    assert lib.mic_atlas_destroy(atlas) == 0


def test_raw_abi_rejects_bad_arguments(abi):
    import torch
    lib, ctx, nat = abi
    objs = {1: np.zeros((4, 4, 4), np.uint8)}
    atlas = _make_atlas(lib, ctx, objs)
    atl = (P * 1)(atlas)
    out = torch.empty((8, 8, 4), dtype=torch.uint8, device="cuda")
    job = nat.Job()
    job.width, job.height, job.out_dev, job.n_placements = 8, 8, out.data_ptr(), 0
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 7, _stream()) < 0 and b"filter" in lib.mic_last_error()
    job.width = 0
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 0, _stream()) < 0 and b"canvas size" in lib.mic_last_error()
    job.width, job.bg_dev = 8, out.data_ptr()
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 0, _stream()) < 0 and b"overlaps the background" in lib.mic_last_error()
    job.bg_dev = None
    pl = (nat.Placement * 1)()
    pl[0].atlas, pl[0].object_id = 3, 1
    job.n_placements, job.placements = 1, pl
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 0, _stream()) < 0 and b"atlas index" in lib.mic_last_error()
    pl[0].atlas = 0
    pl[0].box[0], pl[0].box[1], pl[0].box[2], pl[0].box[3] = 0, 0, 70000, 4
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), 0, _stream()) < 0 and b"too large" in lib.mic_last_error()
    assert lib.mic_composite_batch(None, 1, atl, 1, ctypes.byref(job), 0, _stream()) < 0
    # a corrupt blob is refused, not dereferenced
    junk = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    bad = P()
    assert lib.mic_atlas_from_device_blob(ctx, P(junk.data_ptr()), 4096, None, ctypes.byref(bad)) < 0
    assert b"magic" in lib.mic_last_error()
    assert lib.mic_atlas_destroy(atlas) == 0


def test_raw_abi_render_and_overlay(abi):
    """mic_render (north_star's render(layout_json, objects, canvas) as one ABI call) and
    mic_draw_rect_outlines, against flex.py + the oracle."""
    import json
    import torch
    from image_transformation_amd import flex
    lib, ctx, nat = abi
    rng = np.random.default_rng(99)
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(20, 90)), int(rng.integers(15, 60)), "soft")
            for i in range(5)}
    atlas = _make_atlas(lib, ctx, objs)
    sizes = {k: (v.shape[1], v.shape[0]) for k, v in objs.items()}
    W, H = 333, 211
    layout = {"root": {"type": "flex", "direction": "column", "gap_px": 4, "padding_px": 6, "justify": "center",
                       "children": [{"type": "flex", "direction": "row", "gap_px": 3, "align": "end",
                                     "children": [{"object_id": 1}, {"object_id": "2"}, {"object_id": 77}]},
                                    {"object_id": 3, "padding_px": {"left": 9, "top": 2}},
                                    {"type": "flex", "direction": "row", "justify": "space_between",
                                     "children": [{"object_id": 4}, {"object_id": 5}, {"object_id": 1}]}]}}
    text = json.dumps(layout, indent=1).encode()
    placements = flex.layout_to_placements(layout, sizes, (W, H))
    for bg_kind in ("solid", "image"):
        out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        n = ctypes.c_int32(-1)
        if bg_kind == "solid":
            bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = (12, 200, 99, 255)
            rc = lib.mic_render(ctx, atlas, text, len(text), W, H, None, (ctypes.c_uint8 * 4)(12, 200, 99, 255), 0,
                                P(out.data_ptr()), _stream(), ctypes.byref(n))
        else:
            bg_np = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            bg = torch.from_numpy(bg_np).cuda()
            rc = lib.mic_render(ctx, atlas, text, len(text), W, H, P(bg.data_ptr()), (ctypes.c_uint8 * 4)(), 0,
                                P(out.data_ptr()), _stream(), ctypes.byref(n))
        assert rc == 0, lib.mic_last_error()
        assert n.value == len(placements) == 7
        assert np.array_equal(out.cpu().numpy(), oracle.composite(bg_np, objs, placements))
    # declines what hangs on Python's rules; rejects non-JSON
    odd = json.dumps({"root": {"type": "flex", "direction": "row", "children": [{"object_id": 1, "pin": {"horizontal": "left"}}]}}).encode()
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    rc = lib.mic_render(ctx, atlas, odd, len(odd), W, H, None, (ctypes.c_uint8 * 4)(1, 2, 3, 255), 0, P(out.data_ptr()), _stream(), None)
    assert rc in (0, nat.ERR_UNSUPPORTED)
    assert lib.mic_render(ctx, atlas, b"{nope", 5, W, H, None, (ctypes.c_uint8 * 4)(1, 2, 3, 255), 0, P(out.data_ptr()),
                          _stream(), None) == nat.ERR_FORMAT
    assert lib.mic_render(ctx, atlas, text, len(text), 0, H, None, (ctypes.c_uint8 * 4)(), 0, P(out.data_ptr()), _stream(), None) < 0
    # rectangle outlines
    boxes = np.asarray([[5, 5, 60, 40], [50, 30, 52, 31], [-4, 100, 400, 230], [300, 2, 300, 2]], np.int32)
    cols = np.asarray([[255, 99, 71, 180], [135, 206, 235, 180], [60, 179, 113, 180], [1, 2, 3, 4]], np.uint8)
    ov = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    assert lib.mic_draw_rect_outlines(ctx, P(ov.data_ptr()), W, H, 4, boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                      cols.ctypes.data_as(U8P), 3, _stream()) == 0, lib.mic_last_error()
    assert np.array_equal(ov.cpu().numpy(), oracle.rect_outlines((W, H), boxes, cols, 3))
    bad = np.asarray([[9, 9, 3, 12]], np.int32)
    assert lib.mic_draw_rect_outlines(ctx, P(ov.data_ptr()), W, H, 1, bad.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                      cols.ctypes.data_as(U8P), 3, _stream()) < 0
    assert b"x1 must be greater than or equal to x0" in lib.mic_last_error()
    assert lib.mic_atlas_destroy(atlas) == 0


def test_fragment_cache_eviction_keeps_plans_valid():
    """The resample fragment tables are cached per (in, out, filter) axis with a byte cap; entries are
    ref-counted, so evicting them must not disturb a persistent plan that still points into them."""
    import os
    import torch
    from image_transformation_amd import _native
    lib = _native.lib()
    os.environ["MIC_FRAG_CACHE_MB"] = "1"
    try:
        ctx = P()
        assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(ctx)) == 0, lib.mic_last_error()
    finally:
        del os.environ["MIC_FRAG_CACHE_MB"]
    rng = np.random.default_rng(31)
    objs = {1: cases.synthetic.make_cutout(rng, 300, 200, "soft")}
    atlas = _make_atlas(lib, ctx, objs)
    W, H = 640, 480
    pl = (_native.Placement * 1)(_native.Placement(0, 1, (ctypes.c_int32 * 4)(10, 20, 10 + 411, 20 + 333)))
    job = _native.Job()
    job.width, job.height, job.bg_dev, job.n_placements = W, H, None, 1
    for k, v in enumerate((9, 8, 7, 255)):
        job.bg_rgba[k] = v
    job.placements = ctypes.cast(pl, ctypes.POINTER(_native.Placement))
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    job.out_dev = out.data_ptr()
    plan = P()
    atl = (P * 1)(atlas)
    assert lib.mic_plan_create(ctx, 1, atl, 1, ctypes.byref(job), 0, ctypes.byref(plan)) == 0, lib.mic_last_error()
    bg = np.empty((H, W, 4), np.uint8); bg[:] = (9, 8, 7, 255)
    want = oracle.composite(bg, objs, [{"object_id": 1, "box": [10, 20, 421, 353]}])
    # churn through many other sizes: far more than 1 MB of tables, so the plan's entries leave the cache
    src = torch.from_numpy(objs[1]).cuda()
    for k in range(60):
        dw, dh = 150 + 7 * k, 120 + 5 * k
        dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
        assert lib.mic_resize(ctx, P(src.data_ptr()), 300, 200, P(dst.data_ptr()), dw, dh, 0, _stream()) == 0, lib.mic_last_error()
        if k % 20 == 0:
            assert np.array_equal(dst.cpu().numpy(), oracle.resize(objs[1], (dw, dh)))
    for _ in range(2):
        out.zero_()
        assert lib.mic_plan_run(plan, None, _stream()) == 0, lib.mic_last_error()
        assert np.array_equal(out.cpu().numpy(), want)
    # sampled event brackets: every 3rd of 9 runs is timed
    assert lib.mic_profile_begin_sampled(ctx, 10, 3) == 0, lib.mic_last_error()
    for _ in range(9):
        assert lib.mic_plan_run(plan, None, _stream()) == 0
    n, comp, res = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
    assert lib.mic_profile_end(ctx, _stream(), ctypes.byref(n), ctypes.byref(comp), ctypes.byref(res)) == 0
    assert n.value == 3 and comp.value > 0 and res.value > 0
    assert lib.mic_profile_begin_sampled(ctx, 10, 0) < 0
    assert lib.mic_plan_destroy(plan) == 0 and lib.mic_atlas_destroy(atlas) == 0 and lib.mic_destroy(ctx) == 0


def test_raw_abi_contact_sheet(abi, golden_dir):
    """mic_contact_sheet through the raw ABI against the sheets captured from the reference
    (_build_labeled_contact_sheet, macro_placement_test.py:162-242): the thumbnail band of every cell bit-exact;
    a label strip handed over as a coverage mask lands where it is put, blended as ImageDraw.text blends it
    on the opaque sheet; unknown ids and the empty sheet follow the header's contract."""
    import json
    import os
    import torch
    from PIL import Image
    lib, ctx, nat = abi
    arrays = np.load(os.path.join(golden_dir, "contact_sheet.npz"))
    for bundle in ("squarespace", "audio_book"):
        rj = os.path.join(cases.BUNDLE_DIR, bundle, "results.json")
        with open(rj, encoding="utf-8") as f:
            items = sorted(json.load(f), key=lambda it: int(it["object_id"]))
        objs = {int(it["object_id"]): np.array(Image.open(os.path.join(cases.BUNDLE_DIR, bundle, it["filename"])).convert("RGBA"))
                for it in items}
        atlas = _make_atlas(lib, ctx, objs)
        ids = (ctypes.c_int32 * len(objs))(*objs.keys())
        W, H = ctypes.c_int32(), ctypes.c_int32()
        assert lib.mic_contact_sheet_size(len(objs), 256, 256, 4, 72, ctypes.byref(W), ctypes.byref(H)) == 0
        want = arrays[f"{bundle}_sheet"]
        assert (H.value, W.value) == want.shape[:2]
        out = torch.zeros((H.value, W.value, 4), dtype=torch.uint8, device="cuda")
        # one label: a 5 x 3 coverage mask under the second thumbnail
        mask = np.array([[0, 64, 128, 255, 255], [255, 255, 0, 1, 254], [7, 7, 7, 7, 7]], np.uint8)
        strip = nat.LabelStrip()
        strip.cell, strip.x, strip.y, strip.w, strip.h = 1, 300, 270, 5, 3
        strip.coverage_host = mask.ctypes.data
        assert lib.mic_contact_sheet(ctx, atlas, len(objs), ids, 256, 256, 4, 72, 1, ctypes.byref(strip),
                                     P(out.data_ptr()), _stream()) == 0, lib.mic_last_error()
        got = out.cpu().numpy()
        assert np.array_equal(got[:256], want[:256])  # LANCZOS thumbnails + alpha-over on white, as the reference's
        band = np.full((72, W.value, 4), 255, np.uint8)  # the label band: white, except the strip
        m = mask.astype(np.uint32)
        t = 255 * (255 - m) + 128  # black ink over white, Pillow's blend: div255(dst * (255 - m) + 128)
        band[270 - 256:273 - 256, 300:305, :3] = (((t >> 8) + t) >> 8)[:, :, None]
        assert np.array_equal(got[256:], band)
        # an id the atlas does not hold
        bad = (ctypes.c_int32 * 1)(99)
        assert lib.mic_contact_sheet(ctx, atlas, 1, bad, 256, 256, 4, 72, 0, None, P(out.data_ptr()), _stream()) < 0
        assert b"not in the atlas" in lib.mic_last_error()
        # the empty sheet: one blank cell
        assert lib.mic_contact_sheet_size(0, 256, 256, 4, 72, ctypes.byref(W), ctypes.byref(H)) == 0
        assert (W.value, H.value) == (256, 328)
        blank = torch.zeros((328, 256, 4), dtype=torch.uint8, device="cuda")
        assert lib.mic_contact_sheet(ctx, atlas, 0, None, 256, 256, 4, 72, 0, None, P(blank.data_ptr()), _stream()) == 0
        assert bool((blank == 255).all())
        assert lib.mic_atlas_destroy(atlas) == 0


def test_raw_abi_edges_hardened_in_round_2(abi):
    """Edges of the ABI tightened in round 2, one assertion each: an output that overlaps its background anywhere
    (not just equal pointers) is refused; so are two canvases of one launch that overlap each other; a plan's cached
    job table keeps refusing what it refused; work enqueued on a caller's (non-default) stream -- including the
    atlas' planar copies, which used to be built on the NULL stream -- is ordered on that stream; plans and atlases
    may be destroyed in any order."""
    import torch
    lib, ctx, nat = abi
    rng = np.random.default_rng(3)
    objs = {1: rng.integers(0, 256, (40, 60, 4), dtype=np.uint8), 2: rng.integers(0, 256, (33, 21, 4), dtype=np.uint8)}
    atlas = _make_atlas(lib, ctx, objs)
    atl = (P * 1)(atlas)
    W, H = 128, 64
    buf = torch.zeros(2 * W * H * 4, dtype=torch.uint8, device="cuda")
    pl = (nat.Placement * 1)()
    pl[0].atlas, pl[0].object_id = 0, 1
    for k, v in enumerate((3, 3, 63, 43)):
        pl[0].box[k] = v

    def job(bg_off, out_off):
        j = nat.Job()
        j.width, j.height, j.n_placements, j.placements = W, H, 1, pl
        j.bg_dev = buf.data_ptr() + bg_off if bg_off is not None else None
        j.bg_rgba[3] = 255
        j.out_dev = buf.data_ptr() + out_off
        return j

    # output = background shifted by one row: overlap, not equality
    j = job(0, W * 4)
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(j), 0, _stream()) < 0
    assert b"overlaps the background" in lib.mic_last_error()
    # two canvases of one launch, the second starting inside the first
    jobs = (nat.Job * 2)(job(None, 0), job(None, W * H * 2))
    assert lib.mic_composite_batch(ctx, 1, atl, 2, jobs, 0, _stream()) < 0
    assert b"overlap" in lib.mic_last_error()
    # a persistent plan refuses the bad output set every time it is offered (cached or not), and runs a good one
    plan = P()
    j = job(0, 0)
    j.out_dev = None
    assert lib.mic_plan_create(ctx, 1, atl, 1, ctypes.byref(j), 0, ctypes.byref(plan)) == 0, lib.mic_last_error()
    bad = (P * 1)(buf.data_ptr() + 16)
    good = (P * 1)(buf.data_ptr() + W * H * 4)
    for _ in range(3):
        assert lib.mic_plan_run(plan, bad, _stream()) < 0
        assert lib.mic_plan_run(plan, good, _stream()) == 0, lib.mic_last_error()
    # a resampling call on a side stream: planar copy + resample + composite are all ordered on that stream
    side = torch.cuda.Stream()
    out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    pl2 = (nat.Placement * 1)()
    pl2[0].atlas, pl2[0].object_id = 0, 2
    for k, v in enumerate((10, 5, 10 + 50, 5 + 44)):  # 21x33 -> 50x44: LANCZOS
        pl2[0].box[k] = v
    j2 = nat.Job()
    j2.width, j2.height, j2.n_placements, j2.placements, j2.out_dev = W, H, 1, pl2, out.data_ptr()
    for k, v in enumerate((9, 200, 33, 255)):
        j2.bg_rgba[k] = v
    with torch.cuda.stream(side):
        assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(j2), 0, P(side.cuda_stream)) == 0, lib.mic_last_error()
    side.synchronize()
    want = oracle.composite(oracle.fill_solid((W, H), (9, 200, 33, 255)), objs, [{"object_id": 2, "box": [10, 5, 60, 49]}])
    assert np.array_equal(out.cpu().numpy(), want)
    # destroy order: the atlas first, then the plan that was made from it
    assert lib.mic_atlas_destroy(atlas) == 0
    assert lib.mic_plan_destroy(plan) == 0


def test_pack_blob_into_pinned_memory_is_byte_identical():
    """Atlas() packs the cutouts straight into a pinned block, which the host allocator hands out with whatever a
    previous user left in it: header, table, every gap between the 256-byte aligned images and the guard band at
    the end must come out as zeros, exactly as in the pageable (np.zeros) form that rank 0 broadcasts."""
    import torch
    from PIL import Image
    from image_transformation_amd.compositor import Atlas, pack_blob
    rng = np.random.default_rng(12)
    objs = {7: cases.synthetic.make_cutout(rng, 61, 45, "soft"), 3: cases.synthetic.make_cutout(rng, 33, 80, "binary"),
            11: Image.fromarray(cases.synthetic.make_cutout(rng, 5, 3, "soft"), "RGBA"), 2: cases.synthetic.make_cutout(rng, 1, 1, "soft")}
    want = pack_blob(objs)
    assert not want.is_pinned()
    for _ in range(3):
        junk = torch.empty(want.numel(), dtype=torch.uint8, pin_memory=True)
        junk.fill_(0xAB)
        del junk  # back to the caching host allocator: the next block of this size is this one
        got = pack_blob(objs, pin=True)
        assert got.is_pinned() and torch.equal(got, want)
    atlas = Atlas(objs)
    assert torch.equal(atlas.blob.cpu(), want)
    assert atlas[11].size == (5, 3) and len(atlas) == 4


def test_selftest_canary(abi):
    """mic_selftest: the known-answer kernel behind the v_ashr_pk_u8_i32 workaround (kernels_resample.hip clip8 / clip8x4)
    agrees with host arithmetic on this compiler build."""
    lib, ctx = abi[0], abi[1]
    assert lib.mic_selftest(ctx, _stream()) == 0, lib.mic_last_error()


def test_raw_abi_render_batch(abi):
    """mic_render_batch (1.9): n Flex trees onto n canvases in one call -- different sizes, a solid colour on the host, one
    in device memory, a background image -- against flex.py + the oracle; a tree the native placer declines makes the
    whole call return MIC_ERR_UNSUPPORTED with nothing enqueued; the Python render_batch() falls back per layout then."""
    import json
    import torch
    from image_transformation_amd import flex
    from image_transformation_amd.compositor import Atlas, SolidCanvas, render_batch
    lib, ctx, nat = abi
    rng = np.random.default_rng(515)
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(20, 90)), int(rng.integers(15, 60)), "soft") for i in range(6)}
    atlas = _make_atlas(lib, ctx, objs)
    sizes = {k: (v.shape[1], v.shape[0]) for k, v in objs.items()}
    dims = [(333, 211), (640, 200), (256, 256)]
    layouts = [{"root": {"type": "flex", "direction": d, "gap_px": g, "justify": j, "align": a,
                         "children": [{"object_id": k} for k in order]}}
               for d, g, j, a, order in (("row", 4, "center", "end", [1, 2, 3]), ("row", 0, "space_between", "start", [6, 5, 4, 3, 2, 1]),
                                         ("column", 7, "space_around", "center", [2, "4", 99, 6]))]
    texts = [json.dumps(l).encode() for l in layouts]
    word = torch.tensor((9, 8, 7, 255), dtype=torch.uint8, device="cuda")
    bg_img = rng.integers(0, 256, (dims[2][1], dims[2][0], 4), dtype=np.uint8)
    bg_dev = torch.from_numpy(bg_img).cuda()
    jobs = (nat.Job * 3)()
    outs, bgs = [], []
    for i, (W, H) in enumerate(dims):
        out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        outs.append(out)
        jobs[i].width, jobs[i].height, jobs[i].out_dev = W, H, out.data_ptr()
        b = np.empty((H, W, 4), np.uint8)
        if i == 0:
            for c, v in enumerate((200, 100, 50, 255)):
                jobs[i].bg_rgba[c] = v
            b[:] = (200, 100, 50, 255)
        elif i == 1:
            jobs[i].bg_rgba_dev = word.data_ptr()
            b[:] = (9, 8, 7, 255)
        else:
            jobs[i].bg_dev = bg_dev.data_ptr()
            b = bg_img
        bgs.append(b)
    tarr = (ctypes.c_char_p * 3)(*texts)
    lens = (ctypes.c_size_t * 3)(*[len(t) for t in texts])
    placed = (ctypes.c_int32 * 3)()
    assert lib.mic_render_batch(ctx, atlas, 3, tarr, lens, jobs, 0, _stream(), placed) == 0, lib.mic_last_error()
    for i, (W, H) in enumerate(dims):
        pl = flex.layout_to_placements(layouts[i], sizes, (W, H))
        assert placed[i] == len(pl)
        assert np.array_equal(outs[i].cpu().numpy(), oracle.composite(bgs[i], objs, pl)), i
    # one tree the native placer leaves to the Python mirror: the whole call declines, nothing is written
    # (int("1_0") == 10 in Python: a gap the reference accepts and the native placer leaves to the mirror)
    odd_tree = {"root": {"type": "flex", "direction": "row", "gap_px": "1_0", "children": [{"object_id": 1}, {"object_id": 2}]}}
    odd = json.dumps(odd_tree).encode()
    rc_one = lib.mic_render_job(ctx, atlas, odd, len(odd), ctypes.byref(jobs[0]), 0, _stream(), None)
    assert rc_one == nat.ERR_UNSUPPORTED, rc_one
    if rc_one == nat.ERR_UNSUPPORTED:
        for o in outs:
            o.zero_()
        tarr2 = (ctypes.c_char_p * 3)(texts[0], odd, texts[2])
        lens2 = (ctypes.c_size_t * 3)(len(texts[0]), len(odd), len(texts[2]))
        assert lib.mic_render_batch(ctx, atlas, 3, tarr2, lens2, jobs, 0, _stream(), None) == nat.ERR_UNSUPPORTED
        torch.cuda.synchronize()
        assert not any(bool(o.any()) for o in outs)
    assert lib.mic_render_batch(ctx, atlas, 3, (ctypes.c_char_p * 3)(texts[0], b"{nope", texts[2]), lens, jobs, 0, _stream(), None) == nat.ERR_FORMAT
    assert lib.mic_atlas_destroy(atlas) == 0
    # the package's render_batch: JSON texts, dicts and a layout that needs the Python mirror (pin) in one batch
    a2 = Atlas(objs)
    mixed = [texts[0].decode(), layouts[1], odd_tree]
    cvs = [SolidCanvas(dims[0], (200, 100, 50, 255)), SolidCanvas(dims[1], colour_dev=word), bg_dev]
    got = render_batch(mixed, a2, cvs)
    for i, (W, H) in enumerate(dims):
        lay = json.loads(mixed[i]) if isinstance(mixed[i], str) else mixed[i]
        pl = flex.layout_to_placements(lay, sizes, (W, H))
        assert np.array_equal(got[i].cpu().numpy(), oracle.composite(bgs[i], objs, pl)), ("render_batch", i)
