"""GPU parity: the HIP path (through the C ABI, via the package's reference-shaped call surface)
against the CPU oracle on the same seeded inputs and against the fixtures captured from the
reference.  Bar: bit-exact (8-bit integer arithmetic), which is inside north_star's +-1 LSB.

Nothing here reads /root/reference; inputs are regenerated from seeds or come from
tests/golden/bundles (data files).
"""
import ctypes
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
import oracle  # noqa: E402  (the checker)


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; none is visible")
    from image_transformation_amd import _native
    ctx = _native.context()  # raises loudly if libmic.so is missing
    return ctx


def _load(golden_dir, stem):
    with open(os.path.join(golden_dir, stem + ".json"), encoding="utf-8") as f:
        meta = json.load(f)
    p = os.path.join(golden_dir, stem + ".npz")
    return meta, (np.load(p) if os.path.exists(p) else None)


def _img(a):
    from PIL import Image
    return Image.fromarray(np.ascontiguousarray(a), "RGBA")


def _arr(im):
    return np.array(im)


def _maxdiff(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()) if a.shape == b.shape else -1


# ------------------------------------------------------------------------------------------ composite()
def test_composite_golden_cases(gpu, golden_dir):
    from image_transformation_amd.compositor import composite
    meta, arrays = _load(golden_dir, "composite")
    by_name = {c["name"]: c for c in cases.composite_cases()}
    for row in meta["cases"]:
        c = by_name[row["name"]]
        bg = _img(c["bg"])
        before = bg.tobytes()
        got = _arr(composite(bg, {k: _img(v) for k, v in c["objects"].items()}, c["placements"]))
        assert bg.tobytes() == before, "background must not be modified (compositor.py:11)"
        want = arrays[row["name"]]
        assert np.array_equal(got, want), (row["name"], _maxdiff(got, want))
        assert np.array_equal(got, oracle.composite(c["bg"], c["objects"], c["placements"]))


def test_reference_unit_test_restated(gpu):
    """tests/test_compositor.py:5-11 of the reference, against this package."""
    from PIL import Image
    from image_transformation_amd.compositor import composite
    bg = Image.new("RGBA", (10, 10), (255, 0, 0, 255))
    obj = Image.new("RGBA", (2, 2), (0, 255, 0, 255))
    out = composite(bg, {1: obj}, [{"object_id": 1, "box": [4, 4, 6, 6]}])
    assert out.getpixel((4, 4))[:3] == (0, 255, 0)
    assert out.mode == "RGBA" and out.size == (10, 10)


def test_composite_error_behaviour(gpu):
    from PIL import Image
    from image_transformation_amd.compositor import composite
    bg = Image.new("RGBA", (8, 8), (1, 2, 3, 255))
    obj = {1: Image.new("RGBA", (2, 2), (0, 255, 0, 255))}
    with pytest.raises(ValueError):
        composite(bg, obj, [{"object_id": "one", "box": [0, 0, 2, 2]}])       # int() of a non-number
    with pytest.raises(ValueError):
        composite(bg, obj, [{"object_id": 1, "box": [0, 0, 2]}])               # unpack
    with pytest.raises((ValueError, TypeError)):
        composite(bg, obj, [{"object_id": 1, "box": [0, 0, None, 2]}])
    with pytest.raises(KeyError):
        composite(bg, obj, [{"box": [0, 0, 2, 2]}])
    with pytest.raises(ValueError, match="wrong mode"):
        composite(bg.convert("RGB"), obj, [{"object_id": 1, "box": [0, 0, 2, 2]}])
    with pytest.raises(ValueError, match="wrong mode"):
        composite(bg, {1: obj[1].convert("RGB")}, [{"object_id": 1, "box": [0, 0, 2, 2]}])
    out = composite(bg, obj, [{"object_id": 5, "box": [0, 0, 2, 2]}])          # unknown id: skipped
    assert out.tobytes() == bg.tobytes() and out is not bg


# ------------------------------------------------------------------------------------------ resize
def test_resize_golden_cases(gpu, golden_dir):
    import ctypes
    import torch
    from image_transformation_amd import _native
    meta, arrays = _load(golden_dir, "resize")
    lib = _native.lib()
    for i, row in enumerate(meta["cases"]):
        c = cases.resize_case(i)
        src = torch.from_numpy(c["src"]).to(gpu.torch_device)
        for filt, key in ((_native.LANCZOS, row["name"]), (_native.BILINEAR, row["name"] + "_bilinear")):
            dw, dh = c["size"]
            dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device=gpu.torch_device)
            _native.check(lib.mic_resize(gpu.handle, ctypes.c_void_p(src.data_ptr()), c["src"].shape[1],
                                         c["src"].shape[0], ctypes.c_void_p(dst.data_ptr()), dw, dh, filt,
                                         ctypes.c_void_p(gpu.stream_ptr())))
            got = dst.cpu().numpy()
            assert np.array_equal(got, arrays[key]), (key, _maxdiff(got, arrays[key]))


def test_resize_random_shapes_vs_oracle(gpu):
    """The matrix-core resample kernel against the (golden-pinned) oracle on shapes that reach its
    corners: many tiles per layer, windows cut by the right edge (width % 4 != 0), windows wider than
    one 64-sample chunk (shrinks below 1/3), the LDS-heavy tiles, one-axis resizes (identity table on
    the other axis), binary alpha (select path), tiny sources, and the two-pass fallback (extreme
    shrink)."""
    import ctypes
    import torch
    from image_transformation_amd import _native
    lib = _native.lib()
    rng = np.random.default_rng(77)
    shapes = [((301, 203), (457, 311)), ((457, 311), (301, 203)), ((643, 97), (211, 97)), ((97, 643), (97, 211)),
              ((1000, 800), (256, 205)), ((513, 259), (1026, 518)), ((130, 70), (1301, 707)), ((66, 66), (67, 65)),
              ((799, 601), (160, 121)), ((1201, 5), (300, 5)), ((3, 900), (3, 100)), ((2, 2), (97, 33)),
              ((1, 7), (50, 3)), ((4000, 16), (40, 16)), ((16, 4000), (17, 33)), ((257, 255), (255, 257)),
              ((19, 23), (640, 480)), ((1023, 767), (511, 383)),
              ((2000, 1500), (256, 192)), ((1501, 1999), (101, 123)), ((3000, 40), (97, 40)),  # deep shrinks: many chunks
              ((59, 450), (114, 25)), ((450, 59), (25, 114)), ((64, 1000), (64, 48))]  # rings of 20-32 slots (found by the soak)
    for i, ((sw, sh), (dw, dh)) in enumerate(shapes):
        src = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
        if i % 3 == 1:    # binary alpha, like the reference's bundles
            src[:, :, 3] = np.where(rng.random((sh, sw)) < 0.45, 0, 255)
        elif i % 3 == 2:  # mostly opaque with soft edges
            src[:, :, 3] = np.where(rng.random((sh, sw)) < 0.8, 255, src[:, :, 3])
        dev = torch.from_numpy(src).to(gpu.torch_device)
        for filt in (_native.LANCZOS, _native.BILINEAR):
            dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device=gpu.torch_device)
            _native.check(lib.mic_resize(gpu.handle, ctypes.c_void_p(dev.data_ptr()), sw, sh,
                                         ctypes.c_void_p(dst.data_ptr()), dw, dh, filt, ctypes.c_void_p(gpu.stream_ptr())))
            want = oracle.resize(src, (dw, dh), filt)
            got = dst.cpu().numpy()
            assert np.array_equal(got, want), ((sw, sh), (dw, dh), filt, _maxdiff(got, want))


def test_resize_transparent_margins_vs_oracle(gpu):
    """Cutout-shaped sources (an opaque or soft blob inside wide transparent margins, like the reference's
    bundles): the marching kernel skips the horizontal pass of source bands without a pixel of alpha > 0
    and stores output tiles that only see such bands as transparent black -- including sources whose
    transparent pixels carry colour (premultiplying zeroes it), a fully transparent source, and blobs
    that start/end inside a 16-row band."""
    import ctypes
    import torch
    from image_transformation_amd import _native
    lib = _native.lib()
    rng = np.random.default_rng(99)
    cases_ = [((640, 480), (400, 300), (200, 150, 90, 60)), ((640, 480), (961, 719), (330, 250, 40, 200)),
              ((500, 700), (250, 349), (250, 100, 200, 17)), ((300, 300), (300, 450), (150, 150, 10, 10)),
              ((1200, 900), (300, 225), (600, 450, 250, 180)), ((257, 513), (500, 1000), (128, 40, 60, 33)),
              ((400, 400), (200, 200), None)]
    for (sw, sh), (dw, dh), blob in cases_:
        src = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)  # colour everywhere, also under alpha 0
        if blob is None:
            src[:, :, 3] = 0
        else:
            cx, cy, rx, ry = blob
            yy, xx = np.mgrid[0:sh, 0:sw]
            inside = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
            soft = rng.integers(1, 255, (sh, sw), dtype=np.uint8)
            src[:, :, 3] = np.where(inside, np.where(rng.random((sh, sw)) < 0.7, 255, soft), 0)
        dev = torch.from_numpy(src).to(gpu.torch_device)
        for filt in (_native.LANCZOS, _native.BILINEAR):
            dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device=gpu.torch_device)
            _native.check(lib.mic_resize(gpu.handle, ctypes.c_void_p(dev.data_ptr()), sw, sh,
                                         ctypes.c_void_p(dst.data_ptr()), dw, dh, filt, ctypes.c_void_p(gpu.stream_ptr())))
            want = oracle.resize(src, (dw, dh), filt)
            got = dst.cpu().numpy()
            assert np.array_equal(got, want), ((sw, sh), (dw, dh), blob, filt, _maxdiff(got, want))


# ------------------------------------------------------------------------------------------ median / fill_solid
def test_median_cases(gpu, golden_dir):
    from image_transformation_amd.background_resizing import _median_color_nontransparent
    meta, _ = _load(golden_dir, "median")
    rows = {r["name"]: r for r in meta["cases"]}
    for i in range(cases.N_MEDIAN):
        c = cases.median_case(i)
        assert list(_median_color_nontransparent(_img(c["rgba"]))) == rows[c["name"]]["rgb"], c["name"]


def test_median_large_random_vs_oracle(gpu):
    import torch
    from image_transformation_amd.background_resizing import median_color_device
    rng = np.random.default_rng(7)
    # run back to back on one context: the kernel's last block must leave its scratch zeroed
    for (h, w, mode) in [(2160, 3840, "noise"), (1081, 1923, "flat"), (777, 1234, "sparse"),
                         (333, 1001, "clear_noise"), (4320, 7680, "clear_flat"), (1080, 1920, "flat_mixed_alpha"),
                         (1, 3, "noise"), (2161, 3841, "blocks"), (2160, 3840, "one_colour"), (1500, 1999, "two_colours"),
                         (2161, 3841, "one_colour_clear"), (600, 1024, "runs")]:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if mode.startswith("one_colour"):  # every chunk of 256 pixels is one colour: the whole-chunk shortcut, both sets
            a[:] = (220, 238, 245, 0 if mode.endswith("clear") else 255)
        if mode == "two_colours":  # the boundary falls inside chunks (1999 px rows), one side transparent
            a[:] = (10, 200, 30, 255)
            a[:, 1000:] = (250, 3, 77, 0)
            a[700:] = (9, 9, 9, 200)
        if mode == "runs":  # runs of 256 equal pixels, chunk-aligned (1024 px rows), next to noise chunks
            flat = rng.integers(0, 256, (h, w // 256, 1, 4), dtype=np.uint8)
            a = np.where((np.arange(w // 256) % 3 != 0)[None, :, None, None], np.repeat(flat, 256, axis=2),
                         a.reshape(h, w // 256, 256, 4)).reshape(h, w, 4)
        if mode in ("flat", "clear_flat", "flat_mixed_alpha"):  # mostly one colour: the wave-aggregation path
            a[:, :, :3] = (38, 73, 115)
            a[::7, ::5, :3] = rng.integers(0, 256, a[::7, ::5, :3].shape, dtype=np.uint8)
        if mode == "sparse":
            a[:, :, 3] = np.where(rng.random((h, w)) < 0.01, 255, 0)
        if mode.startswith("clear"):  # no pixel with alpha > 0: the all-pixels fallback
            a[:, :, 3] = 0
        if mode == "flat_mixed_alpha":  # same colour, alternating alpha class inside every wave
            a[:, ::3, 3] = 0
        if mode == "blocks":  # flat 64x64 blocks with different colours and alpha classes
            a[:] = np.repeat(np.repeat(a[::64, ::64], 64, axis=0), 64, axis=1)[:h, :w]
            a[:, :, 3] = np.where(a[:, :, 3] < 100, 0, a[:, :, 3])
        assert median_color_device(torch.from_numpy(a).to(gpu.torch_device)) == oracle.median_rgb(a), mode


def test_fill_solid_bundles(gpu, golden_dir):
    from image_transformation_amd.background_resizing import fill_solid, solid_canvas
    meta, _ = _load(golden_dir, "median")
    rows = {r["name"]: r for r in meta["cases"]}
    for b in cases.BUNDLES:
        path = os.path.join(cases.BUNDLE_DIR, b, "background.png")
        img = fill_solid(path, (33, 17))
        assert img.mode == "RGBA" and img.size == (33, 17)
        a = _arr(img)
        want = rows[f"bundle_{b}"]["rgb"] + [255]
        assert (a == np.asarray(want, np.uint8)).all()
        assert list(solid_canvas(path, (5, 5)).rgba) == want
    with pytest.raises(FileNotFoundError):
        fill_solid(os.path.join(cases.BUNDLE_DIR, "nope.png"), (4, 4))


# ------------------------------------------------------------------------------------------ render(): C1 + App. A.6
def test_render_bundles_c1(gpu, golden_dir):
    """BASELINE.json configs[0]: squarespace 1:1 (and the other bundle/ratio rows of App. A.6)."""
    from image_transformation_amd.background_resizing import fill_solid, solid_canvas
    from image_transformation_amd.compositor import composite, load_object_images, render
    from image_transformation_amd.layout_constraints import compute_canvas_size
    meta, arrays = _load(golden_dir, "bundles")
    for row in meta["cases"]:
        base = os.path.join(cases.BUNDLE_DIR, row["bundle"])
        objects = load_object_images(os.path.join(base, "results.json"))
        W, H = row["canvas"]
        if "ratio" in row:
            from PIL import Image
            with Image.open(os.path.join(base, "background.png")) as im:
                assert compute_canvas_size(im.size, row["ratio"], quiet=True) == (W, H)
        bgp = os.path.join(base, "background.png")
        if "placements" in row:
            out = composite(fill_solid(bgp, (W, H)), objects, row["placements"])
        else:
            keep = {c["object_id"] for c in row["layout"]["root"]["children"]}
            sub = {k: v for k, v in objects.items() if k in keep}
            out = render(row["layout"], sub, solid_canvas(bgp, (W, H)))
            out2 = render(row["layout"], sub, fill_solid(bgp, (W, H)))  # image background path
            assert out.tobytes() == out2.tobytes()
        got = _arr(out)
        assert cases.sha16(got) == row["sha16"], row["name"]
        if row["name"] in arrays.files:
            assert np.array_equal(got, arrays[row["name"]])
        if "centre_px" in row:
            assert [int(v) for v in got[H // 2, W // 2]] == row["centre_px"]


# ------------------------------------------------------------------------------------------ contact sheet
def test_contact_sheet(gpu, golden_dir):
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet, thumbnail_size
    meta, arrays = _load(golden_dir, "contact_sheet")
    for r in meta["thumbnail_sizes"]:
        assert list(thumbnail_size(r["src"], (256, 256))) == r["size"]
    for row in meta["cases"]:
        if row.get("bundle") is None:
            continue
        b = row["bundle"]
        rj = os.path.join(cases.BUNDLE_DIR, b, "results.json")
        sheet = _arr(build_labeled_contact_sheet(os.path.join(cases.BUNDLE_DIR, b, "objects"), rj))
        want = arrays[f"{b}_sheet"]
        assert list(sheet.shape) == row["sheet_shape"]
        # thumbnail area of every cell: bit-exact (LANCZOS thumbnail + alpha-over on white)
        assert np.array_equal(sheet[:256], want[:256]), _maxdiff(sheet[:256], want[:256])
        # label band: glyphs come from the host's FreeType; identical when the font stack matches
        if cases.sha16(sheet) != row["sheet_sha16"]:
            diff = (sheet != want).any(axis=2)
            assert not diff[:256].any()
            assert diff.sum() < 0.02 * diff.size, "label band differs by more than glyph rasterisation"
    # empty bundle -> one blank cell (macro_placement_test.py:198-199)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "results.json"), "w") as f:
            f.write("[]")
        blank = build_labeled_contact_sheet(d, os.path.join(d, "results.json"))
        assert blank.size == (256, 328) and blank.getpixel((0, 0)) == (255, 255, 255, 255)


def test_big_thumbnail_hash(gpu, golden_dir):
    import ctypes
    import torch
    from image_transformation_amd import _native
    meta, _ = _load(golden_dir, "contact_sheet")
    row = [r for r in meta["cases"] if r.get("bundle") is None][0]
    big = cases.synthetic.make_cutout(np.random.default_rng(50_001), 1000, 800, "soft")
    tw, th = row["size"]
    src = torch.from_numpy(big).to(gpu.torch_device)
    dst = torch.empty((th, tw, 4), dtype=torch.uint8, device=gpu.torch_device)
    _native.check(_native.lib().mic_resize(gpu.handle, ctypes.c_void_p(src.data_ptr()), 1000, 800,
                                           ctypes.c_void_p(dst.data_ptr()), tw, th, _native.LANCZOS,
                                           ctypes.c_void_p(gpu.stream_ptr())))
    assert cases.sha16(dst.cpu().numpy()) == row["sha16"]


# ------------------------------------------------------------------------------------------ full-size configs
def _big(golden_dir):
    meta, _ = _load(golden_dir, "big_hashes")
    return {r["name"]: r for r in meta["cases"]}


def _solid(W, H):
    a = np.empty((H, W, 4), np.uint8)
    a[:] = np.asarray(cases.synthetic.SOLID_BG, np.uint8)
    return a


@pytest.mark.parametrize("alpha_mode", ["binary", "soft"])
def test_c2_c3_flex_full_size(gpu, golden_dir, alpha_mode):
    """BASELINE.json configs[1] and [2] at full size: Flex layout -> render_batch -> hash of the
    reference's output, plus equality with the oracle on the same placements."""
    from image_transformation_amd import flex
    from image_transformation_amd.compositor import Atlas, SolidCanvas, render_batch
    big = _big(golden_dir)
    syn = cases.synthetic
    size, objs, layout = syn.c2_workload(alpha_mode)
    atlas = Atlas(objs)
    out = render_batch([layout], atlas, [SolidCanvas(size, syn.SOLID_BG)])[0].cpu().numpy()
    row = big[f"c2_flex_{alpha_mode}"]
    pl = flex.layout_to_placements(layout, atlas, size)
    assert [p["box"] for p in pl] == row["boxes"]
    assert cases.sha16(out) == row["sha16"]
    assert np.array_equal(out, oracle.composite(_solid(*size), objs, pl))

    size, objs, layouts = syn.c3_workload(alpha_mode, n_layouts=3)
    atlas = Atlas(objs)
    outs = render_batch(layouts, atlas, [SolidCanvas(size, syn.SOLID_BG)] * 3)
    for k, (layout, o) in enumerate(zip(layouts, outs)):
        row = big[f"c3_flex_{alpha_mode}_{k}"]
        pl = flex.layout_to_placements(layout, atlas, size)
        assert [p["box"] for p in pl] == row["boxes"]
        got = o.cpu().numpy()
        assert cases.sha16(got) == row["sha16"], row["name"]
        if k == 0:
            assert np.array_equal(got, oracle.composite(_solid(*size), objs, pl))


def test_placements_mode_lanczos_full_size(gpu, golden_dir):
    """Direct composite() callers: boxes != cutout size -> Pillow-exact LANCZOS, overlaps, overhang."""
    from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
    big = _big(golden_dir)
    syn = cases.synthetic
    for name, W, H, n, seed in [("c2_placements_soft", 1920, 1080, 8, 2), ("c3_placements_soft", 3840, 2160, 32, 3)]:
        size, objs, pl = syn.placements_workload(W, H, n, seed, "soft")
        atlas = Atlas(objs)
        out = composite_device(atlas, [SolidCanvas(size, syn.SOLID_BG)], [coerce_placements(atlas, pl)])[0]
        got = out.cpu().numpy()
        assert cases.sha16(got) == big[name]["sha16"], name
        if n == 8:
            assert np.array_equal(got, oracle.composite(_solid(W, H), objs, pl))
        stats = gpu.stats()
        assert stats["resampled_layers"] > 0 and stats["canvas_pixels"] == W * H


def test_c4_variants_batch(gpu, golden_dir):
    """BASELINE.json configs[3] (first 8 variants): mixed canvas sizes in ONE launch."""
    from image_transformation_amd.compositor import Atlas, SolidCanvas, render_batch
    big = _big(golden_dir)
    syn = cases.synthetic
    objs, variants = syn.c4_workload("binary", n_variants=8)
    atlas = Atlas(objs)
    outs = render_batch([v[1] for v in variants], atlas, [SolidCanvas(v[0], syn.SOLID_BG) for v in variants])
    for v, o in enumerate(outs):
        row = big[f"c4_variant_{v}"]
        assert list(o.shape[1::-1]) == row["canvas"]
        assert cases.sha16(o.cpu().numpy()) == row["sha16"], row["name"]


def test_c5_audio_book_8k(gpu, golden_dir):
    """BASELINE.json configs[4]: audio_book end to end at 7680x4320 -- fill_solid colour, contact
    sheet size, and the 4 composite iterations (LANCZOS x8/x4 upscales)."""
    from image_transformation_amd.background_resizing import solid_canvas
    from image_transformation_amd.compositor import load_object_images, render
    big = _big(golden_dir)
    base = os.path.join(cases.BUNDLE_DIR, "audio_book")
    objects = load_object_images(os.path.join(base, "results.json"))
    canvas = solid_canvas(os.path.join(base, "background.png"), (7680, 4320))
    assert canvas.rgba == (38, 73, 115, 255)
    for it in range(4):
        row = big[f"c5_audio_book_iter{it}"]
        out = render({"placements": row["placements"]}, objects, canvas, as_tensor=True)
        assert cases.sha16(out.cpu().numpy()) == row["sha16"], row["name"]


# ------------------------------------------------------------------------------------------ properties at full size
def test_properties_full_size(gpu):
    """Size-independent properties at the bench size (4K, 32 objects):
    - splitting the placement list into two successive composites equals one composite
      (sequential 8-bit state, compositor.py:12-21);
    - an empty placement list is the identity on the background;
    - a layout translated by (dx, dy) on a larger canvas equals the translated image."""
    import torch
    from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
    syn = cases.synthetic
    (W, H), objs, pl = syn.placements_workload(3840, 2160, 32, 11, "soft", scale_range=(1.0, 1.0))
    atlas = Atlas(objs)
    rows = coerce_placements(atlas, pl)
    solid = SolidCanvas((W, H), syn.SOLID_BG)
    full = composite_device(atlas, [solid], [rows])[0]
    first = composite_device(atlas, [solid], [rows[:13]])[0]
    second = composite_device(atlas, [first], [rows[13:]])[0]
    assert torch.equal(full, second)
    ident = composite_device(atlas, [full], [[]])[0]
    assert torch.equal(ident, full) and ident.data_ptr() != full.data_ptr()
    dx, dy = 37, 21
    moved = [(o, x1 + dx, y1 + dy, x2 + dx, y2 + dy) for (o, x1, y1, x2, y2) in rows]
    wide = composite_device(atlas, [SolidCanvas((W + 64, H + 64), syn.SOLID_BG)], [moved])[0]
    assert torch.equal(wide[dy:dy + H - 64, dx:dx + W - 64], full[:H - 64, :W - 64])
    # order matters where soft layers overlap: reversing must change something
    rev = composite_device(atlas, [solid], [rows[::-1]])[0]
    assert not torch.equal(rev, full)


def test_properties_full_size_lanczos_batches(gpu):
    """Size-independent properties of the placements mode (Pillow-exact LANCZOS layers) at the bench size:
    - a batch of canvases in ONE call (one resample launch over every canvas' layers, one composite launch over every
      canvas' pages) equals the same canvases one call each;
    - splitting a placement list into two successive composites equals one composite, resampled layers included;
    - a persistent plan run twice, and run into a second set of outputs, repeats itself bit for bit."""
    import torch
    from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, composite_device, coerce_placements
    syn = cases.synthetic
    (W, H), objs, pl = syn.placements_workload(3840, 2160, 32, 3, "soft")
    atlas = Atlas(objs)
    sets = [pl] + syn.placement_sets(objs, W, H, 3, 3)
    rows = [coerce_placements(atlas, q) for q in sets]
    solid = SolidCanvas((W, H), syn.SOLID_BG)
    together = composite_device(atlas, [solid] * len(rows), rows)
    for k, r in enumerate(rows):
        alone = composite_device(atlas, [solid], [r])[0]
        assert torch.equal(together[k], alone), k
    first = composite_device(atlas, [solid], [rows[1][:11]])[0]
    second = composite_device(atlas, [first], [rows[1][11:]])[0]
    assert torch.equal(second, together[1])
    plan = CompositeBatch(atlas, [solid] * len(rows), rows)
    a, b = plan.alloc_outputs(), plan.alloc_outputs()
    plan.run(a); plan.run(b); plan.run(a)
    torch.cuda.synchronize()
    for k in range(len(rows)):
        assert torch.equal(a[k], together[k]) and torch.equal(b[k], together[k]), k


def test_alpha_over_every_triple_on_the_device(gpu):
    """Exhaustive on the hardware: one 4096 x 4096 layer over a 4096 x 4096 background image whose pixel i carries
    source alpha i & 255, source red (i >> 8) & 255 and destination red i >> 16 -- all 2^24 (alpha, source, destination)
    triples of AlphaComposite.c's arithmetic go through the composite kernel's packed opaque-destination form
    (kernels_composite.hip over_opaque_dst; the other channels ride along with permuted values) -- and, with a
    translucent destination, through the verbatim formula.  Checked against the oracle."""
    import torch
    from image_transformation_amd.compositor import Atlas, composite_device, coerce_placements
    n = 4096
    i = np.arange(n * n, dtype=np.uint32).reshape(n, n)
    sa, sc, dc = (i & 255).astype(np.uint8), ((i >> 8) & 255).astype(np.uint8), (i >> 16).astype(np.uint8)
    src = np.stack([sc, sc ^ 0x5A, 255 - sc, sa], axis=2)
    atlas = Atlas({1: src})
    rows = coerce_placements(atlas, [{"object_id": 1, "box": [0, 0, n, n]}])
    for dst_alpha in (255, None):
        da = np.full_like(dc, 255) if dst_alpha == 255 else ((dc.astype(np.uint16) * 7 + sa * 3 + 1) & 255).astype(np.uint8)
        bg = np.ascontiguousarray(np.stack([dc, dc ^ 0xA5, 255 - dc, da], axis=2))
        got = composite_device(atlas, [torch.from_numpy(bg).cuda()], [rows])[0].cpu().numpy()
        want = oracle.composite(bg, {1: src}, [{"object_id": 1, "box": [0, 0, n, n]}])
        assert np.array_equal(got, want), dst_alpha


def test_resize_every_colour_alpha_pair(gpu):
    """Every (colour, alpha) pair through premultiply -> the MFMA digit-chain passes -> unpremultiply of the tile
    kernel (mic_resize: single images take it; tests/test_gpu_march.py has the same through the marching kernel):
    pixel (x, y) of a 256 x 256 quadrant is colour x at alpha y, mirrored 2 x 2, resized to shapes on both sides of 1."""
    import torch
    from image_transformation_amd import _native
    x = np.arange(256, dtype=np.uint8)
    c, a = np.meshgrid(x, x)
    quad = np.stack([c, 255 - c, (c.astype(np.uint16) * 7 & 255).astype(np.uint8), a], axis=2)
    top = np.concatenate([quad, quad[:, ::-1]], axis=1)
    src = np.ascontiguousarray(np.concatenate([top, top[::-1]], axis=0))  # 512 x 512
    dev = torch.from_numpy(src).cuda()
    lib, P = _native.lib(), ctypes.c_void_p
    for dw, dh in [(512, 513), (513, 512), (700, 700), (301, 419), (1024, 600), (90, 77)]:
        for filt in (0, 1):
            dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
            _native.check(lib.mic_resize(gpu.handle, P(dev.data_ptr()), 512, 512, P(dst.data_ptr()), dw, dh, filt, P(gpu.stream_ptr())))
            assert np.array_equal(dst.cpu().numpy(), oracle.resize(src, (dw, dh), filt)), (dw, dh, filt)


def test_ragged_and_extreme_shapes(gpu):
    """1-pixel canvases/objects, widths that are not multiples of 4 or 256, > 64 layers per canvas."""
    from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
    rng = np.random.default_rng(99)
    for (W, H, n) in [(1, 1, 3), (3, 2, 5), (257, 17, 9), (4399, 33, 7), (1023, 1025, 150)]:
        objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(1, 90)), int(rng.integers(1, 70)), "soft")
                for i in range(min(n, 12))}
        pl = []
        for k in range(n):
            oid = int(rng.integers(1, len(objs) + 1))
            sh, sw = objs[oid].shape[:2]
            x1, y1 = int(rng.integers(-sw, W + 1)), int(rng.integers(-sh, H + 1))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        bg = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        import torch
        atlas = Atlas(objs)
        out = composite_device(atlas, [torch.from_numpy(bg).to(gpu.torch_device)], [coerce_placements(atlas, pl)])[0]
        want = oracle.composite(bg, objs, pl)
        got = out.cpu().numpy()
        assert np.array_equal(got, want), ((W, H, n), _maxdiff(got, want))


def test_composite_fuzz_page_geometry(gpu):
    """Random canvases against the oracle with everything that moves the 4 KiB page geometry: widths
    that are / are not multiples of 4, sizes that are / are not multiples of a page (tail page through
    the normal path), output and background pointers at odd 4-byte offsets inside a larger buffer
    (first page partial: edge_page), solid / translucent-solid / image backgrounds, binary and soft
    cutouts, and a few resampled layers."""
    import torch
    from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
    rng = np.random.default_rng(2024)
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(3, 200)), int(rng.integers(3, 150)),
                                               "binary" if i % 2 else "soft") for i in range(10)}
    atlas = Atlas(objs)
    dev = gpu.torch_device
    for it in range(40):
        W = int(rng.choice([64, 100, 255, 256, 257, 1000, 1024, 1027, 1365, 2048, 2051]))
        H = int(rng.integers(1, 60))
        n = int(rng.integers(0, 14))
        pl = []
        for _ in range(n):
            oid = int(rng.integers(1, len(objs) + 1))
            sh, sw = objs[oid].shape[:2]
            if rng.random() < 0.2:  # resampled layer
                sw, sh = max(1, int(sw * rng.uniform(0.4, 1.6))), max(1, int(sh * rng.uniform(0.4, 1.6)))
            x1, y1 = int(rng.integers(-sw, W + 1)), int(rng.integers(-sh, H + 1))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        rows = coerce_placements(atlas, pl)
        kind = it % 3
        n_bytes = W * H * 4
        # an output view at a random 4-byte offset inside a larger allocation
        off = 4 * int(rng.integers(0, 1500))
        big = torch.zeros(n_bytes + 8192, dtype=torch.uint8, device=dev)
        out_view = big[off:off + n_bytes].view(H, W, 4)
        if kind == 0:
            colour = (38, 73, 115, 255)
            bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = colour
            canvas = SolidCanvas((W, H), colour)
        elif kind == 1:
            colour = (200, 10, 60, int(rng.integers(0, 255)))
            bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = colour
            canvas = SolidCanvas((W, H), colour)
        else:
            bg_np = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            if it % 2:
                bg_np[:, :, 3] = 255
            boff = 4 * int(rng.integers(0, 1500))
            bbig = torch.zeros(n_bytes + 8192, dtype=torch.uint8, device=dev)
            canvas = bbig[boff:boff + n_bytes].view(H, W, 4)
            canvas.copy_(torch.from_numpy(bg_np).to(dev))
        got = composite_device(atlas, [canvas], [rows], outs=[out_view])[0].cpu().numpy()
        want = oracle.composite(bg_np, objs, pl)
        assert np.array_equal(got, want), (it, W, H, n, kind, off, _maxdiff(got, want))
        # nothing outside the canvas was written
        assert not big[:off].any() and not big[off + n_bytes:].any(), (it, "wrote outside the canvas")


def test_plan_job_table_cache(gpu):
    """A persistent plan keeps device job tables for the last four output sets: cycling over six sets
    (hits, misses and evictions) must give the same pixels every time, and a run must never write into
    a set it was not asked to."""
    import torch
    from image_transformation_amd import flex
    from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
    size, objs, layouts = cases.synthetic.c3_workload("binary", seed=11, n_layouts=3)
    size = (1283, 517)  # small, unaligned width: both kernel classes of a mixed batch stay cheap
    atlas = Atlas(objs)
    rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, size)) for l in layouts]
    canvases = [SolidCanvas(size, (38, 73, 115, 255)), SolidCanvas(size, (1, 2, 3, 200)), SolidCanvas(size, (9, 9, 9, 255))]
    plan = CompositeBatch(atlas, canvases, rows)
    want = None
    sets = [plan.alloc_outputs() for _ in range(6)]
    for s in sets:
        for t in s:
            t.zero_()
    order = [0, 1, 0, 2, 3, 4, 5, 0, 1, 5, 5, 2]
    used = set()
    for k in order:
        plan.run(sets[k])
        used.add(k)
        got = [t.cpu().numpy() for t in sets[k]]
        if want is None:
            want = got
            bgs = [np.empty((size[1], size[0], 4), np.uint8) for _ in canvases]
            for b, c in zip(bgs, canvases):
                b[:] = np.asarray(c.rgba, np.uint8)
            for g, b, l in zip(got, bgs, layouts):
                assert np.array_equal(g, oracle.composite(b, objs, flex.layout_to_placements(l, atlas, size)))
        for g, w in zip(got, want):
            assert np.array_equal(g, w), k
        for j, s in enumerate(sets):  # sets not run yet are still zero
            if j not in used:
                assert not any(bool(t.any()) for t in s), (k, j)
        for t in sets[k]:
            t.zero_()
            used.discard(k)


# ------------------------------------------------------------------------------------------ round 3
def test_median_batch_strided_views_and_long_batches(gpu):
    """mic_median_rgb_batch: several images -- packed, strided views (fill_gradient's edge strips), one-row and
    one-column views, all-transparent, > 16 images (two launches) -- each equal to the oracle on the same pixels; the
    scratch double buffer survives alternating batch sizes."""
    import torch
    from image_transformation_amd.background_resizing import median_color_device, median_colors_device
    rng = np.random.default_rng(31)
    base = rng.integers(0, 256, (250, 970, 4), dtype=np.uint8)
    base[:, :, 3] = np.where(rng.random((250, 970)) < 0.3, 0, base[:, :, 3])
    dev = torch.from_numpy(base).to(gpu.torch_device)
    views = [dev[:, :8], dev[:, -8:], dev[:8], dev[-8:], dev, dev[3:4], dev[:, 5:6], dev[10:200:1, 100:900]]
    want = [oracle.median_rgb(np.ascontiguousarray(v.cpu().numpy())) for v in views]
    for _ in range(3):
        assert median_colors_device(views) == want
        assert median_color_device(dev) == want[4]
    clear = torch.zeros((40, 50, 4), dtype=torch.uint8, device=gpu.torch_device)
    clear[:, :, :3] = torch.randint(0, 256, (40, 50, 3), dtype=torch.uint8, device=gpu.torch_device)
    big = torch.randint(0, 256, (1100, 1900, 4), dtype=torch.uint8, device=gpu.torch_device)  # > 32 blocks: 8 copies
    many = [clear, big, big[:, 7:1500], big[500:]] + [dev[k:k + 9 + k, k:k + 31] for k in range(0, 34, 2)]
    assert len(many) == 21
    got = median_colors_device(many)
    for v, g in zip(many, got):
        assert g == oracle.median_rgb(np.ascontiguousarray(v.cpu().numpy()))
    assert median_colors_device([]) == []
    with pytest.raises(ValueError):
        median_colors_device([dev[:, ::2]])  # pixels not adjacent


@pytest.mark.parametrize("two", ["1", "0"])
def test_median_forced_launch_forms_agree(gpu, two):
    """Both forms of the median -- one launch with a retirement ticket, or histogram kernel + select kernel -- at sizes on
    either side of the size rule that picks between them by default (MIC_MEDIAN_TWO_LAUNCHES=1 / 0 at mic_create forces
    one form for every size): the same colours as the oracle."""
    import ctypes
    import torch
    from image_transformation_amd import _native
    lib = _native.lib()
    os.environ["MIC_MEDIAN_TWO_LAUNCHES"] = two
    try:
        h = ctypes.c_void_p()
        _native.check(lib.mic_create(gpu.device, ctypes.byref(h)))
    finally:
        del os.environ["MIC_MEDIAN_TWO_LAUNCHES"]
    try:
        rng = np.random.default_rng(32)
        for shape in ((492, 492), (2160, 3840), (1, 1), (37, 1001), (3000, 4100)):
            a = rng.integers(0, 256, shape + (4,), dtype=np.uint8)
            if shape[0] == 3000:  # flat with specks, mostly transparent
                a[:, :, :3] = (1, 2, 254)
                a[::11, ::13, :3] = 77
                a[:, :, 3] = np.where(rng.random(shape) < 0.9, 0, a[:, :, 3])
            d = torch.from_numpy(a).to(gpu.torch_device)
            out = (ctypes.c_uint8 * 3)()
            for _ in range(2):
                _native.check(lib.mic_median_rgb(h, ctypes.c_void_p(d.data_ptr()), shape[1], shape[0], out,
                                                 ctypes.c_void_p(gpu.stream_ptr())))
                assert tuple(out) == oracle.median_rgb(a), shape
    finally:
        lib.mic_destroy(h)


def test_atlas_and_plan_built_on_a_side_stream(gpu):
    """an Atlas uploaded and a CompositeBatch created inside `with torch.cuda.stream(s)` (a
    non-blocking side stream), LANCZOS layers included (the marching kernel's planar copies are converted by the first
    RUN, on the run's stream) -- then run on that stream and again on the default stream."""
    import torch
    from image_transformation_amd import _native
    from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
    syn = cases.synthetic
    lib = _native.lib()
    size, objs, pl = syn.placements_workload(1920, 1080, 24, 77, "soft")
    want = oracle.composite(_solid(*size), objs, pl)
    # a context that routes every qualifying layer through the marching kernel (planar copies needed)
    os.environ["MIC_RS_MARCH_MIN_UNITS"] = "0"
    try:
        import ctypes
        h = ctypes.c_void_p()
        _native.check(lib.mic_create(gpu.device, ctypes.byref(h)))
    finally:
        del os.environ["MIC_RS_MARCH_MIN_UNITS"]
    lib.mic_destroy(h)  # (only to prove creation works with the knob; the package context below uses its default)
    s = torch.cuda.Stream(device=gpu.torch_device)
    with torch.cuda.stream(s):
        filler = torch.empty(256 << 20, dtype=torch.uint8, device=gpu.torch_device)
        filler.fill_(1)  # keeps the side stream busy while the upload and the plan are enqueued behind it
        atlas = Atlas(objs)
        plan = CompositeBatch(atlas, [SolidCanvas(size, syn.SOLID_BG)], [coerce_placements(atlas, pl)])
        out_side = plan.run()[0]
    out_default = plan.run()[0]  # torch's default stream: must wait for the upload on `s`
    torch.cuda.synchronize()
    assert np.array_equal(out_default.cpu().numpy(), want)
    assert np.array_equal(out_side.cpu().numpy(), want)
    assert plan.stats()["resampled_layers"] > 0


def test_results_are_ordinary_mutable_images(gpu):
    """The PIL image composite() returns behaves like the reference's: writable through load(), paste, ImageDraw; and
    edits never leak into a later result."""
    from PIL import ImageDraw
    from image_transformation_amd.compositor import composite
    rng = np.random.default_rng(8)
    bg = _img(rng.integers(0, 256, (60, 80, 4), dtype=np.uint8))
    objs = {1: _img(rng.integers(0, 256, (20, 30, 4), dtype=np.uint8))}
    pl = [{"object_id": 1, "box": [5, 6, 35, 26]}]
    want = oracle.composite(np.array(bg), {1: np.array(objs[1])}, pl)
    out = composite(bg, objs, pl)
    assert np.array_equal(np.array(out), want)
    px = out.load()
    px[0, 0] = (1, 2, 3, 4)
    ImageDraw.Draw(out).rectangle([10, 10, 20, 20], fill=(9, 9, 9, 9))
    out.paste((7, 7, 7, 7), (0, 30, 5, 35))
    assert out.getpixel((0, 0)) == (1, 2, 3, 4) and out.getpixel((15, 15)) == (9, 9, 9, 9) and out.getpixel((2, 32)) == (7, 7, 7, 7)
    again = composite(bg, objs, pl)
    assert np.array_equal(np.array(again), want)


def test_in_place_edit_of_a_loaded_cutout_is_seen(gpu, tmp_path):
    """load_object_images() -> putalpha in place -> composite: the edited pixels are composited
    (the shared atlas of the bundle's files must not be used), also when an earlier load of the same files has
    already put that atlas into the cache (the contact sheet does)."""
    from image_transformation_amd.compositor import composite, load_object_images
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
    base = os.path.join(cases.BUNDLE_DIR, "squarespace")
    rj = os.path.join(base, "results.json")
    build_labeled_contact_sheet(os.path.join(base, "objects"), rj)  # fills the cache with the files' atlas
    objs = load_object_images(rj)
    objs[2].putalpha(90)
    bg = _img(_solid(492, 492))
    pl = [{"object_id": 2, "box": [10, 20, 10 + objs[2].size[0], 20 + objs[2].size[1]]}, {"object_id": 1, "box": [0, 0, 300, 90]}]
    want = oracle.composite(np.array(bg), {k: np.array(v) for k, v in objs.items()}, pl)
    assert np.array_equal(np.array(composite(bg, objs, pl)), want)
    shared = load_object_images(rj, shared=True)
    shared[2].putalpha(90)  # copy-on-write view: detected, private atlas
    assert np.array_equal(np.array(composite(bg, shared, pl)), want)


def test_layer_records_in_kernel_arguments_boundary(gpu):
    """A single-canvas launch carries its layer records in the kernel arguments up to 64 of them (kPackLayers) and reads
    the device table beyond: 1, 63, 64, 65 and 130 layers, one-shot (mic_composite_batch) and persistent (mic_plan_*)
    launches, identity and resampled layers, all equal to the oracle."""
    from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, composite_device, coerce_placements
    syn = cases.synthetic
    rng = np.random.default_rng(64)
    objs = syn.make_cutouts(9, (20, 70), (15, 60), seed=640, alpha_mode="soft")
    atlas = Atlas(objs)
    W, H = 801, 333
    for n in (1, 63, 64, 65, 130):
        for resample in (False, True):
            pl = []
            for k in range(n):
                oid = int(rng.integers(1, 10))
                sh, sw = objs[oid].shape[:2]
                if resample and k % 5 == 0:
                    sw, sh = max(1, int(sw * 1.3)), max(1, int(sh * 0.8))
                x1, y1 = int(rng.integers(-10, W)), int(rng.integers(-10, H))
                pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
            rows = coerce_placements(atlas, pl)
            want = oracle.composite(_solid(W, H), objs, pl)
            got = composite_device(atlas, [SolidCanvas((W, H), syn.SOLID_BG)], [rows])[0].cpu().numpy()
            assert np.array_equal(got, want), (n, resample, "one-shot")
            plan = CompositeBatch(atlas, [SolidCanvas((W, H), syn.SOLID_BG)], [rows])
            for _ in range(2):
                assert np.array_equal(plan.run()[0].cpu().numpy(), want), (n, resample, "plan")


def test_large_pil_backgrounds(gpu):
    """A PIL background that is not one colour at full size (uploaded through the pinned staging buffer): 4K and an odd
    4399-wide canvas, several calls in a row, composite() and render(as_tensor=True), equal to the oracle."""
    from image_transformation_amd.compositor import composite, render
    syn = cases.synthetic
    rng = np.random.default_rng(909)
    objs = syn.make_cutouts(6, (200, 500), (150, 400), seed=910, alpha_mode="soft")
    pil_objs = {k: _img(v) for k, v in objs.items()}
    for (W, H) in ((3840, 2160), (4399, 1885)):
        for rep in range(2):
            bg = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            pl = [{"object_id": k, "box": [int(rng.integers(-50, W - 100)), int(rng.integers(-50, H - 100)), 0, 0]} for k in objs]
            for p in pl:
                sh, sw = objs[p["object_id"]].shape[:2]
                p["box"][2], p["box"][3] = p["box"][0] + sw, p["box"][1] + sh
            want = oracle.composite(bg, objs, pl)
            assert np.array_equal(np.array(composite(_img(bg), pil_objs, pl)), want), (W, H, rep)
            if rep == 0:
                assert np.array_equal(render({"placements": pl}, pil_objs, _img(bg), as_tensor=True).cpu().numpy(), want)


def test_solid_canvas_with_a_device_colour_word(gpu):
    """Round 4: a SolidCanvas whose colour lives in device memory (mic_job.bg_rgba_dev: what the median kernel wrote on the
    stream) composites like the same colour given on the host -- single-canvas launches (job in the kernel arguments),
    batches (job table), render() through mic_render_job, aligned / unaligned widths, LANCZOS layers; a colour word that
    is NOT opaque takes the exact per-pixel path; the colour is only downloaded when `.rgba` is read."""
    import torch
    from image_transformation_amd import _native
    from image_transformation_amd.background_resizing import median_color_device
    from image_transformation_amd.compositor import (Atlas, CompositeBatch, SolidCanvas, composite_device, coerce_placements,
                                                     render)
    rng = np.random.default_rng(808)
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(30, 120)), int(rng.integers(20, 90)), "soft") for i in range(5)}
    atlas = Atlas(objs)
    for colour in ((38, 73, 115, 255), (200, 10, 60, 130), (0, 0, 0, 0)):
        word = torch.tensor(colour, dtype=torch.uint8, device=gpu.torch_device)
        sizes = [(333, 97), (256, 64), (1030, 41), (64, 64)]
        pls, cvs_dev, cvs_host, bgs = [], [], [], []
        for (W, H) in sizes:
            pl = []
            for _ in range(int(rng.integers(1, 9))):
                oid = int(rng.integers(1, 6))
                sh, sw = objs[oid].shape[:2]
                if rng.random() < 0.4:
                    sw, sh = max(1, int(sw * rng.uniform(0.5, 1.7))), max(1, int(sh * rng.uniform(0.5, 1.7)))
                x1, y1 = int(rng.integers(-sw // 2, W)), int(rng.integers(-sh // 2, H))
                pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
            pls.append(pl)
            cvs_dev.append(SolidCanvas((W, H), colour_dev=word))
            cvs_host.append(SolidCanvas((W, H), colour))
            bg = np.empty((H, W, 4), np.uint8)
            bg[:] = colour
            bgs.append(bg)
        rows = [coerce_placements(atlas, pl) for pl in pls]
        wants = [oracle.composite(bgs[i], objs, pls[i]) for i in range(len(sizes))]
        for cvs in (cvs_dev, cvs_host):
            for i, o in enumerate(composite_device(atlas, cvs, rows)):            # one launch, job table
                assert np.array_equal(o.cpu().numpy(), wants[i]), (colour, i)
            for i in range(len(sizes)):                                             # one canvas per call, job in the arguments
                got = composite_device(atlas, [cvs[i]], [rows[i]])[0].cpu().numpy()
                assert np.array_equal(got, wants[i]), (colour, "single", i)
            plan = CompositeBatch(atlas, cvs, rows)
            for rep in range(2):
                for i, o in enumerate(plan.run()):
                    assert np.array_equal(o.cpu().numpy(), wants[i]), (colour, "plan", rep, i)
        assert cvs_dev[0]._rgba is None  # nothing above asked for the colour on the host ...
        assert cvs_dev[0].rgba == colour and cvs_dev[0].to_image().getpixel((0, 0)) == colour  # ... this does
    # render() through mic_render_job, colour straight from the median kernel (no host round trip in between)
    W, H = 640, 360
    img = rng.integers(0, 256, (70, 90, 4), dtype=np.uint8)
    dev = torch.from_numpy(img).to(gpu.torch_device)
    word = torch.empty(4, dtype=torch.uint8, device=gpu.torch_device)
    _native.check(_native.lib().mic_median_rgb_dev(gpu.handle, ctypes.c_void_p(dev.data_ptr()), 90, 70,
                                                   ctypes.c_void_p(word.data_ptr()), ctypes.c_void_p(gpu.stream_ptr())))
    layout = {"root": {"type": "flex", "direction": "row", "justify": "space_around", "align": "center", "gap_px": 5,
                       "children": [{"object_id": k} for k in objs]}}
    got = render(layout, atlas, SolidCanvas((W, H), colour_dev=word), as_tensor=True).cpu().numpy()
    med = oracle.median_rgb(img)
    assert median_color_device(dev) == med
    bg = np.empty((H, W, 4), np.uint8)
    bg[:] = med + (255,)
    from image_transformation_amd import flex
    assert np.array_equal(got, oracle.composite(bg, objs, flex.layout_to_placements(layout, atlas, (W, H))))
    # a background image AND a colour word: refused
    lib = _native.lib()
    job = _native.Job()
    out = torch.empty((8, 8, 4), dtype=torch.uint8, device=gpu.torch_device)
    job.width, job.height, job.out_dev, job.bg_dev, job.bg_rgba_dev = 8, 8, out.data_ptr(), dev.data_ptr(), word.data_ptr()
    atl = (ctypes.c_void_p * 1)(atlas.handle)
    assert lib.mic_composite_batch(gpu.handle, 1, atl, 1, ctypes.byref(job), 0, ctypes.c_void_p(gpu.stream_ptr())) < 0
    assert b"both a background image and a device colour word" in lib.mic_last_error()


def test_single_canvas_launch_dispatches_no_surplus_workgroups(gpu):
    """One canvas per call is the reference's own call shape (compositor.py:6-22).  Round 4 shipped a single-job launch
    that dispatched one four-wave workgroup per PAGE (three quarters of its waves returned at once; results were
    right, so no parity test saw it).  mic_stats.composite_blocks is computed from the grid that is launched: for
    every instantiation (aligned x solid) and both single-job modes (layer records in the kernel arguments: <= 64
    layers; job only: more) it must cover the canvas' 4 KiB pages with less than one round of 8 workgroups to spare
    (workgroups of four pages; of one page for canvases up to 2048 pages, round 5)."""
    import torch
    from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
    syn = cases.synthetic
    objs = syn.make_cutouts(6, (30, 90), (20, 70), seed=77, alpha_mode="soft")
    atlas = Atlas(objs)
    rng = np.random.default_rng(5)
    kp = 4  # kPagesPerWorkgroup (mic_internal.h)
    seen = set()
    for (W, H) in [(492, 492), (1920, 1080), (3840, 2160), (7680, 4320), (4399, 1885), (1023, 64)]:
        for solid in (True, False):
            for n_layers in (5, 70):
                pl = []
                for k in range(n_layers):
                    oid = int(rng.integers(1, 7))
                    sh, sw = objs[oid].shape[:2]
                    x1, y1 = int(rng.integers(-20, W - 10)), int(rng.integers(-20, H - 10))
                    pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
                rows = coerce_placements(atlas, pl)
                if solid:
                    bg_np = oracle.fill_solid((W, H), syn.SOLID_BG)
                    canvas = SolidCanvas((W, H), syn.SOLID_BG)
                else:
                    bg_np = np.empty((H, W, 4), np.uint8)
                    bg_np[...] = rng.integers(0, 256, (1, W, 4), dtype=np.uint8)
                    bg_np[..., 3] = 255
                    canvas = torch.from_numpy(bg_np).to(gpu.torch_device)
                out = composite_device(atlas, [canvas], [rows])[0]
                st = gpu.stats()
                out_np = out.cpu().numpy()
                n_pages = (W * H + (out.data_ptr() % 4096) // 4 + 1023) // 1024
                # (canvases of at most 2048 pages -- kSmallCanvasPages -- are launched as one-wave workgroups)
                covered = st["composite_blocks"] * (1 if n_pages <= 2048 else kp)
                assert n_pages <= covered < n_pages + 8 * kp, ((W, H), solid, n_layers, st["composite_blocks"], n_pages)
                assert np.array_equal(out_np, oracle.composite(bg_np, objs, pl)), ((W, H), solid, n_layers)
                seen.add((W % 4 == 0, solid, n_layers <= 64))
    assert len(seen) == 8  # four instantiations x {kAllInArgs, kJobInArgs}
