"""The two big-call MFMA resample kernels -- the lane kernel (kernels_resample_lane.hip, round 5: one wave per piece,
tiled planar source, ring of intermediate rows in registers) and the marching kernel it took over from
(kernels_resample.hip) -- on the shapes that reach their corners.  Every test runs once per kernel.

By default only big calls take them (mic_api.hip: routing), so small parity cases run through the tile kernel; here a
context created with MIC_RS_LANE_MIN_SLOTS=0 (lane) or MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=0 (marching) sends every
layer that qualifies through the kernel under test, and mic_stats.marched_layers says that it did.
Each resize is the only layer of a composite onto a transparent canvas: alpha-over onto alpha 0 returns the layer
wherever its alpha is > 0 (oracle.composite is the expected value either way).
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cases  # noqa: E402,F401
import oracle  # noqa: E402  (the checker)

P = ctypes.c_void_p


@pytest.fixture(scope="module", params=["lane", "lane_two_tiles", "march"])
def forced(request):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; none is visible")
    from image_transformation_amd import _native
    lib = _native.lib()
    env = {"MIC_LAYER_CACHE_MB": "0"}  # every call resamples (a resident layer of an earlier call would skip the kernel under test)
    # (calls below 1500 slots of work run one x-tile per piece: "lane_two_tiles" switches that off, so that these small
    # shapes also reach the two-tile form the big calls use)
    env.update({"MIC_RS_LANE_MIN_SLOTS": "0"} if request.param == "lane" else
               {"MIC_RS_LANE_MIN_SLOTS": "0", "MIC_RS_LANE_SPLIT": "0,0"} if request.param == "lane_two_tiles" else
               {"MIC_RS_LANE": "0", "MIC_RS_MARCH_MIN_UNITS": "0"})
    os.environ.update(env)
    try:
        ctx = P()
        assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(ctx)) == 0, lib.mic_last_error()
    finally:
        for k in env:
            del os.environ[k]
    _native.kernel_under_test = "march" if request.param == "march" else "lane"
    yield lib, ctx, _native
    assert lib.mic_destroy(ctx) == 0


def _atlas(lib, ctx, objs):
    ids = (ctypes.c_int32 * len(objs))(*objs.keys())
    ws = (ctypes.c_int32 * len(objs))(*[a.shape[1] for a in objs.values()])
    hs = (ctypes.c_int32 * len(objs))(*[a.shape[0] for a in objs.values()])
    keep = [np.ascontiguousarray(a) for a in objs.values()]
    ptrs = (P * len(objs))(*[a.ctypes.data for a in keep])
    atlas = P()
    assert lib.mic_atlas_create(ctx, len(objs), ids, ws, hs, ptrs, ctypes.byref(atlas)) == 0, lib.mic_last_error()
    return atlas


def _composite(lib, ctx, nat, atlas, size, bg_rgba, placements, filt):
    import torch
    W, H = size
    pl = (nat.Placement * max(len(placements), 1))()
    for i, p in enumerate(placements):
        pl[i].atlas, pl[i].object_id = 0, p["object_id"]
        for k in range(4):
            pl[i].box[k] = p["box"][k]
    out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    job = nat.Job()
    job.width, job.height, job.bg_dev = W, H, None
    for k, v in enumerate(bg_rgba):
        job.bg_rgba[k] = v
    job.n_placements, job.placements, job.out_dev = len(placements), pl, out.data_ptr()
    atl = (P * 1)(atlas)
    stream = P(torch.cuda.current_stream().cuda_stream)
    assert lib.mic_composite_batch(ctx, 1, atl, 1, ctypes.byref(job), filt, stream) == 0, lib.mic_last_error()
    st = nat.Stats()
    assert lib.mic_last_stats(ctx, ctypes.byref(st)) == 0
    return out.cpu().numpy(), st.as_dict()


def _alpha(rng, src, mode):
    sh, sw = src.shape[:2]
    if mode == 1:    # binary alpha, like the reference's bundles
        src[:, :, 3] = np.where(rng.random((sh, sw)) < 0.45, 0, 255)
    elif mode == 2:  # mostly opaque with soft edges
        src[:, :, 3] = np.where(rng.random((sh, sw)) < 0.8, 255, src[:, :, 3])
    return src


def test_march_single_layer_shapes_vs_oracle(forced):
    lib, ctx, nat = forced
    rng = np.random.default_rng(78)
    shapes = [((301, 203), (457, 311)), ((457, 311), (301, 203)), ((643, 97), (211, 97)), ((97, 643), (97, 211)),
              ((513, 259), (1026, 518)), ((130, 70), (1301, 707)), ((66, 66), (67, 65)), ((2, 2), (97, 33)),
              ((1, 7), (50, 3)), ((257, 255), (255, 257)), ((19, 23), (640, 480)), ((1023, 767), (511, 383)),
              ((16, 4000), (17, 33)), ((59, 450), (114, 25)), ((450, 59), (25, 114)), ((64, 1000), (64, 48)),
              ((640, 480), (639, 481)), ((1000, 37), (350, 37)), ((37, 1000), (37, 350)), ((900, 700), (310, 240)),
              ((333, 333), (1000, 1000)), ((65, 65), (64, 64)), ((1024, 64), (1025, 63))]
    marched = 0
    for i, ((sw, sh), (dw, dh)) in enumerate(shapes):
        src = _alpha(rng, rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8), i % 3)
        objs = {5: src}
        atlas = _atlas(lib, ctx, objs)
        pl = [{"object_id": 5, "box": [0, 0, dw, dh]}]
        for filt in (nat.LANCZOS, nat.BILINEAR):
            got, st = _composite(lib, ctx, nat, atlas, (dw, dh), (0, 0, 0, 0), pl, filt)
            want = oracle.composite(np.zeros((dh, dw, 4), np.uint8), objs, pl, filt)
            assert np.array_equal(got, want), ((sw, sh), (dw, dh), filt, st)
            marched += st["marched_layers"]
        assert lib.mic_atlas_destroy(atlas) == 0
    # most of these have one-chunk tiles on both axes; the rest took the tile kernel (the lane kernel's vertical window is
    # four 16-row bands that END with a tile's last tap row: it leaves a few more of the deep shrinks to the tile kernel)
    assert marched >= (26 if nat.kernel_under_test == "lane" else 30), marched


def test_march_transparent_margins_vs_oracle(forced):
    """Cutout-shaped sources: the marching kernel skips the horizontal pass of source bands without a pixel of
    alpha > 0 and stores output tiles that only see such bands as transparent black."""
    lib, ctx, nat = forced
    rng = np.random.default_rng(99)
    cases_ = [((640, 480), (400, 300), (200, 150, 90, 60)), ((640, 480), (961, 719), (330, 250, 40, 200)),
              ((500, 700), (250, 349), (250, 100, 200, 17)), ((300, 300), (300, 450), (150, 150, 10, 10)),
              ((1200, 900), (500, 375), (600, 450, 250, 180)), ((257, 513), (500, 1000), (128, 40, 60, 33)),
              ((400, 400), (200, 200), None)]
    marched = 0
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
        objs = {1: src}
        atlas = _atlas(lib, ctx, objs)
        pl = [{"object_id": 1, "box": [0, 0, dw, dh]}]
        for filt in (nat.LANCZOS, nat.BILINEAR):
            for bg in ((0, 0, 0, 0), (9, 200, 33, 255)):
                got, st = _composite(lib, ctx, nat, atlas, (dw, dh), bg, pl, filt)
                b = np.empty((dh, dw, 4), np.uint8)
                b[:] = bg
                want = oracle.composite(b, objs, pl, filt)
                assert np.array_equal(got, want), ((sw, sh), (dw, dh), blob, filt, bg, st)
                marched += st["marched_layers"]
        assert lib.mic_atlas_destroy(atlas) == 0
    assert marched == 4 * len(cases_), marched


def test_march_many_layers_overhang_vs_oracle(forced):
    """Several layers of one call through the marching kernel at once (different ring / band sizes in one launch),
    boxes hanging over every canvas edge, one cutout used at two scales, ids missing from the atlas."""
    lib, ctx, nat = forced
    rng = np.random.default_rng(5)
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(20, 400)), int(rng.integers(20, 300)),
                                               ["binary", "soft"][i % 2]) for i in range(7)}
    atlas = _atlas(lib, ctx, objs)
    for (W, H) in ((1023, 517), (640, 360), (257, 1025)):
        pl = []
        for k in range(14):
            oid = int(rng.integers(1, 9))  # 8 is not in the atlas: skipped like the reference skips it
            sh, sw = objs.get(oid, objs[1]).shape[:2]
            s = rng.uniform(0.4, 2.2)
            w, h = max(1, int(sw * s)), max(1, int(sh * rng.uniform(0.4, 2.2)))
            x1, y1 = int(rng.integers(-w // 2, W - w // 2)), int(rng.integers(-h // 2, H - h // 2))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + w, y1 + h]})
        for filt in (nat.LANCZOS, nat.BILINEAR):
            got, st = _composite(lib, ctx, nat, atlas, (W, H), (38, 73, 115, 255), pl, filt)
            b = np.empty((H, W, 4), np.uint8)
            b[:] = (38, 73, 115, 255)
            want = oracle.composite(b, objs, pl, filt)
            assert np.array_equal(got, want), ((W, H), filt, st, pl)
            assert st["marched_layers"] >= 8, st
    assert lib.mic_atlas_destroy(atlas) == 0


def test_march_every_colour_alpha_pair(forced):
    """Every (colour, alpha) pair through the marching kernel's premultiply (planarize_kernel) -> digit-chain passes
    -> unpremultiply: a 256 x 256 source whose pixel (x, y) is colour x at alpha y (the other channels permuted
    values), repeated 2 x 2 so that every pair also sits next to its opposite, resized to shapes on both sides of 1."""
    lib, ctx, nat = forced
    x = np.arange(256, dtype=np.uint8)
    c, a = np.meshgrid(x, x)
    quad = np.stack([c, 255 - c, (c.astype(np.uint16) * 7 & 255).astype(np.uint8), a], axis=2)
    top = np.concatenate([quad, quad[:, ::-1]], axis=1)
    src = np.ascontiguousarray(np.concatenate([top, top[::-1]], axis=0))  # 512 x 512
    objs = {1: src}
    atlas = _atlas(lib, ctx, objs)
    marched = 0
    for dw, dh in [(512, 513), (513, 512), (700, 700), (301, 419), (1024, 600)]:
        pl = [{"object_id": 1, "box": [0, 0, dw, dh]}]
        for filt in (nat.LANCZOS, nat.BILINEAR):
            got, st = _composite(lib, ctx, nat, atlas, (dw, dh), (0, 0, 0, 0), pl, filt)
            want = oracle.composite(np.zeros((dh, dw, 4), np.uint8), objs, pl, filt)
            assert np.array_equal(got, want), ((dw, dh), filt, st)
            marched += st["marched_layers"]
    assert lib.mic_atlas_destroy(atlas) == 0
    assert marched >= 8, marched


@pytest.mark.parametrize("keeps,two_tiles", [(1, False), (1, True), (0, False)])
def test_lane_layers_that_keep_one_axis(keeps, two_tiles):
    """A layer that keeps its width or its height (Pillow skips that pass: Resample.c need_horizontal / need_vertical; the
    call site is compositor.py:20) runs the lane kernel's own instantiation for its class -- one MFMA per channel over the
    kept axis instead of the three-digit chain -- in a launch of its own behind the general one.  Single layers,
    transparent margins, and one call that mixes the three classes; MIC_RS_LANE_KEEPS=0 sends them all through the general
    form.  Same pixels as the oracle either way."""
    import torch
    from image_transformation_amd import _native as nat
    lib = nat.lib()
    env = {"MIC_LAYER_CACHE_MB": "0", "MIC_RS_LANE_MIN_SLOTS": "0", "MIC_RS_LANE_KEEPS": str(keeps)}
    if two_tiles:  # (small calls run one x-tile per piece; this sends them through the two-tile bodies as well)
        env["MIC_RS_LANE_SPLIT"] = "0,0"
    os.environ.update(env)
    try:
        ctx = P()
        assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(ctx)) == 0, lib.mic_last_error()
    finally:
        for k in env:
            del os.environ[k]
    rng = np.random.default_rng(1234 + keeps)
    shapes = [((300, 200), (300, 260)), ((300, 200), (390, 200)), ((513, 259), (513, 400)), ((130, 70), (260, 70)),
              ((64, 64), (64, 100)), ((1000, 37), (1300, 37)), ((37, 1000), (37, 1300)), ((257, 255), (257, 140)),
              ((255, 257), (131, 257)), ((17, 16), (17, 33)), ((16, 17), (31, 17)), ((640, 480), (640, 481)),
              ((640, 480), (641, 480)), ((33, 900), (33, 899))]
    marched = 0
    for i, ((sw, sh), (dw, dh)) in enumerate(shapes):
        src = _alpha(rng, rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8), i % 3)
        if i % 4 == 3:  # a cutout with transparent margins: zero bands, tiles that only see zero bands
            src[: sh // 3, :, 3] = 0
            src[:, sw - sw // 4:, 3] = 0
        objs = {5: src}
        atlas = _atlas(lib, ctx, objs)
        pl = [{"object_id": 5, "box": [0, 0, dw, dh]}]
        for filt in (nat.LANCZOS, nat.BILINEAR):
            for bg in ((0, 0, 0, 0), (9, 200, 33, 255)):
                got, st = _composite(lib, ctx, nat, atlas, (dw, dh), bg, pl, filt)
                b = np.empty((dh, dw, 4), np.uint8)
                b[:] = bg
                want = oracle.composite(b, objs, pl, filt)
                assert np.array_equal(got, want), ((sw, sh), (dw, dh), filt, bg, st)
                marched += st["marched_layers"]
        assert lib.mic_atlas_destroy(atlas) == 0
    assert marched >= 4 * (len(shapes) - 3), marched  # (the smallest ones stay with the tile kernel)
    # the three classes in one call, boxes hanging over the canvas edges
    objs = {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(40, 300)), int(rng.integers(30, 260)), ["binary", "soft"][i % 2])
            for i in range(6)}
    atlas = _atlas(lib, ctx, objs)
    W, H = 1023, 517
    pl = []
    for k in range(18):
        oid = 1 + k % 6
        sh, sw = objs[oid].shape[:2]
        w, h = int(sw * rng.uniform(0.6, 2.0)), int(sh * rng.uniform(0.6, 2.0))
        if k % 3 == 1:
            w = sw
        elif k % 3 == 2:
            h = sh
        x1, y1 = int(rng.integers(-w // 2, W - w // 2)), int(rng.integers(-h // 2, H - h // 2))
        pl.append({"object_id": oid, "box": [x1, y1, x1 + w, y1 + h]})
    for filt in (nat.LANCZOS, nat.BILINEAR):
        got, st = _composite(lib, ctx, nat, atlas, (W, H), (38, 73, 115, 255), pl, filt)
        b = np.empty((H, W, 4), np.uint8)
        b[:] = (38, 73, 115, 255)
        assert np.array_equal(got, oracle.composite(b, objs, pl, filt)), (filt, st)
        assert st["marched_layers"] == 18, st
    assert lib.mic_atlas_destroy(atlas) == 0
    assert lib.mic_destroy(ctx) == 0
