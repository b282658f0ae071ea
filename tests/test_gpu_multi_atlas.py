"""Plans over SEVERAL atlases (CompositeBatch(..., atlas_of=...) -> mic_plan_create / mic_composite_batch with
n_atlases > 1): the path bench.py's `cold_inputs` leg times.  Different cutouts live under the SAME object ids in the
atlases, so a canvas that read the wrong atlas cannot pass.  Every output is compared with the oracle run on that
canvas' own objects (compositor.py:12-21 semantics per canvas)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cases  # noqa: E402
import oracle  # noqa: E402

P = ctypes.c_void_p


def _bundle(rng, n, alpha):
    """ids 1..n, sizes differ per bundle (same ids, other pixels AND other sizes)."""
    return {i + 1: cases.synthetic.make_cutout(rng, int(rng.integers(24, 140)), int(rng.integers(18, 110)), alpha)
            for i in range(n)}


def _placements(rng, objs, W, H, n, lanczos_share):
    pl = []
    for _ in range(n):
        oid = int(rng.integers(1, len(objs) + 2))  # one id past the end: unknown -> skipped
        sh, sw = objs.get(oid, objs[1]).shape[:2]
        if rng.random() < lanczos_share:
            sw, sh = max(1, int(sw * rng.uniform(0.4, 2.2))), max(1, int(sh * rng.uniform(0.4, 2.2)))
        x1, y1 = int(rng.integers(-sw // 2, W)), int(rng.integers(-sh // 2, H))
        pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
    return pl


def _canvas_kinds(rng, sizes):
    """(numpy background, canvas argument factory) per canvas: opaque solid, translucent solid, image, aligned and
    unaligned widths -- all four kernel classes in one plan."""
    import torch
    from image_transformation_amd.compositor import SolidCanvas
    out = []
    for i, (W, H) in enumerate(sizes):
        kind = i % 3
        if kind == 0:
            col = (38, 73, 115, 255)
        elif kind == 1:
            col = tuple(int(v) for v in rng.integers(0, 256, 4))
        if kind < 2:
            bg = np.empty((H, W, 4), np.uint8)
            bg[:] = col
            out.append((bg, SolidCanvas((W, H), col)))
        else:
            bg = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            if i % 2:
                bg[:, :, 3] = 255
            out.append((bg, torch.from_numpy(bg).cuda()))
    return out


@pytest.mark.parametrize("route", ["tile", "march", "lane"])
def test_composite_batch_over_three_atlases(route, monkeypatch):
    import torch
    from image_transformation_amd import _native
    from image_transformation_amd.compositor import Atlas, CompositeBatch, coerce_placements, pack_blob
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X")
    march = route != "tile"
    rng = np.random.default_rng(77001 + march)
    lib = _native.lib()
    # a context whose every qualifying layer takes the tile kernel / the marching kernel (planar copies per atlas) /
    # the lane kernel (tiled planar copies per atlas)
    for k, v in {"tile": {"MIC_RS_LANE": "0"}, "march": {"MIC_RS_LANE": "0", "MIC_RS_MARCH_MIN_UNITS": "0"},
                 "lane": {"MIC_RS_LANE_MIN_SLOTS": "0"}}[route].items():
        monkeypatch.setenv(k, v)
    ctx = _native.Context(torch.cuda.current_device())  # a context of this test's own (mic_create reads the setting)
    bundles = [_bundle(rng, 5, "soft"), _bundle(rng, 7, "binary"), _bundle(rng, 4, "soft")]
    atlases = []
    for b in bundles:
        a = Atlas.__new__(Atlas)
        a.ctx = ctx
        host = pack_blob(b)
        blob = host.cuda()
        torch.cuda.synchronize()
        a._init_from_blob(blob, header=host.numpy()[:32 + 32 * len(b)])
        atlases.append(a)
    sizes = [(333, 97), (256, 64), (1024, 40), (515, 77), (64, 64), (1500, 33), (4, 9), (777, 51), (400, 300)]
    atlas_of = [0, 1, 2, 1, 0, 2, 2, 1, 0]
    kinds = _canvas_kinds(rng, sizes)
    for filt in (_native.LANCZOS, _native.BILINEAR):
        pls = [_placements(rng, bundles[atlas_of[i]], W, H, int(rng.integers(0, 14)), 0.5) for i, (W, H) in enumerate(sizes)]
        rows = [coerce_placements(atlases[atlas_of[i]], pl) for i, pl in enumerate(pls)]
        plan = CompositeBatch(atlases, [k[1] for k in kinds], rows, filter=filt, atlas_of=atlas_of)
        st = plan.stats()
        assert st["resampled_layers"] > 0 and st["identity_layers"] > 0
        if march:
            assert st["marched_layers"] > 0
        for rep in range(2):  # the second run takes the cached job table when the outputs repeat
            outs = plan.run()
            torch.cuda.synchronize()
            for i, o in enumerate(outs):
                want = oracle.composite(kinds[i][0], bundles[atlas_of[i]], pls[i], filt)
                assert np.array_equal(o.cpu().numpy(), want), (filt, rep, i, sizes[i], atlas_of[i])
        del plan
    del atlases
    assert lib.mic_destroy(ctx.handle) == 0


def test_raw_mic_composite_batch_three_atlases():
    """The same through the raw C ABI: mic_atlas_create x 3 (host pointers), placements that name their atlas by index,
    ONE mic_composite_batch call over jobs of different atlases -- and one job that mixes all three."""
    import torch
    from image_transformation_amd import _native
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X")
    lib = _native.lib()
    ctx = P()
    assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(ctx)) == 0, lib.mic_last_error()
    rng = np.random.default_rng(77100)
    bundles = [_bundle(rng, 3, "soft"), _bundle(rng, 3, "binary"), _bundle(rng, 3, "soft")]
    handles, keep = [], []
    for b in bundles:
        ids = (ctypes.c_int32 * len(b))(*b.keys())
        ws = (ctypes.c_int32 * len(b))(*[a.shape[1] for a in b.values()])
        hs = (ctypes.c_int32 * len(b))(*[a.shape[0] for a in b.values()])
        arrs = [np.ascontiguousarray(a) for a in b.values()]
        keep.append(arrs)
        ptrs = (P * len(b))(*[a.ctypes.data for a in arrs])
        h = P()
        assert lib.mic_atlas_create(ctx, len(b), ids, ws, hs, ptrs, ctypes.byref(h)) == 0, lib.mic_last_error()
        handles.append(h)
    atl = (P * 3)(*handles)
    sizes = [(301, 88), (640, 48), (97, 97), (512, 60)]
    owner = [2, 0, 1, None]  # None: a job whose placements name all three atlases
    jobs = (_native.Job * len(sizes))()
    outs, wants, parrs = [], [], []
    stream = P(torch.cuda.current_stream().cuda_stream)
    for j, (W, H) in enumerate(sizes):
        n = 9
        pa = (_native.Placement * n)()
        bg = np.empty((H, W, 4), np.uint8)
        bg[:] = (200, 10, 60, 255)
        want = bg
        for k in range(n):
            a = owner[j] if owner[j] is not None else k % 3
            pl = _placements(rng, bundles[a], W, H, 1, 0.5)[0]
            pa[k].atlas, pa[k].object_id = a, pl["object_id"]
            for c in range(4):
                pa[k].box[c] = pl["box"][c]
            want = oracle.composite(want, bundles[a], [pl])  # layer by layer: each with its own atlas' objects
        out = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        jobs[j].width, jobs[j].height, jobs[j].bg_dev = W, H, None
        for c, v in enumerate((200, 10, 60, 255)):
            jobs[j].bg_rgba[c] = v
        jobs[j].n_placements, jobs[j].placements, jobs[j].out_dev = n, pa, out.data_ptr()
        outs.append(out)
        wants.append(want)
        parrs.append(pa)
    assert lib.mic_composite_batch(ctx, 3, atl, len(sizes), jobs, 0, stream) == 0, lib.mic_last_error()
    torch.cuda.synchronize()
    for j in range(len(sizes)):
        assert np.array_equal(outs[j].cpu().numpy(), wants[j]), (j, sizes[j])
    # an atlas of another context is refused
    other = P()
    assert lib.mic_create(torch.cuda.current_device(), ctypes.byref(other)) == 0
    assert lib.mic_composite_batch(other, 3, atl, len(sizes), jobs, 0, stream) < 0
    assert b"another context" in lib.mic_last_error()
    assert lib.mic_destroy(other) == 0
    for h in handles:
        assert lib.mic_atlas_destroy(h) == 0
    assert lib.mic_destroy(ctx) == 0
