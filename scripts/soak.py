"""Randomised differential soak: HIP path vs the oracle on many random composites and resizes.
Not part of the test suite (minutes); run on the GPU box:  python scripts/soak.py [seconds] [seed]
With MIC_RS_MARCH_MIN_UNITS=0 in the environment every resampled layer that qualifies takes the marching kernel
(by default only calls with >= 512 work units do; small calls take the tile kernel)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import oracle
from image_transformation_amd import _native, synthetic
from PIL import Image
from image_transformation_amd.background_resizing import median_colors_device
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, composite, composite_device, coerce_placements

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.default_rng(seed)
ctx = _native.context(); lib = _native.lib(); P = ctypes.c_void_p
t_end = time.time() + budget
n_comp = n_rs = n_pil = n_med = n_multi = 0
t_say = time.time() + 30
while time.time() < t_end:
    if time.time() > t_say:  # progress line: long silent GPU runs are taken for hung
        print(f"... {n_comp} composites, {n_pil} PIL composites, {n_med} medians, {n_rs} resizes so far", flush=True)
        t_say = time.time() + 30
    # ---- a random atlas + a few canvases
    objs = {i + 1: synthetic.make_cutout(rng, int(rng.integers(1, 260)), int(rng.integers(1, 200)),
                                         ["binary", "soft"][int(rng.integers(0, 2))]) for i in range(int(rng.integers(1, 9)))}
    atlas = Atlas(objs)
    for _ in range(6):
        W = int(rng.choice([1, 2, 3, 5, 64, 255, 256, 257, 511, 1000, 1023, 1024, 1025, 1366, 2049, 4099]))
        H = int(rng.integers(1, 80))
        pl = []
        for _k in range(int(rng.integers(0, 20))):
            oid = int(rng.integers(1, len(objs) + 2))
            sh, sw = objs.get(oid, objs[1]).shape[:2]
            r = rng.random()
            if r < 0.25:
                sw, sh = int(sw * rng.uniform(0.2, 2.5)), int(sh * rng.uniform(0.2, 2.5))
            elif r < 0.3:
                sw, sh = int(rng.integers(0, 3)), int(rng.integers(0, 3))
            elif r < 0.4:  # one axis kept (the pass Pillow skips; the lane kernel's one-digit forms)
                if rng.random() < 0.5:
                    sw = int(sw * rng.uniform(0.5, 2.5))
                else:
                    sh = int(sh * rng.uniform(0.5, 2.5))
            x1, y1 = int(rng.integers(-sw - 2, W + 2)), int(rng.integers(-sh - 2, H + 2))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        kind = int(rng.integers(0, 3))
        if kind == 0:
            col = (38, 73, 115, 255)
        elif kind == 1:
            col = tuple(int(v) for v in rng.integers(0, 256, 4))
        if kind < 2:
            bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = col
            canvas = SolidCanvas((W, H), col)
        else:
            bg_np = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            if rng.random() < 0.5: bg_np[:, :, 3] = 255
            canvas = torch.from_numpy(bg_np).cuda()
        off = 4 * int(rng.integers(0, 1100))
        big = torch.zeros(W * H * 4 + 8192, dtype=torch.uint8, device="cuda")
        out = big[off:off + W * H * 4].view(H, W, 4)
        filt = int(rng.integers(0, 2))
        got = composite_device(atlas, [canvas], [coerce_placements(atlas, pl)], outs=[out], filter=filt)[0].cpu().numpy()
        want = oracle.composite(bg_np, objs, pl, filt)
        if not np.array_equal(got, want) or big[:off].any() or big[off + W * H * 4:].any():
            print("COMPOSITE MISMATCH", dict(seed=seed, W=W, H=H, kind=kind, off=off, filt=filt, pl=pl,
                                             sizes={k: v.shape for k, v in objs.items()}))
            sys.exit(1)
        n_comp += 1
    # ---- round 4: ONE plan over several atlases (CompositeBatch(atlas_of=...), mic_plan_create with n_atlases > 1): the
    # same ids name other cutouts in the second atlas; every canvas against the oracle on its own atlas' objects
    objs2 = {k: synthetic.make_cutout(rng, int(rng.integers(1, 200)), int(rng.integers(1, 160)), ["binary", "soft"][int(rng.integers(0, 2))]) for k in objs}
    atlas2 = Atlas(objs2)
    both, n_cv = [objs, objs2], int(rng.integers(2, 7))
    a_of = [int(rng.integers(0, 2)) for _ in range(n_cv)]
    cvs, bgs, pls = [], [], []
    for i in range(n_cv):
        W, H = int(rng.choice([3, 64, 257, 1000, 1025, 2049])), int(rng.integers(1, 60))
        pl = []
        for _k in range(int(rng.integers(0, 12))):
            oid = int(rng.integers(1, len(objs) + 2))
            sh, sw = both[a_of[i]].get(oid, both[a_of[i]][1]).shape[:2]
            if rng.random() < 0.4:
                sw, sh = max(1, int(sw * rng.uniform(0.3, 2.2))), max(1, int(sh * rng.uniform(0.3, 2.2)))
            x1, y1 = int(rng.integers(-sw - 1, W + 1)), int(rng.integers(-sh - 1, H + 1))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        col = tuple(int(v) for v in rng.integers(0, 256, 4)) if rng.random() < 0.5 else (38, 73, 115, 255)
        bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = col
        cvs.append(SolidCanvas((W, H), col)); bgs.append(bg_np); pls.append(pl)
    filt = int(rng.integers(0, 2))
    mplan = CompositeBatch([atlas, atlas2], cvs, [coerce_placements(both[a_of[i]], pls[i]) for i in range(n_cv)], filter=filt, atlas_of=a_of)
    for i, o in enumerate(mplan.run()):
        if not np.array_equal(o.cpu().numpy(), oracle.composite(bgs[i], both[a_of[i]], pls[i], filt)):
            print("MULTI-ATLAS MISMATCH", dict(seed=seed, canvas=i, atlas_of=a_of, filt=filt, pl=pls[i]))
            sys.exit(1)
        n_multi += 1
    del mplan, atlas2
    # ---- round 3: the PIL-level drop-in (speculative solid background, layer records in the kernel arguments, small
    # canvases written straight into pinned host memory, event-waited downloads) and the batched strided median
    pil_objs = {k: Image.fromarray(v, "RGBA") for k, v in objs.items()}
    for _ in range(3):
        W, H = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        kind = int(rng.integers(0, 4))
        col = tuple(int(v) for v in rng.integers(0, 256, 4)) if kind != 0 else (220, 238, 245, 255)
        bg_np = np.empty((H, W, 4), np.uint8); bg_np[:] = col
        if kind == 2:  # solid but for ONE byte somewhere: the speculative launch must be thrown away
            bg_np[int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 3:
            bg_np = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        pl = []
        for _k in range(int(rng.integers(1, 12))):
            oid = int(rng.integers(1, len(objs) + 2))
            sh, sw = objs.get(oid, objs[1]).shape[:2]
            if rng.random() < 0.3:
                sw, sh = max(1, int(sw * rng.uniform(0.3, 2.0))), max(1, int(sh * rng.uniform(0.3, 2.0)))
            x1, y1 = int(rng.integers(-sw, W + 1)), int(rng.integers(-sh, H + 1))
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        bg_im = Image.fromarray(bg_np, "RGBA")
        got = np.array(composite(bg_im, pil_objs, pl))
        if not np.array_equal(got, oracle.composite(bg_np, objs, pl)) or not np.array_equal(np.array(bg_im), bg_np):
            print("PIL COMPOSITE MISMATCH", dict(seed=seed, W=W, H=H, kind=kind, pl=pl, sizes={k: v.shape for k, v in objs.items()}))
            sys.exit(1)
        n_pil += 1
    mh, mw = int(rng.integers(1, 300)), int(rng.integers(1, 400))
    m_np = rng.integers(0, 256, (mh, mw, 4), dtype=np.uint8)
    m_np[:, :, 3] = np.where(rng.random((mh, mw)) < rng.random(), 0, m_np[:, :, 3])
    m_dev = torch.from_numpy(m_np).cuda()
    views, want_m = [], []
    for _k in range(int(rng.integers(1, 22))):
        y0, x0 = int(rng.integers(0, mh)), int(rng.integers(0, mw))
        y1, x1 = int(rng.integers(y0 + 1, mh + 1)), int(rng.integers(x0 + 1, mw + 1))
        views.append(m_dev[y0:y1, x0:x1])
        want_m.append(oracle.median_rgb(np.ascontiguousarray(m_np[y0:y1, x0:x1])))
    if median_colors_device(views) != want_m:
        print("MEDIAN BATCH MISMATCH", dict(seed=seed, shape=(mh, mw), n=len(views)))
        sys.exit(1)
    n_med += len(views)
    # ---- one packed image through the single-image median: structured content (flat runs, few colours, mixed alpha
    # classes) at sizes on both sides of the one-launch / two-launch rule
    from image_transformation_amd.background_resizing import median_color_device
    big = rng.random() < 0.15
    bh, bw = (int(rng.integers(1500, 3000)), int(rng.integers(1500, 3200))) if big else (int(rng.integers(1, 700)), int(rng.integers(1, 1100)))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        b_np = rng.integers(0, 256, (bh, bw, 4), dtype=np.uint8)
    elif kind == 1:  # one colour with specks
        b_np = np.empty((bh, bw, 4), np.uint8); b_np[:] = rng.integers(0, 256, 4, dtype=np.uint8)
        sp = rng.random((bh, bw)) < 0.002
        b_np[sp] = rng.integers(0, 256, (int(sp.sum()), 4), dtype=np.uint8)
    elif kind == 2:  # a palette of a few colours in runs of random length (chunks of 256 equal pixels and broken ones)
        pal = rng.integers(0, 256, (int(rng.integers(1, 5)), 4), dtype=np.uint8)
        pal[rng.random(len(pal)) < 0.4, 3] = 0
        runs = np.repeat(rng.integers(0, len(pal), bh * bw // 64 + 2), rng.integers(1, 700, bh * bw // 64 + 2))[:bh * bw]
        b_np = pal[runs].reshape(bh, bw, 4)
    elif kind == 3:  # everything transparent: the all-pixels fallback
        b_np = rng.integers(0, 256, (bh, bw, 4), dtype=np.uint8); b_np[:, :, 3] = 0
    else:  # smooth gradient + a little noise (photo-like), alpha mostly opaque
        gx = np.linspace(0, 255, bw)[None, :, None]; gy = np.linspace(0, 255, bh)[:, None, None]
        b_np = np.clip(np.concatenate([gx + 0 * gy, gy + 0 * gx, (gx + gy) / 2, 255 + 0 * gx + 0 * gy], axis=2)
                       + rng.normal(0, 3, (bh, bw, 4)), 0, 255).astype(np.uint8)
    b_np = np.ascontiguousarray(b_np)
    if median_color_device(torch.from_numpy(b_np).cuda()) != oracle.median_rgb(b_np):
        print("MEDIAN MISMATCH", dict(seed=seed, shape=(bh, bw), kind=kind))
        sys.exit(1)
    n_med += 1
    # ---- a random resize
    sw, sh = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    dw, dh = max(1, int(sw * rng.uniform(0.05, 3.0))), max(1, int(sh * rng.uniform(0.05, 3.0)))
    src = synthetic.make_cutout(rng, sw, sh, ["binary", "soft"][int(rng.integers(0, 2))])
    dev = torch.from_numpy(src).cuda()
    dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
    filt = int(rng.integers(0, 2))
    _native.check(lib.mic_resize(ctx.handle, P(dev.data_ptr()), sw, sh, P(dst.data_ptr()), dw, dh, filt, P(ctx.stream_ptr())))
    if not np.array_equal(dst.cpu().numpy(), oracle.resize(src, (dw, dh), filt)):
        print("RESIZE MISMATCH", dict(seed=seed, src=(sw, sh), dst=(dw, dh), filt=filt))
        sys.exit(1)
    # the same resize as the only layer of a composite onto a transparent canvas (the plan path: marching or tile
    # kernel by the routing rule); alpha-over onto alpha 0 returns the layer wherever its alpha is > 0
    a1 = Atlas({1: src})
    got = composite_device(a1, [SolidCanvas((dw, dh), (0, 0, 0, 0))], [coerce_placements(a1, [{"object_id": 1, "box": [0, 0, dw, dh]}])],
                           filter=filt)[0].cpu().numpy()
    want = oracle.composite(np.zeros((dh, dw, 4), np.uint8), {1: src}, [{"object_id": 1, "box": [0, 0, dw, dh]}], filt)
    if not np.array_equal(got, want):
        print("PLAN RESIZE MISMATCH", dict(seed=seed, src=(sw, sh), dst=(dw, dh), filt=filt))
        sys.exit(1)
    n_rs += 1
print(f"soak ok: {n_comp} composites, {n_multi} multi-atlas canvases, {n_pil} PIL drop-in composites, {n_med} batched medians, {n_rs} resizes, seed {seed}, {budget:.0f} s, MIC_RS_MARCH_MIN_UNITS={os.environ.get('MIC_RS_MARCH_MIN_UNITS')}")
