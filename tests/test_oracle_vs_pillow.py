"""The CPU oracle against LIVE Pillow / NumPy (the third-party libraries the reference's arithmetic
lives in) on fresh random inputs -- beyond the committed fixtures.

The loop below is this repo's own restatement of the reference's composite() (compositor.py:6-22:
resize to the box with LANCZOS, alpha_composite at the box origin); nothing is imported from the
reference.  Skipped when Pillow is not installed."""
import numpy as np
import pytest

import oracle

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageDraw  # noqa: E402


def _img(a):
    return Image.fromarray(np.ascontiguousarray(a), "RGBA")


def _pillow_composite(bg, objs, placements):
    canvas = _img(bg).copy()
    for p in placements:
        obj = objs.get(int(p["object_id"]))
        if obj is None:
            continue
        x1, y1, x2, y2 = [int(v) for v in p["box"]]
        o = _img(obj).resize((max(1, x2 - x1), max(1, y2 - y1)), Image.LANCZOS)
        canvas.alpha_composite(o, dest=(x1, y1)) if x1 >= 0 and y1 >= 0 else canvas.paste(
            Image.alpha_composite(canvas.crop((x1, y1, x1 + o.width, y1 + o.height)), o), (x1, y1))
    return np.array(canvas)


def _cutout(rng, w, h, mode):
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    if mode == "binary":
        a[:, :, 3] = np.where(rng.random((h, w)) < 0.4, 0, 255)
    elif mode == "edges":
        a[:, :, 3] = rng.choice(np.asarray([0, 1, 2, 127, 128, 254, 255], np.uint8), (h, w))
    return a


def test_resize_matches_pillow_random_shapes():
    rng = np.random.default_rng(314)
    for i in range(40):
        sw, sh = int(rng.integers(1, 120)), int(rng.integers(1, 90))
        dw, dh = int(rng.integers(1, 160)), int(rng.integers(1, 120))
        src = _cutout(rng, sw, sh, ["soft", "binary", "edges"][i % 3])
        for filt, pf in ((oracle.LANCZOS, Image.LANCZOS), (oracle.BILINEAR, Image.BILINEAR)):
            want = np.array(_img(src).resize((dw, dh), pf))
            assert np.array_equal(oracle.resize(src, (dw, dh), filt), want), ((sw, sh), (dw, dh), filt)


def test_composite_matches_pillow_random_layers():
    rng = np.random.default_rng(2718)
    for it in range(25):
        W, H = int(rng.integers(1, 140)), int(rng.integers(1, 100))
        objs = {k + 1: _cutout(rng, int(rng.integers(1, 70)), int(rng.integers(1, 60)), ["soft", "binary", "edges"][k % 3])
                for k in range(5)}
        bg = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        if it % 2:
            bg[:, :, 3] = 255
        pl = []
        for _ in range(int(rng.integers(0, 9))):
            oid = int(rng.integers(1, 8))  # ids 6, 7 are unknown: skipped
            sh, sw = objs.get(oid, objs[1]).shape[:2]
            if rng.random() < 0.4:
                sw, sh = max(0, int(sw * rng.uniform(0.3, 1.8))), max(0, int(sh * rng.uniform(0.3, 1.8)))
            x1, y1 = int(rng.integers(0, W + 1)), int(rng.integers(0, H + 1))  # Pillow's dest must be >= 0
            pl.append({"object_id": oid, "box": [x1, y1, x1 + sw, y1 + sh]})
        assert np.array_equal(oracle.composite(bg, objs, pl), _pillow_composite(bg, objs, pl)), (it, W, H)


def test_median_matches_numpy():
    rng = np.random.default_rng(1618)
    for it in range(30):
        h, w = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if it % 4 == 0:
            a[:, :, 3] = 0
        elif it % 4 == 1:
            a[:, :, 3] = np.where(rng.random((h, w)) < 0.5, 0, a[:, :, 3])
        elif it % 4 == 2:
            a[:, :, :3] = rng.integers(0, 3, (h, w, 3), dtype=np.uint8) * 127  # many ties
        mask = a[:, :, 3] > 0
        px = a[:, :, :3][mask] if mask.any() else a[:, :, :3].reshape(-1, 3)  # background_resizing.py:17-20
        want = tuple(int(v) for v in np.median(px, axis=0))
        assert oracle.median_rgb(a) == want, it


def test_thumbnail_size_matches_pillow():
    rng = np.random.default_rng(1414)
    for _ in range(300):
        w, h = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        im = Image.new("RGBA", (w, h))
        im.thumbnail((256, 256), Image.LANCZOS)
        assert tuple(oracle.thumbnail_size((w, h), (256, 256))) == im.size, (w, h)


def test_rect_outlines_match_imagedraw():
    rng = np.random.default_rng(1732)
    for it in range(40):
        W, H = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        n = int(rng.integers(0, 8))
        width = int(rng.integers(1, 6))
        boxes, cols = [], []
        im = Image.new("RGBA", (W, H), (0, 0, 0, 0))
        d = ImageDraw.Draw(im)
        for _ in range(n):
            x1, y1 = int(rng.integers(-10, W + 5)), int(rng.integers(-10, H + 5))
            b = [x1, y1, x1 + int(rng.integers(0, 40)), y1 + int(rng.integers(0, 30))]
            c = tuple(int(v) for v in rng.integers(0, 256, 4))
            d.rectangle(b, outline=c, width=width)
            boxes.append(b)
            cols.append(c)
        got = oracle.rect_outlines((W, H), boxes, cols, width) if n else np.zeros((H, W, 4), np.uint8)
        assert np.array_equal(got, np.array(im)), (it, W, H, width, boxes)
