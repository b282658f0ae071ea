"""libmic's PNG writer (csrc/png_encode.cpp, image_transformation_amd/png.py): host-only, so it is tested on the CPU.
Every file is decoded again by Pillow (the reference's readers: Image.open in macro_placement_test.py:1510 and any
viewer of the artifact tree) and, independently, its IDAT stream by zlib and its chunks' CRCs by hand."""
import io
import os
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from image_transformation_amd import png as mic_png


def _images():
    rng = np.random.default_rng(77)
    out = {}
    out["noise"] = rng.integers(0, 256, (97, 131, 4), dtype=np.uint8)
    solid = np.empty((120, 200, 4), np.uint8)
    solid[:] = (220, 238, 245, 255)
    out["solid"] = solid
    canvas = solid.copy()
    canvas[20:70, 30:110] = rng.integers(0, 256, (50, 80, 4), dtype=np.uint8)
    canvas[80:100, 5:190, :3] = (np.arange(185)[None, :, None] * np.array([1, 2, 3])[None, None, :]) % 256
    out["canvas_like"] = canvas
    yy, xx = np.mgrid[0:300, 0:257]
    photo = np.stack([(xx * 0.7 + yy * 0.2) % 256, (128 + 90 * np.sin(xx / 17.0) * np.cos(yy / 23.0)), (xx ^ yy) % 256,
                      np.where((xx - 128) ** 2 + (yy - 150) ** 2 < 110 ** 2, 255, 0)], axis=2)
    out["photo_like"] = (photo + rng.integers(-3, 4, photo.shape)).clip(0, 255).astype(np.uint8)
    out["one_px"] = np.array([[[1, 2, 3, 4]]], np.uint8)
    out["one_row"] = rng.integers(0, 256, (1, 999, 4), dtype=np.uint8)
    out["one_col"] = rng.integers(0, 256, (777, 1, 4), dtype=np.uint8)
    out["zeros"] = np.zeros((64, 64, 4), np.uint8)
    tall = np.zeros((2500, 37, 4), np.uint8)
    tall[::3] = 255
    out["stripes"] = tall
    rep = np.tile(rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), (40, 50, 1))  # long-distance matches
    out["tiled"] = rep
    return out


def _check_structure(data: bytes, shape):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, kinds = 8, b"", []
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(kind + body) == crc, kind
        kinds.append(kind)
        if kind == b"IHDR":
            w, h, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", body)
            assert (h, w) == shape[:2] and (depth, ctype, comp, filt, inter) == (8, 6, 0, 0, 0)
        if kind == b"IDAT":
            idat += body
        pos += 12 + n
    assert pos == len(data) and kinds[0] == b"IHDR" and kinds[-1] == b"IEND"
    raw = zlib.decompress(idat)  # checks the Adler-32 of the whole filtered stream too
    assert len(raw) == shape[0] * (shape[1] * 4 + 1)
    assert set(raw[::shape[1] * 4 + 1]) <= {0, 1, 2}
    return len(idat)


@pytest.mark.parametrize("level", [0, 1])
@pytest.mark.parametrize("threads", [1, 3, 0])
def test_png_round_trip(level, threads, tmp_path):
    for name, a in _images().items():
        data = mic_png.encode(a, level=level, threads=threads)
        _check_structure(data, a.shape)
        back = np.array(Image.open(io.BytesIO(data)).convert("RGBA"))
        assert Image.open(io.BytesIO(data)).mode == "RGBA"
        assert np.array_equal(back, a), (name, level, threads)
        path = tmp_path / f"{name}.png"
        mic_png.save(Image.fromarray(a, "RGBA"), path, level=level, threads=threads)  # PIL in: rows read in place
        assert np.array_equal(np.array(Image.open(path)), a), name
        assert path.read_bytes() == data  # deterministic: the same bytes for the same image, level and threads


def test_png_many_threads_and_strided_input():
    rng = np.random.default_rng(5)
    big = np.empty((1100, 1300, 4), np.uint8)
    big[:] = (38, 73, 115, 255)
    for k in range(12):
        y, x = int(rng.integers(0, 900)), int(rng.integers(0, 1000))
        big[y:y + 200, x:x + 300] = rng.integers(0, 256, (200, 300, 4), dtype=np.uint8)
    for threads in (2, 7, 16):
        data = mic_png.encode(big, threads=threads)
        assert _check_structure(data, big.shape) < big.nbytes
        assert np.array_equal(np.array(Image.open(io.BytesIO(data))), big)
    view = big[100:900:2, 50:1250]  # row stride != width * 4 (every other row): base + stride form
    assert np.array_equal(np.array(Image.open(io.BytesIO(mic_png.encode(view)))), view)
    col = big[:, ::3]  # pixel stride != 4: copied to a contiguous array first
    assert np.array_equal(np.array(Image.open(io.BytesIO(mic_png.encode(col)))), col)


def test_png_large_pil_image_in_several_memory_blocks():
    """Pillow stores images above 16 MB in several blocks: the writer walks Pillow's row-pointer table."""
    rng = np.random.default_rng(6)
    a = np.empty((2300, 2100, 4), np.uint8)  # 19.3 MB
    a[:] = (250, 250, 250, 255)
    a[500:1500, 400:1800] = rng.integers(0, 256, (1000, 1400, 4), dtype=np.uint8)
    im = Image.fromarray(a, "RGBA")
    data = mic_png.encode(im)
    assert np.array_equal(np.array(Image.open(io.BytesIO(data))), a)
    assert len(data) < 0.4 * a.nbytes  # the flat 70 % of the image costs next to nothing


def test_png_compresses_like_a_fast_zlib_level():
    imgs = _images()
    for name in ("canvas_like", "photo_like", "tiled", "solid"):
        a = imgs[name]
        mine = len(mic_png.encode(a))
        buf = io.BytesIO()
        Image.fromarray(a, "RGBA").save(buf, format="PNG", compress_level=1)
        assert mine <= 1.5 * len(buf.getvalue()) + 200, (name, mine, len(buf.getvalue()))


def test_png_errors():
    from image_transformation_amd._native import MicError
    with pytest.raises(ValueError):
        mic_png.encode(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(ValueError):
        mic_png.encode(np.zeros((0, 4, 4), np.uint8))
    with pytest.raises(MicError, match="cannot open"):
        mic_png.save(np.zeros((4, 4, 4), np.uint8), "/nonexistent-dir/x.png")
    rgb = Image.new("RGB", (5, 4), (1, 2, 3))  # other modes are converted like .convert("RGBA")
    assert np.array_equal(np.array(Image.open(io.BytesIO(mic_png.encode(rgb)))), np.array(rgb.convert("RGBA")))


def test_save_like_pil_only_takes_png_names(tmp_path):
    """The reference's helpers call image.save(path) and PIL picks the format from the name: only *.png goes through
    libmic's writer."""
    im = Image.fromarray(_images()["canvas_like"], "RGBA")
    mic_png.save_like_pil(im, tmp_path / "a.PNG")
    assert (tmp_path / "a.PNG").read_bytes() == mic_png.encode(im)
    mic_png.save_like_pil(im, tmp_path / "a.bmp")
    assert Image.open(tmp_path / "a.bmp").format == "BMP"
    mic_png.save_like_pil(im, tmp_path / "b.png", compress_level=9)  # explicit PIL options: PIL's encoder
    assert (tmp_path / "b.png").read_bytes() != mic_png.encode(im)
    assert np.array_equal(np.array(Image.open(tmp_path / "b.png")), np.array(im))


def test_async_saves_on_the_library_threads(tmp_path):
    """mic_png_write_async / mic_png_wait: many saves queued at once, each file identical to the synchronous writer's,
    errors reported by wait(), an id waited for twice refused."""
    from image_transformation_amd._native import MicError, lib
    imgs = _images()
    pend = []
    for rep in range(3):
        for name, a in imgs.items():
            im = Image.fromarray(a, "RGBA") if rep % 2 == 0 else a
            pend.append((name, a, tmp_path / f"{name}_{rep}.png", mic_png.save_async(im, tmp_path / f"{name}_{rep}.png", threads=1 + rep)))
    for name, a, path, p in pend:
        p.wait()
        p.wait()  # idempotent on the Python object
        assert np.array_equal(np.array(Image.open(path)), a), name
    assert (tmp_path / "noise_0.png").read_bytes() == mic_png.encode(imgs["noise"], threads=1)
    bad = mic_png.save_async(imgs["solid"], "/nonexistent-dir/x.png")
    with pytest.raises(MicError, match="cannot open"):
        bad.wait()
    assert lib().mic_png_wait(987654321) != 0  # unknown job


def test_png_seeded_sweep_of_shapes_and_contents():
    """120 seeded images the fixed kinds above do not contain: every size class from 1 px to a few hundred on each
    side (stripe boundaries fall everywhere), five content classes, every level / thread setting -- each decoded by
    Pillow and its stream structure checked."""
    rng = np.random.default_rng(20261004)
    for n in range(120):
        h = int(rng.choice([1, 2, 3, 7, 16, 17, 63, 64, 65, 200, 333]))
        w = int(rng.choice([1, 2, 5, 31, 32, 33, 127, 128, 129, 492, 700]))
        kind = n % 5
        if kind == 0:
            a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        elif kind == 1:  # few colours, long runs
            a = rng.integers(0, 3, (h, w, 1), dtype=np.uint8).repeat(4, axis=2) * 85
        elif kind == 2:  # smooth gradients (Sub / Up filters matter)
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([(xx + yy) % 256, (2 * xx) % 256, (3 * yy) % 256, np.full_like(xx, 255)], axis=2).astype(np.uint8)
        elif kind == 3:  # a cutout: transparent margins around noise
            a = np.zeros((h, w, 4), np.uint8)
            a[h // 4:h - h // 4, w // 4:w - w // 4] = rng.integers(0, 256, (h - 2 * (h // 4), w - 2 * (w // 4), 4), dtype=np.uint8)
        else:  # a repeated row (matches at distance = one row)
            a = np.tile(rng.integers(0, 256, (1, w, 4), dtype=np.uint8), (h, 1, 1))
        level, threads = int(rng.integers(0, 2)), int(rng.choice([0, 1, 2, 5]))
        data = mic_png.encode(a, level=level, threads=threads)
        _check_structure(data, a.shape)
        back = np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"))
        assert back.shape == a.shape and np.array_equal(back, a), (n, h, w, kind, level, threads)
