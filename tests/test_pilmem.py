"""Zero-copy access to PIL pixel rows (image_transformation_amd/_pilmem.py): the fast path must give the
same bytes as np.asarray() on every kind of image the callers hand over, and refuse what it cannot verify."""
import os

import numpy as np
import pytest
from PIL import Image

from image_transformation_amd import _pilmem


def _check(im):
    W, H = im.size
    dst = np.zeros(H * W * 4, np.uint8)
    assert _pilmem.copy_to(im, dst.ctypes.data)
    assert np.array_equal(dst.reshape(H, W, 4), np.asarray(im))


@pytest.mark.parametrize("size", [(3840, 2160), (492, 492), (1, 1), (7, 3), (4399, 1885), (1, 900), (900, 1)])
def test_copy_matches_asarray(size):
    rng = np.random.default_rng(size[0] * 7 + size[1])
    a = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8)
    _check(Image.fromarray(a, "RGBA"))  # one block
    big = Image.new("RGBA", size, (1, 2, 3, 4))  # Pillow's block allocator: several blocks above 16 MB
    big.paste(Image.fromarray(a, "RGBA"))
    _check(big)
    assert sum(n for _, n in _pilmem.row_runs(big)) == size[0] * size[1] * 4


def test_opened_cropped_and_foreign_buffers(golden_dir):
    im = Image.open(os.path.join(golden_dir, "bundles", "squarespace", "background.png")).convert("RGBA")
    _check(im)
    _check(im.crop((10, 10, 200, 100)))
    buf = np.arange(40 * 30 * 4, dtype=np.uint8)
    _check(Image.frombuffer("RGBA", (40, 30), buf, "raw", "RGBA", 0, 1))
    assert _pilmem.row_runs(im.convert("RGB")) is None  # only RGBA is handled
    assert _pilmem.solid_colour(im.convert("RGB")) is None


@pytest.mark.parametrize("size", [(3840, 2160), (492, 492), (2, 1), (4399, 1885)])
def test_solid_colour_is_exact(size):
    W, H = size
    s = Image.new("RGBA", size, (38, 73, 115, 255))
    assert _pilmem.solid_colour(s) == (38, 73, 115, 255)
    for xy in ((W - 1, H - 1), (0, H - 1), (W // 2, H // 2), (W - 1, 0)):
        t = s.copy()
        px = list(t.getpixel(xy))
        px[3] ^= 1
        t.putpixel(xy, tuple(px))
        assert _pilmem.solid_colour(t) is None, xy
    assert _pilmem.solid_colour(Image.new("RGBA", size, (0, 0, 0, 0))) == (0, 0, 0, 0)


def test_row_table_and_exact_solid_scan():
    """row_table() (the cheap form used by the hot drop-in call) agrees with row_runs(); mic_host_rows_solid compares
    every byte: one differing byte anywhere (first / last pixel, a middle row, an alpha byte) is found, row ranges work,
    images in several memory blocks work."""
    import ctypes
    from image_transformation_amd import _native
    lib = _native.lib()

    def solid(im, rgba, y0=0, y1=None):
        table, W, H = _pilmem.row_table(im)
        ok = ctypes.c_int(7)
        col = (ctypes.c_uint8 * 4)(*rgba)
        assert lib.mic_host_rows_solid(ctypes.c_void_p(table), W, y0, H if y1 is None else y1, col, ctypes.byref(ok)) == 0
        return ok.value

    for size in ((492, 492), (3840, 2160), (1, 1), (63, 5), (64, 5), (65, 5), (129, 3)):
        c = (220, 238, 245, 255)
        im = Image.new("RGBA", size, c)
        tab = _pilmem.row_table(im)
        assert tab is not None and tab[1:] == size
        runs = _pilmem.row_runs(im)
        first_row = ctypes.c_uint64.from_address(tab[0]).value
        assert first_row == runs[0][0]
        assert solid(im, c) == 1 and solid(im, (220, 238, 245, 254)) == 0
        W, H = size
        for (x, y) in {(0, 0), (W - 1, H - 1), (W // 2, H // 2), (W - 1, 0), (0, H - 1)}:
            for ch in range(4):
                px = list(c)
                px[ch] ^= 1
                im.putpixel((x, y), tuple(px))
                assert solid(im, c) == 0, (size, x, y, ch)
                if H > 2 and 0 < y < H - 1:
                    assert solid(im, c, 0, y) == 1 and solid(im, c, y + 1, H) == 1 and solid(im, c, y, y + 1) == 0
                im.putpixel((x, y), c)
        assert solid(im, c) == 1
    assert _pilmem.row_table(Image.new("RGB", (4, 4))) is None
    assert _pilmem.solid_colour(Image.new("RGBA", (300, 200), (1, 2, 3, 4))) == (1, 2, 3, 4)


def test_same_pixels_is_an_exact_byte_comparison():
    rng = np.random.default_rng(12)
    for size in ((7, 5), (492, 492), (2300, 2100)):
        a = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8)
        im1 = Image.fromarray(a, "RGBA")
        im2 = Image.new("RGBA", size)
        im2.paste(im1)  # other memory layout (Pillow's block allocator above 16 MB)
        assert _pilmem.same_pixels(im1, im2) and _pilmem.same_pixels(im2, im1.copy())
        for (x, y) in ((0, 0), (size[0] - 1, size[1] - 1), (size[0] // 2, size[1] // 2)):
            px = list(im2.getpixel((x, y)))
            px[3] ^= 1
            im2.putpixel((x, y), tuple(px))
            assert not _pilmem.same_pixels(im1, im2)
            px[3] ^= 1
            im2.putpixel((x, y), tuple(px))
        assert _pilmem.same_pixels(im1, im2)
    assert not _pilmem.same_pixels(Image.new("RGBA", (4, 4)), Image.new("RGBA", (4, 5)))
    assert not _pilmem.same_pixels(Image.new("RGBA", (4, 4)), Image.new("RGB", (4, 4)))


class _FakeCore:
    def __init__(self, capsule):
        self.ptr = capsule


class _FakeImage:
    """What _pilmem reads of a PIL image (mode, size, load, getpixel, im.ptr), over a hand-built ImagingMemoryInstance."""

    def __init__(self, pixels, capsule):
        self.mode, self.size, self._px, self.im, self.readonly = "RGBA", (pixels.shape[1], pixels.shape[0]), pixels, _FakeCore(capsule), 0

    def load(self):
        return None

    def getpixel(self, xy):
        return tuple(int(v) for v in self._px[xy[1], xy[0]])


def _synthetic_imaging(pixels, layout, blocks=2):
    """A struct laid out like ImagingMemoryInstance of the given _pilmem._LAYOUTS entry (bands, xsize, ysize, char **image,
    pixelsize, linesize at those offsets; everything else junk), its rows in `blocks` separately allocated pieces, wrapped
    in a PyCapsule named like Pillow's."""
    import ctypes
    H, W = pixels.shape[:2]
    o_bands, o_x, o_y, o_image, o_px, o_line = layout
    keep = []
    rows = (ctypes.c_uint64 * H)()
    per = -(-H // blocks)
    for b0 in range(0, H, per):
        piece = pixels[b0:b0 + per].copy()  # a block of its own, as Pillow's allocator hands them out
        keep.append(piece)
        for y in range(piece.shape[0]):
            rows[b0 + y] = piece.ctypes.data + y * W * 4
    struct_bytes = (ctypes.c_uint8 * 128)(*([0xAB] * 128))
    base = ctypes.addressof(struct_bytes)
    for off, val in ((o_bands, 4), (o_x, W), (o_y, H), (o_px, 4), (o_line, 4 * W)):
        ctypes.c_int32.from_address(base + off).value = val
    ctypes.c_uint64.from_address(base + o_image).value = ctypes.addressof(rows)
    new = ctypes.pythonapi.PyCapsule_New
    new.restype, new.argtypes = ctypes.py_object, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
    name = ctypes.c_char_p(b"PIL Imaging")
    keep += [rows, struct_bytes, name]
    return new(base, name, None), keep


@pytest.mark.parametrize("which", [0, 1])
def test_both_struct_layouts_on_a_synthetic_imaging_struct(which, monkeypatch):
    """Only one Pillow is installed here (12.x: layout 0); the reference pins 11.3.0 (layout 1: the mode is a char[7]
    in front of the fields instead of a 4-byte enum).  Both layouts are exercised against a hand-built struct: the rows
    are found, copied and compared exactly, the report names the layout, and a struct that fits neither is refused."""
    rng = np.random.default_rng(5 + which)
    px = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    monkeypatch.setattr(_pilmem, "_good_layout", None)
    monkeypatch.setattr(_pilmem, "counters", {"zero_copy": 0, "fallback": 0})
    cap, keep = _synthetic_imaging(px, _pilmem._LAYOUTS[which])
    fake = _FakeImage(px, cap)
    runs = _pilmem.row_runs(fake)
    assert runs is not None and sum(n for _, n in runs) == px.size and 1 <= len(runs) <= 2
    assert _pilmem.path_report()["layout"] == _pilmem._LAYOUT_NAMES[which]
    dst = np.zeros(px.size, np.uint8)
    assert _pilmem.copy_to(fake, dst.ctypes.data) and np.array_equal(dst.reshape(px.shape), px)
    tab = _pilmem.row_table(fake)
    assert tab is not None and tab[1:] == (53, 37)
    cap2, keep2 = _synthetic_imaging(px.copy(), _pilmem._LAYOUTS[which], blocks=3)
    other = _FakeImage(px.copy(), cap2)
    assert _pilmem.same_pixels(fake, other)
    assert _pilmem.path_report()["zero_copy"] >= 3 and _pilmem.path_report()["fallback"] == 0
    # a struct of neither layout (fields shifted by 4 bytes): refused, counted as a fallback, nothing read through it
    monkeypatch.setattr(_pilmem, "_good_layout", None)
    shifted = tuple(o + 4 for o in _pilmem._LAYOUTS[which])
    if shifted not in _pilmem._LAYOUTS:
        cap3, keep3 = _synthetic_imaging(px, shifted)
        bad = _FakeImage(px, cap3)
        assert _pilmem.row_runs(bad) is None and _pilmem.row_table(bad) is None
        assert _pilmem.path_report()["fallback"] >= 2 and _pilmem.path_report()["layout"] is None
    del keep, keep2
