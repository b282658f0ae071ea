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
