"""PNG artifacts without PIL's encoder: libmic's threaded writer (csrc/png_encode.cpp) behind the reference's
`draft.save(path)` / `canvas_img.save(canvas_path)` / overlay saves (macro_placement_test.py:1428-1430, 1513-1514,
1699-1700).

    save(image, path)        PIL RGBA image, (H, W, 4) uint8 array, or device tensor -> PNG file
    encode(image) -> bytes   the same, to memory

On-disk format stays 8-bit RGBA PNG; the bytes differ from Pillow's (another deflate), the decoded pixels are identical
(tests/test_png.py re-opens every file with Pillow).  Host-side code: it needs libmic.so but no GPU.  A PIL image's
rows are read in place (Pillow's own row-pointer table, _pilmem); nothing is copied before the encoder's filter pass.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Optional, Tuple

import numpy as np
from PIL import Image

from . import _native, _pilmem

_P = ctypes.c_void_p
DEFAULT_LEVEL = 1


def _rows_of(img: Image.Image) -> Optional[Tuple[Any, int, int]]:
    """(ctypes array of H row addresses, W, H) for an RGBA PIL image whose memory _pilmem can locate."""
    runs = _pilmem.row_runs(img)
    if runs is None:
        return None
    W, H = img.size
    line = 4 * W
    table = np.empty(H, np.uint64)
    y = 0
    for addr, n in runs:
        k = n // line
        table[y:y + k] = addr + np.arange(k, dtype=np.uint64) * np.uint64(line)
        y += k
    if y != H:
        return None
    return table, W, H


def _source(image):
    """-> (kind, keepalive, pointer-ish, W, H, stride): kind 'rows' (pointer table) or 'flat' (base + stride)."""
    if isinstance(image, Image.Image):
        if image.mode != "RGBA":
            image = image.convert("RGBA")
        got = _rows_of(image)
        if got is not None:
            table, W, H = got
            return "rows", (image, table), table.ctypes.data, W, H, 0
        arr = np.asarray(image, dtype=np.uint8)
    elif hasattr(image, "detach") and hasattr(image, "cpu"):  # a torch tensor (device or host)
        arr = image.detach().cpu().numpy()
    else:
        arr = np.asarray(image)
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 4:
        raise ValueError("PNG writer takes RGBA8 images: (H, W, 4) uint8")
    if arr.strides[2] != 1 or arr.strides[1] != 4 or arr.strides[0] < arr.shape[1] * 4:
        arr = np.ascontiguousarray(arr)
    H, W = int(arr.shape[0]), int(arr.shape[1])
    if H == 0 or W == 0:
        raise ValueError("cannot write an empty image")
    return "flat", arr, arr.ctypes.data, W, H, int(arr.strides[0])


def save(image, path, level: int = DEFAULT_LEVEL, threads: int = 0) -> None:
    """Write `image` to `path` as a PNG (RGBA8).  threads=0: sized by the image (one worker per ~384 KiB of pixels,
    at most 16); threads=1 keeps the call on the calling thread (callers that already run one save per pool thread)."""
    kind, keep, ptr, W, H, stride = _source(image)
    lib = _native.lib()
    p = os.fsencode(os.fspath(path))
    if kind == "rows":
        _native.check(lib.mic_png_write_rows(p, _P(ptr), W, H, int(level), int(threads)))
    else:
        _native.check(lib.mic_png_write(p, _P(ptr), W, H, stride, int(level), int(threads)))
    del keep


class PendingSave:
    """A save queued on libmic's own worker threads (mic_png_write_async): wait() blocks until the file is written and
    raises if it could not be.  Keeps the image alive until then."""
    __slots__ = ("_job", "_keep")

    def __init__(self, job: int, keep):
        self._job, self._keep = job, keep

    def wait(self) -> None:
        if self._job is not None:
            job, self._job = self._job, None
            try:
                _native.check(_native.lib().mic_png_wait(job))
            finally:
                self._keep = None

    def __del__(self):  # never leave a job reading freed pixels behind
        try:
            self.wait()
        except Exception:
            pass


def save_async(image, path, level: int = DEFAULT_LEVEL, threads: int = 1) -> PendingSave:
    """save() without waiting: the encode + write run on worker threads inside libmic (no Python thread, no GIL
    hand-offs on the caller's critical path).  Returns a PendingSave; call .wait() before relying on the file."""
    kind, keep, ptr, W, H, stride = _source(image)
    if kind != "rows":
        arr = keep
        table = np.empty(H, np.uint64)
        table[:] = arr.ctypes.data + np.arange(H, dtype=np.uint64) * np.uint64(stride)
        keep, ptr = (arr, table), table.ctypes.data
    job = ctypes.c_int64()
    _native.check(_native.lib().mic_png_write_async(os.fsencode(os.fspath(path)), _P(ptr), W, H, int(level), int(threads),
                                                    ctypes.byref(job)))
    return PendingSave(job.value, keep)


def encode(image, level: int = DEFAULT_LEVEL, threads: int = 0) -> bytes:
    """`image` as PNG bytes."""
    kind, keep, ptr, W, H, stride = _source(image)
    lib = _native.lib()
    cap = int(lib.mic_png_bound(W, H))
    buf = np.empty(cap, np.uint8)
    n = ctypes.c_size_t()
    if kind == "rows":
        _native.check(lib.mic_png_encode_rows(_P(ptr), W, H, int(level), int(threads), _P(buf.ctypes.data), cap, ctypes.byref(n)))
    else:
        _native.check(lib.mic_png_encode(_P(ptr), W, H, stride, int(level), int(threads), _P(buf.ctypes.data), cap, ctypes.byref(n)))
    del keep
    return buf[:n.value].tobytes()


# ------------------------------------------------------------------------------------------ reading
_DECODE_OFF = os.environ.get("MIC_PNG_DECODE") == "0"  # MIC_PNG_DECODE=0: every file goes to Pillow (A/B, debugging)


def _new_rgba(w: int, h: int):
    """An uninitialised RGBA PIL image of its own + the address of Pillow's row-pointer table for it (or None)."""
    im = Image.new("RGBA", (w, h), None)
    tab = _pilmem.row_table(im)
    return im, (tab[0] if tab is not None else None)


def decode_many(blobs, threads: int = 0):
    """PNG files in memory -> RGBA PIL images, decoded by libmic (csrc/png_decode.cpp) straight into the images' own
    memory, several files at a time on the library's threads.  An entry is None where the decoder DECLINES the file
    (16-bit, interlaced, a colour key, anything irregular): the caller hands that one to Pillow, which decodes it or
    raises its own error.  Equals Image.open(f).convert("RGBA") byte for byte (tests/test_png_decode.py)."""
    n = len(blobs)
    out = [None] * n
    if n == 0 or _DECODE_OFF:
        return out
    lib = _native.lib()
    w, h = ctypes.c_int32(), ctypes.c_int32()
    todo, ims, tables = [], [], []
    for i, b in enumerate(blobs):
        if lib.mic_png_info(b, len(b), ctypes.byref(w), ctypes.byref(h)) != 0:
            continue
        # Pillow's decompression-bomb guard belongs to the reference's Image.open(path) (compositor.py:34): above
        # MAX_IMAGE_PIXELS it warns, above twice that it raises DecompressionBombError.  Such a file is DECLINED here,
        # before anything is allocated for it, so Pillow opens it and warns / raises exactly as it always did.
        if Image.MAX_IMAGE_PIXELS is not None and w.value * h.value > Image.MAX_IMAGE_PIXELS:
            continue
        im, table = _new_rgba(w.value, h.value)
        if table is None:  # Pillow's memory cannot be located: decode into an array, then wrap it
            arr = np.empty((h.value, w.value, 4), np.uint8)
            if lib.mic_png_decode(b, len(b), _P(arr.ctypes.data), 4 * w.value, w.value, h.value) == 0:
                out[i] = Image.fromarray(arr, "RGBA")
            continue
        todo.append(i)
        ims.append(im)
        tables.append(table)
    if not todo:
        return out
    k = len(todo)
    if k == 1:
        i, im = todo[0], ims[0]
        if lib.mic_png_decode_rows(blobs[i], len(blobs[i]), _P(tables[0]), im.size[0], im.size[1]) == 0:
            out[i] = im
        return out
    pngs = (ctypes.c_char_p * k)(*[blobs[i] for i in todo])
    sizes = (ctypes.c_size_t * k)(*[len(blobs[i]) for i in todo])
    rows = (_P * k)(*tables)
    ws = (ctypes.c_int32 * k)(*[im.size[0] for im in ims])
    hs = (ctypes.c_int32 * k)(*[im.size[1] for im in ims])
    status = (ctypes.c_int32 * k)()
    if threads <= 0:
        # a bundle's handful of small cutouts (~100 KB of PNG in all) decode in about a millisecond on one core: starting
        # threads costs more than it saves; big files are dealt over the library's threads
        threads = 1 if sum(len(blobs[i]) for i in todo) < (256 << 10) else min(k, 8)
    lib.mic_png_decode_many(k, pngs, sizes, rows, ws, hs, int(threads), status)
    for j, i in enumerate(todo):
        if status[j] == 0:
            out[i] = ims[j]
    return out


def decode(blob: bytes) -> Optional[Image.Image]:
    """One PNG file in memory -> RGBA PIL image, or None where libmic's decoder declines it."""
    return decode_many([blob])[0]


def decode_counts() -> Tuple[int, int]:
    """(files libmic has decoded, files it has declined) in this process: the fallback meter."""
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    _native.lib().mic_png_decode_counts(ctypes.byref(a), ctypes.byref(b))
    return int(a.value), int(b.value)


def save_like_pil(image: Image.Image, path, **pil_kwargs) -> None:
    """image.save(path) as the reference's helpers call it (_save_overlay_debug, _compose_candidates_grid): PIL picks
    the format from the file name, so only *.png goes through libmic's writer; any other name (or explicit PIL
    options) is PIL's business."""
    if not pil_kwargs and isinstance(image, Image.Image) and image.mode == "RGBA" and os.fspath(path).lower().endswith(".png"):
        save(image, path)
    else:
        image.save(path, **pil_kwargs)
