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


def save_like_pil(image: Image.Image, path, **pil_kwargs) -> None:
    """image.save(path) as the reference's helpers call it (_save_overlay_debug, _compose_candidates_grid): PIL picks
    the format from the file name, so only *.png goes through libmic's writer; any other name (or explicit PIL
    options) is PIL's business."""
    if not pil_kwargs and isinstance(image, Image.Image) and image.mode == "RGBA" and os.fspath(path).lower().endswith(".png"):
        save(image, path)
    else:
        image.save(path, **pil_kwargs)
