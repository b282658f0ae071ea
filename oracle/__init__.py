"""CPU oracle for the compositor hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (image_transformation_amd) never does; it
fails loudly when its HIP library is missing instead of falling back here.

The arithmetic lives in mic_oracle.c (plain C, built by oracle/Makefile); this
module is the ctypes/numpy binding plus the few pieces of the reference's Python
semantics that sit between its call surface and that arithmetic
(compositor.py:12-19: id/box coercion, unknown-id skip, degenerate boxes).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmic_oracle.so")
_lib = None

LANCZOS = 0
BILINEAR = 1


def build(force: bool = False) -> str:
    """Compile libmic_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "mic_oracle.c")

    def stale() -> bool:
        return not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)

    if force or stale():
        import fcntl
        with open(_LIB_PATH + ".lock", "w") as lock:  # the ranks of a multi-process test import this together
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or stale():
                subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmic_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.orc_alpha_over_at.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_alpha_over_at.restype = None
        L.orc_premultiply.argtypes = [u8p, u8p, ctypes.c_size_t]
        L.orc_premultiply.restype = None
        L.orc_unpremultiply.argtypes = [u8p, u8p, ctypes.c_size_t]
        L.orc_unpremultiply.restype = None
        L.orc_resize.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int]
        L.orc_resize.restype = ctypes.c_int
        L.orc_composite.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, i32p,
                                    ctypes.POINTER(u8p), ctypes.c_int, i32p, i32p, ctypes.c_int, u8p]
        L.orc_composite.restype = ctypes.c_int
        L.orc_median_rgb.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.orc_median_rgb.restype = None
        L.orc_fill_solid.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p]
        L.orc_fill_solid.restype = None
        L.orc_rect_outlines.argtypes = [u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, u8p, ctypes.c_int]
        L.orc_rect_outlines.restype = None
        L.orc_thumbnail_size.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int)] * 2
        L.orc_thumbnail_size.restype = None
        L.orc_resample_coeffs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(i32p), ctypes.POINTER(i32p)]
        L.orc_resample_coeffs.restype = ctypes.c_int
        _lib = L
    return _lib


def _u8(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _i32(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _rgba(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("expected an (H, W, 4) uint8 RGBA array")
    return a


def alpha_over_at(canvas: np.ndarray, overlay: np.ndarray, dx: int, dy: int) -> None:
    """In place Image.alpha_composite(overlay, dest=(dx, dy)) (compositor.py:21)."""
    assert canvas.flags.c_contiguous and canvas.dtype == np.uint8
    ov = _rgba(overlay)
    lib().orc_alpha_over_at(_u8(canvas), canvas.shape[1], canvas.shape[0], _u8(ov), ov.shape[1],
                            ov.shape[0], int(dx), int(dy))


def premultiply(a: np.ndarray) -> np.ndarray:
    a = _rgba(a)
    out = np.empty_like(a)
    lib().orc_premultiply(_u8(a), _u8(out), a.shape[0] * a.shape[1])
    return out


def unpremultiply(a: np.ndarray) -> np.ndarray:
    a = _rgba(a)
    out = np.empty_like(a)
    lib().orc_unpremultiply(_u8(a), _u8(out), a.shape[0] * a.shape[1])
    return out


def resize(a: np.ndarray, size: Tuple[int, int], filter: int = LANCZOS) -> np.ndarray:
    """Image.resize((w, h), LANCZOS) on RGBA (compositor.py:20)."""
    a = _rgba(a)
    w, h = int(size[0]), int(size[1])
    out = np.empty((h, w, 4), np.uint8)
    rc = lib().orc_resize(_u8(a), a.shape[1], a.shape[0], _u8(out), w, h, filter)
    if rc != 0:
        raise ValueError(f"orc_resize failed rc={rc}")
    return out


def resample_coeffs(in_size: int, out_size: int, filter: int = LANCZOS):
    """(bounds[out,2], coeffs[out,ksize]) int32 tables of one resample axis."""
    b = ctypes.POINTER(ctypes.c_int32)()
    k = ctypes.POINTER(ctypes.c_int32)()
    ksize = lib().orc_resample_coeffs(in_size, out_size, filter, ctypes.byref(b), ctypes.byref(k))
    bounds = np.ctypeslib.as_array(b, shape=(out_size, 2)).copy()
    kk = np.ctypeslib.as_array(k, shape=(out_size, ksize)).copy()
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free(ctypes.cast(b, ctypes.c_void_p))
    libc.free(ctypes.cast(k, ctypes.c_void_p))
    return bounds, kk


def coerce_placements(object_ids: Iterable[int], placements: Sequence[dict]):
    """Python-level semantics of compositor.py:12-18 -> (place_obj, boxes) int32 arrays.

    object_id: int, or anything int() accepts; unknown ids are skipped (-1).
    box: exactly four values, each coerced with int() (truncation toward zero).
    """
    index = {oid: i for i, oid in enumerate(object_ids)}
    place_obj: List[int] = []
    boxes: List[List[int]] = []
    for p in placements:
        oid = int(p["object_id"]) if not isinstance(p["object_id"], int) else p["object_id"]
        if oid not in index:
            continue
        x1, y1, x2, y2 = [int(v) for v in p["box"]]
        place_obj.append(index[oid])
        boxes.append([x1, y1, x2, y2])
    return (np.asarray(place_obj, np.int32).reshape(-1),
            np.asarray(boxes, np.int32).reshape(-1, 4))


def composite(bg: np.ndarray, objects: Dict[int, np.ndarray], placements: Sequence[dict],
              filter: int = LANCZOS) -> np.ndarray:
    """compositor.py:6-22 on numpy RGBA arrays; returns a new canvas."""
    bg = _rgba(bg)
    ids = list(objects.keys())
    arrs = [_rgba(objects[i]) for i in ids]
    place_obj, boxes = coerce_placements(ids, placements)
    return composite_arrays(bg, arrs, place_obj, boxes, filter)


def composite_arrays(bg: np.ndarray, arrs: List[np.ndarray], place_obj: np.ndarray,
                     boxes: np.ndarray, filter: int = LANCZOS) -> np.ndarray:
    bg = _rgba(bg)
    n = len(arrs)
    ws = np.asarray([a.shape[1] for a in arrs], np.int32)
    hs = np.asarray([a.shape[0] for a in arrs], np.int32)
    ptrs = (ctypes.POINTER(ctypes.c_uint8) * max(n, 1))(*[_u8(a) for a in arrs])
    place_obj = np.ascontiguousarray(place_obj, np.int32)
    boxes = np.ascontiguousarray(boxes, np.int32)
    out = np.empty_like(bg)
    rc = lib().orc_composite(_u8(bg), bg.shape[1], bg.shape[0], n, _i32(ws), _i32(hs), ptrs,
                             len(place_obj), _i32(place_obj), _i32(boxes), filter, _u8(out))
    if rc != 0:
        raise MemoryError(f"orc_composite failed rc={rc}")
    return out


def median_rgb(rgba: np.ndarray) -> Tuple[int, int, int]:
    """background_resizing.py:11-22."""
    a = _rgba(rgba)
    out = np.zeros(3, np.uint8)
    lib().orc_median_rgb(_u8(a), a.shape[0] * a.shape[1], _u8(out))
    return int(out[0]), int(out[1]), int(out[2])


OVERLAY_COLORS = [(255, 99, 71, 180), (135, 206, 235, 180), (60, 179, 113, 180), (238, 130, 238, 180),
                  (255, 215, 0, 180), (30, 144, 255, 180)]  # macro_placement_test.py:971-978


def rect_outlines(size: Tuple[int, int], boxes: Sequence[Sequence[int]], colors: Sequence[Sequence[int]],
                  width: int = 3) -> np.ndarray:
    """ImageDraw.rectangle(box, outline=colour, width=width) per box, in order, on a transparent
    RGBA image of `size` (macro_placement_test.py:967-983)."""
    w, h = int(size[0]), int(size[1])
    out = np.empty((h, w, 4), np.uint8)
    b = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(-1, 4))
    c = np.ascontiguousarray(np.asarray(colors, np.uint8).reshape(-1, 4))
    assert len(b) == len(c)
    lib().orc_rect_outlines(_u8(out), w, h, len(b), _i32(b), _u8(c), int(width))
    return out


def overlay_debug(placements: Sequence[dict], canvas_size: Tuple[int, int]) -> np.ndarray:
    """_save_overlay_debug's image (before the PNG encoder)."""
    boxes = [[int(v) for v in p["box"]] for p in placements]
    colors = [OVERLAY_COLORS[i % len(OVERLAY_COLORS)] for i in range(len(boxes))]
    if not boxes:
        return np.zeros((int(canvas_size[1]), int(canvas_size[0]), 4), np.uint8)
    return rect_outlines(canvas_size, boxes, colors, 3)


def candidates_grid(images: Sequence[np.ndarray]) -> np.ndarray:
    """_compose_candidates_grid (macro_placement_test.py:1332-1345): every image resized (LANCZOS) to
    the first one's size, alpha-composited onto a white 2x2 grid; more than four are ignored (zip)."""
    ref_h, ref_w = images[0].shape[:2]
    bg = np.full((2 * ref_h, 2 * ref_w, 4), 255, np.uint8)
    pos = [(0, 0), (ref_w, 0), (0, ref_h), (ref_w, ref_h)]
    objs = {i + 1: im for i, im in enumerate(images[:4])}
    pl = [{"object_id": i + 1, "box": [x, y, x + ref_w, y + ref_h]} for i, (x, y) in enumerate(pos[:len(objs)])]
    return composite(bg, objs, pl)


def fill_solid(size: Tuple[int, int], rgba: Sequence[int]) -> np.ndarray:
    w, h = int(size[0]), int(size[1])
    out = np.empty((h, w, 4), np.uint8)
    col = np.asarray(rgba, np.uint8)
    lib().orc_fill_solid(_u8(out), w, h, _u8(col))
    return out


def edge_strip_medians(rgba: np.ndarray, strip_px: int = 8):
    """background_resizing.py:36-57: medians of the left, right, top, bottom edge strips."""
    a = _rgba(rgba)
    h, w = a.shape[:2]
    return (median_rgb(a[:, :min(strip_px, w)]), median_rgb(a[:, max(0, w - strip_px):]),
            median_rgb(a[:min(strip_px, h)]), median_rgb(a[max(0, h - strip_px):]))


def fill_gradient(rgba: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """background_resizing.py:60-98 restated with the same NumPy float32 semantics: t is a Python
    float, (1 - t) and t are cast to float32 by the float32 colour arrays, products and sum are
    float32, astype(uint8) truncates."""
    left, right, top, bottom = edge_strip_medians(rgba)
    var = lambda p, q: float(sum((int(p[i]) - int(q[i])) ** 2 for i in range(3)))  # noqa: E731
    W, H = int(size[0]), int(size[1])
    out = np.zeros((H, W, 4), np.uint8)
    horizontal = var(left, right) <= var(top, bottom)
    c1 = np.array(left if horizontal else top, dtype=np.float32)
    c2 = np.array(right if horizontal else bottom, dtype=np.float32)
    n = W if horizontal else H
    for i in range(n):
        t = i / max(1, n - 1)
        rgb = (np.float32(1 - t) * c1 + np.float32(t) * c2).astype(np.uint8)
        if horizontal:
            out[:, i, :3] = rgb
        else:
            out[i, :, :3] = rgb
    out[:, :, 3] = 255
    return out


def thumbnail_size(size: Tuple[int, int], req: Tuple[int, int] = (256, 256)) -> Tuple[int, int]:
    ow, oh = ctypes.c_int(), ctypes.c_int()
    lib().orc_thumbnail_size(int(size[0]), int(size[1]), int(req[0]), int(req[1]),
                             ctypes.byref(ow), ctypes.byref(oh))
    return ow.value, oh.value


def thumbnail(a: np.ndarray, req: Tuple[int, int] = (256, 256), filter: int = LANCZOS) -> np.ndarray:
    """im.copy().thumbnail(req, LANCZOS) (macro_placement_test.py:192-195)."""
    a = _rgba(a)
    tw, th = thumbnail_size((a.shape[1], a.shape[0]), req)
    return resize(a, (tw, th), filter)
