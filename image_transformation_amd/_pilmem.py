"""Zero-copy access to a PIL image's pixel rows.

The reference's callers hand PIL images to composite() (compositor.py:6) and .save() what comes back
(macro_placement_test.py:1513).  Pillow's public export paths all copy -- Image.tobytes() / np.asarray()
run the "raw" encoder over the image (14 ms for a 4K RGBA image on the GPU box's host, three times the
whole GPU composite + both PCIe transfers) and the Arrow export refuses images stored in several memory
blocks (every image above 16 MB).  What Pillow does expose is the capsule of its Imaging struct
(ImagingCore.ptr, the handle its own ImageTk / third-party C extensions use): the struct's `image`
member is the table of row pointers.  This module reads that table so that rows can be memmove'd
straight into a pinned staging buffer, or compared for the "one solid colour" case.

The struct layout is validated field by field against what the Python object reports (bands, size,
pixelsize, linesize, first and last pixel); anything unexpected -> None and the caller takes the
np.asarray() path.  No pixel arithmetic happens here.
"""
from __future__ import annotations

import concurrent.futures
import ctypes
import threading
from typing import List, Optional, Tuple

import numpy as np
from PIL import Image

_get_ptr = ctypes.pythonapi.PyCapsule_GetPointer
_get_ptr.restype = ctypes.c_void_p
_get_ptr.argtypes = [ctypes.py_object, ctypes.c_char_p]
_get_name = ctypes.pythonapi.PyCapsule_GetName
_get_name.restype = ctypes.c_char_p
_get_name.argtypes = [ctypes.py_object]
_libc = ctypes.CDLL(None)
_libc.memcmp.restype = ctypes.c_int
_libc.memcmp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]

# (offset of bands, xsize, ysize, image (char **), pixelsize, linesize) in struct ImagingMemoryInstance:
# Pillow >= 12 keeps the mode as an enum (4 bytes), earlier versions as char[7] (padded to 8).
_LAYOUTS = ((12, 16, 20, 48, 72, 76), (16, 20, 24, 56, 80, 84))
_PARALLEL_BYTES = 4 << 20
_pool: Optional[concurrent.futures.ThreadPoolExecutor] = None
_pool_lock = threading.Lock()


def _workers() -> concurrent.futures.ThreadPoolExecutor:
    global _pool
    if _pool is None:
        with _pool_lock:
            if _pool is None:
                _pool = concurrent.futures.ThreadPoolExecutor(4, thread_name_prefix="mic-pil")
    return _pool


_good_layout: Optional[int] = None  # index into _LAYOUTS that row_runs() has validated in this process
# Which path the callers of this module took, per process: "zero_copy" = Pillow's rows were reached in place,
# "fallback" = the image's memory could not be located (an unknown struct layout, a lazy image) and the caller went
# through np.asarray / Image.tobytes instead -- correct, but ~14 ms per 4K image.  path_report() names the layout.
counters = {"zero_copy": 0, "fallback": 0}
_LAYOUT_NAMES = ("Pillow >= 12 (mode kept as an enum)", "Pillow < 12 (mode kept as char[7])")


def path_report() -> dict:
    """{"layout": which ImagingMemoryInstance layout was validated in this process (None: none yet / none fits),
    "zero_copy": calls that reached Pillow's rows in place, "fallback": calls that had to copy through Pillow}."""
    return {"layout": None if _good_layout is None else _LAYOUT_NAMES[_good_layout], **counters}


def row_table(img: Image.Image) -> Optional[Tuple[int, int, int]]:
    """(address of Pillow's row-pointer table, W, H) of an RGBA image -- the cheap form of row_runs() for hot calls:
    once row_runs() has proven a struct layout in this process (field by field and against getpixel), later images
    only have their size fields checked against what the Python object reports (a handful of ctypes reads)."""
    global _good_layout
    try:
        if img.mode != "RGBA":
            return None
        if _good_layout is None and row_runs(img) is None:
            return None
        img.load()
        W, H = img.size
        if W <= 0 or H <= 0 or _good_layout is None:
            return None
        cap = img.im.ptr
        base = _get_ptr(cap, _get_name(cap))
        if not base:
            return None
        o_bands, o_x, o_y, o_image, o_px, o_line = _LAYOUTS[_good_layout]
        head = (ctypes.c_int32 * 24).from_address(base)
        if (head[o_bands >> 2], head[o_x >> 2], head[o_y >> 2], head[o_px >> 2], head[o_line >> 2]) != (4, W, H, 4, 4 * W):
            return None
        table = ctypes.c_uint64.from_address(base + o_image).value
        if table:
            counters["zero_copy"] += 1
            return table, W, H
        counters["fallback"] += 1
        return None
    except Exception:
        counters["fallback"] += 1
        return None


def row_runs(img: Image.Image) -> Optional[List[Tuple[int, int]]]:
    """RGBA image -> [(address, bytes)] of maximal runs of rows that are contiguous in memory, in row
    order (a 4K image: three 16 MB blocks), or None if the image's memory cannot be located safely."""
    try:
        if img.mode != "RGBA":
            return None
        img.load()
        W, H = img.size
        if W <= 0 or H <= 0:
            return None
        cap = img.im.ptr
        base = _get_ptr(cap, _get_name(cap))
        if not base:
            return None
        head = np.frombuffer((ctypes.c_uint8 * 96).from_address(base), np.uint8)
        global _good_layout
        for li, (o_bands, o_x, o_y, o_image, o_px, o_line) in enumerate(_LAYOUTS):
            f = lambda o: int(head[o:o + 4].view(np.int32)[0])  # noqa: E731
            if (f(o_bands), f(o_x), f(o_y), f(o_px), f(o_line)) != (4, W, H, 4, 4 * W):
                continue
            table = int(head[o_image:o_image + 8].view(np.uint64)[0])
            if not table:
                continue
            rows = np.frombuffer((ctypes.c_uint64 * H).from_address(table), np.uint64).astype(np.int64)
            if (rows <= 0).any():
                continue
            # the first and the last pixel read through the table must be what Pillow itself reports
            first = tuple(np.frombuffer((ctypes.c_uint8 * 4).from_address(int(rows[0])), np.uint8).tolist())
            last = tuple(np.frombuffer((ctypes.c_uint8 * 4).from_address(int(rows[-1]) + 4 * (W - 1)), np.uint8).tolist())
            if first != tuple(img.getpixel((0, 0))) or last != tuple(img.getpixel((W - 1, H - 1))):
                continue
            _good_layout = li
            counters["zero_copy"] += 1
            line = 4 * W
            brk = np.nonzero(np.diff(rows) != line)[0] + 1
            starts = np.concatenate([[0], brk])
            ends = np.concatenate([brk, [H]])
            return [(int(rows[s]), int(e - s) * line) for s, e in zip(starts, ends)]
        counters["fallback"] += 1
        return None
    except Exception:
        counters["fallback"] += 1
        return None


def copy_to(img: Image.Image, dst_addr: int) -> bool:
    """memmove the image's W*H*4 bytes to dst_addr (row-major, tightly packed).  False: use np.asarray."""
    runs = row_runs(img)
    if runs is None:
        return False
    jobs = []
    off = 0
    for addr, n in runs:
        # pieces of a few MB: ctypes releases the GIL around memmove, a 33 MB image moves on 4 cores
        for p in range(0, n, _PARALLEL_BYTES):
            jobs.append((dst_addr + off + p, addr + p, min(_PARALLEL_BYTES, n - p)))
        off += n
    if len(jobs) <= 1:
        for d, s, n in jobs:
            ctypes.memmove(d, s, n)
    else:
        list(_workers().map(lambda j: ctypes.memmove(*j), jobs))
    return True


def same_pixels(a: Image.Image, b: Image.Image) -> bool:
    """Do two RGBA images hold the same size and bytes?  memcmp over Pillow's own rows (no tobytes() copies); False
    also when either image's memory cannot be located."""
    if a.size != b.size or a.mode != "RGBA" or b.mode != "RGBA":
        return False
    ta, tb = row_table(a), row_table(b)
    if ta is None or tb is None:
        return False
    W, H = a.size
    ra = (ctypes.c_uint64 * H).from_address(ta[0])
    rb = (ctypes.c_uint64 * H).from_address(tb[0])
    line = 4 * W
    # rows of one image are contiguous in long runs: compare run by run
    y = 0
    while y < H:
        n = 1
        while y + n < H and ra[y + n] == ra[y] + n * line and rb[y + n] == rb[y] + n * line:
            n += 1
        if _libc.memcmp(ra[y], rb[y], n * line) != 0:
            return False
        y += n
    return True


def copy_from(img: Image.Image, src_addr: int) -> bool:
    """memmove W*H*4 tightly packed bytes at src_addr INTO the image's own rows (an Image.new the caller owns).
    False: the image's memory could not be located (the caller builds the image another way)."""
    if getattr(img, "readonly", 0):
        return False
    runs = row_runs(img)
    if runs is None:
        return False
    jobs = []
    off = 0
    for addr, n in runs:
        for p in range(0, n, _PARALLEL_BYTES):
            jobs.append((addr + p, src_addr + off + p, min(_PARALLEL_BYTES, n - p)))
        off += n
    if len(jobs) <= 1:
        for d, s, n in jobs:
            ctypes.memmove(d, s, n)
    else:
        list(_workers().map(lambda j: ctypes.memmove(*j), jobs))
    return True


def solid_colour(img: Image.Image) -> Optional[Tuple[int, int, int, int]]:
    """(r, g, b, a) if every pixel of the RGBA image has that value -- what background_resizing.fill_solid
    returns and run_macro_only re-opens from canvas.png every iteration (macro_placement_test.py:1510) --
    else None (also when the memory cannot be located: the caller then simply uploads the image).
    Exact: every byte is compared (memcmp of each run of rows against itself shifted by one row)."""
    runs = row_runs(img)
    if runs is None:
        return None
    W = img.size[0]
    line = 4 * W
    r0 = runs[0][0]
    first = np.frombuffer((ctypes.c_uint32 * W).from_address(r0), np.uint32)
    if not (first == first[0]).all():
        return None
    # quick rejection before the full scan: the first pixel of every run
    for addr, _ in runs:
        if _libc.memcmp(addr, r0, 4) != 0:
            return None
    jobs = []
    for addr, n in runs:
        jobs.append((addr, r0, line))  # the run's first row equals the image's first row
        for p in range(0, n - line, _PARALLEL_BYTES):  # and every row equals the next one
            jobs.append((addr + p, addr + p + line, min(_PARALLEL_BYTES, n - line - p)))
    if len(jobs) <= 2:
        same = all(_libc.memcmp(*j) == 0 for j in jobs)
    else:
        same = all(r == 0 for r in _workers().map(lambda j: _libc.memcmp(*j), jobs))
    if not same:
        return None
    px = np.frombuffer((ctypes.c_uint8 * 4).from_address(r0), np.uint8)
    return int(px[0]), int(px[1]), int(px[2]), int(px[3])
