"""Debug artifacts of the reference's orchestrator on the device path.

    _save_overlay_debug(placements, canvas_size, path)   macro_placement_test.py:967-983
    _compose_candidates_grid(image_paths, out_path)      macro_placement_test.py:1332-1345

The overlay is ImageDraw.rectangle outlines (width 3, six cycling colours) on a transparent canvas:
one HIP pass (mic_draw_rect_outlines) instead of per-rectangle host drawing.  The candidates grid is
a composite() call: every image resized (Pillow-exact LANCZOS) to the first one's size and
alpha-composited onto a white 2x2 sheet.
"""
from __future__ import annotations

import ctypes
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image

from . import _native
from . import png as mic_png
from .compositor import SolidCanvas, _to_pil, composite_device, Atlas, coerce_placements

_P = ctypes.c_void_p

# macro_placement_test.py:971-978
OVERLAY_COLORS = [(255, 99, 71, 180), (135, 206, 235, 180), (60, 179, 113, 180), (238, 130, 238, 180),
                  (255, 215, 0, 180), (30, 144, 255, 180)]
OUTLINE_WIDTH = 3


def rect_outlines_device(canvas_size: Tuple[int, int], boxes: Sequence[Sequence[int]],
                         colors: Sequence[Sequence[int]], width: int = OUTLINE_WIDTH, device: Optional[int] = None):
    """Transparent RGBA (H, W, 4) uint8 device tensor with ImageDraw.rectangle(box, outline=colour,
    width=width) applied per box in order.  Raises ValueError like ImageDraw for x2 < x1 / y2 < y1."""
    import torch
    W, H = int(canvas_size[0]), int(canvas_size[1])
    b = np.ascontiguousarray(np.asarray(boxes, dtype=np.int64).reshape(-1, 4))
    c = np.ascontiguousarray(np.asarray(colors, dtype=np.uint8).reshape(-1, 4))
    if len(b) != len(c):
        raise ValueError("one colour per box")
    for (x1, y1, x2, y2) in b:
        if x2 < x1:
            raise ValueError("x1 must be greater than or equal to x0")  # ImageDraw.rectangle's message
        if y2 < y1:
            raise ValueError("y1 must be greater than or equal to y0")
    b32 = np.ascontiguousarray(np.clip(b, -(1 << 30), 1 << 30).astype(np.int32))
    ctx = _native.context(device)
    out = torch.empty((H, W, 4), dtype=torch.uint8, device=ctx.torch_device)
    _native.check(_native.lib().mic_draw_rect_outlines(
        ctx.handle, _P(out.data_ptr()), W, H, len(b32), b32.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
        c.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), int(width), _P(ctx.stream_ptr())))
    return out


def overlay_debug_device(placements: Sequence[Dict], canvas_size: Tuple[int, int], device: Optional[int] = None):
    """_save_overlay_debug's image as a device tensor."""
    boxes = [[v for v in p["box"]] for p in placements]  # x1, y1, x2, y2 = p["box"] (:980)
    for bx in boxes:
        if len(bx) != 4:
            raise ValueError(f"not enough values to unpack (expected 4, got {len(bx)})" if len(bx) < 4
                             else "too many values to unpack (expected 4)")
    colors = [OVERLAY_COLORS[i % len(OVERLAY_COLORS)] for i in range(len(boxes))]
    return rect_outlines_device(canvas_size, boxes or np.zeros((0, 4), np.int64),
                                colors or np.zeros((0, 4), np.uint8), OUTLINE_WIDTH, device)


def overlay_debug(placements: Sequence[Dict], canvas_size: Tuple[int, int]) -> Image.Image:
    return _to_pil(overlay_debug_device(placements, canvas_size))


def save_overlay_debug(placements: Sequence[Dict], canvas_size: Tuple[int, int], path) -> None:
    """Drop-in for _save_overlay_debug(placements, canvas_size, path)."""
    mic_png.save_like_pil(overlay_debug(placements, canvas_size), path)  # (*.png: libmic's writer)


def candidates_grid_device(images: Sequence[Image.Image]):
    """The 2x2 grid of _compose_candidates_grid as a device tensor; images beyond the fourth are
    ignored like zip() does there."""
    if not images:
        raise ValueError("no images")
    imgs = [im.convert("RGBA") for im in images[:4]]
    ref_w, ref_h = imgs[0].size
    positions = [(0, 0), (ref_w, 0), (0, ref_h), (ref_w, ref_h)]
    atlas = Atlas({i + 1: im for i, im in enumerate(imgs)})
    rows = coerce_placements(atlas, [{"object_id": i + 1, "box": [x, y, x + ref_w, y + ref_h]}
                                     for i, (x, y) in enumerate(positions[:len(imgs)])])
    sheet = SolidCanvas((2 * ref_w, 2 * ref_h), (255, 255, 255, 255))
    return composite_device(atlas, [sheet], [rows])[0]


def compose_candidates_grid(image_paths: Sequence, out_path) -> None:
    """Drop-in for _compose_candidates_grid(image_paths, out_path): missing files are skipped, nothing
    is written when none exists."""
    imgs: List[Image.Image] = [Image.open(p).convert("RGBA") for p in image_paths if Path(p).exists()]
    if not imgs:
        return
    mic_png.save_like_pil(_to_pil(candidates_grid_device(imgs)), out_path)
