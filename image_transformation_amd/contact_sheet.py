"""Labelled contact sheet on the GPU: macro_placement_test.py:162-242 (_build_labeled_contact_sheet).

Thumbnails (Pillow Image.thumbnail size rule + LANCZOS) and the tiling/alpha-over run in the HIP
kernels: the whole sheet is ONE composite job over a solid white background whose layers are, in
the reference's order, thumb_0, label_0, thumb_1, label_1, ...  Glyph rasterisation stays on the
host (FreeType through Pillow's ImageDraw, as SURVEY.md section 2 row 3 scopes it): each label is
drawn once as an 8-bit coverage mask and uploaded as a black RGBA strip whose alpha is that mask.
Blending it with the alpha-over kernel equals ImageDraw.text's own mask blend on an opaque sheet:
both are div255(dst * (255 - m) + 128) per channel with alpha 255.
"""
from __future__ import annotations

import ctypes
import functools
import json
import threading
from pathlib import Path
from typing import List, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

from . import _native
from .compositor import open_rgba, load_object_images, Atlas, SolidCanvas, _device_guard, _to_pil, LANCZOS

_P = ctypes.c_void_p
_MARGIN = 32


def thumbnail_size(size: Tuple[int, int], req: Tuple[int, int]) -> Tuple[int, int]:
    """Pillow's Image.thumbnail size rule (macro_placement_test.py:194)."""
    ow, oh = ctypes.c_int32(), ctypes.c_int32()
    _native.check(_native.lib().mic_thumbnail_size(int(size[0]), int(size[1]), int(req[0]), int(req[1]),
                                                   ctypes.byref(ow), ctypes.byref(oh)))
    return ow.value, oh.value


_fonts = threading.local()  # a FreeType face is not safe to share between threads: one font object per thread and size


def _resolve_font(font_size: int):
    """Same fallback chain as macro_placement_test.py:176-186 (one font object per size and thread)."""
    cache = _fonts.__dict__.setdefault("by_size", {})
    if font_size in cache:
        return cache[font_size]
    font = None
    for name in ("DejaVuSans.ttf", "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"):
        try:
            font = ImageFont.truetype(name, size=font_size)
            break
        except Exception:
            continue
    if font is None:
        try:
            font = ImageFont.load_default()
        except Exception:
            font = None
    cache[font_size] = font
    return font


@functools.lru_cache(maxsize=512)
def _label_mask(label: str, font):
    """Coverage mask of draw.text((0, 0), label) cropped to its ink, with the crop's offset: (mask, dx, dy), or None
    if the label leaves no ink.  Pure in (label, font object): a bundle's labels are rasterised once per process."""
    probe = ImageDraw.Draw(Image.new("L", (1, 1), 0))
    try:
        bbox = probe.textbbox((0, 0), label, font=font)
        w, h = bbox[2], bbox[3]
    except Exception:
        w, h = len(label) * 16, 48
    cw, ch = max(1, int(w) + 2 * _MARGIN), max(1, int(h) + 2 * _MARGIN)
    mask_img = Image.new("L", (cw, ch), 0)
    ImageDraw.Draw(mask_img).text((_MARGIN, _MARGIN), label, fill=255, font=font)
    box = mask_img.getbbox()
    if box is None:
        return None
    m = np.ascontiguousarray(np.asarray(mask_img.crop(box), np.uint8))
    m.setflags(write=False)
    return m, box[0] - _MARGIN, box[1] - _MARGIN


def _label_strip(label: str, font, tx: int, ty: int):
    """Coverage mask of draw.text((tx, ty), label) as (mask, x, y) or None if no ink."""
    got = _label_mask(label, font)
    if got is None:
        return None
    m, dx, dy = got
    return m, tx + dx, ty + dy


def _measure_label(draw, label: str, font) -> Tuple[int, int]:
    """Text size with the reference's fallbacks (macro_placement_test.py:222-238)."""
    return _measure_cached(label, font)


@functools.lru_cache(maxsize=512)
def _measure_cached(label: str, font) -> Tuple[int, int]:
    draw = ImageDraw.Draw(Image.new("RGBA", (1, 1)))
    try:
        bbox = draw.textbbox((0, 0), label, font=font)
        return bbox[2] - bbox[0], bbox[3] - bbox[1]
    except Exception:
        try:
            if font is not None and hasattr(font, "getsize"):
                return font.getsize(label)
        except Exception:
            pass
        return int(len(label) * 7), 12


def build_labeled_contact_sheet(objects_dir: str, results_json_path: str,
                                thumb_size: Tuple[int, int] = (256, 256), cols: int = 4,
                                label_height: int = 72, font_size: int = 24,
                                as_tensor: bool = False, view: bool = False):
    """Drop-in for _build_labeled_contact_sheet (objects_dir is unused there too: :191).

    The cutouts come through load_object_images (per-process decode cache) and its resident atlas -- the same
    upload every later composite of the bundle uses; thumbnails, tiling and the label blend are ONE
    mic_contact_sheet call (a single composite job inside libmic)."""
    with open(results_json_path, "r", encoding="utf-8") as f:
        items = json.load(f)
    items = sorted(items, key=lambda it: int(it["object_id"]))
    font = _resolve_font(font_size)
    tw_req, th_req, label_height, cols = int(thumb_size[0]), int(thumb_size[1]), int(label_height), int(cols)
    cell_w, cell_h = tw_req, th_req + label_height
    if not items:
        return SolidCanvas((cell_w, cell_h), (255, 255, 255, 255)).to_image()

    objects = load_object_images(results_json_path, shared=True)  # {id: image}; files missing -> FileNotFoundError like the reference
    atlas = objects.atlas()
    ids = [int(it["object_id"]) for it in items]
    labels = [str(it.get("label", f"id_{it['object_id']}")) for it in items]
    if len(set(ids)) != len(ids):
        # duplicate ids: the reference draws each file it lists; a dict keeps one per id -> private atlas by position
        cutouts = [open_rgba(Path(results_json_path).parent / it["filename"]) for it in items]
        atlas = Atlas({i: im for i, im in enumerate(cutouts)})
        ids = list(range(len(cutouts)))

    probe = ImageDraw.Draw(Image.new("RGBA", (1, 1)))
    strips = (_native.LabelStrip * max(len(ids), 1))()
    keep: List[np.ndarray] = []
    n_strips = 0
    for idx, (oid, label) in enumerate(zip(ids, labels)):
        r, c = divmod(idx, cols)
        x_cell, y_cell = c * cell_w, r * cell_h
        tw, th_text = _measure_label(probe, label, font)
        tx = x_cell + (cell_w - tw) // 2
        ty = y_cell + th_req + max(0, (label_height - th_text) // 2)
        strip = _label_strip(label, font, tx, ty)
        if strip is None:
            continue
        mask, sx, sy = strip
        keep.append(mask)
        st = strips[n_strips]
        st.cell, st.x, st.y, st.w, st.h = idx, int(sx), int(sy), int(mask.shape[1]), int(mask.shape[0])
        st.coverage_host = mask.ctypes.data
        n_strips += 1

    lib = _native.lib()
    W, H = ctypes.c_int32(), ctypes.c_int32()
    _native.check(lib.mic_contact_sheet_size(len(ids), tw_req, th_req, cols, label_height, ctypes.byref(W), ctypes.byref(H)))
    import torch
    ctx = atlas.ctx
    out = torch.empty((H.value, W.value, 4), dtype=torch.uint8, device=ctx.torch_device)
    ids_a = np.asarray(ids, np.int32)
    atlas.wait_ready()
    with _device_guard(ctx):
        _native.check(lib.mic_contact_sheet(ctx.handle, atlas.handle, len(ids), ids_a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                            tw_req, th_req, cols, label_height, n_strips, ctypes.cast(strips, ctypes.c_void_p),
                                            _P(out.data_ptr()), _P(ctx.stream_ptr())))
    del keep
    return out if as_tensor else _to_pil(out, view=view)


# the reference's (module-private) name, for callers that import it by that name
_build_labeled_contact_sheet = build_labeled_contact_sheet
