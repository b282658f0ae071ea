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
import json
from pathlib import Path
from typing import List, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

from . import _native
from .compositor import open_rgba, Atlas, SolidCanvas, _to_pil, LANCZOS

_P = ctypes.c_void_p
_MARGIN = 32


def thumbnail_size(size: Tuple[int, int], req: Tuple[int, int]) -> Tuple[int, int]:
    """Pillow's Image.thumbnail size rule (macro_placement_test.py:194)."""
    ow, oh = ctypes.c_int32(), ctypes.c_int32()
    _native.check(_native.lib().mic_thumbnail_size(int(size[0]), int(size[1]), int(req[0]), int(req[1]),
                                                   ctypes.byref(ow), ctypes.byref(oh)))
    return ow.value, oh.value


def _resolve_font(font_size: int):
    """Same fallback chain as macro_placement_test.py:176-186."""
    for name in ("DejaVuSans.ttf", "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf"):
        try:
            return ImageFont.truetype(name, size=font_size)
        except Exception:
            continue
    try:
        return ImageFont.load_default()
    except Exception:
        return None


def _label_strip(label: str, font, tx: int, ty: int):
    """Coverage mask of draw.text((tx, ty), label) as (rgba strip, x, y) or None if no ink."""
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
    m = np.asarray(mask_img.crop(box), np.uint8)
    strip = np.zeros(m.shape + (4,), np.uint8)
    strip[:, :, 3] = m
    return strip, tx - _MARGIN + box[0], ty - _MARGIN + box[1]


def _measure_label(draw, label: str, font) -> Tuple[int, int]:
    """Text size with the reference's fallbacks (macro_placement_test.py:222-238)."""
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
                                as_tensor: bool = False):
    """Drop-in for _build_labeled_contact_sheet (objects_dir is unused there too: :191)."""
    with open(results_json_path, "r", encoding="utf-8") as f:
        items = json.load(f)
    items = sorted(items, key=lambda it: int(it["object_id"]))
    font = _resolve_font(font_size)

    cutouts: List[Image.Image] = []
    labels: List[str] = []
    for it in items:
        cutouts.append(open_rgba(Path(results_json_path).parent / it["filename"]))
        labels.append(str(it.get("label", f"id_{it['object_id']}")))

    cell_w, cell_h = int(thumb_size[0]), int(thumb_size[1]) + int(label_height)
    if not cutouts:
        sheet = SolidCanvas((cell_w, cell_h), (255, 255, 255, 255))
        return sheet.to_image()

    rows_n = (len(cutouts) + cols - 1) // cols
    sheet = SolidCanvas((cols * cell_w, rows_n * cell_h), (255, 255, 255, 255))
    probe = ImageDraw.Draw(Image.new("RGBA", (1, 1)))

    # layers: cutouts keep indices 0..n-1, label strips n..2n-1, all in one atlas
    pixels = {i: im for i, im in enumerate(cutouts)}
    rows: List[Tuple[int, int, int, int, int]] = []
    n = len(cutouts)
    for idx, (im, label) in enumerate(zip(cutouts, labels)):
        r, c = divmod(idx, cols)
        x_cell, y_cell = c * cell_w, r * cell_h
        tw_, th_ = thumbnail_size(im.size, (int(thumb_size[0]), int(thumb_size[1])))
        x = x_cell + (cell_w - tw_) // 2
        y = y_cell + (int(thumb_size[1]) - th_) // 2
        rows.append((idx, x, y, x + tw_, y + th_))
        tw, th_text = _measure_label(probe, label, font)
        tx = x_cell + (cell_w - tw) // 2
        ty = y_cell + int(thumb_size[1]) + max(0, (label_height - th_text) // 2)
        strip = _label_strip(label, font, tx, ty)
        if strip is not None:
            arr, sx, sy = strip
            pixels[n + idx] = arr
            rows.append((n + idx, sx, sy, sx + arr.shape[1], sy + arr.shape[0]))

    from .compositor import composite_device

    atlas = Atlas(pixels)
    out = composite_device(atlas, [sheet], [rows], filter=LANCZOS)[0]
    return out if as_tensor else _to_pil(out)


# the reference's (module-private) name, for callers that import it by that name
_build_labeled_contact_sheet = build_labeled_contact_sheet
