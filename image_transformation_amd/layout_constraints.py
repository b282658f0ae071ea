"""Canvas sizing for an aspect-ratio variant.

Mirrors layout_constraints.py:44-52 (parse_ratio) and :55-86 (compute_canvas_size) of
the reference: keep the pixel count, change the aspect ratio.  FP64 sqrt followed by
Python's round (half-to-even) -- the integer results are part of the bit-exact contract
because they define every variant's canvas.
"""
from __future__ import annotations

import math
from typing import Tuple


def parse_ratio(ratio: str) -> float:
    """'W:H' -> W/H.  ValueError on malformed or non-positive input (layout_constraints.py:44-52)."""
    fields = ratio.split(":")
    if len(fields) != 2:
        raise ValueError(f"Invalid ratio '{ratio}', expected W:H")
    num, den = float(fields[0]), float(fields[1])
    if num <= 0 or den <= 0:
        raise ValueError("Ratio components must be positive")
    return num / den


def compute_canvas_size(original_size: Tuple[int, int], ratio: str, quiet: bool = False) -> Tuple[int, int]:
    """(ow, oh), 'W:H' -> (tw, th) with tw*th ~= ow*oh and tw/th ~= W/H (layout_constraints.py:55-86).

    The reference prints a 'Canvas sizing:' line on every call (:84); `quiet=True` is a
    build-side extension for batch callers."""
    ow, oh = original_size
    pixels = ow * oh
    r = parse_ratio(ratio)
    tw = max(1, int(round(math.sqrt(pixels * r))))
    th = max(1, int(round(math.sqrt(pixels / r))))
    if not quiet:
        print(f"Canvas sizing: {ow}x{oh} ({pixels:,} px) → {tw}x{th} ({tw*th:,} px, ratio {tw/th:.3f})")
    return tw, th
