"""Flex-DSL -> placement boxes: the integer layout half of render().

Host-side mirror of the reference's layout maths:

    _measure_flex_node        macro_placement_test.py:637-686  -> measure()
    _place_flex_container     macro_placement_test.py:689-951  -> place_container()
      (inner place_object_node  :706-847)                      -> _place_object()
    _sanitize_padding/_pin/_offset/_stick_to   :255-372        -> _norm_*()
    _clamp_boxes_to_canvas    macro_placement_test.py:954-964  -> clamp_boxes()

Everything here is integer maths on a JSON tree plus cutout *sizes*; no pixels.
The contract is bit-exact boxes (floor division for centring, Python's
round-half-even where the reference rounds floats), the same placement-dict
keys, and the same exception types/messages for malformed object fields.

Unlike the reference the two directions are not written out twice: a container is
laid out along a (main, cross) axis pair and mapped back to (x, y) at the end.
`images` may map id -> anything with a `.size` (PIL image, Atlas entry) or id -> (w, h).
"""
from __future__ import annotations

from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

ALIGN_VALUES = ("start", "center", "end")
_EDGES = ("left", "right", "top", "bottom")
_ZERO_PAD = {"left": 0, "right": 0, "top": 0, "bottom": 0}


def _size_of(images: Mapping[int, Any], oid: int) -> Optional[Tuple[int, int]]:
    ent = images.get(oid)
    if ent is None:
        return None
    size = getattr(ent, "size", ent)
    return int(size[0]), int(size[1])


# --------------------------------------------------------------------------- object fields
def _norm_padding(value: Any, oid: int) -> Dict[str, int]:
    """macro_placement_test.py:255-283 (bool passes as int, as in Python)."""
    if isinstance(value, int):
        if value < 0:
            raise ValueError(f"padding_px for object_id {oid} must be non-negative")
        return {k: value for k in _EDGES}
    if not isinstance(value, dict):
        raise ValueError(
            f"padding_px for object_id {oid} must be int or dict with left/right/top/bottom")
    unknown = sorted(set(value) - set(_EDGES))
    if unknown:
        raise ValueError(f"padding_px for object_id {oid} has unsupported keys: {unknown}")
    out = {}
    for side in _EDGES:
        v = value.get(side, 0)
        if not isinstance(v, int):
            raise ValueError(f"padding_px[{side}] for object_id {oid} must be an integer")
        if v < 0:
            raise ValueError(f"padding_px[{side}] for object_id {oid} must be non-negative")
        out[side] = v
    return out


def _norm_pin(value: Any, oid: int) -> Dict[str, str]:
    """macro_placement_test.py:286-306."""
    if value is None:
        return {}
    if not isinstance(value, dict):
        raise ValueError(f"pin for object_id {oid} must be an object with axis keys")
    unknown = sorted(set(value) - {"horizontal", "vertical"})
    if unknown:
        raise ValueError(f"pin for object_id {oid} has unsupported keys: {unknown}")
    out: Dict[str, str] = {}
    for axis in ("horizontal", "vertical"):
        v = value.get(axis)
        if v is None:
            continue
        if v not in ALIGN_VALUES:
            raise ValueError(
                f"pin.{axis} for object_id {oid} must be one of {sorted(ALIGN_VALUES)}")
        out[axis] = v
    return out


def _norm_offset(value: Any, oid: int) -> Dict[str, int]:
    """macro_placement_test.py:309-325."""
    if value is None:
        return {"x": 0, "y": 0}
    if not isinstance(value, dict):
        raise ValueError(f"offset_px for object_id {oid} must be an object with x/y")
    unknown = sorted(set(value) - {"x", "y"})
    if unknown:
        raise ValueError(f"offset_px for object_id {oid} has unsupported keys: {unknown}")
    out = {}
    for axis in ("x", "y"):
        v = value.get(axis, 0)
        if not isinstance(v, int):
            raise ValueError(f"offset_px.{axis} for object_id {oid} must be an integer")
        out[axis] = v
    return out


def _norm_stick_to(value: Any, oid: int) -> Dict[str, Any]:
    """macro_placement_test.py:328-372."""
    if value is None:
        return {}
    if not isinstance(value, dict):
        raise ValueError(f"stick_to for object_id {oid} must be an object with edges and margin_px")
    unknown = sorted(set(value) - {"edges", "margin_px"})
    if unknown:
        raise ValueError(f"stick_to for object_id {oid} has unsupported keys: {unknown}")
    edges = value.get("edges")
    if not isinstance(edges, list) or not edges:
        raise ValueError(f"stick_to.edges for object_id {oid} must be a non-empty list")
    seen: List[str] = []
    for e in edges:
        if not isinstance(e, str):
            raise ValueError(f"stick_to.edges entries for object_id {oid} must be strings")
        low = e.lower()
        if low not in _EDGES:
            raise ValueError(f"stick_to.edge '{e}' for object_id {oid} is not supported")
        if low in seen:
            raise ValueError(f"stick_to.edges for object_id {oid} contains duplicate '{low}'")
        seen.append(low)
    for a, b in (("left", "right"), ("top", "bottom")):
        if a in seen and b in seen:
            raise ValueError(
                f"stick_to.edges for object_id {oid} cannot include both '{a}' and '{b}'")
    margin = value.get("margin_px", 0)
    if not isinstance(margin, int):
        raise ValueError(f"stick_to.margin_px for object_id {oid} must be an integer")
    if margin < 0:
        raise ValueError(f"stick_to.margin_px for object_id {oid} must be non-negative")
    return {"edges": seen, "margin_px": margin}


# --------------------------------------------------------------------------- measure
def measure(node: Any, images: Mapping[int, Any]) -> Tuple[int, int]:
    """Intrinsic (w, h) of an object or container node (macro_placement_test.py:637-686)."""
    if isinstance(node, dict) and "object_id" in node:
        try:
            oid = int(node["object_id"])
        except Exception:
            return 0, 0
        raw = node.get("padding_px")
        pad = _norm_padding(raw, oid) if raw is not None else _ZERO_PAD
        w, h = _size_of(images, oid) or (0, 0)
        return max(0, w + pad["left"] + pad["right"]), max(0, h + pad["top"] + pad["bottom"])

    gap = int(node.get("gap_px", 0))
    pad = int(node.get("padding_px", 0))
    kids = node.get("children", []) or []
    if not kids:
        return max(0, 2 * pad), max(0, 2 * pad)
    sizes = [measure(k, images) if isinstance(k, dict) else (0, 0) for k in kids]
    gaps = gap * (len(sizes) - 1) if len(sizes) > 1 else 0
    if node.get("direction", "row") == "row":
        w = sum(s[0] for s in sizes) + gaps
        h = max(s[1] for s in sizes)
    else:
        w = max(s[0] for s in sizes)
        h = sum(s[1] for s in sizes) + gaps
    grow = 2 * max(0, pad)
    return int(max(0, w + grow)), int(max(0, h + grow))


# --------------------------------------------------------------------------- place
def _settle(lo, hi, target, mode, stick_near, stick_far, margin, shift):
    """One axis of an object inside its padded slot [lo, hi] (macro_placement_test.py:767-828).

    Float centring followed by round-half-even, then pushed back inside the slot."""
    spare = max(0, (hi - lo) - target)
    if mode == "center":
        pos = lo + spare / 2
    elif mode == "end":
        pos = hi - target
    else:
        pos = lo
    if stick_near:
        pos = lo + margin
    elif stick_far:
        pos = hi - margin - target
    pos += shift
    upper = hi - target
    if upper < lo:
        upper = lo
    pos = min(max(pos, lo), upper)
    a = int(round(pos))
    b = a + int(target)
    if b > hi:
        a, b = a - (b - hi), hi
    if a < lo:
        a, b = lo, b + (lo - a)
    return a, b


def _place_object(node: dict, slot_xy, slot_wh, images, out: List[dict], cell: str,
                  direction: str, align: str) -> None:
    oid = int(node.get("object_id", -1))
    size = _size_of(images, oid)
    iw, ih = size if size is not None else (0, 0)

    pad_raw = node.get("padding_px")
    pad = _norm_padding(pad_raw, oid) if pad_raw is not None else dict(_ZERO_PAD)
    pin_raw = node.get("pin")
    pin = _norm_pin(pin_raw, oid) if pin_raw is not None else {}
    off_raw = node.get("offset_px")
    off = _norm_offset(off_raw, oid) if off_raw is not None else {"x": 0, "y": 0}
    stick_raw = node.get("stick_to")
    stick = _norm_stick_to(stick_raw, oid) if stick_raw is not None else {}

    x_lo = slot_xy[0] + pad["left"]
    y_lo = slot_xy[1] + pad["top"]
    x_hi = max(x_lo, slot_xy[0] + slot_wh[0] - pad["right"])
    y_hi = max(y_lo, slot_xy[1] + slot_wh[1] - pad["bottom"])
    avail_w = max(0, x_hi - x_lo)
    avail_h = max(0, y_hi - y_lo)

    scale = 1.0
    if size is not None and iw > 0 and ih > 0:
        cands = [1.0]
        if avail_w > 0:
            cands.append(avail_w / iw)
        if avail_h > 0:
            cands.append(avail_h / ih)
        scale = max(0.0, min(cands))
        tw = int(round(iw * scale))
        th = int(round(ih * scale))
    else:
        tw, th = avail_w, avail_h
    tw = max(0, min(tw, avail_w))
    th = max(0, min(th, avail_h))

    h_mode = pin.get("horizontal")
    if h_mode is None:
        h_mode = align if direction == "column" else "start"
    v_mode = pin.get("vertical")
    if v_mode is None:
        v_mode = align if direction == "row" else "start"

    edges = stick.get("edges", []) if stick else []
    margin = stick.get("margin_px", 0) if stick else 0
    x1, x2 = _settle(x_lo, x_hi, tw, h_mode, "left" in edges, "right" in edges, margin,
                     off.get("x", 0))
    y1, y2 = _settle(y_lo, y_hi, th, v_mode, "top" in edges, "bottom" in edges, margin,
                     off.get("y", 0))

    entry: Dict[str, Any] = {"object_id": oid, "cell": cell,
                             "box": [int(x1), int(y1), int(x2), int(y2)], "scale": float(scale)}
    if pad_raw is not None:
        entry["padding_px"] = pad
    if pin_raw is not None and pin:
        entry["pin"] = pin
    if off_raw is not None or off.get("x", 0) or off.get("y", 0):
        entry["offset_px"] = off
    if stick_raw is not None and stick:
        entry["stick_to"] = stick
    out.append(entry)


def _main_start_and_gap(justify: str, n: int, inner: int, total: int, sum_sizes: int, gap: int):
    """Main-axis start offset and inter-child gap (macro_placement_test.py:863-886, :909-931)."""
    if justify == "start":
        return 0, gap
    if justify == "center":
        return max(0, (inner - total) // 2), gap
    if justify == "end":
        return max(0, inner - total), gap
    if justify == "space_between" and n > 1:
        return 0, max(0, (inner - sum_sizes) // (n - 1))
    if justify == "space_around" and n > 0:
        g = max(0, (inner - sum_sizes) // n)
        return g // 2, g
    return 0, gap


def place_container(node: dict, origin: Tuple[int, int], size: Tuple[int, int],
                    images: Mapping[int, Any], placements: List[dict], parent_cell: str) -> None:
    """Recursive placement, depth-first child order (macro_placement_test.py:689-951)."""
    direction = node.get("direction", "row")
    justify = node.get("justify", "center")
    align = node.get("align", "center")
    gap = int(node.get("gap_px", 0))
    pad = int(node.get("padding_px", 0))
    is_row = direction == "row"
    m, c = (0, 1) if is_row else (1, 0)  # index of the main / cross axis in (x, y)

    inner_org = (origin[0] + pad, origin[1] + pad)
    inner_len = (max(0, size[0] - 2 * pad), max(0, size[1] - 2 * pad))

    kids = node.get("children", [])
    sizes: List[Tuple[int, int]] = []
    for k in kids:
        if "object_id" in k:
            try:
                int(k["object_id"])
            except Exception:
                sizes.append((0, 0))
                continue
        sizes.append(measure(k, images))

    n = len(kids)
    sum_main = sum(s[m] for s in sizes)
    total = sum_main + gap * (n - 1 if n > 0 else 0)
    start, step_gap = _main_start_and_gap(justify, n, inner_len[m], total, sum_main, gap)
    cur = inner_org[m] + start

    for k, s in zip(kids, sizes):
        if align == "start":
            cross = inner_org[c]
        elif align == "end":
            cross = inner_org[c] + (inner_len[c] - s[c])
        else:  # "center" and any unknown value
            cross = inner_org[c] + (inner_len[c] - s[c]) // 2
        pos = (cur, cross) if is_row else (cross, cur)
        if "object_id" in k:
            _place_object(k, pos, s, images, placements, parent_cell, direction, align)
        else:
            place_container(k, pos, s, images, placements, parent_cell)
        cur = cur + s[m] + step_gap


def clamp_boxes(placements: List[dict], canvas_size: Tuple[int, int]) -> None:
    """In place: keep w,h, push the box back inside the canvas (macro_placement_test.py:954-964)."""
    W, H = canvas_size
    for p in placements:
        x1, y1, x2, y2 = p["box"]
        w, h = x2 - x1, y2 - y1
        x1 = max(0, min(x1, W - w))
        y1 = max(0, min(y1, H - h))
        p["box"] = [int(x1), int(y1), int(x1 + w), int(y1 + h)]


def native_boxes(layout: Any, images: Mapping[int, Any],
                 canvas_size: Tuple[int, int]) -> Optional[List[Tuple[int, int, int, int, int]]]:
    """Placement boxes [(object_id, x1, y1, x2, y2), ...] from libmic's native placer
    (mic_flex_place, csrc/flex_place.cpp), or None when the layout is not a {"root": ...} tree or
    uses something the native placer leaves to this module (then call layout_to_placements, which
    also raises the reference's errors).  `layout` may be the JSON text itself (a VLM reply) or the
    parsed dict.  Host-only: no GPU is touched."""
    import ctypes
    import json

    import numpy as np

    from . import _native

    if isinstance(layout, (bytes, str)):
        text = layout.encode("utf-8") if isinstance(layout, str) else layout
    elif isinstance(layout, dict) and "root" in layout:
        try:
            text = json.dumps(layout, separators=(",", ":")).encode("utf-8")
        except (TypeError, ValueError):
            return None
    else:
        return None
    i32p = ctypes.POINTER(ctypes.c_int32)
    tab = getattr(images, "_native_table", None)
    if tab is None:
        ids, ws, hs = [], [], []
        for oid in images:
            size = _size_of(images, oid)
            if not isinstance(oid, int) or isinstance(oid, bool) or size is None:
                return None
            ids.append(oid)
            ws.append(size[0])
            hs.append(size[1])
        arrs = (np.asarray(ids, np.int32), np.asarray(ws, np.int32), np.asarray(hs, np.int32))
        tab = (len(ids), arrs, tuple(a.ctypes.data_as(i32p) for a in arrs))
        try:
            images._native_table = tab  # Atlas / ObjectImages keep it; plain dicts recompute
        except AttributeError:
            pass
    n_obj, _, (ids_p, ws_p, hs_p) = tab
    cap = 4 * n_obj + 64
    lib = _native.lib()
    while True:
        buf = _scratch(cap)
        n = ctypes.c_int32(-1)
        rc = lib.mic_flex_place(text, len(text), n_obj, ids_p, ws_p, hs_p, int(canvas_size[0]), int(canvas_size[1]),
                                cap, buf[2], buf[3], ctypes.byref(n))
        if rc == 0:
            k = n.value
            o, b = buf[0][:k].tolist(), buf[1][:4 * k].tolist()
            return [(o[i], b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]) for i in range(k)]
        if n.value > cap:  # a layout that mentions objects many times over
            cap = n.value
            continue
        return None  # unsupported or malformed: the Python placer decides what that means


_scratch_buf: List[Any] = []


def _scratch(cap: int):
    """Reusable output arrays (ids, boxes) and their ctypes pointers for native_boxes."""
    import ctypes

    import numpy as np

    if not _scratch_buf or len(_scratch_buf[0]) < cap:
        i32p = ctypes.POINTER(ctypes.c_int32)
        ids, boxes = np.empty(cap, np.int32), np.empty(4 * cap, np.int32)
        _scratch_buf[:] = [ids, boxes, ids.ctypes.data_as(i32p), boxes.ctypes.data_as(i32p)]
    return _scratch_buf


def layout_to_placements(layout: Any, images: Mapping[int, Any],
                         canvas_size: Tuple[int, int]) -> List[dict]:
    """layout_json -> placements, the way run_macro_only does it before composite()
    (macro_placement_test.py:1495-1498): {"root": ...} is placed from (0,0) with the
    canvas size as the root container's size, then clamped; {"placements": [...]} or a
    bare list is used as-is.  A dict without "root"/"placements" raises KeyError('root')
    like the reference's caller."""
    if isinstance(layout, (list, tuple)):
        return list(layout)
    if "root" not in layout and "placements" in layout:
        return list(layout["placements"])
    root = layout["root"]
    out: List[dict] = []
    place_container(root, (0, 0), (int(canvas_size[0]), int(canvas_size[1])), images, out,
                    "flex_root")
    clamp_boxes(out, canvas_size)
    return out


# The reference's (module-private) names, for callers that import the box maths by those names
# (macro_placement_test.py:255-372, 637-964): same signatures, same results, same error messages.
_measure_flex_node = measure
_place_flex_container = place_container
_clamp_boxes_to_canvas = clamp_boxes
_sanitize_padding, _sanitize_pin, _sanitize_offset, _sanitize_stick_to = _norm_padding, _norm_pin, _norm_offset, _norm_stick_to
