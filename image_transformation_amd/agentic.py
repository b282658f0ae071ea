"""Second caller of the pixel path: the reference's experimental LangGraph variant (SURVEY.md
section 8f row 2).  Only its deterministic pieces are mirrored -- the start-packed Flex placer
(agentic/utils/layout.py:23-121) and the compositor node's pixel work
(agentic/nodes/compositor.py:14-54: fill_solid + alpha-over with NO resizing, ValueError on a size
mismatch).  The graph, the VLM nodes and the artifact writers are out of scope.

Parity: `placements_from_flex` is pinned by tests/golden/agentic.json -- the reference's own
function run by tests/golden/make_golden.py on 489 seeded trees (placements in dict order, or the
exception type and message) -- and cross-checked against the main Flex placer (flex.py) on trees
where the two DSL dialects coincide (justify=start, align=start).  The pixel work is the
identity-size subset of compositor.composite, which is pinned.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Mapping, Tuple

from PIL import Image


@dataclass
class ObjectMeta:
    """agentic/state.py:13-21."""
    object_id: int
    name: str
    filename: str
    width: int
    height: int


@dataclass
class PlacementState:
    """agentic/state.py:24-42: absolute placement of one object."""
    object_id: int
    name: str
    x: int
    y: int
    width: int
    height: int

    def move_dx(self, delta: int) -> None:
        self.x += delta

    def move_dy(self, delta: int) -> None:
        self.y += delta


def _non_negative(value: int, label: str) -> int:
    if value < 0:
        raise ValueError(f"{label} cannot be negative")
    return value


def _place(node: Dict, origin: Tuple[int, int], objects: Mapping[int, ObjectMeta],
           out: Dict[int, PlacementState]) -> Tuple[int, int]:
    """Place `node` with its top-left corner at `origin`; returns its (w, h).  Children are packed
    from the start of the main axis and NOT aligned on the cross axis (agentic/utils/layout.py:55-104)."""
    if "object_id" in node:
        oid = int(node["object_id"])
        meta = objects[oid]  # KeyError for unknown ids, as in the reference
        out[oid] = PlacementState(object_id=oid, name=meta.name, x=origin[0], y=origin[1],
                                  width=meta.width, height=meta.height)
        return meta.width, meta.height
    direction = node.get("direction")
    gap = _non_negative(int(node.get("gap_px", 0)), "gap_px")
    pad = _non_negative(int(node.get("padding_px", 0)), "padding_px")
    children = node.get("children", [])
    if not children:
        raise ValueError("container must have at least one child")
    x, y = origin[0] + pad, origin[1] + pad
    sizes: List[Tuple[int, int]] = []
    for child in children:
        w, h = _place(child, (x, y), objects, out)
        sizes.append((w, h))
        if direction == "row":
            x += w + gap
        else:
            y += h + gap
    if direction == "row":
        total = (sum(s[0] for s in sizes) + gap * (len(sizes) - 1), max(s[1] for s in sizes))
    else:
        total = (max(s[0] for s in sizes), sum(s[1] for s in sizes) + gap * (len(sizes) - 1))
    return total[0] + 2 * pad, total[1] + 2 * pad


def placements_from_flex(flex: Dict, canvas_size: Tuple[int, int],
                         objects: Mapping[int, ObjectMeta]) -> Dict[int, PlacementState]:
    """agentic/utils/layout.py:106-121: place from (0,0); reject layouts larger than the canvas and
    layouts that miss an object.  (A bad `direction` only raises where the reference's measure
    helper would be reached; the place helper treats any non-"row" value as a column, as there.)"""
    if "root" not in flex:
        raise ValueError("Flex JSON must include 'root'")
    out: Dict[int, PlacementState] = {}
    w, h = _place(flex["root"], (0, 0), objects, out)
    if w > canvas_size[0] or h > canvas_size[1]:
        raise ValueError("Flex DSL produces placements larger than canvas; revise macro layout")
    missing = set(objects.keys()) - set(out.keys())
    if missing:
        raise ValueError(f"Placement missing required object ids: {sorted(missing)}")
    return out


def composite_placements(canvas: Any, object_images: Mapping[int, Any],
                         placements: Mapping[int, PlacementState], *, as_tensor: bool = False):
    """The pixel part of the compositor node (agentic/nodes/compositor.py:36-43): for each placement
    in dict order, the cutout must already have the placement's size (ValueError otherwise -- the
    node never scales) and is alpha-composited at (x, y).  `canvas` is anything render() accepts
    (RGBA image, SolidCanvas from background_resizing.solid_canvas, device tensor)."""
    from .compositor import render  # deferred: importing this module must not touch the GPU

    boxes = []
    for pl in placements.values():
        img = object_images[pl.object_id]
        size = tuple(img.size) if not isinstance(img, tuple) else img
        if size != (pl.width, pl.height):
            raise ValueError("Placement size mismatch; scaling objects is not permitted")
        boxes.append({"object_id": pl.object_id, "box": [pl.x, pl.y, pl.x + pl.width, pl.y + pl.height]})
    return render({"placements": boxes}, object_images, canvas, as_tensor=as_tensor)


def compositor_node(background_path: str, canvas_size: Tuple[int, int], object_images: Mapping[int, Image.Image],
                    placements: Mapping[int, PlacementState]) -> Image.Image:
    """fill_solid + composite_placements: what the node renders before it saves the PNG
    (agentic/nodes/compositor.py:17, :36-46)."""
    from .background_resizing import solid_canvas

    return composite_placements(solid_canvas(background_path, canvas_size), object_images, placements)
