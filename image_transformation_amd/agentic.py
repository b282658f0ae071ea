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
from typing import Any, Dict, Mapping, Tuple

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


def _spacing(node: Mapping, key: str) -> int:
    """`gap_px` / `padding_px` of a container: int() of whatever the JSON holds (so "7" is 7 and "x", None or [1] raise
    int()'s own ValueError / TypeError), negative values refused with the reference's message."""
    value = int(node.get(key, 0))
    if value < 0:
        raise ValueError(f"{key} cannot be negative")
    return value


def placements_from_flex(flex: Dict, canvas_size: Tuple[int, int],
                         objects: Mapping[int, ObjectMeta]) -> Dict[int, PlacementState]:
    """agentic/utils/layout.py:52-121 (`_place_node` + `placements_from_flex`): children are packed from the start of the
    main axis, NOT aligned on the cross axis, the whole tree starts at (0, 0); a layout larger than the canvas or one
    that misses an object is refused.  (A bad `direction` never raises here: anything that is not "row" is a column,
    as in the reference's place helper.)

    The reference recurses; this walks the tree with an explicit stack of open containers, which visits the nodes --
    and therefore meets the errors, and fills the result dict -- in the same order: a container's gap and padding are
    read when it is opened (gap first), an empty one raises right then, a leaf looks its object up when it is reached
    (KeyError for an unknown id), a later leaf with the same id replaces the earlier entry in place."""
    if "root" not in flex:
        raise ValueError("Flex JSON must include 'root'")
    out: Dict[int, PlacementState] = {}

    class _Open:  # one container being filled: where its next child goes, what its children add up to so far
        __slots__ = ("row", "gap", "pad", "todo", "x", "y", "main", "cross", "count")

        def __init__(self, node: Mapping, x: int, y: int):
            self.row = node.get("direction") == "row"
            self.gap = _spacing(node, "gap_px")
            self.pad = _spacing(node, "padding_px")
            children = node.get("children", [])
            if not children:
                raise ValueError("container must have at least one child")
            self.todo = iter(children)
            self.x, self.y = x + self.pad, y + self.pad
            self.main = self.cross = self.count = 0

        def add(self, w: int, h: int) -> None:  # a finished child of size (w, h): advance the cursor along the main axis
            along, across = (w, h) if self.row else (h, w)
            self.main += along
            self.cross = max(self.cross, across)
            self.count += 1
            if self.row:
                self.x += w + self.gap
            else:
                self.y += h + self.gap

        def size(self) -> Tuple[int, int]:
            main = self.main + self.gap * (self.count - 1) + 2 * self.pad
            cross = self.cross + 2 * self.pad
            return (main, cross) if self.row else (cross, main)

    def leaf(node: Mapping, x: int, y: int) -> Tuple[int, int]:
        oid = int(node["object_id"])
        meta = objects[oid]
        out[oid] = PlacementState(object_id=oid, name=meta.name, x=x, y=y, width=meta.width, height=meta.height)
        return meta.width, meta.height

    root = flex["root"]
    if "object_id" in root:
        total = leaf(root, 0, 0)
    else:
        stack = [_Open(root, 0, 0)]
        total = (0, 0)
        while stack:
            top = stack[-1]
            child = next(top.todo, None)
            if child is None:  # every child placed: the container's own size goes to its parent
                stack.pop()
                total = top.size()
                if stack:
                    stack[-1].add(*total)
            elif "object_id" in child:
                top.add(*leaf(child, top.x, top.y))
            else:
                stack.append(_Open(child, top.x, top.y))
    if total[0] > canvas_size[0] or total[1] > canvas_size[1]:
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
