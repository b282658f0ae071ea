"""Seeded synthetic workloads for the compositor path (bench.py, tests, golden fixtures).

The reference ships no benchmark inputs; these generators define
the BASELINE.json configurations C2-C5 so that the CPU oracle,
the golden fixtures and the GPU path all see identical bytes.  NumPy PCG64 with a
fixed draw order: do not reorder the draws, committed fixtures depend on them.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

SOLID_BG = (38, 73, 115, 255)  # audio_book median colour (background_resizing.py:11-22)


def make_cutout(rng: np.random.Generator, w: int, h: int, alpha_mode: str = "binary") -> np.ndarray:
    """One RGBA cutout: random RGB; alpha 'binary' (elliptical blob, 0/255 as in the
    reference's bundles), 'soft' (uniform 0..255, worst case for rounding) or 'opaque'."""
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    if alpha_mode == "soft":
        return px
    if alpha_mode == "opaque":
        px[:, :, 3] = 255
        return px
    if alpha_mode != "binary":
        raise ValueError(f"unknown alpha_mode {alpha_mode!r}")
    yy = (np.arange(h, dtype=np.float64) + 0.5) / h - 0.5
    xx = (np.arange(w, dtype=np.float64) + 0.5) / w - 0.5
    inside = (xx[None, :] / 0.42) ** 2 + (yy[:, None] / 0.42) ** 2 <= 1.0
    px[:, :, 3] = np.where(inside, 255, 0).astype(np.uint8)
    return px


def make_cutouts(n: int, w_range: Tuple[int, int], h_range: Tuple[int, int], seed: int,
                 alpha_mode: str = "binary") -> Dict[int, np.ndarray]:
    """n cutouts with ids 1..n, sizes uniform in the inclusive ranges."""
    rng = np.random.default_rng(seed)
    out: Dict[int, np.ndarray] = {}
    for i in range(n):
        w = int(rng.integers(w_range[0], w_range[1] + 1))
        h = int(rng.integers(h_range[0], h_range[1] + 1))
        out[i + 1] = make_cutout(rng, w, h, alpha_mode)
    return out


def flex_row(ids: List[int], **kw) -> dict:
    node = {"type": "flex", "direction": "row", "children": [{"object_id": i} for i in ids]}
    node.update(kw)
    return node


def c2_workload(alpha_mode: str = "binary", seed: int = 2):
    """C2: 1920x1080, 8 objects, depth-1 row container (identity scale)."""
    W, H = 1920, 1080
    objs = make_cutouts(8, (160, 220), (300, 700), seed, alpha_mode)
    layout = {"root": flex_row(list(objs), justify="space_around", align="center", gap_px=8)}
    return (W, H), objs, layout


def c3_layout(ids: List[int], rng: np.random.Generator, rows: int = 4, transpose: bool = False) -> dict:
    """Depth-2 Flex tree: root column of `rows` row containers (or the transpose)."""
    per = (len(ids) + rows - 1) // rows
    justs = ["start", "center", "end", "space_between", "space_around"]
    aligns = ["start", "center", "end"]
    inner_dir, outer_dir = ("column", "row") if transpose else ("row", "column")
    children = []
    for r in range(rows):
        sub = ids[r * per:(r + 1) * per]
        if not sub:
            continue
        children.append({
            "type": "flex", "direction": inner_dir,
            "justify": justs[int(rng.integers(0, len(justs)))],
            "align": aligns[int(rng.integers(0, len(aligns)))],
            "gap_px": int(rng.integers(0, 17)),
            "children": [{"object_id": i} for i in sub],
        })
    return {"root": {"type": "flex", "direction": outer_dir,
                     "justify": justs[int(rng.integers(0, len(justs)))],
                     "align": aligns[int(rng.integers(0, len(aligns)))],
                     "gap_px": int(rng.integers(0, 17)), "children": children}}


def c3_workload(alpha_mode: str = "binary", seed: int = 3, n_layouts: int = 1):
    """C3: 3840x2160, 32 objects, depth-2 row/column Flex-DSL (identity scale).

    Returns ((W, H), objects, [layout...]); layout k permutes the children and redraws
    justify/align/gap so a batch of layouts is a batch of distinct canvases."""
    W, H = 3840, 2160
    objs = make_cutouts(32, (240, 440), (300, 500), seed, alpha_mode)
    rng = np.random.default_rng(seed + 1000)
    layouts = []
    ids = list(objs)
    for k in range(n_layouts):
        order = ids if k == 0 else [ids[j] for j in rng.permutation(len(ids))]
        layouts.append(c3_layout(order, rng))
    return (W, H), objs, layouts


def placements_workload(W: int, H: int, n: int, seed: int, alpha_mode: str = "soft",
                        scale_range: Tuple[float, float] = (0.5, 1.5), div: Tuple[int, int] = (8, 3)):
    """Placements mode (direct composite() callers): cutouts U[W/8,W/3]xU[H/8,H/3], boxes
    = cutout size x U[0.5,1.5] at random positions, overlaps and edge overhang allowed."""
    objs = make_cutouts(n, (max(1, W // div[0]), max(1, W // div[1])),
                        (max(1, H // div[0]), max(1, H // div[1])), seed, alpha_mode)
    rng = np.random.default_rng(seed + 2000)
    placements = []
    for oid, a in objs.items():
        sh, sw = a.shape[:2]
        s = float(rng.uniform(scale_range[0], scale_range[1]))
        w = max(1, int(round(sw * s)))
        h = max(1, int(round(sh * s)))
        x1 = int(rng.integers(-w // 4, W - (3 * w) // 4 + 1))
        y1 = int(rng.integers(-h // 4, H - (3 * h) // 4 + 1))
        placements.append({"object_id": oid, "box": [x1, y1, x1 + w, y1 + h]})
    return (W, H), objs, placements


def placement_sets(objs: Dict[int, np.ndarray], W: int, H: int, seed: int, n_sets: int,
                   scale_range: Tuple[float, float] = (0.5, 1.5)) -> List[List[dict]]:
    """n_sets further placement lists over the same cutouts (placements_workload's draw, other seeds): a batch of
    canvases whose layers all differ in scale, i.e. nothing for the resample dedup to share."""
    out = []
    for k in range(n_sets):
        rng = np.random.default_rng(seed + 3000 + k)
        placements = []
        for oid, a in objs.items():
            sh, sw = a.shape[:2]
            s = float(rng.uniform(scale_range[0], scale_range[1]))
            w = max(1, int(round(sw * s)))
            h = max(1, int(round(sh * s)))
            x1 = int(rng.integers(-w // 4, W - (3 * w) // 4 + 1))
            y1 = int(rng.integers(-h // 4, H - (3 * h) // 4 + 1))
            placements.append({"object_id": oid, "box": [x1, y1, x1 + w, y1 + h]})
        out.append(placements)
    return out


RATIOS_C4 = ("9:16", "1:1", "16:9", "21:9")


def c4_workload(alpha_mode: str = "binary", seed: int = 4, n_variants: int = 64):
    """C4: one 32-object bundle, n_variants = ratios x Flex JSONs; canvas sizes come from
    compute_canvas_size((3840, 2160), ratio) (layout_constraints.py:55-86)."""
    from .layout_constraints import compute_canvas_size

    objs = make_cutouts(32, (240, 440), (300, 500), seed, alpha_mode)
    rng = np.random.default_rng(seed + 1000)
    ids = list(objs)
    variants = []
    for v in range(n_variants):
        ratio = RATIOS_C4[v % len(RATIOS_C4)]
        W, H = compute_canvas_size((3840, 2160), ratio, quiet=True)
        order = [ids[j] for j in rng.permutation(len(ids))]
        portrait = H > W
        layout = c3_layout(order, rng, rows=8 if portrait else 4, transpose=False)
        variants.append(((W, H), layout))
    return objs, variants
