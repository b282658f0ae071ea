"""Kernel and wall time of single-canvas composites (the reference's call shape: one composite() per call)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements, render
for name, (size, objs, layouts) in (("C3 4K/32", synthetic.c3_workload("binary", 3, 1)), ("C2 1080p/8", synthetic.c2_workload("binary", 2))):
    if isinstance(layouts, dict):
        layouts = [layouts]
    atlas = Atlas(objs)
    ctx = atlas.ctx
    rows = [coerce_placements(atlas, flex.layout_to_placements(layouts[0], atlas, size))]
    solid = SolidCanvas(size, synthetic.SOLID_BG)
    one = CompositeBatch(atlas, [solid], rows)
    outs = [one.alloc_outputs() for _ in range(12)]
    for k in range(10):
        one.run(outs[k % 12])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        one.run(outs[k % 12], check=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 200
    ctx.profile_begin(100)
    for k in range(100):
        one.run(outs[k % 12], check=False)
    n, c, _ = ctx.profile_end()
    st = one.stats()
    b = 4 * (st["canvas_pixels"] + st["layer_pixels"])
    print(f"{name}: kernel {c / n * 1e3:.2f} us ({b / (c / n * 1e-3) / 8e12:.3f} of HBM peak), wall {wall * 1e6:.1f} us per canvas")
    text = json.dumps(layouts[0])
    for _ in range(5):
        render(text, atlas, solid, as_tensor=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        render(text, atlas, solid, as_tensor=True)
    torch.cuda.synchronize()
    print(f"   render(json text, atlas, SolidCanvas, as_tensor=True): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
