"""Latency of the reference-shaped entry points (one canvas per call, like run_macro_only's loop)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from image_transformation_amd import synthetic, flex
from image_transformation_amd.compositor import Atlas, SolidCanvas, render, composite, CompositeBatch, coerce_placements
from image_transformation_amd.background_resizing import solid_canvas

def timeit(fn, iters=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

for name, (size, objs, layouts) in {"C3 4K/32": synthetic.c3_workload("binary", seed=3, n_layouts=4),
                                     "C2 1080p/8": synthetic.c2_workload("binary", seed=2) if hasattr(synthetic, "c2_workload") else None}.items():
    if size is None: continue
    if not isinstance(layouts, list): layouts = [layouts]
    atlas = Atlas(objs)
    canvas = SolidCanvas(size, synthetic.SOLID_BG)
    text = [json.dumps(l) for l in layouts]
    k = [0]
    def f_dict():
        render(layouts[k[0] % len(layouts)], atlas, canvas, as_tensor=True); k[0] += 1
    def f_text():
        render(text[k[0] % len(text)], atlas, canvas, as_tensor=True); k[0] += 1
    print(f"{name}: render(dict, as_tensor) {timeit(f_dict):8.1f} us   render(json text, as_tensor) {timeit(f_text):8.1f} us")
    pil = {k_: Image.fromarray(v, "RGBA") for k_, v in objs.items()}
    from image_transformation_amd.compositor import ObjectImages
    oi = ObjectImages(pil)
    pl = flex.layout_to_placements(layouts[0], oi, size)
    bgimg = Image.new("RGBA", size, tuple(synthetic.SOLID_BG))
    print(f"{name}: composite(PIL bg, dict, placements) -> PIL {timeit(lambda: composite(bgimg, oi, pl), iters=20, warm=3):10.1f} us")
    print(f"{name}: render(dict, PIL out)                   {timeit(lambda: render(layouts[0], oi, canvas), iters=20, warm=3):10.1f} us")
