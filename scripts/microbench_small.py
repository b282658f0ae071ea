"""The reference's own call at the reference's own size: composite(PIL background, {id: PIL}, placements) -> PIL on the
492 x 492 / 4-object bundle canvases, this package against the same loop through Pillow on the host."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, cases
from PIL import Image
from image_transformation_amd.compositor import composite, load_object_images, render, SolidCanvas
from image_transformation_amd.background_resizing import fill_solid
from image_transformation_amd import flex

def pillow_composite(bg, imgs, placements):
    canvas = bg.copy()
    for p in placements:
        im = imgs.get(int(p["object_id"]))
        if im is None:
            continue
        x1, y1, x2, y2 = [int(v) for v in p["box"]]
        w, h = max(1, x2 - x1), max(1, y2 - y1)
        r = im.resize((w, h), Image.LANCZOS)
        canvas.alpha_composite(r, dest=(x1, y1))
    return canvas

def timeit(fn, iters=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

with open(os.path.join(os.path.dirname(cases.BUNDLE_DIR), "bundles.json")) as f:
    rows = {r["name"]: r for r in json.load(f)["cases"]}
for b, case in (("squarespace", "squarespace_1x1"), ("audio_book", None)):
    base = os.path.join(cases.BUNDLE_DIR, b)
    rj = os.path.join(base, "results.json")
    objs = load_object_images(rj)
    row = rows.get(case) if case else next(r for n, r in rows.items() if n.startswith(b))
    size = tuple(row["canvas_size"]) if "canvas_size" in row else (492, 492)
    bg = fill_solid(os.path.join(base, "background.png"), size)
    pl = flex.layout_to_placements(row["layout"], objs, size)
    pil_objs = {k: objs[k] for k in objs}
    # placements mode: the same boxes grown by 20 % (forces LANCZOS)
    pl2 = [{"object_id": p["object_id"], "box": [p["box"][0], p["box"][1], p["box"][0] + int((p["box"][2] - p["box"][0]) * 1.2),
                                                  p["box"][1] + int((p["box"][3] - p["box"][1]) * 1.2)]} for p in pl]
    print(f"{b} {size} {len(pl)} objects:")
    print(f"   identity scale: this package {timeit(lambda: composite(bg, objs, pl)):7.1f} us   Pillow {timeit(lambda: pillow_composite(bg, pil_objs, pl)):7.1f} us")
    print(f"   LANCZOS x1.2  : this package {timeit(lambda: composite(bg, objs, pl2)):7.1f} us   Pillow {timeit(lambda: pillow_composite(bg, pil_objs, pl2)):7.1f} us")
    print(f"   render(layout, objs, SolidCanvas) -> PIL {timeit(lambda: render(row['layout'], objs, SolidCanvas(size, (220, 238, 245, 255)))):7.1f} us")
