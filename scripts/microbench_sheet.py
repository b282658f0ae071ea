"""Contact sheet / fill_solid / run_layouts wall times on the committed bundles."""
import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, cases
from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
from image_transformation_amd.background_resizing import fill_solid, solid_canvas
from image_transformation_amd.pipeline import run_layouts

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for b in cases.BUNDLES:
    base = os.path.join(cases.BUNDLE_DIR, b)
    rj = os.path.join(base, "results.json")
    print(b, "contact sheet %.2f ms" % timeit(lambda: build_labeled_contact_sheet(os.path.join(base, "objects"), rj)),
          " fill_solid(492x492) %.2f ms" % timeit(lambda: fill_solid(os.path.join(base, "background.png"), (492, 492))),
          " solid_canvas %.2f ms" % timeit(lambda: solid_canvas(os.path.join(base, "background.png"), (492, 492))))
with open(os.path.join(os.path.dirname(cases.BUNDLE_DIR), "bundles.json")) as f:
    rows = {r["name"]: r for r in json.load(f)["cases"]}
lay = rows["squarespace_1x1"]["layout"]
base = os.path.join(cases.BUNDLE_DIR, "squarespace")
with tempfile.TemporaryDirectory() as td:
    print("run_layouts(squarespace, 1:1, 3 iterations, save=True) %.1f ms" % timeit(lambda: run_layouts(base, "1:1", [lay] * 3, output_root=td), iters=5, warm=1))
    print("run_layouts(squarespace, 1:1, 3 iterations, save=False) %.1f ms" % timeit(lambda: run_layouts(base, "1:1", [lay] * 3, save=False), iters=5, warm=1))
