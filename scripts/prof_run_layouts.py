"""cProfile of pipeline.run_layouts (save=False) on the squarespace bundle: where the host time of an iteration goes."""
import cProfile, json, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, cases
from image_transformation_amd.pipeline import run_layouts
with open(os.path.join(os.path.dirname(cases.BUNDLE_DIR), "bundles.json")) as f:
    rows = {r["name"]: r for r in json.load(f)["cases"]}
lay = rows["squarespace_1x1"]["layout"]
base = os.path.join(cases.BUNDLE_DIR, "squarespace")
n_it = int(os.environ.get("MIC_ITERS", "8"))
for _ in range(3):
    run_layouts(base, "1:1", [lay] * n_it, save=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run_layouts(base, "1:1", [lay] * n_it, save=False)
torch.cuda.synchronize()
print(f"run_layouts, {n_it} iterations, save=False: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    run_layouts(base, "1:1", [lay] * n_it, save=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
