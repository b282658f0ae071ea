"""cProfile of pipeline.run_layouts WITH its PNG artifacts on the squarespace bundle (3 iterations): where the ~6 ms go."""
import cProfile, json, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from image_transformation_amd.pipeline import run_layouts
with open(os.path.join(ROOT, "tests", "golden", "bundles.json")) as f:
    rows = {r["name"]: r for r in json.load(f)["cases"]}
lay = rows["squarespace_1x1"]["layout"]
base = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace")
td = tempfile.mkdtemp()
for _ in range(5):
    run_layouts(base, "1:1", [lay] * 3, output_root=td)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); run_layouts(base, "1:1", [lay] * 3, output_root=td); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"run_layouts, 3 iterations, save=True: median {ts[len(ts) // 2] * 1e3:.2f} ms, min {ts[0] * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    run_layouts(base, "1:1", [lay] * 3, output_root=td)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
