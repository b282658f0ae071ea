import cProfile, pstats, json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import torch, cases
from image_transformation_amd.compositor import composite, load_object_images
from image_transformation_amd.background_resizing import fill_solid
from image_transformation_amd import flex
with open(os.path.join(os.path.dirname(cases.BUNDLE_DIR), "bundles.json")) as f:
    row = next(r for r in json.load(f)["cases"] if r["name"] == "squarespace_1x1")
base = os.path.join(cases.BUNDLE_DIR, "squarespace")
objs = load_object_images(os.path.join(base, "results.json"))
bg = fill_solid(os.path.join(base, "background.png"), (492, 492))
pl = flex.layout_to_placements(row["layout"], objs, (492, 492))
for _ in range(20): composite(bg, objs, pl)
pr = cProfile.Profile(); pr.enable()
for _ in range(500): composite(bg, objs, pl)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
