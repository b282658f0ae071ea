import os, sys, cProfile, pstats, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, cases
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, SolidCanvas, render
from image_transformation_amd.background_resizing import fill_solid
size, objs, layouts = synthetic.c3_workload("binary", seed=3, n_layouts=4)
atlas = Atlas(objs); canvas = SolidCanvas(size, synthetic.SOLID_BG)
k = [0]
def f():
    render(layouts[k[0] % 4], atlas, canvas, as_tensor=True); k[0] += 1
for _ in range(20): f()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): f()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
bg = os.path.join(cases.BUNDLE_DIR, "squarespace", "background.png")
g = lambda: fill_solid(bg, (492, 492))
g(); g()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): g()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
