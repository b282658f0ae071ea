import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, cases
from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
base = os.path.join(cases.BUNDLE_DIR, sys.argv[1] if len(sys.argv) > 1 else "squarespace")
rj = os.path.join(base, "results.json")
f = lambda: build_labeled_contact_sheet(os.path.join(base, "objects"), rj)
f(); f()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): f()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
