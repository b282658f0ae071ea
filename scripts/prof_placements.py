"""A few runs of the C3 placements-mode (LANCZOS) composite for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, os.environ.get("MIC_ALPHA", "soft"))
atlas = Atlas(objs)
plan = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(atlas, pl)])
out = plan.alloc_outputs()
cold = os.environ.get("MIC_WARM") != "1"  # default: every run resamples again (mic_plan_invalidate); MIC_WARM=1: layers resident
for _ in range(int(os.environ.get("MIC_ITERS", "6"))):
    if cold:
        plan.invalidate()
    plan.run(out)
torch.cuda.synchronize()
print(plan.stats())
