"""A few launches of the C3 batch composite (for rocprofv3 --kernel-trace / --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements

B = int(os.environ.get("MIC_B", "16"))
alpha = os.environ.get("MIC_ALPHA", "binary")
iters = int(os.environ.get("MIC_ITERS", "10"))
size, objs, layouts = synthetic.c3_workload(alpha, seed=3, n_layouts=B)
atlas = Atlas(objs)
rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, size)) for l in layouts]
plan = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)] * B, rows)
outs = [plan.alloc_outputs() for _ in range(2)]
for k in range(iters):
    plan.run(outs[k % 2])
torch.cuda.synchronize()
print("done", atlas.ctx.stats())
