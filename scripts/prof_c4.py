"""A few launches of one C4 aspect-ratio batch (16 variants of the 32-object bundle) for rocprofv3.
MIC_RATIO = index into synthetic.RATIOS_C4; prints the algorithmic bytes per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements

ratio_idx = int(os.environ.get("MIC_RATIO", "3"))
alpha = os.environ.get("MIC_ALPHA", "binary")
objs, variants = synthetic.c4_workload(alpha, n_variants=64)
atlas = Atlas(objs)
vs = [v for i, v in enumerate(variants) if i % 4 == ratio_idx]
rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, sz)) for (sz, l) in vs]
plan = CompositeBatch(atlas, [SolidCanvas(sz, synthetic.SOLID_BG) for (sz, _) in vs], rows)
outs = [plan.alloc_outputs() for _ in range(2)]
for k in range(int(os.environ.get("MIC_ITERS", "10"))):
    plan.run(outs[k % 2])
torch.cuda.synchronize()
st = plan.stats()
print("ratio", synthetic.RATIOS_C4[ratio_idx], vs[0][0], "alg_bytes", 4 * st["canvas_pixels"] + 4 * st["layer_pixels"])
