"""The bench's cold-inputs leg alone, for rocprofv3: 16 canvases per launch, an atlas of its own per canvas, two
such sets and three output sets rotating (> 512 MB of inputs, > 1 GB of outputs: nothing is re-used from a cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
B = 16
size, objs, layouts = synthetic.c3_workload("binary", seed=3, n_layouts=B)
probe = Atlas(objs)
rows = [coerce_placements(probe, flex.layout_to_placements(l, probe, size)) for l in layouts]
solid = SolidCanvas(size, synthetic.SOLID_BG)
sets = [CompositeBatch([Atlas(objs) for _ in range(B)], [solid] * B, rows, atlas_of=list(range(B))) for _ in range(2)]
outs = [sets[0].alloc_outputs() for _ in range(3)]
for k in range(int(os.environ.get("MIC_ITERS", "24"))):
    sets[k % 2].run(outs[k % 3], check=False)
torch.cuda.synchronize()
print(sets[0].stats())
