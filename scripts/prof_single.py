"""Single-canvas composites (the reference's call shape) for rocprofv3: 4K / 32 objects and 1080p / 8 objects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
for size, objs, layouts in (synthetic.c3_workload("binary", 3, 1), synthetic.c2_workload("binary", 2)):
    layout = layouts[0] if isinstance(layouts, list) else layouts
    atlas = Atlas(objs)
    one = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(atlas, flex.layout_to_placements(layout, atlas, size))])
    outs = [one.alloc_outputs() for _ in range(12)]
    for k in range(40):
        one.run(outs[k % 12], check=False)
    torch.cuda.synchronize()
    print(size, one.stats())
