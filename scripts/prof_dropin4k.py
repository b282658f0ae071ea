"""cProfile of the PIL drop-in at 4K / 32 objects: composite(PIL solid bg), composite(PIL image bg), render -> PIL."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from image_transformation_amd import synthetic, flex
from image_transformation_amd.compositor import ObjectImages, composite
size, objs, layouts = synthetic.c3_workload("binary", seed=3, n_layouts=1)
W, H = size
imgs = ObjectImages({k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in objs.items()})
pl = flex.layout_to_placements(layouts[0], imgs, size)
bg_solid = Image.new("RGBA", size, tuple(synthetic.SOLID_BG))
noise = np.random.default_rng(1).integers(0, 256, (H, W, 4), dtype=np.uint8); noise[:, :, 3] = 255
bg_image = Image.fromarray(noise, "RGBA")
for name, bg in (("solid", bg_solid), ("image", bg_image)):
    for _ in range(5): composite(bg, imgs, pl)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): composite(bg, imgs, pl)
    pr.disable()
    print("==", name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(10)
