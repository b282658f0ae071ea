"""Event-bracketed timing of the C3 placements-mode (LANCZOS) composite: resample and composite kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
for alpha in os.environ.get("MIC_ALPHAS", "soft,binary").split(","):
    size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, alpha)
    atlas = Atlas(objs)
    ctx = atlas.ctx
    plan = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(atlas, pl)])
    out = plan.alloc_outputs()
    for _ in range(5):
        plan.run(out)
    torch.cuda.synchronize()
    n = int(os.environ.get("MIC_N", "30"))
    ctx.profile_begin(n)
    for _ in range(n):
        plan.run(out)
    k, c, r = ctx.profile_end()
    st = plan.stats()
    rs_bytes = 4 * (st["source_pixels"] + sum(max(1, p["box"][2] - p["box"][0]) * max(1, p["box"][3] - p["box"][1]) for p in pl))
    print(f"{alpha}: resample {r / k * 1e3:.1f} us ({rs_bytes / (r / k * 1e-3) / 1e9:.0f} GB/s, {rs_bytes / (r / k * 1e-3) / 8e12:.3f} of HBM peak), "
          f"composite {c / k * 1e3:.1f} us, unit_px env {os.environ.get('MIC_RS_UNIT_PX')}")
