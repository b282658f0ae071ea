"""Where the lane kernel overtakes the tile kernel: 4 layers of side s resized x1.25, event-bracketed resample time behind
mic_plan_invalidate, by route (MIC_RS_LANE_MIN_SLOTS=0: lane for everything; MIC_RS_LANE=0 MIC_RS_MARCH_MIN_UNITS=1e8: tile)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements


def lane_cost(sh, dw, dh):  # lane_partition.h: lane_layer_cost
    tx, ty, bands = (dw + 15) // 16, (dh + 15) // 16, (sh + 15) // 16
    return tx * (1200 * bands + 400 * ty) + tx / 2 * (200 * bands + 3500 * ty)


tag = "lane" if os.environ.get("MIC_RS_LANE_MIN_SLOTS") == "0" else "tile" if os.environ.get("MIC_RS_LANE") == "0" else "default"
for n in (2, 4, 8):
    for s in (100, 150, 200, 250, 300, 400, 500):
        objs = synthetic.make_cutouts(n, (s, s), (s, s), seed=7, alpha_mode="binary")
        a = Atlas(objs)
        d = int(s * 1.25)
        pl = [{"object_id": k + 1, "box": [20 * k, 10 * k, 20 * k + d, 10 * k + d]} for k in range(n)]
        plan = CompositeBatch(a, [SolidCanvas((1920, 1080), synthetic.SOLID_BG)], [coerce_placements(a, pl)])
        outs = plan.alloc_outputs()
        ctx = a.ctx
        for k in range(3):
            plan.invalidate(); plan.run(outs)
        ctx.profile_begin(40)
        for k in range(40):
            plan.invalidate(); plan.run(outs, check=False)
        torch.cuda.synchronize()
        calls, c_ms, r_ms = ctx.profile_end()
        slots = n * lane_cost(s, d, d) / 15000
        print(f"{tag:7s} {n} layers {s}x{s} -> {d}x{d}: resample {r_ms / calls * 1e3:6.2f} us  (~{slots:.0f} slots of lane work, lane layers {plan.stats()['marched_layers']})")
        del plan
