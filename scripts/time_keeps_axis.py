"""Layers that keep one axis (the pass Pillow skips) through the lane kernel: event-bracketed resample time, every run behind
mic_plan_invalidate.  A/B by environment: MIC_RS_LANE_KEEPS=0 (general three-digit form for every layer) / 1 (shipped)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements


def bracket(ctx, plan, outs, n=30):
    for k in range(3):
        plan.invalidate(); plan.run(outs)
    ctx.profile_begin(n)
    for k in range(n):
        plan.invalidate(); plan.run(outs, check=False)
    torch.cuda.synchronize()
    calls, c_ms, r_ms = ctx.profile_end()
    return r_ms / calls * 1e3


tag = f"[MIC_RS_LANE_KEEPS={os.environ.get('MIC_RS_LANE_KEEPS', '1')}]"
for n in (12, 32):
    objs = synthetic.make_cutouts(n, (700, 700), (500, 500), seed=5, alpha_mode="soft")
    a = Atlas(objs)
    for name, (w, h) in (("both axes 700x500 -> 900x640", (900, 640)), ("keeps height 700x500 -> 900x500", (900, 500)),
                         ("keeps width 700x500 -> 700x640", (700, 640))):
        pl = [{"object_id": k + 1, "box": [10 * k, 5 * k, 10 * k + w, 5 * k + h]} for k in range(n)]
        plan = CompositeBatch(a, [SolidCanvas((3840, 2160), synthetic.SOLID_BG)], [coerce_placements(a, pl)])
        r = bracket(a.ctx, plan, plan.alloc_outputs())
        print(f"{n} layers, {name} ({n * w * h / 1e6:.1f} Mpx out): resample {r:.2f} us, stats {plan.stats()['marched_layers']} lane layers {tag}")
        del plan
# a mixed call: a third of each class
n = 30
objs = synthetic.make_cutouts(n, (700, 700), (500, 500), seed=6, alpha_mode="soft")
a = Atlas(objs)
pl = [{"object_id": k + 1, "box": [10 * k, 5 * k, 10 * k + (700 if k % 3 == 1 else 900), 5 * k + (500 if k % 3 == 2 else 640)]} for k in range(n)]
plan = CompositeBatch(a, [SolidCanvas((3840, 2160), synthetic.SOLID_BG)], [coerce_placements(a, pl)])
print(f"30 layers, a third of each class: resample {bracket(a.ctx, plan, plan.alloc_outputs()):.2f} us {tag}")
