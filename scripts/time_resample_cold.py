"""COLD resample timing (mic_plan_invalidate before every run: the resampled layers are never resident): event-bracketed
resample / composite kernel times of the C3 placements canvas (soft / binary alpha), a 16-canvas placements batch, the four
C5 composites (audio_book at 7680x4320, x8 LANCZOS upscales) and the reference-sized C1 call with scales != 1.
A/B two builds inside one gpurun call with MIC_LIB=<variant>."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.background_resizing import solid_canvas
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements, load_object_images


def bracket(ctx, plan, outs, n=30):
    for k in range(3):
        plan.invalidate()
        plan.run(outs[k % len(outs)])
    ctx.profile_begin(n)
    for k in range(n):
        plan.invalidate()
        plan.run(outs[k % len(outs)], check=False)
    torch.cuda.synchronize()
    calls, c_ms, r_ms = ctx.profile_end()
    return r_ms / calls * 1e3, c_ms / calls * 1e3


for amode in ("soft", "binary"):
    size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, amode)
    a = Atlas(objs)
    plan = CompositeBatch(a, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(a, pl)])
    r, c = bracket(a.ctx, plan, [plan.alloc_outputs()])
    print(f"C3 placements ({amode}): resample {r:.2f} us, composite {c:.2f} us")
    if amode == "soft":
        rng = np.random.default_rng(11)
        lists = []
        for _ in range(16):
            q = []
            for p in pl:
                x1, y1, x2, y2 = p["box"]
                w = max(8, int((x2 - x1) * rng.uniform(0.8, 1.25))); h = max(8, int((y2 - y1) * rng.uniform(0.8, 1.25)))
                dx, dy = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
                q.append({"object_id": p["object_id"], "box": [x1 + dx, y1 + dy, x1 + dx + w, y1 + dy + h]})
            lists.append(coerce_placements(a, q))
        plan = CompositeBatch(a, [SolidCanvas(size, synthetic.SOLID_BG)] * 16, lists)
        r, c = bracket(a.ctx, plan, [plan.alloc_outputs()], n=10)
        print(f"16-canvas placements batch (soft): resample {r / 16:.2f} us per canvas, composite {c / 16:.2f} us per canvas")
    del plan

gold = os.path.join(ROOT, "tests", "golden")
base = os.path.join(gold, "bundles", "audio_book")
with open(os.path.join(gold, "big_hashes.json")) as f:
    big = {r["name"]: r for r in json.load(f)["cases"]}
canvas = solid_canvas(os.path.join(base, "background.png"), (7680, 4320))
objects = load_object_images(os.path.join(base, "results.json"))
atlas = objects.atlas()
for i in range(4):
    plan = CompositeBatch(atlas, [canvas], [coerce_placements(atlas, big[f"c5_audio_book_iter{i}"]["placements"])])
    outs = [plan.alloc_outputs() for _ in range(2)]
    r, c = bracket(atlas.ctx, plan, outs, n=20)
    print(f"C5 iter {i}: resample {r:.2f} us, composite {c:.2f} us")
    del plan, outs

for n in (2, 4, 8, 12):
    objs = synthetic.make_cutouts(n, (700, 700), (500, 500), seed=5, alpha_mode="soft")
    a = Atlas(objs)
    pl = [{"object_id": k + 1, "box": [10 * k, 5 * k, 10 * k + 900, 5 * k + 640]} for k in range(n)]
    plan = CompositeBatch(a, [SolidCanvas((3840, 2160), synthetic.SOLID_BG)], [coerce_placements(a, pl)])
    r, c = bracket(a.ctx, plan, [plan.alloc_outputs()])
    print(f"{n} layers 700x500 -> 900x640 ({n * 0.576:.1f} Mpx out): resample {r:.2f} us")
