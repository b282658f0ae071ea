"""Host cost of ONE-SHOT batch calls (no persistent plan): render_batch(16 Flex JSON texts) and composite_device(16
placement lists) at 4K / 32 objects, against the 112 us the launch itself takes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements, composite_device, render_batch
size, objs, layouts = synthetic.c3_workload("binary", 3, 16)
atlas = Atlas(objs)
cv = [SolidCanvas(size, synthetic.SOLID_BG)] * 16
texts = [json.dumps(l) for l in layouts]
rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, size)) for l in layouts]
outs = [torch.empty((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(16)]
def t(fn, n=30):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    ts.sort(); return ts[len(ts) // 2]
h, c = t(lambda: render_batch(texts, atlas, cv, outs=outs)); print(f"render_batch(16 JSON texts): host {h*1e3:.3f} ms, to completion {c*1e3:.3f} ms")
h, c = t(lambda: render_batch(layouts, atlas, cv, outs=outs)); print(f"render_batch(16 dict layouts): host {h*1e3:.3f} ms, to completion {c*1e3:.3f} ms")
h, c = t(lambda: composite_device(atlas, cv, rows, outs=outs)); print(f"composite_device(16 row lists): host {h*1e3:.3f} ms, to completion {c*1e3:.3f} ms")
h, c = t(lambda: CompositeBatch(atlas, cv, rows)); print(f"CompositeBatch(...) creation: {h*1e3:.3f} ms")
plan = CompositeBatch(atlas, cv, rows)
h, c = t(lambda: plan.run(outs, check=False)); print(f"plan.run: host {h*1e3:.3f} ms, to completion {c*1e3:.3f} ms")
