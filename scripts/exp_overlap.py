"""Experiment: do the (instruction-bound) resample kernel and the (memory-bound) composite kernel overlap when a
placements-mode batch is cut into chunks that run on two streams?  16 canvases x 32 LANCZOS layers, nothing shared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import synthetic, _native
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements, pack_blob, _upload
W, H, N = 3840, 2160, 16
size, objs, pl0 = synthetic.placements_workload(W, H, 32, 3, os.environ.get("MIC_ALPHA", "soft"))
sets = [pl0] + synthetic.placement_sets(objs, W, H, 3, N - 1)
ctxs = [_native.context(), _native.Context(torch.cuda.current_device())]
atlases = []
for c in ctxs:
    a = Atlas.__new__(Atlas)
    a.ctx = c
    host = pack_blob(objs)
    a._init_from_blob(_upload(host.numpy(), c), header=host.numpy())
    atlases.append(a)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def build(chunks):
    plans = []
    per = N // chunks
    for i in range(chunks):
        a = atlases[i % 2]
        ss = sets[i * per:(i + 1) * per]
        p = CompositeBatch(a, [SolidCanvas(size, synthetic.SOLID_BG)] * per, [coerce_placements(a, q) for q in ss])
        plans.append((p, p.alloc_outputs(), streams[i % 2]))
    return plans

def run(plans):
    for p, o, s in plans:
        with torch.cuda.stream(s):
            p.run(o, check=False)

for chunks in (1, 2, 4, 8):
    plans = build(chunks)
    for _ in range(3):
        run(plans)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        run(plans)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{chunks} chunk(s) on {min(chunks, 2)} stream(s): {dt * 1e6 / N:.1f} us per canvas ({dt * 1e3:.3f} ms per 16)")
    del plans
