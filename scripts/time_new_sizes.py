"""Host + GPU cost of a composite() call whose LANCZOS box sizes have never been seen (every axis table is built on the
host in double precision, Resample.c's precompute_coeffs, and uploaded) against the same call repeated (tables cached,
layers resident) -- the C3 placements workload, transient entry point (mic_composite_batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, SolidCanvas, composite_device, coerce_placements
W, H = 3840, 2160
psize, pobjs, ppl = synthetic.placements_workload(W, H, 32, 3, "soft")
atlas = Atlas(pobjs)
cv = SolidCanvas(psize, synthetic.SOLID_BG)
sets = [ppl] + synthetic.placement_sets(pobjs, W, H, 3, 12)
out = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")]
def call(pl):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    composite_device(atlas, [cv], [coerce_placements(atlas, pl)], outs=out)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3
for i, pl in enumerate(sets):
    a = call(pl); b = call(pl); c = call(pl)
    print(f"set {i}: new sizes: host {a[0]:.3f} ms, to completion {a[1]:.3f} ms | again: host {b[0]:.3f} / {b[1]:.3f} | third: {c[0]:.3f} / {c[1]:.3f}", flush=True)
