"""Stage stamps of the lane resample kernel (probe build: scripts/build_variant.sh probe -p profiles/r05_lane_stage_probe.patch,
run with MIC_LIB=build/var_probe.bin): per piece the 100 MHz clock at kernel entry, piece start, record in registers, first
band arrived, first tile of output rows, end; plus cycles spent waiting for bands / taps and inside the horizontal passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from image_transformation_amd import synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements

OUT = "/tmp/lane_probe.bin"
os.environ["MIC_LANE_PROBE_OUT"] = OUT  # (read when the plan is built: the records then carry a stamp buffer; every run dumps it)


def report(name, plan):
    for _ in range(3):
        plan.invalidate(); plan.run()
    torch.cuda.synchronize()
    plan.invalidate(); plan.run()
    torch.cuda.synchronize()
    p = np.fromfile(OUT, dtype=np.uint64).reshape(-1, 8)
    if os.environ.get("MIC_LANE_PROBE_SAVE"):
        np.save(os.path.join(os.environ["MIC_LANE_PROBE_SAVE"], name.split(":")[0].replace(" ", "_").replace("(", "").replace(")", "") + ".npy"), p)
    p = p[p[:, 5] > 0]
    t0 = p[:, 0].min()
    us = lambda a: a.astype(np.float64) / 100.0
    entry, start, rec, band0, emit0, end = (us(p[:, i] - t0) for i in range(6))
    bandwait = (p[:, 6] & 0xFFFFFFFF).astype(np.float64); tapwait = (p[:, 6] >> 32).astype(np.float64)
    hcyc = (p[:, 7] & 0xFFFFFFFF).astype(np.float64); bands = ((p[:, 7] >> 32) & 0xFFFF).astype(np.float64); tiles = (p[:, 7] >> 48).astype(np.float64)
    q = lambda a: f"mean {a.mean():6.2f} p10 {np.percentile(a, 10):6.2f} p50 {np.percentile(a, 50):6.2f} p90 {np.percentile(a, 90):6.2f} max {a.max():6.2f}"
    print(f"== {name}: {len(p)} pieces, span first entry -> last end {end.max():.2f} us; bands/piece {bands.mean():.1f}, tile rows/piece {tiles.mean():.1f}")
    print(f"   kernel entry (dispatch ramp)        {q(entry)}")
    print(f"   entry -> piece start (barrier, earlier piece) {q(start - entry)}")
    print(f"   start -> record in registers        {q(rec - start)}")
    print(f"   record -> first band arrived        {q(band0 - rec)}")
    print(f"   first band -> first tile of rows    {q(emit0 - band0)}")
    print(f"   first tile -> end (stores drained)  {q(end - emit0)}")
    print(f"   piece life start -> end             {q(end - start)}")
    print(f"   piece end (from the first entry)    {q(end)}")
    life_cyc = (end - start) * 1e-6
    print(f"   cycles: waiting for bands {bandwait.mean():.0f}, waiting for taps {tapwait.mean():.0f}, inside horizontal passes {hcyc.mean():.0f} per piece")


for n in (2, 12):
    objs = synthetic.make_cutouts(n, (700, 700), (500, 500), seed=5, alpha_mode="soft")
    a = Atlas(objs)
    pl = [{"object_id": k + 1, "box": [10 * k, 5 * k, 10 * k + 900, 5 * k + 640]} for k in range(n)]
    plan = CompositeBatch(a, [SolidCanvas((3840, 2160), synthetic.SOLID_BG)], [coerce_placements(a, pl)])
    report(f"{n} layers 700x500 -> 900x640", plan)
    del plan
for amode in ("soft", "binary"):
    size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, amode)
    a = Atlas(objs)
    plan = CompositeBatch(a, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(a, pl)])
    report(f"C3 placements ({amode})", plan)
    del plan

# C5: audio_book's cutouts upscaled x8 onto the 8K canvas (BASELINE configs[4]); a call below one round
import json
from image_transformation_amd.background_resizing import solid_canvas
from image_transformation_amd.compositor import load_object_images
gold = os.path.join(ROOT, "tests", "golden")
base = os.path.join(gold, "bundles", "audio_book")
with open(os.path.join(gold, "big_hashes.json")) as f:
    big = {r["name"]: r for r in json.load(f)["cases"]}
canvas = solid_canvas(os.path.join(base, "background.png"), (7680, 4320))
objects = load_object_images(os.path.join(base, "results.json"))
atlas = objects.atlas()
plan = CompositeBatch(atlas, [canvas], [coerce_placements(atlas, big["c5_audio_book_iter0"]["placements"])])
report("C5 iter 0 (x8 upscales)", plan)
