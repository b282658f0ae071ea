"""Timeline of the marching resample kernel's workgroups (tuning build: git apply profiles/r04_tuning_scaffolding.patch, then scripts/build_variant.sh probe -DMIC_RS_PROBE,
run with MIC_LIB=scripts/var_probe.bin): per workgroup the start / end (100 MHz clock) of its first wave -> resident
workgroups over time, duration spread, tail, and what list scheduling of those durations could reach."""
import ctypes, heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import _native, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, os.environ.get("MIC_ALPHA", "soft"))
atlas = Atlas(objs)
plan = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(atlas, pl)])
out = plan.alloc_outputs()
for _ in range(4):
    plan.run(out)
torch.cuda.synchronize()
n = 16384
buf = np.zeros((n, 12), np.uint64)
assert _native.lib().mic_debug_rs_probe(ctypes.c_void_p(buf.ctypes.data), n) == 0
live = buf[buf[:, 1] > 0]
t0 = live[:, 0].min()
st = (live[:, 0] - t0).astype(np.float64) / 100.0  # us
en = (live[:, 1] - t0).astype(np.float64) / 100.0
dur = en - st
bands = (live[:, 3] & 0xFFFFFFFF).astype(np.int64); tiles = (live[:, 3] >> 32).astype(np.int64)
print(f"{len(live)} workgroups; kernel span {en.max():.1f} us; duration per workgroup: mean {dur.mean():.2f}, p10 {np.percentile(dur, 10):.2f}, "
      f"median {np.median(dur):.2f}, p90 {np.percentile(dur, 90):.2f}, max {dur.max():.2f} us; bands/unit mean {bands.mean():.1f} max {bands.max()}, tiles mean {tiles.mean():.1f}")
A = np.stack([bands, tiles, np.ones_like(bands)], 1).astype(np.float64)
coef = np.linalg.lstsq(A, dur, rcond=None)[0]
print(f"duration ~ {coef[0]:.2f} us x bands + {coef[1]:.2f} us x tiles + {coef[2]:.2f} us (residual std {np.std(dur - A @ coef):.2f})")
ph = live[:, 4:10].astype(np.float64)
tot = ph[:, 5]
for nm, i in (("setup (entry -> loop)", 0), ("barrier 1 (others still reading)", 1), ("band -> LDS + barrier 2", 2), ("horizontal section", 3), ("vertical tiles", 4), ("total", 5)):
    print(f"   {nm:34s} mean {ph[:, i].mean():8.0f} cycles  {100 * ph[:, i].sum() / tot.sum():5.1f} %   per band {ph[:, i].sum() / bands.sum():7.0f}")
print(f"   clock: {tot.mean() / dur.mean():.0f} s_memtime ticks per us")
def sim(durs, slots=1280):
    h = [0.0] * slots; heapq.heapify(h); end = 0.0
    for x in durs:
        t = heapq.heappop(h); end = max(end, t + x); heapq.heappush(h, t + x)
    return end
print(f"sum of durations / 1280 slots = {dur.sum() / 1280:.1f} us; list scheduling in start order {sim(dur[np.argsort(st)]):.1f} us, longest first {sim(np.sort(dur)[::-1]):.1f} us")
for t in np.arange(0, en.max() + 2, 4.0):
    resident = int(((st <= t) & (en > t)).sum())
    print(f"  t={t:5.1f} us: {resident:5d} workgroups resident, {int((st > t).sum()):5d} not started")
