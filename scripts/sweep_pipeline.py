"""Round 4: the LANCZOS path as one path -- sweep of the pipelining knobs (MIC_PIPE_BANDS / _CHUNK / _STREAMS / _PRIO, read
by mic_create, so every setting gets a context of its own) on the C3 placements canvas (32 LANCZOS layers, soft alpha),
one canvas per call and 16 canvases per call.  GPU time = HIP events on the caller's stream around the whole call
(the stream's last composite waits for the last resample group, so the bracket covers the side streams' work).
Every run is COLD: the plan is invalidated first (mic_plan_invalidate), i.e. all layers are resampled again.
    python scripts/sweep_pipeline.py [single|batch|all]"""
import itertools, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import _native, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements

what = sys.argv[1] if len(sys.argv) > 1 else "all"
W, H = 3840, 2160
psize, pobjs, ppl = synthetic.placements_workload(W, H, 32, 3, "soft")
sets = [ppl] + synthetic.placement_sets(pobjs, W, H, 3, 15)
KNOBS = ("MIC_FUSE_CHUNK", "MIC_PIPE_BANDS", "MIC_PIPE_CHUNK", "MIC_PIPE_STREAMS", "MIC_PIPE_PRIO", "MIC_PIPE_EVFLAGS")


def make_ctx(**env):
    for k in KNOBS:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    return _native.Context(0)


def span_us(plan, outs, reps, cold=True):
    for _ in range(3):
        if cold: plan.invalidate()
        plan.run(outs, check=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if cold: plan.invalidate()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(outs, check=False); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 1), round(ts[0], 1)


ref_single = ref_batch = None
EVF = ("2", "40000002", "20000002", "60000002")  # DisableTiming, + ReleaseToDevice, + DisableSystemFence, + both
rows = []
if what in ("single", "all"):
    for bands, streams, prio, evf in [(1, 1, 0, "2")] + [(b, s, 0, f) for f in EVF for b in (2, 4) for s in (1, 2)]:
        ctx = make_ctx(MIC_PIPE_BANDS=bands, MIC_PIPE_STREAMS=streams, MIC_PIPE_PRIO=prio, MIC_PIPE_EVFLAGS=evf)
        atlas = Atlas(pobjs, ctx=ctx)
        plan = CompositeBatch(atlas, [SolidCanvas(psize, synthetic.SOLID_BG)], [coerce_placements(atlas, ppl)])
        outs = plan.alloc_outputs()
        med, best = span_us(plan, outs, 40)
        got = plan.run(outs)[0].cpu().numpy()
        if ref_single is None: ref_single = got
        same = bool(np.array_equal(got, ref_single))
        warm = span_us(plan, outs, 20, cold=False)
        rows.append(dict(mode="single", evflags=evf, bands=bands, streams=streams, prio=prio, cold_us=med, cold_best_us=best, warm_us=warm[0],
                         groups=plan.stats()["pipeline_groups"], same_pixels=same))
        print(json.dumps(rows[-1]), flush=True)
        del plan, atlas, outs
        _native.lib().mic_destroy(ctx.handle)
if what in ("batch", "all"):
    for nb in (16, 4):
        for fuse in (0, 1, 2, 3, 4, 8):
            if fuse >= nb: continue
            ctx = make_ctx(MIC_FUSE_CHUNK=fuse, MIC_PIPE_CHUNK=0)
            atlas = Atlas(pobjs, ctx=ctx)
            plan = CompositeBatch(atlas, [SolidCanvas(psize, synthetic.SOLID_BG)] * nb, [coerce_placements(atlas, q) for q in sets[:nb]])
            outs = plan.alloc_outputs()
            med, best = span_us(plan, outs, 12)
            plan.invalidate()
            got = [o.cpu().numpy() for o in plan.run(outs)]
            groups = plan.stats()["pipeline_groups"]
            h = [hash(g.tobytes()) for g in got]
            if fuse == 0: ref_batch = h
            warm = span_us(plan, outs, 10, cold=False)
            rows.append(dict(mode=f"batch{nb}", fuse_chunk=fuse, cold_us_per_canvas=round(med / nb, 2),
                             cold_best_us_per_canvas=round(best / nb, 2), warm_us_per_canvas=round(warm[0] / nb, 2),
                             groups=groups, same_pixels=h == ref_batch))
            print(json.dumps(rows[-1]), flush=True)
            del plan, atlas, outs, got
            _native.lib().mic_destroy(ctx.handle)
print("SWEEP_DONE", len(rows))
