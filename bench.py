#!/usr/bin/env python3
"""bench.py -- composited Mpixels/s at 4K canvas, 32 objects (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload = "C3"): synthetic 3840x2160 canvas, 32 RGBA cutouts (binary alpha, as
the reference's bundles), depth-2 row/column Flex-DSL layouts (seed 3).
One step = one pass of the hot path over one batch: B distinct Flex layouts of the bundle are
composited onto B canvases by ONE mic_composite_batch call per GPU (table upload + one kernel
launch).  Timed region: placements + atlas resident on the device -> canvases complete in HBM
(SURVEY.md section 8d); Flex box maths, atlas upload/broadcast and D2H are outside and are reported
separately.  Output canvases rotate over > 256 MiB so the Infinity Cache cannot hold them.

N > 1: one process per GPU, variants sharded v -> GPU v mod N, the atlas is broadcast once over
RCCL before the timed region, no collective on the data path ("scaling": "weak": B per GPU fixed).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(objs, placements, size, budget_s=12.0, max_reps=5000):
    """The CPU oracle ("port" of the reference's Pillow path, 1 thread) on the same workload:
    bounded sample of whole-canvas composites of layout 0."""
    import numpy as np
    import oracle
    from image_transformation_amd.synthetic import SOLID_BG

    W, H = size
    bg = np.empty((H, W, 4), np.uint8)
    bg[:] = np.asarray(SOLID_BG, np.uint8)
    oracle.composite(bg, objs, placements)  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < max_reps and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        oracle.composite(bg, objs, placements)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(W * H / med / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} reps of one 3840x2160/32-object Flex composite (layout 0), median "
                      f"{med * 1e3:.1f} ms, oracle/mic_oracle.c single thread, nproc={os.cpu_count()}"}


def cpu_pillow(objs, placements, size, budget_s=5.0):
    """when Pillow is installed on the box, the same composite through Pillow itself
    (this file's own restatement of the reference's loop, compositor.py:6-22 -- the reference's files
    are not here): the speed a user of the reference sees, and a cross-check of the port's number."""
    try:
        from PIL import Image
    except ImportError:
        return None
    import numpy as np
    from image_transformation_amd.synthetic import SOLID_BG

    W, H = size
    bg = Image.new("RGBA", (W, H), tuple(SOLID_BG))
    imgs = {k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in objs.items()}

    def run():
        canvas = bg.copy()
        for p in placements:
            obj = imgs.get(int(p["object_id"]))
            if obj is None:
                continue
            x1, y1, x2, y2 = [int(v) for v in p["box"]]
            o = obj.resize((max(1, x2 - x1), max(1, y2 - y1)), Image.LANCZOS)
            canvas.alpha_composite(o, dest=(x1, y1))
        return canvas

    run()
    times = []
    t_all = time.perf_counter()
    while (time.perf_counter() - t_all) < budget_s and len(times) < 200:
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    import PIL
    return {"value": round(W * H / med / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "pillow",
            "sample": f"{len(times)} reps of the same composite through Pillow {PIL.__version__} (resize + alpha_composite "
                      f"loop), median {med * 1e3:.1f} ms"}


def cpu_baseline_threads(objs, placements, size, budget_s=8.0):
    """the same port, one image per core (the C call releases the GIL), different
    layouts of the batch round-robin.  Reported beside cpu_baseline, not instead of it."""
    import threading

    import numpy as np
    import oracle
    from image_transformation_amd.synthetic import SOLID_BG

    W, H = size
    try:
        n_thr = len(os.sched_getaffinity(0))
    except AttributeError:
        n_thr = os.cpu_count() or 1
    n_thr = max(1, min(n_thr, 32))
    done = [0] * n_thr
    t_end = time.perf_counter() + budget_s

    def work(i):
        bg = np.empty((H, W, 4), np.uint8)
        bg[:] = np.asarray(SOLID_BG, np.uint8)
        k = i
        while time.perf_counter() < t_end:
            oracle.composite(bg, objs, placements[k % len(placements)])
            done[i] += 1
            k += n_thr

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    el = time.perf_counter() - t0
    return {"value": round(sum(done) * W * H / el / 1e6, 1), "unit": "Mpixels/s", "cores": n_thr, "kind": "port",
            "sample": f"{sum(done)} whole-canvas composites over {n_thr} threads in {el:.1f} s, one image per thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="canvases per step per GPU")
    ap.add_argument("--alpha", default="binary", choices=["binary", "soft", "opaque"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # MIC_BENCH_REHEARSAL=1: all ranks share cuda:0 over gloo (to exercise the N>1 code path on a
    # one-GPU box); never set by the driver, whose ranks get one GPU each over RCCL.
    rehearsal = os.environ.get("MIC_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from image_transformation_amd import _native, flex, synthetic
    from image_transformation_amd.batch import broadcast_atlas, shard_indices
    from image_transformation_amd.compositor import CompositeBatch, SolidCanvas, coerce_placements

    B = args.batch
    size, objs, layouts = synthetic.c3_workload(args.alpha, seed=3, n_layouts=B * world)
    W, H = size

    # ---- atlas: packed on rank 0, broadcast once over RCCL, resident afterwards ----
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    atlas = broadcast_atlas(objs if rank == 0 or world == 1 else None, src=0)
    torch.cuda.synchronize()
    atlas_ms = (time.perf_counter() - t0) * 1e3
    ctx = atlas.ctx

    # ---- host Flex box maths (layout_json -> boxes), outside the timed region ----
    mine = shard_indices(len(layouts), rank, world)
    t0 = time.perf_counter()
    placements = [flex.layout_to_placements(layouts[v], atlas, size) for v in mine]
    layout_ms = (time.perf_counter() - t0) * 1e3 / max(len(mine), 1)
    rows = [coerce_placements(atlas, pl) for pl in placements]
    # the same maths through the C++ placer (mic_flex_place) on the JSON text, as render() does it
    texts = [json.dumps(layouts[v]) for v in mine]
    t0 = time.perf_counter()
    native_rows = [flex.native_boxes(t, atlas, size) for t in texts]
    native_ms = (time.perf_counter() - t0) * 1e3 / max(len(mine), 1)
    assert all(nr is None or [tuple(r) for r in nr] == [tuple(r) for r in row] for nr, row in zip(native_rows, rows))
    box_px = sum(max(1, r[3] - r[1]) * max(1, r[4] - r[2]) for row in rows for r in row)
    plan = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)] * len(mine), rows)

    # rotating output sets, > 256 MiB in total
    set_bytes = B * W * H * 4
    n_sets = max(2, -(-(320 << 20) // set_bytes))
    out_sets = [plan.alloc_outputs() for _ in range(n_sets)]

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(args.warmup):
        plan.run(out_sets[k % n_sets])
    torch.cuda.synchronize()
    stats = plan.stats()

    # ---- timed region: exactly K steps ----
    # HIP events around the composite kernel of every 8th step: an event pair between two back-to-back
    # kernels leaves the GPU idle for a few microseconds, so bracketing every step would slow the very
    # loop being timed; the sampled brackets still average >= 25 launches of the default run.
    prof_every = 8 if args.steps >= 64 else 1
    ctx.profile_begin(args.steps, every=prof_every)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        plan.run(out_sets[k % n_sets], check=False)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    n_prof, comp_ms, _ = ctx.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HBM bytes per launch from the PMC counters: measured separately with rocprofv3 (bench.py cannot
    # run under --pmc and time itself) and committed under profiles/; only quoted for the workload
    # it was measured on.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        if tj["workload"] == {"batch": B, "alpha": args.alpha, "canvas": [W, H], "objects": 32}:
            traffic = tj["per_launch"]["hbm_bytes"]

    px_per_step = B * W * H * world
    value = px_per_step * args.steps / elapsed / 1e6
    kernel_ms = comp_ms / max(n_prof, 1)
    # algorithmic bytes of ONE launch: every canvas written once + every visible cutout pixel read once
    b_alg = 4 * stats["canvas_pixels"] + 4 * stats["layer_pixels"]
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    result = {
        "metric": "composited Mpixels/s at 4K canvas, 32 objects",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "C3: 3840x2160 canvas, 32 RGBA cutouts (binary alpha), depth-2 row/column Flex-DSL",
                   "canvases_per_step_per_gpu": B, "alpha": args.alpha, "parallelism": f"variants sharded v mod {world}",
                   "background": "solid, synthesised in-kernel", "filter": "identity scale (Flex pipeline)"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)"
                     if traffic else None,
                     "kernel": "composite_kernel", "kernel_ms": round(kernel_ms, 4), "kernel_launches_timed": n_prof,
                     "algorithmic_bytes_per_launch": b_alg,
                     "read_frac_of_peak": round(4 * stats["layer_pixels"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                     if kernel_ms > 0 else None},
        "atlas": {"bytes": atlas.nbytes, "upload_or_broadcast_ms": round(atlas_ms, 3)},
        "host_layout_ms_per_image": {"python_mirror": round(layout_ms, 3),
                                     "native_mic_flex_place": round(native_ms, 4) if all(r is not None for r in native_rows) else None},
        "box_area_Mpixels_per_s": round(box_px * world * args.steps / elapsed / 1e6, 1),
    }

    if rank == 0 and world == 1 and not args.no_extras:
        # single-canvas launches (latency view of the same workload)
        one = CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)], rows[:1])
        outs1 = [one.alloc_outputs() for _ in range(12)]
        for k in range(10):
            one.run(outs1[k % 12])
        ctx.profile_begin(100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(100):
            one.run(outs1[k % 12])
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t0
        n1, c1, _ = ctx.profile_end()
        s1 = one.stats()
        result["single_canvas"] = {"ms_per_canvas_wall": round(e1 / 100 * 1e3, 4), "kernel_ms": round(c1 / n1, 4),
                                   "Mpixels_per_s": round(W * H * 100 / e1 / 1e6, 1),
                                   "roofline_frac": round((4 * s1["canvas_pixels"] + 4 * s1["layer_pixels"]) /
                                                          (c1 / n1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # placements mode (direct composite() callers): Pillow-exact LANCZOS resample + overlaps
        psize, pobjs, ppl = synthetic.placements_workload(W, H, 32, 3, "soft")
        from image_transformation_amd.compositor import Atlas
        patlas = Atlas(pobjs)
        pplan = CompositeBatch(patlas, [SolidCanvas(psize, synthetic.SOLID_BG)], [coerce_placements(patlas, ppl)])
        pout = pplan.alloc_outputs()
        pplan.run(pout)
        torch.cuda.synchronize()
        ctx.profile_begin(10)
        t0 = time.perf_counter()
        for _ in range(10):
            pplan.run(pout)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        n2, c2, r2 = ctx.profile_end()
        ps = pplan.stats()
        rs_bytes = 4 * (ps["source_pixels"] + sum(max(1, p["box"][2] - p["box"][0]) * max(1, p["box"][3] - p["box"][1])
                                                  for p in ppl))  # every cutout read once + every resampled pixel written once
        result["placements_mode_lanczos"] = {"ms_per_canvas_wall": round(e2 / 10 * 1e3, 3),
                                             "resample_ms": round(r2 / n2, 3), "composite_ms": round(c2 / n2, 4),
                                             "Mpixels_per_s": round(W * H * 10 / e2 / 1e6, 1),
                                             "resample_roofline": {"bound": "valu (instruction issue), not hbm",
                                                                   "algorithmic_bytes": rs_bytes,
                                                                   "achieved_GBps": round(rs_bytes / (r2 / n2 * 1e-3) / 1e9, 1),
                                                                   "frac_of_hbm_peak": round(rs_bytes / (r2 / n2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}

        # PCIe-inclusive step: job-table upload + composite + D2H of every canvas into pinned memory
        # (SURVEY 8d's end-to-end figure; never `value`)
        host = [torch.empty((H, W, 4), dtype=torch.uint8, pin_memory=True) for _ in range(len(rows))]
        for _ in range(2):
            for h, o in zip(host, plan.run(out_sets[0])):
                h.copy_(o, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(5):
            for h, o in zip(host, plan.run(out_sets[k % n_sets])):
                h.copy_(o, non_blocking=True)
        torch.cuda.synchronize()
        e3 = (time.perf_counter() - t0) / 5
        result["pcie_inclusive"] = {"ms_per_step": round(e3 * 1e3, 3), "Mpixels_per_s": round(B * W * H / e3 / 1e6, 1),
                                    "d2h_GBps": round(B * W * H * 4 / e3 / 1e9, 1)}
        del host
        # background synthesis on the same canvas size: median colour of a 4K RGBA image + solid fill
        import ctypes
        P = ctypes.c_void_p
        lib = _native.lib()
        img = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device=dev)
        res = torch.empty(4, dtype=torch.uint8, device=dev)
        sp = P(ctx.stream_ptr())
        for name, fn in (("median_4k_noise_us", lambda: lib.mic_median_rgb_dev(ctx.handle, P(img.data_ptr()), W, H, P(res.data_ptr()), sp)),):
            for _ in range(3):
                _native.check(fn())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record()
            torch.cuda.synchronize()
            result[name] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(objs, placements[0], size)
        result["cpu_baseline_all_cores"] = cpu_baseline_threads(objs, placements, size)
        result["cpu_pillow"] = cpu_pillow(objs, placements[0], size)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
