#!/usr/bin/env python3
"""bench.py -- composited Mpixels/s at 4K canvas, 32 objects (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload c3|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` from a plain shell starts the N ranks itself (child processes through
torch.distributed.run, before this process touches the GPU) and relays rank 0's JSON line.

Workload (config.workload = "C3"): synthetic 3840x2160 canvas, 32 RGBA cutouts (binary alpha, as
the reference's bundles), depth-2 row/column Flex-DSL layouts (seed 3).
One step = one pass of the hot path over one batch: B distinct Flex layouts of the bundle are
composited onto B canvases by ONE mic_plan_run call per GPU (one kernel launch).  Timed region:
placements + atlas resident on the device -> canvases complete in HBM; Flex
box maths, atlas upload/broadcast and D2H are outside and are reported separately.  Output canvases
rotate over > 256 MiB so the Infinity Cache cannot hold them.

The timed loop carries no instrumentation.  The kernel duration behind `roofline` comes from a
SEPARATE pass of the same launches bracketed by HIP events on the launch stream (an event pair
between two back-to-back launches leaves the GPU idle for a few microseconds, which would slow the
very loop being timed).

`roofline` reports the shared-atlas batch the metric is quoted on, split honestly: its cutout reads
are re-reads of one 16 MB atlas that lives in L2 / the Infinity Cache, so `fabric` (algorithmic
bytes over kernel time) is not a DRAM number; `dram` counts only what must cross the HBM pins
(canvases written once + the atlas once).  `cold_inputs` is the leg in which the reads cannot be
cached either: every canvas of the batch has an atlas of its own and the sets rotate over > 512 MB.

N > 1: one process per GPU, variants sharded v -> GPU v mod N, the atlas is broadcast once over
RCCL before the timed region, no collective on the data path ("scaling": "weak": B per GPU fixed).
Rank 0 prints ONE JSON line; it names the backend, dist.get_world_size() and every rank's device.

--workload c4 (and the `c4_strong` object inside every default line): BASELINE.json configs[3] exactly,
a FIXED batch of 64 aspect-ratio variants of one 32-object bundle split v mod N, one launch of 64/N
canvases per rank per step ("scaling": "strong"), atlas broadcast time reported separately.
"""
from __future__ import annotations

import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_quota_cores():
    """CPU time this process's cgroup may use, in cores (cpu.max / cfs quota), or None when unlimited.  A GPU box hands a
    one-GPU job a share of the host (its 256 logical CPUs are visible, a fraction of them is usable)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, p = f.read().split()[:2]
            return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, p = int(f.read()), int(g.read())
            return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


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
            "cpu": cpu_model(), "nproc": os.cpu_count(),
            "sample": f"{len(times)} reps of one 3840x2160/32-object Flex composite (layout 0), median "
                      f"{med * 1e3:.1f} ms, oracle/mic_oracle.c single thread"}


class _Sized:
    """What the Flex placer needs of a cutout: .size = (w, h)."""

    def __init__(self, a):
        self.size = (a.shape[1], a.shape[0])


def _pillow_composite(bg, imgs, placements):
    """This file's own restatement of the reference's loop (compositor.py:6-22); the reference's files
    are not on the GPU box."""
    from PIL import Image

    canvas = bg.copy()
    for p in placements:
        obj = imgs.get(int(p["object_id"]))
        if obj is None:
            continue
        x1, y1, x2, y2 = [int(v) for v in p["box"]]
        o = obj.resize((max(1, x2 - x1), max(1, y2 - y1)), Image.LANCZOS)
        canvas.alpha_composite(o, dest=(x1, y1))
    return canvas


def _median_time(fn, budget_s, max_reps):
    fn()
    times = []
    t_all = time.perf_counter()
    while (time.perf_counter() - t_all) < budget_s and len(times) < max_reps:
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_pillow(objs, placements, size, budget_s=5.0):
    """when Pillow is installed on the box, the same composite through Pillow itself:
    the speed a user of the reference sees, and a cross-check of the port's number."""
    try:
        import PIL
        from PIL import Image
    except ImportError:
        return None
    import numpy as np
    from image_transformation_amd.synthetic import SOLID_BG

    W, H = size
    bg = Image.new("RGBA", (W, H), tuple(SOLID_BG))
    imgs = {k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in objs.items()}
    med, n = _median_time(lambda: _pillow_composite(bg, imgs, placements), budget_s, 200)
    return {"value": round(W * H / med / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "pillow",
            "ms_per_canvas": round(med * 1e3, 2),
            "sample": f"{n} reps of the same composite through Pillow {PIL.__version__} (resize + alpha_composite "
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
    # every CPU this process may use: its affinity mask, capped by the cgroup's CPU quota when there is one (threads
    # beyond the quota are only throttled); each thread holds one 33 MB background and one 33 MB result at a time
    affinity, quota = n_thr, cpu_quota_cores()
    if quota is not None:
        n_thr = min(n_thr, max(1, int(quota + 0.999)))
    n_thr = max(1, min(n_thr, 512))
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
            "nproc": os.cpu_count(), "affinity_cpus": affinity, "cgroup_cpu_quota_cores": quota,
            "sample": f"{sum(done)} whole-canvas composites over {n_thr} threads (= every CPU this process may use: "
                      f"affinity {affinity}, cgroup quota {quota}) in {el:.1f} s, one image per thread"}


def bracketed(ctx, run, n):
    """n launches bracketed by HIP events on the launch stream -> (composite ms, resample ms) per launch."""
    import torch

    ctx.profile_begin(n)
    for k in range(n):
        run(k)
    torch.cuda.synchronize()
    calls, c_ms, r_ms = ctx.profile_end()
    return c_ms / max(calls, 1), r_ms / max(calls, 1)


def plan_bytes(stats):
    """Algorithmic bytes of one launch: every canvas written once + every visible cutout pixel read once."""
    return 4 * stats["canvas_pixels"] + 4 * stats["layer_pixels"]


def frac(nbytes, ms):
    return round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None


def batch_leg(ctx, atlas, canvases, rows, n_sets_bytes=320 << 20, reps=30):
    """Kernel-only measurement of one composite batch: event-bracketed launches over rotating outputs."""
    from image_transformation_amd.compositor import CompositeBatch

    plan = CompositeBatch(atlas, canvases, rows)
    st = plan.stats()
    set_bytes = 4 * st["canvas_pixels"]
    n_sets = max(2, -(-n_sets_bytes // max(set_bytes, 1)))
    outs = [plan.alloc_outputs() for _ in range(n_sets)]
    for k in range(3):
        plan.run(outs[k % n_sets])
    c_ms, _ = bracketed(ctx, lambda k: plan.run(outs[k % n_sets], check=False), reps)
    b = plan_bytes(st)
    return {"kernel_ms": round(c_ms, 4), "algorithmic_bytes": b, "frac_of_hbm_peak": frac(b, c_ms),
            "Mpixels_per_s": round(st["canvas_pixels"] / (c_ms * 1e-3) / 1e6, 1)}


def _free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` from a plain shell: this process has not touched the GPU (no HIP call, no
    torch.cuda.is_available()); it starts the N ranks as CHILD processes through torch.distributed.run (one rank
    per GPU, rendezvous on 127.0.0.1), relays rank 0's single JSON line and returns the children's status.
    Nothing is exec'ed over a process that has initialised the GPU."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, cwd=ROOT)
    for line in proc.stdout:
        # the one JSON line goes to stdout; the launcher's own chatter to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


class Dist:
    """The N-rank protocol of the bench contract: barrier + synchronize on both sides of a timed region, MAX
    (and MIN) over ranks.  world == 1: everything is local."""

    def __init__(self, world, rank, dev, rehearsal, active=None):
        self.world, self.rank, self.dev, self.rehearsal = world, rank, dev, rehearsal
        # active: a process group exists and every collective below goes through it -- always at world > 1, and at
        # world == 1 under MIC_BENCH_FORCE_DIST=1 (a one-rank RCCL communicator: the N > 1 code path on a one-GPU box)
        self.active = world > 1 if active is None else active
        self.backend = None
        if self.active:
            import torch.distributed as dist

            self.backend = dist.get_backend()

    def barrier(self):
        if self.active:
            import torch.distributed as dist

            dist.barrier()

    def max_min(self, x: float):
        if not self.active:
            return x, x
        import torch
        import torch.distributed as dist

        t = torch.tensor([x, -x], dtype=torch.float64, device="cpu" if self.rehearsal else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0].item()), -float(t[1].item())

    def sum(self, x: float) -> float:
        if not self.active:
            return x
        import torch
        import torch.distributed as dist

        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.rehearsal else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0].item())

    def gather(self, obj):
        if not self.active:
            return [obj]
        import torch.distributed as dist

        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out


def timed_steps(D: Dist, ctx, plan, out_sets, steps, warmup):
    """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize; the clock stops after this rank's
    own synchronize (the closing barrier is not this rank's work).  -> (max over ranks s, min over ranks s,
    event-bracketed kernel ms of this rank from a SEPARATE pass, n bracketed)"""
    import torch

    n_sets = len(out_sets)
    for k in range(warmup):
        plan.run(out_sets[k % n_sets])
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    # ONE pair of HIP events on the launch stream around the K launches of the timed region (torch's current stream is
    # the stream libmic launches on): the kernel's average launch duration over exactly the launches that are timed,
    # without an event pair between back-to-back launches (each such pair leaves the GPU idle for ~2 us)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(steps):
        plan.run(out_sets[k % n_sets], check=False)
    ev1.record()
    torch.cuda.synchronize()
    elapsed_rank = time.perf_counter() - t0
    D.barrier()
    elapsed, elapsed_min = D.max_min(elapsed_rank)
    kernel_ms = ev0.elapsed_time(ev1) / steps
    n_br = max(10, min(steps, 50))
    bracket_ms, _ = bracketed(ctx, lambda k: plan.run(out_sets[k % n_sets], check=False), n_br)
    timed_steps.last_bracket_ms = bracket_ms  # (per-launch brackets of a separate pass: reported beside, never the roofline's basis)
    return elapsed, elapsed_min, kernel_ms, n_br


def _golden_hashes(name: str):
    """sha16 (first 16 hex digits of SHA-256 over the RGBA bytes) of canvases the REFERENCE rendered, captured by
    tests/golden/make_golden.py in the build container: {fixture name: sha16}."""
    with open(os.path.join(ROOT, "tests", "golden", name), encoding="utf-8") as f:
        return {r["name"]: r["sha16"] for r in json.load(f)["cases"]}


def verify_sets(D: Dist, out_sets, names, want):
    """After a timed region, outside it: hash every canvas of every rotating output set the timed launches wrote whose
    fixture name (names[i] for canvas i of a set; None = no fixture) is known, against the reference's own hash.
    Every rank checks its own canvases -- on a multi-GPU run these are pixels of an atlas that crossed xGMI -- and
    the verdicts are gathered.  -> {"verified": bool over all ranks, "canvases": distinct canvases checked over all
    ranks, "checks": hashes compared (canvases x sets), "mismatches": [...]}"""
    import hashlib

    import torch

    torch.cuda.synchronize()
    bad, checks, seen = [], 0, set()
    for s, outs in enumerate(out_sets):
        for i, out in enumerate(outs):
            if names[i] is None:
                continue
            got = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
            checks += 1
            seen.add(names[i])
            if got != want[names[i]]:
                bad.append({"rank": D.rank, "set": s, "canvas": names[i], "got": got, "want": want[names[i]]})
    rows = D.gather({"bad": bad, "checks": checks, "seen": sorted(seen)})
    mism = [b for r in rows for b in r["bad"]]
    return {"verified": not mism and sum(r["checks"] for r in rows) > 0,
            "canvases": len({n for r in rows for n in r["seen"]}), "checks": sum(r["checks"] for r in rows),
            "per_rank_checks": [r["checks"] for r in rows], "mismatches": mism[:8]}


def c4_partition(n_var: int, world: int, sizes):
    """Who renders what in the strong-scaling leg: variant v -> rank v mod G (batch.shard_indices).  Pure: the CPU
    tests check the G = 8 split the driver's node will run (64 canvases, 8 per rank, ONE canvas class per rank: the
    ratios cycle with period 4, so ranks 3 and 7 hold every 4399-wide canvas and decide the step time)."""
    from image_transformation_amd.batch import shard_indices

    out = []
    for r in range(world):
        mine = shard_indices(n_var, r, world)
        out.append({"rank": r, "variants": mine, "canvas_sizes": sorted({tuple(sizes[v]) for v in mine})})
    return out


def c4_strong_leg(D: Dist, args, steps, warmup):
    """BASELINE.json configs[3] exactly: ONE 32-object bundle, a FIXED batch of 64 aspect-ratio variants (4 ratios x
    16 Flex JSONs: canvases 2160x3840 / 2880x2880 / 3840x2160 / 4399x1885), variant v -> rank v mod G, one launch of
    64/G canvases per rank per step.  Total work does not grow with G: "scaling": "strong".  The atlas is broadcast
    once from rank 0 (RCCL) before the timed region and reported separately."""
    import torch
    from image_transformation_amd import flex, synthetic
    from image_transformation_amd.batch import broadcast_atlas, shard_indices
    from image_transformation_amd.compositor import CompositeBatch, SolidCanvas, coerce_placements

    n_var = 64
    objs, variants = synthetic.c4_workload(args.alpha, seed=4, n_variants=n_var)
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    atlas = broadcast_atlas(objs if D.rank == 0 or D.world == 1 else None, src=0, force=D.active)
    torch.cuda.synchronize()
    bcast_first_ms = (time.perf_counter() - t0) * 1e3
    D.barrier()
    t0 = time.perf_counter()
    atlas = broadcast_atlas(objs if D.rank == 0 or D.world == 1 else None, src=0, force=D.active)
    torch.cuda.synchronize()
    bcast_warm_ms = (time.perf_counter() - t0) * 1e3
    ctx = atlas.ctx
    part = c4_partition(n_var, D.world, [v[0] for v in variants])
    mine = part[D.rank]["variants"]
    assert mine == shard_indices(n_var, D.rank, D.world)
    rows = [coerce_placements(atlas, flex.layout_to_placements(variants[v][1], atlas, variants[v][0])) for v in mine]
    plan = CompositeBatch(atlas, [SolidCanvas(variants[v][0], synthetic.SOLID_BG) for v in mine], rows)
    st = plan.stats()
    set_bytes = 4 * st["canvas_pixels"]
    n_sets = max(2, -(-(320 << 20) // max(set_bytes, 1)))  # outputs rotate over > 320 MB (here: 2 x 64/G canvases)
    out_sets = [plan.alloc_outputs() for _ in range(n_sets)]
    elapsed, elapsed_min, kernel_ms, n_br = timed_steps(D, ctx, plan, out_sets, steps, warmup)
    # the pixels of the timed launches, after the clock has stopped: every canvas of every output set on every rank
    # against the hash of the REFERENCE's render of that variant (tests/golden/c4_hashes.json; binary alpha, seed 4)
    ver = None
    if args.alpha == "binary":
        ver = verify_sets(D, out_sets, [f"c4_variant_{v}" for v in mine], _golden_hashes("c4_hashes.json"))
    px_total = D.sum(float(st["canvas_pixels"]))
    canv_total = int(round(D.sum(float(len(mine)))))
    k_max, k_min = D.max_min(kernel_ms)
    b_alg = plan_bytes(st)
    f_max, f_min = D.max_min(frac(b_alg, kernel_ms) or 0.0)
    b_max, _ = D.max_min(bcast_warm_ms)
    bf_max, _ = D.max_min(bcast_first_ms)
    sizes = D.gather(sorted({tuple(variants[v][0]) for v in mine}))
    del out_sets, plan
    return {
        "workload": "C4: one 32-object bundle, 64 aspect-ratio variants (9:16 / 1:1 / 16:9 / 21:9 x 16 Flex JSONs), "
                    "variant v -> rank v mod G, one launch of 64/G canvases per rank per step",
        "scaling": "strong", "value": round(px_total * steps / elapsed / 1e6, 1), "unit": "Mpixels/s",
        "canvases_total": canv_total, "canvases_per_rank": len(mine), "steps": steps, "warmup": warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "verified": ver["verified"] if ver else None, "verified_canvases": ver["canvases"] if ver else 0,
        "verification": dict(ver, against="tests/golden/c4_hashes.json: the reference's own render of each variant "
                                          "(compositor.py:6-22), every output set of every rank, after the timed region")
        if ver else "no reference hashes for this alpha mode",
        "per_rank": {"timed_region_s_max": round(elapsed, 6), "timed_region_s_min": round(elapsed_min, 6),
                     "kernel_ms_max": round(k_max, 4), "kernel_ms_min": round(k_min, 4),
                     "roofline_frac_max": round(f_max, 4), "roofline_frac_min": round(f_min, 4),
                     "canvas_sizes": [[list(s) for s in ss] for ss in sizes]},
        "roofline_rank0": {"kernel": "composite_kernel", "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": b_alg,
                           "frac": frac(b_alg, kernel_ms), "kernel_ms_source": "HIP events on the launch stream around the timed launches"},
        "atlas_broadcast": {"bytes": atlas.nbytes, "first_ms_max": round(bf_max, 3), "warm_ms_max": round(b_max, 3),
                            "note": "one broadcast per BUNDLE from rank 0 (RCCL over xGMI when ranks > 1; a plain upload "
                                    "at 1 rank), outside the timed region; first_ms includes context / communicator set-up"},
    }, atlas


def cold_start_child():
    """Runs in a FRESH child process (bench.py --cold-start-child): what the reference's caller sees the first time --
    import, context + code object, the first load_object_images / contact sheet / fill_solid / composite on the
    squarespace bundle (macro_placement_test.py:1397-1511) -- stage by stage, each stage's SECOND call beside it, then the
    same sequence through Pillow / NumPy in the same process.  Prints one JSON object."""
    t_start = time.perf_counter()
    out = {}

    def stage(name, fn):
        t0 = time.perf_counter()
        r = fn()
        out[name] = round((time.perf_counter() - t0) * 1e3, 3)
        return r
    import numpy as np  # noqa: F401
    t0 = time.perf_counter()
    import torch
    out["import_torch_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    t0 = time.perf_counter()
    from image_transformation_amd import _native, flex
    from image_transformation_amd.background_resizing import fill_solid
    from image_transformation_amd.compositor import composite, load_object_images
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
    out["import_package_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    bdir = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace")
    rj, bgp = os.path.join(bdir, "results.json"), os.path.join(bdir, "background.png")
    with open(os.path.join(ROOT, "tests", "golden", "bundles.json"), encoding="utf-8") as f:
        layout = next(r for r in json.load(f)["cases"] if r["name"] == "squarespace_1x1")["layout"]
    size = (492, 492)
    # what used to be ONE stage ("context_and_code_object_ms", ~158 ms), taken apart: the HIP runtime's own start
    # (hipInit + the primary context), torch's lazy CUDA init on top of it (the package takes its buffers and the
    # current stream from torch), loading libmic.so (the code object is only REGISTERED by the loader), mic_create
    # (pinned staging ring, scratch, tables), the first launch of any libmic kernel (the runtime loads the code object
    # -- all ~35 kernel instantiations of the library are one module), and a second launch of the same kernel
    import ctypes

    def _hip_init():
        hip = ctypes.CDLL("libamdhip64.so")
        assert hip.hipInit(0) == 0 and hip.hipSetDevice(0) == 0 and hip.hipFree(None) == 0
    stage("hip_init_ms", _hip_init)
    stage("torch_cuda_init_ms", lambda: (torch.cuda.is_available(), torch.cuda.current_device(), torch.cuda.synchronize()))
    # the process' FIRST kernel launch, whoever makes it, pays for the hardware queue and the launching module: taken here with
    # one of torch's own kernels, so that what mic_create costs afterwards is libmic's share alone (MIC_BENCH_COLD_ORDER=mic:
    # the other way round -- mic_create launches first, as in a caller that never touches a torch kernel)
    torch_first = os.environ.get("MIC_BENCH_COLD_ORDER", "torch") == "torch"
    if torch_first:
        stage("first_torch_kernel_ms", lambda: (torch.zeros(16, device="cuda"), torch.cuda.synchronize()))
    stage("dlopen_libmic_ms", _native.lib)
    ctx0 = stage("mic_create_ms", _native.context)   # its one launch (zeroing the median scratch) loads libmic's code object
    stage("mic_create_drained_ms", torch.cuda.synchronize)
    stage("selftest_first_ms", ctx0.selftest)        # another libmic kernel + a read-back, code object resident
    stage("selftest_second_ms", ctx0.selftest)
    if not torch_first:
        stage("first_torch_kernel_ms", lambda: (torch.zeros(16, device="cuda"), torch.cuda.synchronize()))
    out["cold_order"] = "torch kernel first" if torch_first else "mic_create first"
    out["cold_note"] = ("what a fresh process waits for is the HIP runtime: hipInit + primary context (~60 ms) and the process' FIRST kernel launch, "
                        "whoever makes it (~85-100 ms: hardware queue, first module); libmic's own code object (one module, ~35 kernel "
                        "instantiations) loads in < 1 ms -- mic_create after a torch kernel: 0.7 ms -- so there is nothing to split into a "
                        "second library; of the first contact sheet's ~12 ms, ~7.5 are the process' first device allocation + host-to-device "
                        "copy (pinned or pageable alike), ~2 FreeType label masks + font (profiles/r05_cold_start.txt)")
    out["context_and_code_object_ms"] = round(sum(out[k] for k in ("hip_init_ms", "torch_cuda_init_ms", "first_torch_kernel_ms", "dlopen_libmic_ms",
                                                                   "mic_create_ms", "mic_create_drained_ms")), 3)
    for tag in ("first", "second"):
        objs = stage(f"load_object_images_{tag}_ms", lambda: load_object_images(rj))
        stage(f"contact_sheet_{tag}_ms", lambda: build_labeled_contact_sheet(os.path.join(bdir, "objects"), rj))
        bg = stage(f"fill_solid_{tag}_ms", lambda: fill_solid(bgp, size))
        pl = flex.layout_to_placements(layout, objs, size)
        stage(f"composite_{tag}_ms", lambda: composite(bg, objs, pl))
    out["total_first_pass_ms"] = round(sum(v for k, v in out.items() if k.endswith("_first_ms")) + out["context_and_code_object_ms"], 3)
    # the same sequence through Pillow / NumPy (the files are in the page cache by now, Pillow is imported)
    from PIL import Image
    pil = {}

    def pstage(name, fn):
        t0 = time.perf_counter()
        r = fn()
        pil[name] = round((time.perf_counter() - t0) * 1e3, 3)
        return r
    for tag in ("first", "second"):
        def load():
            with open(rj, encoding="utf-8") as f:
                return {int(it["object_id"]): Image.open(os.path.join(bdir, it["filename"])).convert("RGBA") for it in json.load(f)}
        pobjs = pstage(f"load_object_images_{tag}_ms", load)
        pstage(f"contact_sheet_{tag}_ms", lambda: _pillow_contact_sheet(rj))
        pbg = pstage(f"fill_solid_{tag}_ms", lambda: Image.new("RGBA", size, _numpy_median_colour(Image.open(bgp).convert("RGBA")) + (255,)))
        ppl = flex.layout_to_placements(layout, pobjs, size)
        pstage(f"composite_{tag}_ms", lambda: _pillow_composite(pbg, pobjs, ppl))
    pil["total_first_pass_ms"] = round(sum(v for k, v in pil.items() if k.endswith("_first_ms")), 3)
    out["pillow_numpy"] = pil
    out["process_wall_ms"] = round((time.perf_counter() - t_start) * 1e3, 1)
    print(json.dumps(out), flush=True)


def cold_start():
    """Spawn the child (a subprocess, not an exec of this process) and return its record."""
    import subprocess

    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-start-child"], capture_output=True, text=True,
                             timeout=300, cwd=ROOT)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")]
        if res.returncode != 0 or not line:
            return {"error": (res.stderr or res.stdout)[-400:]}
        rec = json.loads(line[-1])
        rec["what"] = ("a fresh process on the squarespace bundle (492x492, 4 cutouts): first call of every stage of the "
                       "reference's sequence, the second call beside it, Pillow / NumPy's same sequence in the same process")
        return rec
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc)}


def main():
    if "--cold-start-child" in sys.argv:
        return cold_start_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="canvases per step per GPU (workload c3)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4"],
                    help="c3: weak scaling, --batch distinct 4K layouts per GPU (the headline); "
                         "c4: strong scaling, BASELINE configs[3]'s fixed 64 variants split v mod G")
    ap.add_argument("--alpha", default="binary", choices=["binary", "soft", "opaque"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work the cpu_baseline sample may take")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started from a plain shell: become the launcher (before anything touches the GPU)
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # MIC_BENCH_REHEARSAL=1: all ranks share cuda:0 over gloo (to exercise the N>1 code path on a
    # one-GPU box); never set by the driver, whose ranks get one GPU each over RCCL.
    rehearsal = os.environ.get("MIC_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {dev_index} on this node ({torch.cuda.device_count()} visible); "
                         "MIC_BENCH_REHEARSAL=1 shares cuda:0 between the ranks")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # MIC_BENCH_FORCE_DIST=1 (under torch.distributed.run with ONE rank): initialise the process group anyway, so that
    # init_process_group("nccl", device_id=...), the NCCL barrier, GPU-tensor all_reduce / all_gather_object and the atlas
    # broadcast run through RCCL on a one-GPU box -- the exact calls of an N > 1 run, with a one-rank communicator
    dist_on = world > 1 or (os.environ.get("MIC_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if dist_on:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    D = Dist(world, rank, dev, rehearsal, active=dist_on)
    devices = D.gather({"rank": rank, "device_index": dev_index, "name": torch.cuda.get_device_name(dev_index),
                        "pid": os.getpid()})
    ranks_info = {"ranks": dist.get_world_size() if dist_on else 1,
                  "backend": (D.backend + (" (rehearsal: ranks share cuda:0)" if rehearsal else " (RCCL)" if D.backend == "nccl" else ""))
                  if dist_on else "none (single process)",
                  "devices": devices}

    from image_transformation_amd import _native, flex, synthetic
    from image_transformation_amd.batch import broadcast_atlas, shard_indices
    from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements

    if args.workload == "c4":
        leg, _ = c4_strong_leg(D, args, args.steps, args.warmup)
        result = {
            "metric": "composited Mpixels/s at 4K canvas, 32 objects",
            "value": leg["value"], "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": leg["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": leg["workload"], "canvases_per_step_total": leg["canvases_total"],
                       "canvases_per_step_per_gpu": leg["canvases_per_rank"], "alpha": args.alpha,
                       "parallelism": f"variants sharded v mod {world}", "background": "solid, synthesised in-kernel",
                       "filter": "identity scale (Flex pipeline)"},
            "roofline": {"bound": "hbm", "achieved": round((leg["roofline_rank0"]["frac"] or 0) * HBM_PEAK_GBS, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": leg["roofline_rank0"]["frac"], "traffic": None,
                         **leg["roofline_rank0"], "verified": leg["verified"], "verification": leg["verification"]},
            "per_rank": leg["per_rank"], "atlas": leg["atlas_broadcast"], "cpu_baseline": None, **ranks_info,
        }
        if dist_on:
            D.barrier()
            dist.destroy_process_group()  # every collective is done: what follows is rank 0's own CPU time
        if rank == 0 and not args.no_cpu_baseline:
            c4objs, c4vars = synthetic.c4_workload(args.alpha, seed=4, n_variants=4)
            sz, lay = c4vars[2]  # the 16:9 class: 3840 x 2160, the canvas the metric names
            result["cpu_baseline"] = cpu_baseline(c4objs, flex.layout_to_placements(lay, {k: _Sized(v) for k, v in c4objs.items()}, sz),
                                                  sz, budget_s=args.cpu_budget)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if leg["verified"] is False:  # (every rank holds the gathered verdict: all of them leave non-zero)
            raise SystemExit("bench.py: canvases of the timed launches differ from the reference's hashes: "
                             + json.dumps(leg["verification"]["mismatches"]))
        return

    B = args.batch
    size, objs, layouts = synthetic.c3_workload(args.alpha, seed=3, n_layouts=B * world)
    W, H = size

    # ---- atlas: packed on rank 0, broadcast once over RCCL, resident afterwards.  The first call also
    # creates the HIP context, loads the code object and the library: reported as `first_ms`; what an upload
    # (or broadcast) costs once the process is warm is measured by doing it again.
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    atlas = broadcast_atlas(objs if rank == 0 or world == 1 else None, src=0, force=dist_on)
    torch.cuda.synchronize()
    atlas_first_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    atlas = broadcast_atlas(objs if rank == 0 or world == 1 else None, src=0, force=dist_on)
    torch.cuda.synchronize()
    atlas_warm_ms = (time.perf_counter() - t0) * 1e3
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
    canvases = [SolidCanvas(size, synthetic.SOLID_BG)] * len(mine)
    plan = CompositeBatch(atlas, canvases, rows)

    # rotating output sets, > 256 MiB in total
    set_bytes = B * W * H * 4
    n_sets = max(2, -(-(320 << 20) // set_bytes))
    out_sets = [plan.alloc_outputs() for _ in range(n_sets)]

    # ---- timed region: exactly K steps, nothing else; then a separate, event-bracketed pass of the same launches ----
    elapsed, elapsed_min, kernel_ms, n_br = timed_steps(D, ctx, plan, out_sets, args.steps, args.warmup)
    stats = plan.stats()
    kernel_ms_max, kernel_ms_min = D.max_min(kernel_ms)
    # the pixels of the timed launches, checked after the clock has stopped (before any output set is released):
    # layouts 0-2 of this workload were rendered by the REFERENCE in the build container (tests/golden/big_hashes.json);
    # layout v lives on rank v mod G, so at G >= 3 three different GPUs each prove one canvas
    head_ver = None
    if args.alpha in ("binary", "soft") and B * world >= 3:
        head_ver = verify_sets(D, out_sets, [f"c3_flex_{args.alpha}_{v}" if v < 3 else None for v in mine],
                               _golden_hashes("big_hashes.json"))

    # HBM bytes per launch from the PMC counters, and the kernel's average duration by rocprofv3's kernel trace:
    # measured separately with rocprofv3 (bench.py cannot run under --pmc and time itself) and committed under
    # profiles/; only quoted for the workload they were measured on.
    traffic, traffic_src, rocprof_kernel_ms = None, None, None
    for name in ("r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("workload") == {"batch": B, "alpha": args.alpha, "canvas": [W, H], "objects": 32}:
                traffic = tj["per_launch"]["hbm_bytes"]
                traffic_src = f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                if tj.get("kernel_trace"):
                    rocprof_kernel_ms = round(tj["kernel_trace"]["average_ns"] * 1e-6, 4)
                break

    px_per_step = B * W * H * world
    value = px_per_step * args.steps / elapsed / 1e6
    # algorithmic bytes of ONE launch: every canvas written once + every visible cutout pixel read once
    b_alg = plan_bytes(stats)
    b_write, b_read = 4 * stats["canvas_pixels"], 4 * stats["layer_pixels"]
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # what has to cross the HBM pins per launch: the canvases (written once, 531 MB per launch of a rotating
    # > 1 GB set) and at most one pass over the shared atlas; the other B - 1 reads of every cutout are served
    # by L2 / the Infinity Cache (FETCH_SIZE counts them: it sits on the L2's fabric side)
    b_dram = b_write + min(b_read, atlas.nbytes)

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
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "composite_kernel", "kernel_ms": round(kernel_ms, 4),
                     "kernel_ms_source": f"one HIP event pair on the launch stream around the {args.steps} launches of the timed region "
                                         "(average launch period: kernel + the ~1 us between back-to-back launches)",
                     "kernel_ms_bracketed_separate_pass": round(getattr(timed_steps, "last_bracket_ms", 0.0), 4),
                     "kernel_ms_rocprof": rocprof_kernel_ms,
                     "frac_rocprof": frac(b_alg, rocprof_kernel_ms) if rocprof_kernel_ms else None,
                     "algorithmic_bytes_per_launch": b_alg,
                     "fabric": {"bytes": b_alg, "GBps": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4),
                                "note": "algorithmic bytes / kernel time; the cutout reads (a third of the bytes) are "
                                        "re-reads of one shared 16 MB atlas served by L2 / Infinity Cache, so this is "
                                        "memory-system (fabric) throughput, not DRAM bandwidth"},
                     "dram": {"bytes": b_dram, "GBps": round(b_dram / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms > 0 else None,
                              "frac": frac(b_dram, kernel_ms),
                              "note": "bytes that must cross the HBM pins: canvases written once + the atlas read once"},
                     "read_frac_of_peak": frac(b_read, kernel_ms),
                     "verified": head_ver["verified"] if head_ver else None,
                     "verification": dict(head_ver, against="tests/golden/big_hashes.json c3_flex_<alpha>_{0,1,2}: the "
                                          "reference's own render of layouts 0-2, in every output set the timed launches wrote")
                     if head_ver else "no reference hashes for this alpha mode / batch"},
        "atlas": {"bytes": atlas.nbytes, "first_ms": round(atlas_first_ms, 3), "warm_upload_or_broadcast_ms": round(atlas_warm_ms, 3),
                  "note": "first_ms includes HIP context creation and code-object load"},
        "host_layout_ms_per_image": {"python_mirror": round(layout_ms, 3),
                                     "native_mic_flex_place": round(native_ms, 4) if all(r is not None for r in native_rows) else None},
        "box_area_Mpixels_per_s": round(box_px * world * args.steps / elapsed / 1e6, 1),
        **ranks_info,
    }
    if dist_on:
        result["per_rank"] = {"timed_region_s_max": round(elapsed, 6), "timed_region_s_min": round(elapsed_min, 6),
                              "kernel_ms_max": round(kernel_ms_max, 4), "kernel_ms_min": round(kernel_ms_min, 4)}

    # ---- the strong-scaling leg (BASELINE configs[3]) rides in the same line at every N, so that the driver's
    # N = 1, 2, 4, 8 runs carry both curves: `value` (weak, B canvases per GPU) and `c4_strong.value` (64 fixed)
    if not args.no_extras or world > 1:
        del out_sets[1:]  # (memory: the C4 leg allocates 2 x 64/G canvases of its own)
        # (an extra, not the contract line: it does not inherit --steps.  At the driver's --steps 20 the G = 8 leg would be
        # 20 x ~60 us = 1.2 ms of timed region, where one 50 us barrier skew is 4 %: >= 200 steps / >= 20 warm-up always)
        c4, c4_atlas = c4_strong_leg(D, args, max(200, args.steps), max(20, args.warmup))
        result["c4_strong"] = c4
        del c4_atlas
        while len(out_sets) < n_sets:
            out_sets.append(plan.alloc_outputs())

    if rank == 0 and world == 1 and not args.no_extras:
        extras(result, args, ctx, atlas, objs, layouts, placements, rows, plan, out_sets, n_sets, size, dev)
        result["cold_start"] = cold_start()

    if dist_on:
        D.barrier()
        dist.destroy_process_group()  # every collective is done: the CPU legs below are rank 0's own time
    if rank == 0 and not args.no_cpu_baseline:
        # at every N, so that a SCALE record is self-contained (the other ranks have left; nothing waits for this)
        result["cpu_baseline"] = cpu_baseline(objs, placements[0], size, budget_s=args.cpu_budget)
        if world == 1:
            result["cpu_baseline_all_cores"] = cpu_baseline_threads(objs, placements, size)
            result["cpu_pillow"] = cpu_pillow(objs, placements[0], size)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    failed = [name for name, v in (("headline", head_ver), ("c4_strong", result.get("c4_strong", {}).get("verification")))
              if isinstance(v, dict) and not v["verified"]]
    if failed:  # (every rank holds the gathered verdicts: all of them leave non-zero)
        raise SystemExit(f"bench.py: canvases of the timed launches differ from the reference's hashes in {failed}")


def _numpy_median_colour(img):
    """This file's restatement of background_resizing.py:11-22 (per-channel np.median over alpha > 0, int())."""
    import numpy as np

    a = np.array(img)
    mask = a[:, :, 3] > 0
    px = a[:, :, :3][mask] if mask.any() else a[:, :, :3].reshape(-1, 3)
    med = np.median(px, axis=0)
    return tuple(int(v) for v in med)


def _pillow_contact_sheet(results_json, thumb=(256, 256), cols=4, label_h=72, font_size=24):
    """This file's restatement of the reference's labelled contact sheet (macro_placement_test.py:162-242): thumbnails by
    Image.thumbnail(LANCZOS), centred on white cells, label text centred below (DejaVuSans when present)."""
    from PIL import Image, ImageDraw, ImageFont

    with open(results_json, encoding="utf-8") as f:
        items = sorted(json.load(f), key=lambda it: int(it["object_id"]))
    try:
        font = ImageFont.truetype("DejaVuSans.ttf", size=font_size)
    except OSError:
        font = ImageFont.load_default()
    base = os.path.dirname(results_json)
    thumbs = []
    for it in items:
        im = Image.open(os.path.join(base, it["filename"])).convert("RGBA")
        im.thumbnail(thumb, Image.LANCZOS)
        thumbs.append((im, str(it.get("label", it["object_id"]))))
    cw, chh = thumb[0], thumb[1] + label_h
    rows = max(1, -(-len(thumbs) // cols))
    sheet = Image.new("RGBA", (cols * cw if thumbs else cw, rows * chh), (255, 255, 255, 255))
    draw = ImageDraw.Draw(sheet)
    for i, (im, label) in enumerate(thumbs):
        r, c = divmod(i, cols)
        sheet.alpha_composite(im, dest=(c * cw + (cw - im.size[0]) // 2, r * chh + (thumb[1] - im.size[1]) // 2))
        bb = draw.textbbox((0, 0), label, font=font)
        tw, th = bb[2] - bb[0], bb[3] - bb[1]
        draw.text((c * cw + (cw - tw) // 2, r * chh + thumb[1] + max(0, (label_h - th) // 2)), label, fill=(0, 0, 0, 255), font=font)
    return sheet


def _pillow_run_layouts(bundle, canvas_size, layouts, out_dir, flex):
    """This file's restatement of the deterministic steps of run_macro_only (macro_placement_test.py:1414-1430,
    1493-1513, 1679-1699) through Pillow / NumPy: contact sheet, fill_solid -> canvas.png, and per iteration: decode the
    cutouts again, place, re-open canvas.png, composite, save the draft (out_dir None: nothing is written or re-opened)."""
    from PIL import Image

    rj = os.path.join(bundle, "results.json")
    sheet = _pillow_contact_sheet(rj)
    bg = Image.open(os.path.join(bundle, "background.png")).convert("RGBA")
    canvas = Image.new("RGBA", canvas_size, _numpy_median_colour(bg) + (255,))
    if out_dir:
        sheet.save(os.path.join(out_dir, "contact_sheet.png"))
        canvas.save(os.path.join(out_dir, "canvas.png"))
    drafts = []
    for i, layout in enumerate(layouts):
        with open(rj, encoding="utf-8") as f:
            objs = {int(it["object_id"]): Image.open(os.path.join(bundle, it["filename"])).convert("RGBA") for it in json.load(f)}
        pl = flex.layout_to_placements(layout, objs, canvas_size)
        cv = Image.open(os.path.join(out_dir, "canvas.png")).convert("RGBA") if out_dir else canvas
        draft = _pillow_composite(cv, objs, pl)
        if out_dir:
            draft.save(os.path.join(out_dir, f"draft_macro_iter_{i:02d}.png"))
        drafts.append(draft)
    return drafts


def pipeline_extras(result, ctx, dev):
    """BASELINE configs[4] (C5: audio_book end to end at 7680x4320) stage by stage, the contact sheet, and the
    deterministic harness (run_layouts) with and without its PNG artifacts -- wall ms of this package, kernel ms and
    roofline fraction where one kernel dominates, and the same stage through Pillow / NumPy on this host."""
    import tempfile

    import numpy as np
    import torch
    from PIL import Image
    from image_transformation_amd import _native, flex
    from image_transformation_amd.background_resizing import solid_canvas
    from image_transformation_amd.compositor import (CompositeBatch, SolidCanvas, coerce_placements, load_object_images, render)
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
    from image_transformation_amd.pipeline import run_layouts

    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "big_hashes.json"), encoding="utf-8") as f:
        big = {r["name"]: r for r in json.load(f)["cases"]}
    with open(os.path.join(gold, "bundles.json"), encoding="utf-8") as f:
        bundles = {r["name"]: r for r in json.load(f)["cases"]}

    # ---------------- C5 ----------------
    base = os.path.join(gold, "bundles", "audio_book")
    rj, bgp = os.path.join(base, "results.json"), os.path.join(base, "background.png")
    size = (7680, 4320)
    W, H = size
    sync = torch.cuda.synchronize
    t_solid, _ = _median_time(lambda: solid_canvas(bgp, size), 1.0, 200)
    canvas = solid_canvas(bgp, size)
    bg_img = Image.open(bgp).convert("RGBA")
    t_solid_cpu, _ = _median_time(lambda: _numpy_median_colour(bg_img), 1.0, 50)
    assert _numpy_median_colour(bg_img) == tuple(canvas.rgba[:3])
    t_sheet, _ = _median_time(lambda: build_labeled_contact_sheet(os.path.join(base, "objects"), rj), 1.0, 200)
    t_sheet_cpu, _ = _median_time(lambda: _pillow_contact_sheet(rj), 1.0, 50)
    objects = load_object_images(rj)
    atlas = objects.atlas()
    iters = []
    pil_objs = {k: v for k, v in objects.items()}
    bg8k = Image.new("RGBA", size, tuple(canvas.rgba))
    for it in range(4):
        pl = big[f"c5_audio_book_iter{it}"]["placements"]

        def one():
            out = render({"placements": pl}, objects, canvas, as_tensor=True)
            sync()
            return out
        # the refine loop as the reference runs it (macro_placement_test.py:1679-1697): iteration 0 meets an empty layer
        # cache (cleared here), iterations 1-3 move the boxes without resizing them and find the x8 upscales resident
        if it == 0:
            _native.check(_native.lib().mic_layer_cache_clear(ctx.handle))
        sync()
        t0 = time.perf_counter()
        one()  # ONE call in loop order: iteration 0 resamples, 1-3 find the layers
        t_seq = time.perf_counter() - t0
        hits = ctx.stats()["cached_layers"]
        t_wall, _ = _median_time(one, 0.5, 50)
        plan = CompositeBatch(atlas, [canvas], [coerce_placements(atlas, pl)])
        outs = [plan.alloc_outputs() for _ in range(3)]  # 3 x 133 MB: beyond the Infinity Cache
        for k in range(3):
            plan.run(outs[k])

        def cold(k):
            plan.invalidate()  # resample the x8 upscales again, then composite
            plan.run(outs[k % 3], check=False)
        c_ms, r_ms = bracketed(ctx, cold, 12)
        c_warm, _ = bracketed(ctx, lambda k: plan.run(outs[k % 3], check=False), 12)
        st = plan.stats()
        out_px = sum(max(1, q["box"][2] - q["box"][0]) * max(1, q["box"][3] - q["box"][1]) for q in pl)
        rs_bytes = 4 * (st["source_pixels"] + out_px)
        t0 = time.perf_counter()
        _pillow_composite(bg8k, pil_objs, pl)
        t_cpu = time.perf_counter() - t0
        iters.append({"iteration": it, "render_to_device_wall_ms": round(t_wall * 1e3, 3),
                      "render_to_device_wall_ms_in_loop_order": round(t_seq * 1e3, 3), "cached_layers_in_loop_order": hits,
                      "composite_kernel_ms_layers_resident": round(c_warm, 4), "resample_kernel_ms": round(r_ms, 4),
                      "composite_kernel_ms": round(c_ms, 4), "composite_algorithmic_bytes": plan_bytes(st),
                      "composite_roofline_frac": frac(plan_bytes(st), c_ms), "resample_algorithmic_bytes": rs_bytes,
                      "resample_frac_of_hbm_peak": frac(rs_bytes, r_ms), "marched_layers": st["marched_layers"],
                      "pillow_ms": round(t_cpu * 1e3, 1)})
        del plan, outs
    # PIL in -> PIL out (what the harness hands to .save()): includes the 133 MB download
    t_pil, _ = _median_time(lambda: render({"placements": big["c5_audio_book_iter0"]["placements"]}, objects, canvas), 2.0, 20)
    gpu_total = t_solid + t_sheet + sum(i["render_to_device_wall_ms"] for i in iters) * 1e-3
    cpu_total = t_solid_cpu + t_sheet_cpu + sum(i["pillow_ms"] for i in iters) * 1e-3
    result["c5_end_to_end"] = {
        "what": "BASELINE configs[4]: audio_book bundle at 7680x4320 -- solid_canvas(background.png) (median colour), "
                "labelled contact sheet, 4 composites (LANCZOS x8 upscales of the 3 cutouts; tests/golden/big_hashes.json)",
        "solid_canvas_ms": round(t_solid * 1e3, 3), "solid_canvas_numpy_ms": round(t_solid_cpu * 1e3, 3),
        "contact_sheet_ms": round(t_sheet * 1e3, 3), "contact_sheet_pillow_ms": round(t_sheet_cpu * 1e3, 3),
        "composites": iters, "render_to_pil_wall_ms_iter0": round(t_pil * 1e3, 2),
        "total_ms_device_resident": round(gpu_total * 1e3, 2), "total_ms_pillow_numpy": round(cpu_total * 1e3, 1),
        "note": "wall ms are medians of warm calls (files in the decode cache, atlas resident); kernel ms from event-bracketed "
                "launches of a persistent plan over 3 rotating 133 MB outputs; pillow_ms is ONE call per iteration"}

    # ---------------- contact sheet at both bundle sizes ----------------
    sq = os.path.join(gold, "bundles", "squarespace")
    t_sq, _ = _median_time(lambda: build_labeled_contact_sheet(os.path.join(sq, "objects"), os.path.join(sq, "results.json")), 1.0, 200)
    t_sq_cpu, _ = _median_time(lambda: _pillow_contact_sheet(os.path.join(sq, "results.json")), 1.0, 50)
    result["contact_sheet"] = {"squarespace_ms": round(t_sq * 1e3, 3), "squarespace_pillow_ms": round(t_sq_cpu * 1e3, 3),
                               "audio_book_ms": round(t_sheet * 1e3, 3), "audio_book_pillow_ms": round(t_sheet_cpu * 1e3, 3),
                               "note": "PIL sheet returned, 1024x328; warm (decode cache, label masks, resident atlas)"}
    result["contact_sheet_ms"] = round(t_sq * 1e3, 3)

    # ---------------- run_layouts: the deterministic half of run_macro_only ----------------
    lay = bundles["squarespace_1x1"]["layout"]
    layouts = [lay] * 3
    with tempfile.TemporaryDirectory() as td:
        t_nosave, _ = _median_time(lambda: run_layouts(sq, "1:1", layouts, save=False), 1.0, 200)
        t_save, _ = _median_time(lambda: run_layouts(sq, "1:1", layouts, output_root=td), 2.0, 100)
        os.makedirs(os.path.join(td, "pil"), exist_ok=True)
        t_cpu_nosave, _ = _median_time(lambda: _pillow_run_layouts(sq, (492, 492), layouts, None, flex), 1.0, 50)
        t_cpu_save, _ = _median_time(lambda: _pillow_run_layouts(sq, (492, 492), layouts, os.path.join(td, "pil"), flex), 2.0, 50)
        # a 4K draft through the package's PNG writer vs PIL's encoder (the artifact the harness saves per iteration)
        from image_transformation_amd import png as mic_png
        _, o4, l4 = __import__("image_transformation_amd.synthetic", fromlist=["c3_workload"]).c3_workload("binary", seed=3, n_layouts=1)
        draft4k = render(l4[0], o4, SolidCanvas((3840, 2160), (38, 73, 115, 255)))
        p_mine, p_pil = os.path.join(td, "d_mine.png"), os.path.join(td, "d_pil.png")
        t_png, _ = _median_time(lambda: mic_png.save(draft4k, p_mine), 2.0, 50)
        t_png_pil, _ = _median_time(lambda: draft4k.save(p_pil), 3.0, 5)
        # what a 4K caller of draft.save waits for in all: render (enqueue + kernel + 33 MB download) + encode + write
        canvas4k = SolidCanvas((3840, 2160), (38, 73, 115, 255))
        t_render_save, _ = _median_time(lambda: mic_png.save(render(l4[0], o4, canvas4k), p_mine), 2.0, 50)
        sizes = (os.path.getsize(p_mine), os.path.getsize(p_pil))
    result["run_layouts"] = {
        "what": "squarespace bundle, ratio 1:1 (492x492), 3 iterations: contact sheet + fill_solid + 3 x (place, clamp, composite)",
        "ms_without_saving": round(t_nosave * 1e3, 3), "ms_with_png_artifacts": round(t_save * 1e3, 3),
        "pillow_ms_without_saving": round(t_cpu_nosave * 1e3, 3), "pillow_ms_with_png_artifacts": round(t_cpu_save * 1e3, 3),
        "png_4k_draft": {"libmic_writer_ms": round(t_png * 1e3, 2), "pil_save_ms": round(t_png_pil * 1e3, 1),
                         "render_to_saved_png_wall_ms": round(t_render_save * 1e3, 2),
                         "bytes_libmic": sizes[0], "bytes_pil": sizes[1]},
        "note": "the Pillow figures restate the reference's sequence (cutouts decoded every iteration, canvas.png written and "
                "re-opened) in this file; this package's artifacts: contact sheet, canvas, 3 drafts, 3 overlays, JSONs"}
    result["run_layouts_ms"] = {"no_save": round(t_nosave * 1e3, 3), "save": round(t_save * 1e3, 3)}


def extras(result, args, ctx, atlas, objs, layouts, placements, rows, plan, out_sets, n_sets, size, dev):
    """Secondary measurements (N = 1 only): everything the headline line does not show."""
    import ctypes

    import numpy as np
    import torch
    from image_transformation_amd import _native, flex, synthetic
    from image_transformation_amd.compositor import (Atlas, CompositeBatch, ObjectImages, SolidCanvas, coerce_placements,
                                                     composite, render)

    W, H = size
    B = args.batch
    solid = SolidCanvas(size, synthetic.SOLID_BG)

    # ---- cold inputs: one atlas per canvas of the batch, two such sets alternating with the output sets, so
    # that neither the reads (2 x 256 MB of atlases) nor the writes (> 1 GB of canvases) can live in a cache:
    # every byte of `roofline` crosses the HBM pins
    t0 = time.perf_counter()
    cold_sets = []

    def variant(k):  # atlas k of a set: the same cutout sizes, other colours (alpha untouched) -- a canvas that read
        # another canvas' atlas cannot pass the check below
        if k == 0:
            return objs
        out = {}
        for oid, a in objs.items():
            b = a.copy()
            b[:, :, :3] ^= np.uint8(k)
            out[oid] = b
        return out
    for s in range(2):
        atl = [Atlas(variant(k)) for k in range(B)]
        cold_sets.append(CompositeBatch(atl, [solid] * B, rows, atlas_of=list(range(B))))
    torch.cuda.synchronize()
    for k in range(4):
        cold_sets[k % 2].run(out_sets[k % n_sets])
    # parity before timing: the first and the last canvas of a multi-atlas launch against the CPU oracle, each with
    # its own atlas' pixels (the checker, never the thing timed)
    import oracle
    verified = True
    for s in range(2):
        got = cold_sets[s].run(out_sets[s % n_sets])
        torch.cuda.synchronize()
        for k in (0, B - 1):
            bg = np.empty((H, W, 4), np.uint8)
            bg[:] = np.asarray(synthetic.SOLID_BG, np.uint8)
            verified = verified and bool(np.array_equal(got[k].cpu().numpy(), oracle.composite(bg, variant(k), placements[k])))
    assert verified, "cold_inputs: a multi-atlas launch differs from the oracle"
    c_ms, _ = bracketed(ctx, lambda k: cold_sets[k % 2].run(out_sets[k % n_sets], check=False), 30)
    st = cold_sets[0].stats()
    b = plan_bytes(st)
    result["cold_inputs"] = {
        "what": f"{B} canvases per launch, each reading an atlas of its own ({B} x {atlas.nbytes >> 20} MB), two such sets "
                "and the output sets alternating: inputs 2 x 256 MB + outputs > 1 GB never re-used within 256 MB of traffic",
        "verified": verified,  # canvases 0 and B-1 of both sets == the oracle on their own atlas' pixels, checked above
        "kernel_ms": round(c_ms, 4), "Mpixels_per_s": round(st["canvas_pixels"] / (c_ms * 1e-3) / 1e6, 1),
        "roofline": {"bound": "hbm", "achieved": round(b / (c_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": frac(b, c_ms), "algorithmic_bytes_per_launch": b,
                     "read_frac_of_peak": frac(4 * st["layer_pixels"], c_ms)}}
    del cold_sets

    # ---- single-canvas launches: the reference's own call shape (one composite() per call)
    one = CompositeBatch(atlas, [solid], rows[:1])
    outs1 = [one.alloc_outputs() for _ in range(12)]
    for k in range(10):
        one.run(outs1[k % 12])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100):
        one.run(outs1[k % 12], check=False)
    torch.cuda.synchronize()
    e1 = time.perf_counter() - t0
    c1, _ = bracketed(ctx, lambda k: one.run(outs1[k % 12], check=False), 50)
    s1 = one.stats()
    result["single_canvas"] = {"ms_per_canvas_wall": round(e1 / 100 * 1e3, 4), "kernel_ms": round(c1, 4),
                               "Mpixels_per_s": round(W * H * 100 / e1 / 1e6, 1),
                               "roofline_frac": frac(plan_bytes(s1), c1)}
    del one, outs1

    # ---- the hard classes, kernel-only, 16-canvas batches: soft alpha, a width that is not a multiple of 4,
    # and all four C4 canvas classes in one launch
    _, sobjs, slayouts = synthetic.c3_workload("soft", seed=3, n_layouts=B)
    satlas = Atlas(sobjs)
    srows = [coerce_placements(satlas, flex.layout_to_placements(l, satlas, size)) for l in slayouts]
    result["soft_alpha_batch"] = batch_leg(ctx, satlas, [solid] * B, srows)
    c4objs, variants = synthetic.c4_workload(args.alpha, seed=4, n_variants=64)
    c4atlas = Atlas(c4objs)
    wide = [v for v in variants if v[0][0] % 4 != 0][:B]
    result["unaligned_width_batch"] = dict(
        canvas=list(wide[0][0]),
        **batch_leg(ctx, c4atlas, [SolidCanvas(s, synthetic.SOLID_BG) for s, _ in wide],
                    [coerce_placements(c4atlas, flex.layout_to_placements(l, c4atlas, s)) for s, l in wide]))
    mixed = variants  # BASELINE configs[3]'s whole batch (what `--workload c4` / `c4_strong` time by the wall clock at
    # N = 1): ratios cycle 9:16, 1:1, 16:9, 21:9, sixteen canvases of each class in one launch
    result["mixed_c4_batch"] = dict(
        canvases=sorted({tuple(s) for s, _ in mixed}),
        **batch_leg(ctx, c4atlas, [SolidCanvas(s, synthetic.SOLID_BG) for s, _ in mixed],
                    [coerce_placements(c4atlas, flex.layout_to_placements(l, c4atlas, s)) for s, l in mixed]))
    del satlas, c4atlas

    # ---- placements mode (direct composite() callers): Pillow-exact LANCZOS resample + overlaps.  "soft" =
    # uniform random alpha in every pixel (the worst case for both kernels); "binary" = cutout-shaped alpha as in
    # the reference's bundles (resampled layers are then soft only along their edges)
    # Resampled layers stay resident since round 4 (a persistent plan keeps them in its scratch): every leg reports COLD
    # (mic_plan_invalidate before each run: all 32 layers resampled again, then composited) and WARM (the layers are
    # there: composite only) -- what a refine loop pays for its first and for its later iterations -- and the same
    # composite through Pillow on this host
    def cold_run(plan, out):
        plan.invalidate()
        plan.run(out, check=False)

    for key, amode in (("placements_mode_lanczos", "soft"), ("placements_mode_lanczos_binary_cutouts", "binary")):
        psize, pobjs, ppl = synthetic.placements_workload(W, H, 32, 3, amode)
        patlas = Atlas(pobjs)
        pplan = CompositeBatch(patlas, [SolidCanvas(psize, synthetic.SOLID_BG)], [coerce_placements(patlas, ppl)])
        pout = pplan.alloc_outputs()
        for _ in range(3):
            cold_run(pplan, pout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            cold_run(pplan, pout)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        c2, r2 = bracketed(ctx, lambda k: cold_run(pplan, pout), 20)
        t0 = time.perf_counter()
        for _ in range(10):
            pplan.run(pout, check=False)
        torch.cuda.synchronize()
        e2w = time.perf_counter() - t0
        c2w, r2w = bracketed(ctx, lambda k: pplan.run(pout, check=False), 20)
        ps = pplan.stats()
        rs_bytes = 4 * (ps["source_pixels"] + sum(max(1, p["box"][2] - p["box"][0]) * max(1, p["box"][3] - p["box"][1])
                                                  for p in ppl))  # every cutout read once + every resampled pixel written once
        path_bytes = 4 * (ps["canvas_pixels"] + ps["source_pixels"])  # canvas written once + every cutout read once
        try:
            from PIL import Image
            pil_objs = {k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in pobjs.items()}
            pil_bg = Image.new("RGBA", psize, tuple(synthetic.SOLID_BG))
            t_pil, _ = _median_time(lambda: _pillow_composite(pil_bg, pil_objs, ppl), 4.0, 5)
            pillow_ms = round(t_pil * 1e3, 1)
        except ImportError:
            pillow_ms = None
        result[key] = {
            "alpha": amode, "pillow_ms": pillow_ms,
            "cold": {"ms_per_canvas_wall": round(e2 / 10 * 1e3, 3), "resample_ms": round(r2, 4), "composite_ms": round(c2, 4),
                     "gpu_ms": round(r2 + c2, 4), "Mpixels_per_s": round(W * H * 10 / e2 / 1e6, 1),
                     "path_bytes_survey_8d": path_bytes, "path_frac_of_hbm_peak": frac(path_bytes, r2 + c2)},
            "warm": {"ms_per_canvas_wall": round(e2w / 10 * 1e3, 3), "resample_ms": round(r2w, 4), "composite_ms": round(c2w, 4),
                     "Mpixels_per_s": round(W * H * 10 / e2w / 1e6, 1), "path_frac_of_hbm_peak": frac(path_bytes, r2w + c2w),
                     "note": "the plan's resampled layers are resident from its first run on: later runs (boxes moved onto other "
                             "canvases, refine iterations) only composite"},
            "ms_per_canvas_wall": round(e2 / 10 * 1e3, 3), "resample_ms": round(r2, 4), "composite_ms": round(c2, 4),
            "Mpixels_per_s": round(W * H * 10 / e2 / 1e6, 1),
            "composite_roofline_frac": frac(plan_bytes(ps), c2),
            "resample_roofline": {"bound": "vector instruction issue (the lane kernel, round 5: ~150 vector instructions + 24 MFMA per band and x-tile; the "
                                           "issue port is busy most of the launch), then the prologue ramp of a one-round launch: profiles/r05_lane_kernel.txt",
                                  "algorithmic_bytes": rs_bytes, "achieved_GBps": round(rs_bytes / (r2 * 1e-3) / 1e9, 1),
                                  "frac_of_hbm_peak": frac(rs_bytes, r2)}}
        if amode == "soft":
            # the same in a BATCH (the headline's call shape): 16 canvases over the same cutouts, every canvas with its
            # own scales and positions, so no resampled layer is shared -- one resample launch + one composite launch
            nb = 16
            sets = [ppl] + synthetic.placement_sets(pobjs, W, H, 3, nb - 1)
            bplan = CompositeBatch(patlas, [SolidCanvas(psize, synthetic.SOLID_BG)] * nb,
                                   [coerce_placements(patlas, q) for q in sets])
            bout = bplan.alloc_outputs()
            for _ in range(3):
                cold_run(bplan, bout)
            torch.cuda.synchronize()
            c3, r3 = bracketed(ctx, lambda k: cold_run(bplan, bout), 10)
            c3w, r3w = bracketed(ctx, lambda k: bplan.run(bout, check=False), 10)
            bs = bplan.stats()
            out_px = sum(max(1, q["box"][2] - q["box"][0]) * max(1, q["box"][3] - q["box"][1]) for qs in sets for q in qs)
            result["placements_mode_lanczos_batch"] = {
                "alpha": amode, "canvases": nb, "resample_ms_per_canvas": round(r3 / nb, 4),
                "composite_ms_per_canvas": round(c3 / nb, 4), "gpu_ms_per_canvas_cold": round((r3 + c3) / nb, 4),
                "gpu_ms_per_canvas_warm": round((r3w + c3w) / nb, 4),
                "Mpixels_per_s_kernels": round(nb * W * H / ((r3 + c3) * 1e-3) / 1e6, 1),
                "Mpixels_per_s_kernels_warm": round(nb * W * H / ((r3w + c3w) * 1e-3) / 1e6, 1),
                "composite_roofline_frac": frac(plan_bytes(bs), c3),
                "resample_frac_of_hbm_peak": frac(4 * (bs["source_pixels"] + out_px), r3),
                "pillow_ms_per_canvas": pillow_ms,
                "note": "per-canvas times of ONE 16-canvas call (cold: every layer resampled in that call; warm: layers resident); "
                        "the single-canvas legs above are ONE partial generation of waves (a 4K canvas: 2032 four-wave workgroups = 8128 waves for 8100 pages, "
                        "on 8192 wave slots at this kernel's occupancy) and so bound by launch + ramp + dependent round trips, not by bandwidth"}
            del bplan, bout
        del pplan, patlas

    # ---- PCIe-inclusive step: job-table upload + composite + D2H of every canvas into pinned memory
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

    # ---- the PIL-level drop-in (what a user of the reference calls), beside cpu_pillow: PIL in, PIL out,
    # host transfers and Python included, one 4K canvas per call
    try:
        from PIL import Image

        imgs = ObjectImages({k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in objs.items()})
        bg_solid = Image.new("RGBA", size, tuple(synthetic.SOLID_BG))
        noise = np.random.default_rng(1).integers(0, 256, (H, W, 4), dtype=np.uint8)
        noise[:, :, 3] = 255
        bg_image = Image.fromarray(noise, "RGBA")
        t_solid, _ = _median_time(lambda: composite(bg_solid, imgs, placements[0]), 3.0, 100)
        t_image, _ = _median_time(lambda: composite(bg_image, imgs, placements[0]), 3.0, 100)
        t_render, _ = _median_time(lambda: render(layouts[0], imgs, solid), 3.0, 100)
        result["pil_dropin"] = {
            "composite_pil_solid_bg_ms": round(t_solid * 1e3, 3), "composite_pil_image_bg_ms": round(t_image * 1e3, 3),
            "render_solidcanvas_to_pil_ms": round(t_render * 1e3, 3),
            "note": "composite(PIL bg, {id: PIL}, placements) -> PIL and render(layout, ObjectImages, SolidCanvas) -> PIL "
                    "at 3840x2160 / 32 objects, median wall time per call, cutouts resident (second call on)"}
        # C1 (BASELINE configs[0], the reference's own CPU-runnable case): the committed squarespace bundle, 492 x 492,
        # 4 cutouts -- the same call at the reference's own size, beside the same loop through Pillow on this host
        try:
            bdir = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace")
            with open(os.path.join(ROOT, "tests", "golden", "bundles.json"), encoding="utf-8") as f:
                c1_row = next(r for r in json.load(f)["cases"] if r["name"] == "squarespace_1x1")
            from image_transformation_amd.background_resizing import fill_solid
            from image_transformation_amd.compositor import load_object_images
            c1_objs = load_object_images(os.path.join(bdir, "results.json"))
            c1_size = (492, 492)
            c1_bg = fill_solid(os.path.join(bdir, "background.png"), c1_size)
            c1_pl = flex.layout_to_placements(c1_row["layout"], c1_objs, c1_size)
            c1_pl2 = [{"object_id": q["object_id"], "box": [q["box"][0], q["box"][1], q["box"][0] + int((q["box"][2] - q["box"][0]) * 1.2),
                                                            q["box"][1] + int((q["box"][3] - q["box"][1]) * 1.2)]} for q in c1_pl]
            c1_pil = {k: c1_objs[k] for k in c1_objs}
            c1 = {}
            for key, q in (("identity_scale", c1_pl), ("lanczos_x1.2", c1_pl2)):
                for _ in range(300):  # (steady state: clocks up, allocator and caches warm)
                    composite(c1_bg, c1_objs, q)
                t_mine, _ = _median_time(lambda: composite(c1_bg, c1_objs, q), 1.0, 5000)
                t_pil, _ = _median_time(lambda: _pillow_composite(c1_bg, c1_pil, q), 1.0, 300)
                c1[key] = {"this_package_us": round(t_mine * 1e6, 1), "pillow_us": round(t_pil * 1e6, 1)}
                if key != "identity_scale":
                    # the number above finds the four resampled cutouts resident (the same boxes call after call); this
                    # one forgets them before every call: resample + composite each time
                    def _cold_call():
                        _native.check(_native.lib().mic_layer_cache_clear(c1_objs.atlas().ctx.handle))
                        composite(c1_bg, c1_objs, q)
                    t_cold, _ = _median_time(_cold_call, 1.0, 5000)
                    c1[key]["this_package_us_layers_not_resident"] = round(t_cold * 1e6, 1)
            # where the identity-scale call's time goes (stages timed on their own; profiles/r03_c1_breakdown.json has more)
            from image_transformation_amd import _pilmem as _pm
            from image_transformation_amd import compositor as _C
            c1_rows = _C.coerce_placements(c1_objs, c1_pl)
            c1_atlas = c1_objs.atlas()
            tab = _pm.row_table(c1_bg)
            colour = c1_bg.getpixel((0, 0))

            def _enq_wait():
                p = _C._composite_one(c1_atlas, SolidCanvas(c1_size, colour), c1_rows, 0)
                _native.check(_native.lib().mic_download_wait(c1_atlas.ctx.handle, p.ticket))
            c1["breakdown_us"] = {
                "coerce_placements": round(_median_time(lambda: _C.coerce_placements(c1_objs, c1_pl), 0.2, 2000)[0] * 1e6, 1),
                "row_table_and_3_getpixel": round(_median_time(lambda: (_pm.row_table(c1_bg), c1_bg.getpixel((0, 0)), c1_bg.getpixel((491, 491)), c1_bg.getpixel((246, 246))), 0.2, 2000)[0] * 1e6, 1),
                "exact_solid_scan_of_the_background_(overlaps_the_gpu)": round(_median_time(lambda: _C._rows_solid(tab[0], 492, 492, colour), 0.2, 2000)[0] * 1e6, 1) if tab else None,
                "enqueue_composite_and_download_then_wait_(no_scan)": round(_median_time(_enq_wait, 0.3, 2000)[0] * 1e6, 1),
                "device_only_enqueue": round(_median_time(lambda: _C._composite_one(c1_atlas, SolidCanvas(c1_size, colour), c1_rows, 0, download=False), 0.2, 2000)[0] * 1e6, 1),
                "pcie_floor_us_for_0.97_MB_at_55_GBps": 17.6}
            c1["note"] = "composite(PIL bg, load_object_images(results.json), placements) -> PIL on the squarespace bundle, 492x492 / 4 objects, median wall time"
            result["c1_bundle_dropin"] = c1
        except (OSError, StopIteration, KeyError) as exc:
            result["c1_bundle_dropin"] = {"skipped": repr(exc)}
    except ImportError:
        result["pil_dropin"] = None

    # ---- BASELINE configs[4] (C5) stage by stage, the contact sheet and the deterministic harness, each beside the
    # same stage through Pillow / NumPy on this host
    try:
        pipeline_extras(result, ctx, dev)
    except ImportError as exc:  # no Pillow on this box
        result["c5_end_to_end"] = result["contact_sheet"] = result["run_layouts"] = {"skipped": repr(exc)}

    # ---- background synthesis: median colour of RGBA images (noise: every bin populated)
    P = ctypes.c_void_p
    lib = _native.lib()
    res = torch.empty(4, dtype=torch.uint8, device=dev)
    sp = P(ctx.stream_ptr())
    med = {}
    for label, (mw, mh) in (("492x492", (492, 492)), ("1080p", (1920, 1080)), ("4k", (W, H)), ("8k", (7680, 4320))):
        img = torch.randint(0, 256, (mh, mw, 4), dtype=torch.uint8, device=dev)
        fn = lambda: lib.mic_median_rgb_dev(ctx.handle, P(img.data_ptr()), mw, mh, P(res.data_ptr()), sp)  # noqa: E731
        for _ in range(3):
            _native.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        med[label] = {"us": round(us, 1), "frac_of_hbm_peak": round(4 * mw * mh / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        del img
    result["median_noise"] = med
    result["median_4k_noise_us"] = med["4k"]["us"]


if __name__ == "__main__":
    main()
