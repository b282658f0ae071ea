"""Where the wall time of the reference-sized drop-in call goes: composite(PIL bg, load_object_images(results.json),
placements) -> PIL on the squarespace bundle, 492x492 / 4 cutouts (BASELINE configs[0]).  Every stage of the call is
timed on its own (median of many repetitions), then the call as a whole -> profiles/rNN_c1_breakdown.json."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from image_transformation_amd import _native, _pilmem, compositor as C, flex
from image_transformation_amd.background_resizing import fill_solid


def med(fn, n=3000, warm=200):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter_ns(); fn(); ts.append(time.perf_counter_ns() - t0)
    ts.sort()
    return ts[len(ts) // 2] / 1e3


bdir = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace")
with open(os.path.join(ROOT, "tests", "golden", "bundles.json")) as f:
    row = next(r for r in json.load(f)["cases"] if r["name"] == "squarespace_1x1")
objs = C.load_object_images(os.path.join(bdir, "results.json"))
bg = fill_solid(os.path.join(bdir, "background.png"), (492, 492))
pl = flex.layout_to_placements(row["layout"], objs, (492, 492))
out = {}
C.composite(bg, objs, pl)
atlas = objs.atlas()
ctx = atlas.ctx
rows = C.coerce_placements(objs, pl)
out["whole_call_us"] = med(lambda: C.composite(bg, objs, pl))
out["coerce_placements_us"] = med(lambda: C.coerce_placements(objs, pl))
out["as_atlas_us"] = med(lambda: C._as_atlas(objs))
out["solid_colour_scan_us"] = med(lambda: _pilmem.solid_colour(bg))
canvas = C.SolidCanvas(bg.size, _pilmem.solid_colour(bg))
out["solid_canvas_object_us"] = med(lambda: C.SolidCanvas(bg.size, (220, 238, 245, 255)))
out["build_jobs_us"] = med(lambda: C._build_jobs(atlas, [canvas], [rows]))
dev_out = torch.empty((492, 492, 4), dtype=torch.uint8, device=ctx.torch_device)
out["torch_empty_out_us"] = med(lambda: torch.empty((492, 492, 4), dtype=torch.uint8, device=ctx.torch_device))
def dev_call():
    C.composite_device(atlas, [canvas], [rows], outs=[dev_out])
out["composite_device_enqueue_us"] = med(dev_call)
def dev_call_sync():
    C.composite_device(atlas, [canvas], [rows], outs=[dev_out]); torch.cuda.current_stream().synchronize()
out["composite_device_plus_sync_us"] = med(dev_call_sync)
out["stream_ptr_us"] = med(lambda: ctx.stream_ptr())
out["pinned_alloc_us"] = med(lambda: C._pinned(492 * 492 * 4))
pin = C._pinned(492 * 492 * 4)
def d2h():
    pin.copy_(dev_out.reshape(-1), non_blocking=True); torch.cuda.current_stream().synchronize()
out["d2h_copy_plus_sync_us"] = med(d2h)
out["frombuffer_us"] = med(lambda: Image.frombuffer("RGBA", (492, 492), pin.numpy(), "raw", "RGBA", 0, 1))
out["to_pil_us"] = med(lambda: C._to_pil(dev_out))
out["empty_sync_us"] = med(lambda: torch.cuda.current_stream().synchronize())
# ---- the round-3 path: speculative solid background, per-thread cached job arrays, event-waited download
tab = _pilmem.row_table(bg)
out["r03_row_table_us"] = med(lambda: _pilmem.row_table(bg))
out["r03_three_getpixel_us"] = med(lambda: (bg.getpixel((0, 0)), bg.getpixel((491, 491)), bg.getpixel((246, 246))))
out["r03_rows_solid_scan_us"] = med(lambda: C._rows_solid(tab[0], 492, 492, (220, 238, 245, 255)))
out["r03_stream_of_us"] = med(lambda: C._stream_of(ctx))
def one_enqueue():
    p = C._composite_one(atlas, canvas, rows, 0)
    _native.check(_native.lib().mic_download_wait(ctx.handle, p.ticket))
def one_enqueue_only():
    return C._composite_one(atlas, canvas, rows, 0)
t_both = med(one_enqueue)
pend = []
def enq():
    pend.append(C._composite_one(atlas, canvas, rows, 0))
    if len(pend) >= 8:
        for q in pend:
            _native.check(_native.lib().mic_download_wait(ctx.handle, q.ticket))
        pend.clear()
out["r03_composite_one_enqueue_plus_wait_us"] = t_both
p0 = C._composite_one(atlas, canvas, rows, 0)
out["r03_pending_image_us"] = med(lambda: p0.image())
out["r03_device_only_composite_one_us"] = med(lambda: C._composite_one(atlas, canvas, rows, 0, download=False))
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "c1_breakdown.json"), "w") as f:
    json.dump(out, f, indent=1)
