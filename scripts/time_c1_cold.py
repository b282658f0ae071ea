"""Wall time of the reference-sized drop-in call with LANCZOS boxes and NO resident layers (the layer cache is cleared before
every call): composite(PIL bg, load_object_images(results.json), placements x1.2) on the squarespace bundle (492 x 492, 4
cutouts), and the same for the contact sheet (thumbnails not resident).  A/B of the resample routing by environment:
MIC_RS_LANE=0 (tile kernel), MIC_RS_LANE_MIN_SLOTS=<n>."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from image_transformation_amd import _native, flex
from image_transformation_amd.background_resizing import fill_solid
from image_transformation_amd.compositor import composite, load_object_images
from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
bdir = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace")
with open(os.path.join(ROOT, "tests", "golden", "bundles.json"), encoding="utf-8") as f:
    row = next(r for r in json.load(f)["cases"] if r["name"] == "squarespace_1x1")
objs = load_object_images(os.path.join(bdir, "results.json"))
bg = fill_solid(os.path.join(bdir, "background.png"), (492, 492))
pl = flex.layout_to_placements(row["layout"], objs, (492, 492))
pl2 = [{"object_id": q["object_id"], "box": [q["box"][0], q["box"][1], q["box"][0] + int((q["box"][2] - q["box"][0]) * 1.2),
                                            q["box"][1] + int((q["box"][3] - q["box"][1]) * 1.2)]} for q in pl]
ctx = objs.atlas().ctx


def med(fn, n=600):
    for _ in range(100):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


def cold():
    _native.check(_native.lib().mic_layer_cache_clear(ctx.handle))
    composite(bg, objs, pl2)


def sheet_cold():
    _native.check(_native.lib().mic_layer_cache_clear(ctx.handle))
    build_labeled_contact_sheet(os.path.join(bdir, "objects"), os.path.join(bdir, "results.json"), as_tensor=True)
    torch.cuda.synchronize()


print(f"C1 x1.2 layers resident {med(lambda: composite(bg, objs, pl2)):.1f} us, not resident {med(cold):.1f} us; "
      f"contact sheet (thumbnails not resident, to device) {med(sheet_cold, 300):.1f} us; stats {ctx.stats()['marched_layers']} marched layers "
      f"[MIC_RS_LANE={os.environ.get('MIC_RS_LANE', '1')} MIN_SLOTS={os.environ.get('MIC_RS_LANE_MIN_SLOTS', 'default')}]")
