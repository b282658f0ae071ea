"""Where a fresh process' FIRST contact sheet goes (cold_start.contact_sheet_first_ms in the bench line): stage by stage."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t = {}
def lap(name, t0): t[name] = round((time.perf_counter() - t0) * 1e3, 3)
t0 = time.perf_counter(); import torch; lap("import_torch", t0)
t0 = time.perf_counter()
from image_transformation_amd import _native, contact_sheet as cs
from image_transformation_amd.compositor import load_object_images, _to_pil
lap("import_package", t0)
t0 = time.perf_counter(); ctx = _native.context(); torch.cuda.synchronize(); lap("context", t0)
bdir = os.path.join(ROOT, "tests", "golden", "bundles", "squarespace"); rj = os.path.join(bdir, "results.json")
t0 = time.perf_counter(); objs = load_object_images(rj, shared=True); lap("load_object_images", t0)
t0 = time.perf_counter(); atlas = objs.atlas(); torch.cuda.synchronize(); lap("atlas_upload", t0)
t0 = time.perf_counter(); font = cs._resolve_font(24); lap("font", t0)
items = sorted(json.load(open(rj)), key=lambda it: int(it["object_id"]))
t0 = time.perf_counter()
for it in items:
    cs._measure_cached(str(it["label"]), font); cs._label_mask(str(it["label"]), font)
lap("label_masks", t0)
t0 = time.perf_counter(); out = cs.build_labeled_contact_sheet("", rj, as_tensor=True); lap("sheet_enqueue_first", t0)
t0 = time.perf_counter(); torch.cuda.synchronize(); lap("sheet_sync_first", t0)
t0 = time.perf_counter(); im = _to_pil(out); lap("to_pil_first", t0)
t0 = time.perf_counter(); out = cs.build_labeled_contact_sheet("", rj, as_tensor=True); torch.cuda.synchronize(); lap("sheet_second", t0)
t0 = time.perf_counter(); im = _to_pil(out); lap("to_pil_second", t0)
print(json.dumps(t))
