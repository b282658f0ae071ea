"""Single-canvas composite launches (the reference's call shape, compositor.py:6-22) for a rocprofv3 kernel trace:
a fixed sequence of cases, each MIC_ITERS launches of composite_kernel, written to $MIC_CASES_JSON so that
scripts/trace_cases.py can cut the trace's composite dispatches (in time order) back into the cases.

  4k_flex        C3: 3840x2160, 32 binary cutouts, depth-2 Flex layout, identity scale
  4k_layers32    C3 placements: 32 LANCZOS layers (resident after the first run), overlapping
  8k_c5_word     C5 iter 0 on the audio_book bundle at 7680x4320, solid colour as a DEVICE word (solid_canvas)
  8k_c5_host     the same with the colour in the job record (SolidCanvas)
  1080p          C2: 1920x1080, 8 cutouts
  sheet          the 1024x328 contact sheet of the squarespace bundle (one composite per sheet)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from image_transformation_amd import flex, synthetic
from image_transformation_amd.background_resizing import solid_canvas
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements, load_object_images
from image_transformation_amd.contact_sheet import build_labeled_contact_sheet

n = int(os.environ.get("MIC_ITERS", "40"))
gold = os.path.join(ROOT, "tests", "golden")
seq = []


def run(name, plan, n_out):
    outs = [plan.alloc_outputs() for _ in range(n_out)]
    for k in range(n):
        plan.run(outs[k % n_out], check=False)
    torch.cuda.synchronize()
    st = plan.stats()
    seq.append({"case": name, "launches": n, "bytes": 4 * (st["canvas_pixels"] + st["layer_pixels"]),
                "workgroups": st["composite_blocks"]})


size, objs, layouts = synthetic.c3_workload("binary", 3, 1)
atlas = Atlas(objs)
run("4k_flex", CompositeBatch(atlas, [SolidCanvas(size, synthetic.SOLID_BG)],
                              [coerce_placements(atlas, flex.layout_to_placements(layouts[0], atlas, size))]), 12)
size, objs, pl = synthetic.placements_workload(3840, 2160, 32, 3, "soft")
a2 = Atlas(objs)
run("4k_layers32", CompositeBatch(a2, [SolidCanvas(size, synthetic.SOLID_BG)], [coerce_placements(a2, pl)]), 12)
base = os.path.join(gold, "bundles", "audio_book")
with open(os.path.join(gold, "big_hashes.json")) as f:
    big = {r["name"]: r for r in json.load(f)["cases"]}
word = solid_canvas(os.path.join(base, "background.png"), (7680, 4320))
objects = load_object_images(os.path.join(base, "results.json"))
a3 = objects.atlas()
rows = [coerce_placements(a3, big["c5_audio_book_iter0"]["placements"])]
run("8k_c5_word", CompositeBatch(a3, [word], rows), 3)
run("8k_c5_host", CompositeBatch(a3, [SolidCanvas((7680, 4320), tuple(word.rgba))], rows), 3)
size, objs, layout = synthetic.c2_workload("binary", 2)
a4 = Atlas(objs)
run("1080p", CompositeBatch(a4, [SolidCanvas(size, synthetic.SOLID_BG)],
                            [coerce_placements(a4, flex.layout_to_placements(layout, a4, size))]), 12)
sq = os.path.join(gold, "bundles", "squarespace")
for _ in range(n):
    build_labeled_contact_sheet(os.path.join(sq, "objects"), os.path.join(sq, "results.json"), as_tensor=True)
torch.cuda.synchronize()
seq.append({"case": "sheet", "launches": n, "bytes": 4 * 1024 * 328, "workgroups": None})
with open(os.environ.get("MIC_CASES_JSON", "/dev/stdout"), "w") as f:
    json.dump(seq, f)
