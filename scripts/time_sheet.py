import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
for b in ("squarespace", "audio_book"):
    d = os.path.join(ROOT, "tests", "golden", "bundles", b); rj = os.path.join(d, "results.json")
    for as_tensor in (True, False):
        fn = lambda: build_labeled_contact_sheet("", rj, as_tensor=as_tensor)
        for _ in range(5): fn()
        torch.cuda.synchronize(); ts = []
        for _ in range(200):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort(); print(b, "as_tensor" if as_tensor else "PIL out", f"{ts[100]*1e3:.3f} ms")
