"""The drop-in calls are thread-safe like the Pillow calls they replace (the reference's Streamlit app runs
every session on its own thread): composite(), render(), fill_solid-style fills and resizes issued from
several threads at once on ONE context must each give the result they give alone.  libmic serialises its
entry points per context (include/mic.h, "Threads"); the Python layer shares no staging buffers."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402  (the checker)


def test_concurrent_pil_composites_match_the_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; none is visible")
    from PIL import Image
    from image_transformation_amd import synthetic
    from image_transformation_amd.compositor import ObjectImages, SolidCanvas, composite, render

    n_threads, n_iters = 4, 6
    jobs = []
    for t in range(n_threads):
        W, H = 700 + 64 * t, 420 + 40 * t
        # every thread its own bundle, background kind and mode (resampled layers use the context's arena)
        size, objs, pl = synthetic.placements_workload(W, H, 6, 100 + t, "soft" if t % 2 else "binary")
        imgs = ObjectImages({k: Image.fromarray(np.ascontiguousarray(v), "RGBA") for k, v in objs.items()})
        if t % 2:
            bg = np.random.default_rng(t).integers(0, 256, (H, W, 4), dtype=np.uint8)
        else:
            bg = np.empty((H, W, 4), np.uint8)
            bg[:] = (9 * t, 200, 33, 255)
        want = oracle.composite(bg, objs, pl)
        jobs.append((Image.fromarray(bg, "RGBA"), imgs, pl, want))

    errors = []
    start = threading.Barrier(n_threads)

    def work(t):
        bg, imgs, pl, want = jobs[t]
        try:
            start.wait()
            for _ in range(n_iters):
                got = np.array(composite(bg, imgs, pl))
                if not np.array_equal(got, want):
                    errors.append((t, "composite differs"))
                    return
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_concurrent_contact_sheets_are_identical(golden_dir):
    """Contact sheets built from several threads at once (each thread rasterises labels with a FreeType face of its
    own; masks and measurements are cached per process) equal the sheet a single thread builds."""
    import os
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; none is visible")
    import cases
    from image_transformation_amd.contact_sheet import build_labeled_contact_sheet
    bundles = [os.path.join(cases.BUNDLE_DIR, b) for b in cases.BUNDLES]
    want = [np.array(build_labeled_contact_sheet(os.path.join(b, "objects"), os.path.join(b, "results.json"))) for b in bundles]
    errors = []
    start = threading.Barrier(4)

    def work(t):
        try:
            start.wait()
            for k in range(6):
                b = bundles[(t + k) % len(bundles)]
                got = np.array(build_labeled_contact_sheet(os.path.join(b, "objects"), os.path.join(b, "results.json"),
                                                          font_size=24))
                if not np.array_equal(got, want[(t + k) % len(bundles)]):
                    errors.append((t, k, "sheet differs"))
                    return
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
