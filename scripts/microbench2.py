"""Median / resize / general-composite timings on one MI355X (run via gpurun)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import _native, flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
ctx = _native.context(); lib = _native.lib(); P = ctypes.c_void_p


def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for (w, h, kind) in [(3840, 2160, "noise"), (7680, 4320, "noise"), (7680, 4320, "flat")]:
    a = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
    if kind == "flat":
        a[:, :, 0] = 38; a[:, :, 1] = 73; a[:, :, 2] = 115; a[:, :, 3] = 255
    out = torch.empty(4, dtype=torch.uint8, device="cuda")
    t = timeit(lambda: _native.check(lib.mic_median_rgb_dev(ctx.handle, P(a.data_ptr()), w, h, P(out.data_ptr()), P(ctx.stream_ptr()))))
    print(f"median {w}x{h} {kind:6s}: {t*1e3:8.1f} us  {w*h*4/t/1e6:7.0f} GB/s")

for (sw, sh, dw, dh) in [(1280, 720, 640, 360), (1280, 720, 1920, 1080), (440, 500, 660, 750), (440, 500, 220, 250), (1000, 800, 256, 205)]:
    s = torch.randint(0, 256, (sh, sw, 4), dtype=torch.uint8, device="cuda")
    d = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: _native.check(lib.mic_resize(ctx.handle, P(s.data_ptr()), sw, sh, P(d.data_ptr()), dw, dh, 0, P(ctx.stream_ptr()))))
    print(f"resize {sw}x{sh}->{dw}x{dh}: {t*1e3:8.1f} us   in+out {(sw*sh+dw*dh)*4/t/1e6:7.1f} GB/s  out {dw*dh/t/1e3:8.1f} Mpx/s")

# general kernel: C4 variants by ratio
objs, variants = synthetic.c4_workload("binary", n_variants=64)
atlas = Atlas(objs)
for ratio_idx, name in enumerate(synthetic.RATIOS_C4):
    vs = [v for i, v in enumerate(variants) if i % 4 == ratio_idx]
    rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, sz)) for (sz, l) in vs]
    plan = CompositeBatch(atlas, [SolidCanvas(sz, synthetic.SOLID_BG) for (sz, _) in vs], rows)
    outs = [plan.alloc_outputs() for _ in range(2)]
    k = [0]
    def f():
        plan.run(outs[k[0] % 2], check=False); k[0] += 1
    t = timeit(f)
    st = plan.stats(); balg = 4 * st["canvas_pixels"] + 4 * st["layer_pixels"]
    print(f"C4 ratio {name:5s} {vs[0][0]} x{len(vs)}: {t*1e3:8.1f} us  {balg/t/1e6:7.0f} GB/s  frac {balg/t/1e6/8000:.3f}")
for alpha in ("soft",):
    size, objs2, layouts = synthetic.c3_workload(alpha, seed=3, n_layouts=16)
    at2 = Atlas(objs2)
    rows = [coerce_placements(at2, flex.layout_to_placements(l, at2, size)) for l in layouts]
    plan = CompositeBatch(at2, [SolidCanvas(size, synthetic.SOLID_BG)] * 16, rows)
    outs = [plan.alloc_outputs() for _ in range(2)]
    k = [0]
    def f2():
        plan.run(outs[k[0] % 2], check=False); k[0] += 1
    t = timeit(f2)
    st = plan.stats(); balg = 4 * st["canvas_pixels"] + 4 * st["layer_pixels"]
    print(f"C3 flex {alpha} alpha x16: {t*1e3:8.1f} us  {balg/t/1e6:7.0f} GB/s  frac {balg/t/1e6/8000:.3f}")
