"""Ceilings and ablations for the composite kernel on one MI355X (run via gpurun)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import _native, flex, synthetic
from image_transformation_amd.compositor import Atlas, CompositeBatch, SolidCanvas, coerce_placements
from image_transformation_amd.background_resizing import fill_solid_device

ctx = _native.context()
lib = _native.lib()
W, H, B = 3840, 2160, 16


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


nbytes = B * W * H * 4
a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
b = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
a32, b32 = a.view(torch.int32), b.view(torch.int32)
t = timeit(lambda: a32.fill_(7))
print(f"torch fill   {nbytes/1e6:.0f} MB: {t*1e3:.1f} us  {nbytes/t/1e6:.0f} GB/s (write)")
t = timeit(lambda: b32.copy_(a32))
print(f"torch copy   {nbytes/1e6:.0f} MB: {t*1e3:.1f} us  {2*nbytes/t/1e6:.0f} GB/s (read+write)")
col = (ctypes.c_uint8 * 4)(1, 2, 3, 255)
t = timeit(lambda: _native.check(lib.mic_fill_solid(ctx.handle, ctypes.c_void_p(a.data_ptr()), W, H * B, col, ctypes.c_void_p(ctx.stream_ptr()))))
print(f"mic_fill     {nbytes/1e6:.0f} MB: {t*1e3:.1f} us  {nbytes/t/1e6:.0f} GB/s (write)")

size, objs, layouts = synthetic.c3_workload("binary", seed=3, n_layouts=B)
atlas = Atlas(objs)
rows = [coerce_placements(atlas, flex.layout_to_placements(l, atlas, size)) for l in layouts]
solid = [SolidCanvas(size, synthetic.SOLID_BG)] * B
outs = [[torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(B)] for _ in range(2)]
k = [0]


def run(plan):
    def f():
        plan.run(outs[k[0] % 2]); k[0] += 1
    return f


def report(name, plan):
    t = timeit(run(plan))
    plan.run(outs[0]); torch.cuda.synchronize()
    st = ctx.stats()
    balg = 4 * st["canvas_pixels"] + 4 * st["layer_pixels"]
    ctx.profile_begin(50)
    for _ in range(50):
        plan.run(outs[k[0] % 2]); k[0] += 1
    n, c, r = ctx.profile_end()
    print(f"{name:28s} wall {t*1e3:7.1f} us  kernel {c/n*1e3:7.1f} us  B_alg {balg/1e6:6.0f} MB  {balg/(c/n)/1e6:6.0f} GB/s  frac {balg/(c/n)/1e6/8000:.3f}")


report("composite 0 layers (fill)", CompositeBatch(atlas, solid, [[] for _ in range(B)]))
report("composite C3 flex binary", CompositeBatch(atlas, solid, rows))
bgs = [torch.randint(0, 255, (H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(B)]
report("composite 0 layers, bg image", CompositeBatch(atlas, bgs, [[] for _ in range(B)]))
report("composite C3 flex, bg image", CompositeBatch(atlas, bgs, rows))
for nb in ():
    report(f"composite C3 flex, batch {nb}", CompositeBatch(atlas, solid[:nb], rows[:nb]))
