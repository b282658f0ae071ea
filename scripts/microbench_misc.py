"""fill / gradient / overlay kernels at 4K (events, back-to-back launches)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_transformation_amd import _native
ctx = _native.context(); lib = _native.lib(); P = ctypes.c_void_p
W, H = 3840, 2160
out = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
sp = P(ctx.stream_ptr())
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
U8 = ctypes.c_uint8
print("fill_solid 4K      %.1f us" % timeit(lambda: lib.mic_fill_solid(ctx.handle, P(out.data_ptr()), W, H, (U8 * 4)(1, 2, 3, 255), sp)))
print("fill_gradient 4K   %.1f us" % timeit(lambda: lib.mic_fill_gradient(ctx.handle, P(out.data_ptr()), W, H, (U8 * 3)(10, 20, 30), (U8 * 3)(200, 100, 50), 0, sp)))
rng = np.random.default_rng(0)
for n in (4, 32, 150):
    b = np.zeros((n, 4), np.int32)
    b[:, 0] = rng.integers(0, W - 400, n); b[:, 1] = rng.integers(0, H - 300, n)
    b[:, 2] = b[:, 0] + rng.integers(50, 900, n); b[:, 3] = b[:, 1] + rng.integers(50, 700, n)
    c = rng.integers(0, 256, (n, 4)).astype(np.uint8)
    f = lambda: lib.mic_draw_rect_outlines(ctx.handle, P(out.data_ptr()), W, H, n, b.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), c.ctypes.data_as(ctypes.POINTER(U8)), 3, sp)
    print("overlay 4K, %3d outlines  %.1f us" % (n, timeit(f)))
