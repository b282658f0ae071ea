import ctypes, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from image_transformation_amd import _native
ctx = _native.context(); lib = _native.lib(); P = ctypes.c_void_p
def t(sw, sh, dw, dh, iters=20):
    s = torch.randint(0, 256, (sh, sw, 4), dtype=torch.uint8, device="cuda")
    d = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
    f = lambda: _native.check(lib.mic_resize(ctx.handle, P(s.data_ptr()), sw, sh, P(d.data_ptr()), dw, dh, 0, P(ctx.stream_ptr())))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize()
    print(f"resize {sw}x{sh}->{dw}x{dh}: {(time.perf_counter()-t0)/iters*1e6:8.1f} us")
for a in [(4000, 3000, 256, 192), (2000, 1500, 256, 192), (1000, 800, 256, 205), (3000, 2000, 1500, 1000), (8000, 6000, 512, 384)]:
    t(*a)
