import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from image_transformation_amd import _native
ctx = _native.context()
lib = _native.lib()
rng = np.random.default_rng(0)
for (sw, sh, dw, dh) in [(8, 6, 5, 6), (8, 6, 8, 4), (8, 6, 5, 4), (8, 6, 13, 9), (37, 21, 12, 7)]:
    src = rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)
    d = torch.zeros((dh, dw, 4), dtype=torch.uint8, device="cuda")
    s = torch.from_numpy(src).cuda()
    _native.check(lib.mic_resize(ctx.handle, ctypes.c_void_p(s.data_ptr()), sw, sh, ctypes.c_void_p(d.data_ptr()), dw, dh, 0, ctypes.c_void_p(ctx.stream_ptr())))
    got = d.cpu().numpy(); want = oracle.resize(src, (dw, dh))
    bad = (got != want).any(axis=2)
    print((sw, sh, dw, dh), "mismatch px:", int(bad.sum()), "of", bad.size)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print(" first bad", xs[:5], ys[:5], got[ys[0], xs[0]], want[ys[0], xs[0]])
        print(" bad map:\n", bad.astype(int))
