"""A/B of mic_resize between two builds of libmic (raw ctypes, one process per library):
   python scripts/ab_resize.py <libmic.so>      -> prints us per call for a list of shapes"""
import ctypes, sys
import numpy as np
import torch
lib = ctypes.CDLL(sys.argv[1])
lib.mic_last_error.restype = ctypes.c_char_p
P = ctypes.c_void_p
ctx = P()
assert lib.mic_create(0, ctypes.byref(ctx)) == 0, lib.mic_last_error()
rng = np.random.default_rng(1)
shapes = [((186, 237), (1488, 1896)), ((447, 116), (1788, 464)), ((231, 88), (1848, 704)),   # C5: x8 / x4 upscales
          ((1000, 800), (256, 205)), ((2000, 1500), (256, 192)), ((4000, 3000), (256, 192)),  # thumbnails
          ((1280, 720), (1920, 1080)), ((1920, 1080), (1280, 720)), ((800, 600), (800, 601)), ((640, 480), (641, 480)),
          ((357, 207), (256, 148)), ((3840, 2160), (1920, 1080)), ((512, 512), (2048, 2048))]
stream = P(torch.cuda.current_stream().cuda_stream)
for (sw, sh), (dw, dh) in shapes:
    src = torch.from_numpy(rng.integers(0, 256, (sh, sw, 4), dtype=np.uint8)).cuda()
    dst = torch.empty((dh, dw, 4), dtype=torch.uint8, device="cuda")
    f = lambda: lib.mic_resize(ctx, P(src.data_ptr()), sw, sh, P(dst.data_ptr()), dw, dh, 0, stream)
    for _ in range(3):
        assert f() == 0, lib.mic_last_error()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{sw}x{sh} -> {dw}x{dh}: {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us")
