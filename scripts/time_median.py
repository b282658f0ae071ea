"""Median-colour kernel (mic_median_rgb_dev) time per image kind and size."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import _native
lib = _native.lib(); ctx = _native.context(0); P = ctypes.c_void_p
res = torch.empty(4, dtype=torch.uint8, device="cuda")
for label, (w, h) in (("492x492", (492, 492)), ("970x250", (970, 250)), ("1080p", (1920, 1080)), ("4k", (3840, 2160)), ("8k", (7680, 4320))):
    for kind in ("noise", "flat", "photo", "photo2"):
        if kind == "noise":
            img = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
        elif kind == "flat":
            img = torch.full((h, w, 4), 117, dtype=torch.uint8, device="cuda")
        else:
            # photo-like: smooth gradients + sensor noise (sigma 2 / 6): neighbouring pixels land in a handful of
            # adjacent bins but are rarely equal -- the case between "flat" and "noise"
            yy = torch.linspace(0, 1, h, device="cuda")[:, None]
            xx = torch.linspace(0, 1, w, device="cuda")[None, :]
            base = torch.stack([60 + 120 * xx * torch.ones_like(yy), 90 + 80 * yy * torch.ones_like(xx), 200 - 100 * xx * yy], dim=2)
            base = base + torch.randn_like(base) * (2.0 if kind == "photo" else 6.0)
            img = torch.cat([base.clamp(0, 255).to(torch.uint8), torch.full((h, w, 1), 255, dtype=torch.uint8, device="cuda")], dim=2).contiguous()
        fn = lambda: lib.mic_median_rgb_dev(ctx.handle, P(img.data_ptr()), w, h, P(res.data_ptr()), P(ctx.stream_ptr()))
        for _ in range(5):
            _native.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        print(f"{label:8s} {kind:5s} {us:6.1f} us  {4 * w * h / (us * 1e-6) / 8e12:.3f} of HBM peak")
