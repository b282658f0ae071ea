import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import _native
ctx = _native.context(); torch.cuda.synchronize()
def lap(name, fn):
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True); return r
a = lap("first pinned 2 MB", lambda: torch.empty(2 << 20, dtype=torch.uint8, pin_memory=True))
b = lap("second pinned 2 MB", lambda: torch.empty(2 << 20, dtype=torch.uint8, pin_memory=True))
d = lap("first device 2 MB", lambda: torch.empty(2 << 20, dtype=torch.uint8, device="cuda"))
e = lap("second device 2 MB", lambda: torch.empty(2 << 20, dtype=torch.uint8, device="cuda"))
lap("first copy_", lambda: d.copy_(a, non_blocking=True))
lap("second copy_", lambda: e.copy_(b, non_blocking=True))
lap("first Event", lambda: torch.cuda.Event().record())
lap("second Event", lambda: torch.cuda.Event().record())
