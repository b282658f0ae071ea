"""A few launches of mic_median_rgb_dev on one image (for rocprofv3 --kernel-trace / --pmc runs).
MIC_CASE = "<w>x<h>:<noise|flat|sprinkle>"."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_transformation_amd import _native

case = os.environ.get("MIC_CASE", "3840x2160:noise")
dims, kind = case.split(":")
w, h = (int(v) for v in dims.split("x"))
ctx = _native.context()
lib = _native.lib()
P = ctypes.c_void_p
a = torch.randint(0, 256, (h, w, 4), dtype=torch.uint8, device="cuda")
if kind in ("flat", "sprinkle"):
    a[:, :, 0] = 38; a[:, :, 1] = 73; a[:, :, 2] = 115; a[:, :, 3] = 255
if kind == "sprinkle":
    a[::7, ::5, :3] = torch.randint(0, 256, a[::7, ::5, :3].shape, dtype=torch.uint8, device="cuda")
out = torch.empty(4, dtype=torch.uint8, device="cuda")
for _ in range(int(os.environ.get("MIC_ITERS", "10"))):
    _native.check(lib.mic_median_rgb_dev(ctx.handle, P(a.data_ptr()), w, h, P(out.data_ptr()), P(ctx.stream_ptr())))
torch.cuda.synchronize()
print("done", case, out.tolist())
