"""Host -> device upload of a 4K PIL image: one memmove pass (4 Python threads) + ONE DMA, against mic_upload_rows (the
call's own threads, DMA per 4 MB piece)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from image_transformation_amd import _native, _pilmem, compositor as C
ctx = _native.context()
W, H = 3840, 2160
noise = np.random.default_rng(1).integers(0, 256, (H, W, 4), dtype=np.uint8)
im = Image.fromarray(noise, "RGBA")
big = Image.new("RGBA", (W, H)); big.paste(im)
P = ctypes.c_void_p


def old():
    pin = C._pinned(H * W * 4)
    _pilmem.copy_to(big, pin.data_ptr())
    dev = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    dev.view(-1).copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    return dev


def new():
    dev = C._upload(big, ctx)
    torch.cuda.synchronize()
    return dev


def memmove_only():
    pin = C._pinned(H * W * 4)
    _pilmem.copy_to(big, pin.data_ptr())


pin0 = C._pinned(H * W * 4); dev0 = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
def dma_only():
    dev0.view(-1).copy_(pin0, non_blocking=True); torch.cuda.synchronize()


for name, fn in (("memmove only (4 Python threads)", memmove_only), ("one DMA only", dma_only), ("memmove + one DMA", old), ("mic_upload_rows", new)):
    for _ in range(5): fn()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); print(f"{name}: median {ts[20] * 1e3:.3f} ms")
assert np.array_equal(new().cpu().numpy(), noise)
