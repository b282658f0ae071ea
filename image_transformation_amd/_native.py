"""ctypes binding of libmic.so (include/mic.h).  PyTorch-ROCm is only the buffer handoff:
device memory is owned by torch uint8 tensors and enters the C ABI as data_ptr().

There is no CPU fallback: if the library is missing or no MI355X is visible, the first use
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIC_LIB") or os.path.join(_HERE, "libmic.so")  # MIC_LIB: an alternative build (tuning runs)

ABI_VERSION = (1, 9)  # mic_version(): include/mic.h as this file binds it
LANCZOS = 0
BILINEAR = 1
ERR_FORMAT = -5
ERR_UNSUPPORTED = -6


class MicError(RuntimeError):
    """A libmic call failed (carries the library's status code)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libmic error {code}: {message}")
        self.code = code


class Placement(ctypes.Structure):
    _fields_ = [("atlas", ctypes.c_int32), ("object_id", ctypes.c_int32), ("box", ctypes.c_int32 * 4)]


class Job(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("bg_dev", ctypes.c_void_p), ("bg_rgba", ctypes.c_uint8 * 4),
                ("n_placements", ctypes.c_int32), ("placements", ctypes.POINTER(Placement)),
                ("out_dev", ctypes.c_void_p), ("bg_rgba_dev", ctypes.c_void_p)]


class LabelStrip(ctypes.Structure):
    _fields_ = [("cell", ctypes.c_int32), ("x", ctypes.c_int32), ("y", ctypes.c_int32), ("w", ctypes.c_int32),
                ("h", ctypes.c_int32), ("coverage_host", ctypes.c_void_p)]


class ImageView(ctypes.Structure):
    _fields_ = [("rgba_dev", ctypes.c_void_p), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("stride_bytes", ctypes.c_int64)]


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("canvas_pixels", "layer_pixels", "source_pixels", "resampled_layers", "identity_layers",
                 "skipped_placements", "composite_blocks", "marched_layers", "cached_layers")]

    def as_dict(self) -> Dict[str, int]:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()

# name -> (restype, argtypes); every symbol include/mic.h declares
_P = ctypes.c_void_p
_I32P = ctypes.POINTER(ctypes.c_int32)
SYMBOLS = {
    "mic_last_error": (ctypes.c_char_p, []),
    "mic_version": (ctypes.c_int, []),
    "mic_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_P)]),
    "mic_destroy": (ctypes.c_int, [_P]),
    "mic_sync": (ctypes.c_int, [_P, _P]),
    "mic_selftest": (ctypes.c_int, [_P, _P]),
    "mic_atlas_create": (ctypes.c_int, [_P, ctypes.c_int, _I32P, _I32P, _I32P, ctypes.POINTER(_P),
                                        ctypes.POINTER(_P)]),
    "mic_atlas_blob_size": (ctypes.c_int, [ctypes.c_int, _I32P, _I32P, ctypes.POINTER(ctypes.c_size_t)]),
    "mic_atlas_blob_layout": (ctypes.c_int, [ctypes.c_int, _I32P, _I32P, _I32P, _P, ctypes.c_size_t,
                                             ctypes.POINTER(ctypes.c_uint64)]),
    "mic_atlas_from_device_blob": (ctypes.c_int, [_P, _P, ctypes.c_size_t, _P, ctypes.POINTER(_P)]),
    "mic_atlas_device_blob": (ctypes.c_int, [_P, ctypes.POINTER(_P), ctypes.POINTER(ctypes.c_size_t)]),
    "mic_atlas_count": (ctypes.c_int, [_P]),
    "mic_atlas_lookup": (ctypes.c_int, [_P, ctypes.c_int32, _I32P, _I32P, ctypes.POINTER(_P)]),
    "mic_atlas_destroy": (ctypes.c_int, [_P]),
    "mic_composite_batch": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(_P), ctypes.c_int,
                                           ctypes.POINTER(Job), ctypes.c_int, _P]),
    "mic_plan_create": (ctypes.c_int, [_P, ctypes.c_int, ctypes.POINTER(_P), ctypes.c_int, ctypes.POINTER(Job),
                                       ctypes.c_int, ctypes.POINTER(_P)]),
    "mic_plan_run": (ctypes.c_int, [_P, ctypes.POINTER(_P), _P]),
    "mic_plan_destroy": (ctypes.c_int, [_P]),
    "mic_plan_invalidate": (ctypes.c_int, [_P]),
    "mic_layer_cache_clear": (ctypes.c_int, [_P]),
    "mic_plan_stats": (ctypes.c_int, [_P, ctypes.POINTER(Stats)]),
    "mic_resize": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int32,
                                  ctypes.c_int, _P]),
    "mic_median_rgb": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8), _P]),
    "mic_median_rgb_dev": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    "mic_median_rgb_batch": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ImageView), ctypes.POINTER(ctypes.c_uint8), _P]),
    "mic_fill_solid": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8), _P]),
    "mic_fill_gradient": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8),
                                         ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, _P]),
    "mic_draw_rect_outlines": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _I32P,
                                              ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32, _P]),
    "mic_flex_place": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, _I32P, _I32P, _I32P,
                                      ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _I32P, _I32P, _I32P]),
    "mic_render": (ctypes.c_int, [_P, _P, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, _P,
                                  ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, _P, _P, _I32P]),
    "mic_render_job": (ctypes.c_int, [_P, _P, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(Job), ctypes.c_int, _P, _I32P]),
    "mic_render_batch": (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.POINTER(Job), ctypes.c_int, _P, _I32P]),
    "mic_thumbnail_size": (ctypes.c_int, [ctypes.c_int32] * 4 + [_I32P, _I32P]),
    "mic_contact_sheet_size": (ctypes.c_int, [ctypes.c_int32] * 5 + [_I32P, _I32P]),
    "mic_contact_sheet": (ctypes.c_int, [_P, _P, ctypes.c_int32, _I32P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, _P, _P]),
    "mic_host_rows_solid": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint8),
                                           ctypes.POINTER(ctypes.c_int)]),
    "mic_download": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, _P, _I32P]),
    "mic_download_wait": (ctypes.c_int, [_P, ctypes.c_int32]),
    "mic_png_info": (ctypes.c_int, [_P, ctypes.c_size_t, _I32P, _I32P]),
    "mic_png_decode": (ctypes.c_int, [_P, ctypes.c_size_t, _P, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]),
    "mic_png_decode_rows": (ctypes.c_int, [_P, ctypes.c_size_t, _P, ctypes.c_int32, ctypes.c_int32]),
    "mic_png_decode_many": (ctypes.c_int, [ctypes.c_int32, _P, _P, _P, _I32P, _I32P, ctypes.c_int, _I32P]),
    "mic_png_decode_counts": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "mic_png_bound": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "mic_png_encode": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _P,
                                      ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "mic_png_encode_rows": (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int, _P,
                                           ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "mic_png_write": (ctypes.c_int, [ctypes.c_char_p, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_int,
                                     ctypes.c_int]),
    "mic_png_write_rows": (ctypes.c_int, [ctypes.c_char_p, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int]),
    "mic_png_write_async": (ctypes.c_int, [ctypes.c_char_p, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_int64)]),
    "mic_png_wait": (ctypes.c_int, [ctypes.c_int64]),
    "mic_last_stats": (ctypes.c_int, [_P, ctypes.POINTER(Stats)]),
    "mic_profile_begin": (ctypes.c_int, [_P, ctypes.c_int]),
    "mic_profile_begin_sampled": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_int]),
    "mic_profile_end": (ctypes.c_int, [_P, _P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(ctypes.c_double)]),
}


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen libmic.so and declare every prototype.  Needs no GPU (symbols only)."""
    if path == LIB_PATH and not os.environ.get("MIC_LIB"):
        # not a fallback: the HIP library itself is (re)built when it is missing or older than its sources and
        # hipcc is at hand (build() is a no-op otherwise); without hipcc a stale library is reported, not used silently
        from . import build as _build
        try:
            if _build._stale():
                _build.build()
        except Exception as exc:  # no hipcc on this machine, or the build failed
            if os.path.exists(path):
                import warnings
                warnings.warn(f"{path} is older than its sources and could not be rebuilt ({exc}); "
                              "run `python -m image_transformation_amd.build`", RuntimeWarning)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -m image_transformation_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the compositor path.")
    # torch bundles its own libamdhip64 (same SONAME); importing it first makes libmic share that
    # runtime, so torch streams and tensors are directly usable by the library.
    import torch  # noqa: F401

    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    # the struct layouts above (Stats, LabelStrip, Job) are those of this ABI version: an older build (MIC_LIB) would
    # fill fewer Stats fields silently, a newer one could write past them
    ver = lib.mic_version()
    if (ver >> 16, ver & 0xffff) != ABI_VERSION:
        raise RuntimeError(f"{path} reports ABI {ver >> 16}.{ver & 0xffff}, this binding is written for "
                           f"{ABI_VERSION[0]}.{ABI_VERSION[1]}: rebuild it (`python -m image_transformation_amd.build`)")
    return lib


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                _lib = load_library()
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().mic_last_error()
        raise MicError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")


class Context:
    """One mic_ctx per (process, device); owns the staging ring, scratch arena and tables."""

    def __init__(self, device: int):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible to torch: the compositor path needs an MI355X "
                               "(there is no CPU fallback)")
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        h = _P()
        with torch.cuda.device(self.device):
            torch.cuda.current_stream()  # make sure torch has initialised the device first
            check(lib().mic_create(self.device, ctypes.byref(h)))
        self.handle = h

    def stream_ptr(self) -> int:
        import torch

        return int(torch.cuda.current_stream(self.torch_device).cuda_stream)

    def stats(self) -> Dict[str, int]:
        s = Stats()
        check(lib().mic_last_stats(self.handle, ctypes.byref(s)))
        return s.as_dict()

    def profile_begin(self, max_calls: int, every: int = 1) -> None:
        """Bracket every `every`-th call's kernels with HIP events (up to max_calls brackets)."""
        check(lib().mic_profile_begin_sampled(self.handle, int(max_calls), int(every)))

    def profile_end(self):
        """-> (calls, composite kernel ms summed, resample passes ms summed); syncs the stream."""
        n, c, r = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        check(lib().mic_profile_end(self.handle, _P(self.stream_ptr()), ctypes.byref(n), ctypes.byref(c),
                                    ctypes.byref(r)))
        return n.value, c.value, r.value

    def selftest(self) -> None:
        """Known-answer canary of the clip / pack helpers (mic_selftest); raises MicError on a mismatch."""
        check(lib().mic_selftest(self.handle, _P(self.stream_ptr())))

    def sync(self) -> None:
        check(lib().mic_sync(self.handle, _P(self.stream_ptr())))


_contexts: Dict[int, Context] = {}


def context(device: Optional[int] = None) -> Context:
    """The process-wide context of `device` (default: torch's current device)."""
    import torch

    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm device visible to torch: the compositor path needs an MI355X "
                               "(there is no CPU fallback)")
        device = torch.cuda.current_device()
    device = int(device)
    ctx = _contexts.get(device)
    if ctx is None:
        lib()  # (takes _lock itself)
        with _lock:  # created under the lock: two threads racing here must not make (and leak) two contexts
            ctx = _contexts.get(device)
            if ctx is None:
                ctx = _contexts[device] = Context(device)
    return ctx
