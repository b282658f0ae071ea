"""MI355X compositor behind the reference's compositor.py call surface.

    composite(background_img, object_images, placements) -> Image      compositor.py:6-22
    load_object_images(results_json_path) -> {int id: RGBA Image}      compositor.py:25-35
    render(layout_json, objects, canvas) -> Image                      (north_star; = place + clamp +
                                                                        composite, macro_placement_test.py:1495-1511)

plus the device-resident forms a batch caller uses (Atlas, render_batch).  The pixel work --
Pillow-exact LANCZOS resampling, ordered 8-bit alpha-over, solid background synthesis -- runs in
the HIP kernels of libmic.so; PIL images are only the handoff type the reference's callers expect
(they .save() the result, macro_placement_test.py:1513).  Nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
from PIL import Image

import threading

from . import _native, _pilmem
from ._native import LANCZOS, BILINEAR, Placement, Job  # noqa: F401
from . import flex

_P = ctypes.c_void_p
_FILTERS = {LANCZOS, BILINEAR}


# ------------------------------------------------------------------------------------------ helpers
def _torch():
    import torch

    return torch


class _DecodeCache:
    """Decoded RGBA images by (path, mtime, size).  The reference decodes every cutout for the contact
    sheet and again for every iteration's load_object_images (macro_placement_test.py:1414, 1493,
    1679); PNG decode is most of the host time left around the path, so a file that has not changed
    is decoded once per process.  Entries are returned as copies (callers may mutate PIL images)."""
    _items: "Dict[Tuple[str, int, int], Image.Image]" = {}
    _bytes = 0
    _lock = threading.Lock()  # the reference's Streamlit app runs every session on its own thread
    LIMIT = 512 << 20

    @staticmethod
    def _decode_files(paths: Sequence[str]) -> List[Image.Image]:
        """Image.open(p).convert("RGBA") for every path: libmic's PNG reader (csrc/png_decode.cpp, several files at a
        time on its own threads, straight into the images' memory) for the kinds of PNG it takes; any file it declines --
        another format, 16-bit, interlaced, anything irregular -- goes to Pillow, which decodes it or raises."""
        from . import png as mic_png
        blobs = []
        for p in paths:
            with open(p, "rb") as f:  # FileNotFoundError like Image.open
                blobs.append(f.read())
        ims = mic_png.decode_many(blobs)
        for i, im in enumerate(ims):
            if im is None:
                ims[i] = Image.open(paths[i]).convert("RGBA")
        return ims

    @classmethod
    def _fetch(cls, paths: Sequence[Any]) -> List[Image.Image]:
        """The cached decodes of `paths` (decoding the missing ones together)."""
        keys = []
        for path in paths:
            p = os.fspath(path)
            st = os.stat(p)  # raises FileNotFoundError like Image.open
            keys.append((os.path.abspath(p), st.st_mtime_ns, st.st_size))
        with cls._lock:
            ims = [cls._items.get(k) for k in keys]
        missing = [i for i, im in enumerate(ims) if im is None]
        if missing:
            fresh = cls._decode_files([keys[i][0] for i in missing])
            with cls._lock:
                for i, im in zip(missing, fresh):
                    nbytes = im.size[0] * im.size[1] * 4
                    if cls._bytes + nbytes > cls.LIMIT:
                        cls._items.clear()
                        cls._bytes = 0
                    cls._items[keys[i]] = im
                    cls._bytes += nbytes
                    ims[i] = im
        return ims

    @classmethod
    def open_many(cls, paths: Sequence[Any], shared: bool = False) -> List[Image.Image]:
        return [cls._hand_out(im, shared) for im in cls._fetch(paths)]

    @classmethod
    def open_rgba(cls, path, shared: bool = False) -> Image.Image:
        return cls._hand_out(cls._fetch([path])[0], shared)

    @staticmethod
    def _hand_out(im: Image.Image, shared: bool) -> Image.Image:
        if shared:
            # a second Python object over the SAME pixel memory, flagged read-only: Pillow's in-place operations
            # (putpixel, paste, alpha_composite, ImageDraw, putalpha ...) copy the pixels before they write, so the
            # cache stays intact; only a write through load()'s pixel access raises.  For callers that never
            # modify the cutouts (this package's own pipeline): no 0.5 MB copy per cutout and call.
            try:
                view = im._new(im.im)
                view.readonly = 1
                return view
            except Exception:  # noqa: BLE001  (a Pillow without _new: the plain copy)
                pass
        return im.copy()

    @classmethod
    def pristine(cls, path) -> Optional[Image.Image]:
        """The cached decode itself (never handed to callers): what a private copy is compared with."""
        p = os.fspath(path)
        try:
            st = os.stat(p)
        except OSError:
            return None
        with cls._lock:
            return cls._items.get((os.path.abspath(p), st.st_mtime_ns, st.st_size))

    @classmethod
    def size(cls, path) -> Tuple[int, int]:
        p = os.fspath(path)
        st = os.stat(p)
        key = (os.path.abspath(p), st.st_mtime_ns, st.st_size)
        with cls._lock:
            im = cls._items.get(key)
        return im.size if im is not None else cls.open_rgba(path).size


def open_rgba(path, shared: bool = False) -> Image.Image:
    """Image.open(path).convert("RGBA") through the per-process decode cache (shared=True: see _DecodeCache)."""
    return _DecodeCache.open_rgba(path, shared)


def rgba_size(path) -> Tuple[int, int]:
    """Image.open(path).convert("RGBA").size: decodes the file (and fails on a broken one) like the reference does,
    but only the first time a given version of the file is seen -- later calls read the decode cache."""
    return _DecodeCache.size(path)


def _image_to_array(img: Image.Image) -> np.ndarray:
    """RGBA PIL image -> (H, W, 4) uint8 (a copy of the raw bytes; no arithmetic)."""
    if img.mode != "RGBA":
        raise ValueError("image has wrong mode")  # what Pillow's core.alpha_composite raises
    return np.asarray(img, dtype=np.uint8)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(ctx: _native.Context):
    """libmic sets the HIP device of the calling thread to its context's; keep torch's notion of the
    current device intact when that is another one (one process driving several GPUs), and skip the
    several-microsecond guard in the usual one-process-per-GPU case."""
    torch = _torch()
    return _NO_GUARD if torch.cuda.current_device() == ctx.device else torch.cuda.device(ctx.torch_device)


def _pinned(nbytes: int):
    """A pinned host buffer of its own for one transfer.  torch's caching host allocator hands a freed
    block back without a new hipHostMalloc (microseconds after the first call of a size) and keeps a
    block alive until the copies enqueued on it have run, so concurrent callers never share a buffer --
    the per-process pair of staging buffers this replaces was not safe for two threads."""
    return _torch().empty(max(int(nbytes), 1), dtype=_torch().uint8, pin_memory=True)


def _upload(arr, ctx: _native.Context):
    """Host RGBA pixels -> device (H, W, 4) uint8 tensor through a pinned buffer.  `arr` is a PIL RGBA image
    (its rows are memmove'd out of Pillow's own memory: no Image.tobytes() pass) or an (H, W, 4) array."""
    torch = _torch()
    if isinstance(arr, Image.Image):
        if arr.mode != "RGBA":
            raise ValueError("image has wrong mode")  # what Pillow's core.alpha_composite raises
        W, H = arr.size
        pin = _pinned(H * W * 4)
        shape = (H, W, 4)
        if not _pilmem.copy_to(arr, pin.data_ptr()):
            pin.numpy()[:] = np.asarray(arr, dtype=np.uint8).reshape(-1)
    else:
        arr = np.ascontiguousarray(arr)
        pin = _pinned(arr.size)
        pin.numpy()[:arr.size] = arr.reshape(-1)
        shape = arr.shape
    n = shape[0] * shape[1] * 4 if len(shape) == 3 else int(np.prod(shape))
    dev = torch.empty(shape, dtype=torch.uint8, device=ctx.torch_device)
    dev.view(-1).copy_(pin[:n], non_blocking=True)  # the allocator keeps `pin` alive until the copy has run
    return dev


_RESULT_COPY = os.environ.get("MIC_RESULT_COPY") == "1"


def _to_pil(canvas_dev, view: bool = False) -> Image.Image:
    """Device (H, W, 4) uint8 -> a PIL RGBA image over a pinned host buffer of its own: the download lands in the
    very memory the image shows (Image.frombuffer), there is no second host copy.  The image is an ordinary mutable
    one, like the reference's result: Pillow flags frombuffer images read-only to protect a buffer it does not own,
    but this buffer belongs to nobody else, so the flag is cleared -- px = im.load(); px[x, y] = v, paste, ImageDraw
    all write into the image's own memory.  The page-locked block (torch's caching host allocator rounds it up to a
    power of two) lives as long as the image; a caller that keeps many results alive can trade the copy back with
    MIC_RESULT_COPY=1 (rows memmove'd into an ordinary Image.new, the pinned block returns to the allocator at once).
    view=True (the package's save-only callers) keeps Pillow's read-only flag."""
    torch = _torch()
    h, w = int(canvas_dev.shape[0]), int(canvas_dev.shape[1])
    pin = _pinned(h * w * 4)
    pin.copy_(canvas_dev.reshape(-1), non_blocking=True)
    torch.cuda.current_stream(canvas_dev.device).synchronize()
    if _RESULT_COPY and not view:
        im = Image.new("RGBA", (w, h), None)  # (colour None: Pillow leaves the pixels uninitialised)
        if _pilmem.copy_from(im, pin.data_ptr()):
            return im
        return Image.frombuffer("RGBA", (w, h), pin.numpy(), "raw", "RGBA", 0, 1).copy()
    im = Image.frombuffer("RGBA", (w, h), pin.numpy(), "raw", "RGBA", 0, 1)
    if not view:
        im.readonly = 0
    return im


class _Entry:
    """What flex needs from an object: `.size` (as a PIL image has)."""
    __slots__ = ("size",)

    def __init__(self, size):
        self.size = size


def pack_blob(objects: Mapping[int, Any], pin: bool = False):
    """{id: RGBA image/array} -> host uint8 tensor in libmic's atlas blob layout (header, table,
    256-byte aligned pixels).  Needs no GPU; this is what rank 0 broadcasts to the other GPUs."""
    lib = _native.lib()
    ids: List[int] = []
    arrs: List[np.ndarray] = []
    for oid, im in objects.items():
        if isinstance(im, Image.Image):
            if im.mode != "RGBA":
                raise ValueError("image has wrong mode")
            arr = im  # copied row by row straight out of Pillow's memory below
        else:
            arr = np.ascontiguousarray(im, np.uint8)
            if arr.ndim != 3 or arr.shape[2] != 4:
                raise ValueError("image has wrong mode")
        ids.append(int(oid))
        arrs.append(arr)
    n = len(ids)
    ids_a = np.asarray(ids, np.int32)
    ws = np.asarray([a.size[0] if isinstance(a, Image.Image) else a.shape[1] for a in arrs], np.int32)
    hs = np.asarray([a.size[1] if isinstance(a, Image.Image) else a.shape[0] for a in arrs], np.int32)
    nbytes = ctypes.c_size_t()
    _native.check(lib.mic_atlas_blob_size(n, _i32p(ws), _i32p(hs), ctypes.byref(nbytes)))
    torch = _torch()
    pinned = bool(pin) and torch.cuda.is_available()
    if pinned:
        # straight into a pinned buffer of the call's own (no second host copy before the upload); a pinned block
        # comes back from the allocator with old contents, so the gaps between the images are cleared by hand
        host = _pinned(nbytes.value)
        host_np = host.numpy()
    else:
        # numpy, not torch.zeros: a torch CPU fill fans out over every core of the host (256 on the GPU
        # boxes) and the thread wake-up alone cost ~40 ms per atlas
        host_np = np.zeros(nbytes.value, np.uint8)
        host = torch.from_numpy(host_np)
    offs = np.zeros(max(n, 1), np.uint64)
    head = 32 + 32 * n
    if pinned:
        host_np[:min(head, nbytes.value)] = 0
    _native.check(lib.mic_atlas_blob_layout(n, _i32p(ids_a), _i32p(ws), _i32p(hs), _P(host_np.ctypes.data),
                                            nbytes.value, offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    end = head  # one past the last byte written so far
    for a, off, w, h in zip(arrs, offs, ws, hs):
        off, size = int(off), int(w) * int(h) * 4
        if pinned and off > end:
            host_np[end:off] = 0
        if isinstance(a, Image.Image):
            if not _pilmem.copy_to(a, host_np.ctypes.data + off):
                host_np[off:off + size] = np.asarray(a, dtype=np.uint8).reshape(-1)
        else:
            host_np[off:off + size] = a.reshape(-1)
        end = max(end, off + size)
    if pinned and nbytes.value > end:
        host_np[end:] = 0
    return host


def parse_blob_header(raw: np.ndarray) -> Dict[int, Tuple[int, int]]:
    """Blob bytes (at least header + table) -> {id: (w, h)}; first occurrence of an id wins."""
    raw = np.ascontiguousarray(raw, np.uint8)
    head = np.frombuffer(raw[:32].tobytes(), np.uint32)
    if head[0] != 0x4143494D:
        raise ValueError("not an atlas blob")
    n = int(head[2])
    tab = np.frombuffer(raw[32:32 + 32 * n].tobytes(), np.int32).reshape(n, 8) if n else np.zeros((0, 8), np.int32)
    sizes: Dict[int, Tuple[int, int]] = {}
    for row in tab:
        sizes.setdefault(int(row[0]), (int(row[1]), int(row[2])))
    return sizes


class Atlas(Mapping):
    """Device-resident packed cutouts: the on-GPU form of load_object_images()'s dict.

    Uploaded once per bundle and reused by every composite / refine iteration / variant
    (the reference re-decodes the PNGs each iteration, macro_placement_test.py:1493,1679).
    Behaves as a read-only mapping id -> entry with `.size`, which is all the layout code
    needs (macro_placement_test.py:645,708 only read img.size).

    The blob is IMMUTABLE while the Atlas exists (also one made by from_blob over a caller's tensor):
    libmic keeps data derived from it -- planar copies of cutouts that get resampled, resampled layers
    in plans and in the context's layer cache -- and cannot see the bytes being rewritten in place.  To
    show other pixels, build a new Atlas (and new CompositeBatch plans) over them.
    """

    def __init__(self, objects: Mapping[int, Any], device: Optional[int] = None, ctx: Optional[_native.Context] = None):
        """ctx: a context other than the process-wide one of the device (tests and tuning scripts make their own,
        with other MIC_* settings read by mic_create)."""
        self.ctx = ctx if ctx is not None else _native.context(device)
        host = pack_blob(objects, pin=True)  # packed straight into pinned memory: one host copy, then the DMA
        torch = _torch()
        blob = torch.empty(host.numel(), dtype=torch.uint8, device=self.ctx.torch_device)
        blob.copy_(host, non_blocking=True)  # (torch's host allocator keeps `host` alive until the copy has run)
        # the upload rides on torch's CURRENT stream; a later call may come on another one (a non-blocking side
        # stream, a worker thread's stream): it waits for this event first (wait_ready)
        self._ready = torch.cuda.Event()
        self._ready.record()
        self._ready_stream = torch.cuda.current_stream(self.ctx.torch_device).cuda_stream
        self._init_from_blob(blob, header=host.numpy()[:32 + 32 * len(objects)])

    def _init_from_blob(self, blob, header: Optional[np.ndarray] = None):
        self.blob = blob  # torch uint8 tensor on the device; owns the memory
        if header is None:
            n = int(np.frombuffer(blob[:32].cpu().numpy().tobytes(), np.uint32)[2])
            header = blob[:32 + 32 * n].cpu().numpy()
        self._sizes = parse_blob_header(header)
        h = _P()
        _native.check(_native.lib().mic_atlas_from_device_blob(
            self.ctx.handle, _P(blob.data_ptr()), blob.numel(), _P(header.ctypes.data), ctypes.byref(h)))
        self.handle = h

    _ready = None
    _ready_stream = None

    def wait_ready(self) -> None:
        """Order torch's current stream behind the atlas upload (a no-op once the upload has completed, or when
        the call comes on the stream that carried it)."""
        ev = self._ready
        if ev is None:
            return
        torch = _torch()
        cur = torch.cuda.current_stream(self.ctx.torch_device)
        if cur.cuda_stream != self._ready_stream:
            if ev.query():
                self._ready = None
            else:
                cur.wait_event(ev)
        # same stream: in order by construction (the flag stays: a later call may come on another stream)

    @classmethod
    def from_blob(cls, blob, device: Optional[int] = None) -> "Atlas":
        """Wrap a device blob (e.g. the result of a torch.distributed broadcast).  The tensor must not be written
        again while this Atlas (or a plan built on it) is in use: see the class docstring."""
        self = cls.__new__(cls)
        self.ctx = _native.context(device if device is not None else blob.device.index)
        self._init_from_blob(blob)
        return self

    # Mapping protocol (id -> entry with .size)
    def __getitem__(self, oid):
        return _Entry(self._sizes[oid])

    def __iter__(self):
        return iter(self._sizes)

    def __len__(self):
        return len(self._sizes)

    @property
    def nbytes(self) -> int:
        return int(self.blob.numel())

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _native.lib().mic_atlas_destroy(h)
            except Exception:
                pass
            self.handle = None


class _AtlasCache:
    """Device atlases of bundles loaded from files, by the files' (path, mtime, size) keys: the reference
    decodes a bundle's cutouts for the contact sheet and again in every iteration's load_object_images
    (macro_placement_test.py:1414, 1493, 1679) -- here all of those share ONE upload for as long as the
    files do not change.  A handful of bundles is kept (least recently used goes first)."""
    _items: "Dict[Any, Atlas]" = {}
    _lock = threading.Lock()
    LIMIT = 4

    @classmethod
    def get(cls, key, device, build):
        k = (key, device)
        with cls._lock:
            a = cls._items.pop(k, None)
            if a is not None:
                cls._items[k] = a  # most recently used last
                return a
        a = build()
        with cls._lock:
            cls._items[k] = a
            while len(cls._items) > cls.LIMIT:
                cls._items.pop(next(iter(cls._items)))
        return a


class ObjectImages(dict):
    """dict {id: RGBA Image} as load_object_images returns it, plus a lazily uploaded Atlas that
    is dropped whenever the dict is modified."""

    _atlas: Optional[Atlas] = None
    # set by load_object_images: identifies the files the dict was decoded from.  A dict shares the process-wide device
    # atlas of its files only while its images provably show the files' pixels: READ-ONLY views (shared=True) are
    # checked on every use (Pillow's copy-on-write -- putalpha, paste, ImageDraw ... -- replaces im.im and clears
    # im.readonly: _views_intact); private copies (the default, mutable like the reference's) are compared byte for byte
    # with the decode cache once, when the dict first needs an atlas (_copies_pristine).  Anything else uploads an
    # atlas built from the dict's own images.
    _source_key = None
    _cores = None
    _paths = None  # private copies (the default load): {id: file path}, to compare with the decode cache on first use

    def _views_intact(self) -> bool:
        cores = self._cores
        if not cores or len(cores) != len(self):
            return False
        for k, im in self.items():
            if not getattr(im, "readonly", 0) or getattr(im, "im", None) is not cores.get(k):
                return False
        return True

    def _copies_pristine(self) -> bool:
        """Private copies (the default load_object_images()): do they still hold exactly the files' pixels?  One memcmp
        per cutout against the decode cache, done ONCE, when the dict first needs an atlas: the reference reloads the
        bundle every iteration (macro_placement_test.py:1493, 1679) and every such dict can then share the ONE device
        atlas of the files instead of uploading its own; a copy that was edited in place before first use fails the
        comparison and the dict uploads what it holds."""
        paths = self._paths
        if not paths or len(paths) != len(self):
            return False
        for k, im in self.items():
            ref = _DecodeCache.pristine(paths.get(k, "")) if k in paths else None
            if ref is None or not _pilmem.same_pixels(im, ref):
                return False
        return True

    def atlas(self, device: Optional[int] = None) -> Atlas:
        if self._atlas is not None and self._cores is not None and not self._views_intact():
            self._touch()  # a view was written to since the atlas was taken from the shared cache
        if self._atlas is None or (device is not None and self._atlas.ctx.device != device):
            shared_ok = self._source_key is not None and (self._views_intact() if self._cores is not None
                                                          else self._copies_pristine())
            if shared_ok:
                dev = _native.context(device).device
                self._atlas = _AtlasCache.get(self._source_key, dev, lambda: Atlas(self, device))
            else:
                self._source_key = None
                self._atlas = Atlas(self, device)
            self._paths = None  # (compared once; later in-place edits are announced with invalidate())
        return self._atlas

    def invalidate(self) -> None:
        """Forget the uploaded atlas: call after editing a cutout's pixels IN PLACE (putalpha, paste, ImageDraw on an
        image of this dict) once the dict has been used -- the dict sees its own item assignments, not writes
        into the images it holds.  (Re-assigning the image, objects[k] = im, has the same effect.)"""
        self._touch()

    _native_table = None  # (ids, widths, heights) arrays cached by flex.native_boxes

    def _touch(self):
        self._atlas = None
        self._native_table = None
        self._source_key = None  # no longer what the files hold
        self._cores = None
        self._paths = None

    def __setitem__(self, k, v):
        self._touch()
        super().__setitem__(k, v)

    def __delitem__(self, k):
        self._touch()
        super().__delitem__(k)

    def update(self, *a, **kw):
        self._touch()
        super().update(*a, **kw)

    def pop(self, *a):
        self._touch()
        return super().pop(*a)

    def clear(self):
        self._touch()
        super().clear()

    def setdefault(self, *a):
        self._touch()
        return super().setdefault(*a)

    def popitem(self):
        self._touch()
        return super().popitem()

    def __ior__(self, other):
        self._touch()
        return super().__ior__(other)


def _i32p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _as_atlas(objects, device: Optional[int] = None) -> Atlas:
    if isinstance(objects, Atlas):
        return objects
    if isinstance(objects, ObjectImages):
        return objects.atlas(device)
    return Atlas(objects, device)


def load_object_images(results_json_path: str, shared: bool = False) -> Dict[int, Image.Image]:
    """results.json -> {object_id: RGBA Image} (compositor.py:25-35).  PNG decode stays on the
    host; the returned dict uploads itself to the GPU once, on first use.  shared=True (this package's own
    pipeline, which never modifies a cutout) hands out copy-on-write views of the decode cache instead of copies."""
    with open(results_json_path, "r", encoding="utf-8") as f:
        items = json.load(f)
    base = os.path.dirname(results_json_path)
    out = ObjectImages()
    keys = []
    # The reference opens the entries one after another (compositor.py:31-34): the FIRST entry that is malformed, names
    # a missing file or holds a broken image is the error the caller hears about.  Here the files not seen before are
    # decoded together, so the entries are walked first (id, file name, the file's existence) up to the first failure,
    # the ones before it are decoded (an earlier broken image raises there), and only then is that failure raised.
    oids, paths, failure = [], [], None
    for it in items:
        try:
            oid = int(it["object_id"])
            path = os.path.join(base, it["filename"])
            os.stat(path)
        except Exception as exc:  # noqa: BLE001  (re-raised below, in the reference's order)
            failure = exc
            break
        oids.append(oid)
        paths.append(path)
    images = _DecodeCache.open_many(paths, shared)
    if failure is not None:
        raise failure
    for oid, path, im in zip(oids, paths, images):
        out[oid] = im
        st = os.stat(path)
        keys.append((oid, os.path.abspath(path), st.st_mtime_ns, st.st_size))
    out._source_key = tuple(keys)  # (after the inserts above, which reset it)
    if shared:
        out._cores = {k: getattr(im, "im", None) for k, im in out.items()}
        if not out._views_intact():  # a Pillow without copy-on-write views: treat them as private copies
            out._cores = None
    if out._cores is None:
        out._paths = {k[0]: k[1] for k in keys}
    return out


def coerce_placements(known_ids, placements: Iterable[Mapping]) -> List[Tuple[int, int, int, int, int]]:
    """The Python-level semantics of compositor.py:12-16: object_id through int() unless already
    an int, unknown ids skipped, box = exactly four values each through int() (truncation)."""
    rows = []
    for p in placements:
        oid = p["object_id"]
        if not isinstance(oid, int):
            oid = int(oid)
        if oid not in known_ids:
            continue
        x1, y1, x2, y2 = [int(v) for v in p["box"]]
        rows.append((oid, x1, y1, x2, y2))
    return rows


_I32_MIN, _I32_MAX = -(2 ** 31), 2 ** 31 - 1


_PLACEMENT_DTYPE = np.dtype([("atlas", np.int32), ("object_id", np.int32), ("box", np.int32, (4,))])
assert _PLACEMENT_DTYPE.itemsize == ctypes.sizeof(Placement)


def _fill_placements(rows: Sequence[Tuple[int, int, int, int, int]], atlas_index: int = 0):
    """Coerced rows -> an array of mic_placement (a ctypes array for a handful of rows, a NumPy structured array of
    the same layout for long lists: 512 placements of a 16-canvas batch are one vectorised conversion instead of 512
    trips through ctypes attribute setters)."""
    n = len(rows)
    if n >= 16:
        try:
            a = np.asarray(rows, dtype=np.int64)
        except OverflowError:
            raise OverflowError("placement coordinates do not fit 32 bits") from None
        if a.shape != (n, 5):
            raise ValueError("placement rows are (object_id, x1, y1, x2, y2)")
        if (a < _I32_MIN).any() or (a > _I32_MAX).any():
            raise OverflowError("placement coordinates do not fit 32 bits")
        arr = np.empty(n, _PLACEMENT_DTYPE)
        arr["atlas"] = atlas_index
        arr["object_id"] = a[:, 0]
        arr["box"] = a[:, 1:5]
        return arr
    arr = (Placement * max(len(rows), 1))()
    for i, (oid, x1, y1, x2, y2) in enumerate(rows):
        if not (_I32_MIN <= x1 <= _I32_MAX and _I32_MIN <= y1 <= _I32_MAX and
                _I32_MIN <= x2 <= _I32_MAX and _I32_MIN <= y2 <= _I32_MAX and _I32_MIN <= oid <= _I32_MAX):
            raise OverflowError("placement coordinates do not fit 32 bits")
        p = arr[i]
        p.atlas = atlas_index
        p.object_id = oid
        p.box[0], p.box[1], p.box[2], p.box[3] = x1, y1, x2, y2
    return arr


class SolidCanvas:
    """A canvas that is one colour: what fill_solid() produces (background_resizing.py:25-33).
    Passing it to render() lets the kernel synthesise the background instead of reading it.

    The colour is either known on the host (`rgba`) or lives in device memory (`colour_dev`: a 4-byte uint8 tensor
    r, g, b, a that a kernel enqueued earlier on the stream writes -- the median kernel's result, solid_canvas()): the
    composite reads it when it runs, so background synthesis -> render never waits for the GPU.  Reading `.rgba` of
    such a canvas downloads the colour (one stream wait), once."""

    def __init__(self, size: Tuple[int, int], rgba: Optional[Sequence[int]] = None, *, colour_dev=None):
        self.size = (int(size[0]), int(size[1]))
        self.colour_dev = colour_dev
        self._rgba = None
        if rgba is not None:
            rgba = tuple(int(v) for v in rgba)
            if len(rgba) == 3:
                rgba = rgba + (255,)
            if len(rgba) != 4 or any(not 0 <= v <= 255 for v in rgba):
                raise ValueError("colour must be 3 or 4 values in 0..255")
            self._rgba = rgba
        elif colour_dev is None:
            raise ValueError("a SolidCanvas needs a colour: rgba, or colour_dev (4 bytes on the device)")
        elif colour_dev.numel() != 4 or colour_dev.element_size() != 1 or not colour_dev.is_contiguous():
            raise ValueError("colour_dev must be 4 contiguous bytes r, g, b, a on the device")

    @property
    def rgba(self) -> Tuple[int, int, int, int]:
        if self._rgba is None:
            self._rgba = tuple(int(v) for v in self.colour_dev.cpu().tolist())  # (waits for the kernel that writes it)
        return self._rgba

    def _job_colour(self, job) -> None:
        """Fill a ctypes Job's background fields."""
        job.bg_dev = None
        if self._rgba is not None or self.colour_dev is None:
            r, c = self.rgba, job.bg_rgba
            c[0], c[1], c[2], c[3] = r[0], r[1], r[2], r[3]
            job.bg_rgba_dev = None
        else:
            job.bg_rgba_dev = self.colour_dev.data_ptr()

    def to_image(self) -> Image.Image:
        return Image.new("RGBA", self.size, self.rgba)


def _make_job(size, bg_dev_ptr, bg_rgba, placement_arr, n_place, out_ptr) -> Job:
    j = Job()
    j.width, j.height = size
    if isinstance(bg_rgba, SolidCanvas):
        bg_rgba._job_colour(j)
    else:
        j.bg_dev = bg_dev_ptr
        for k in range(4):
            j.bg_rgba[k] = bg_rgba[k]
    j.n_placements = n_place
    j.placements = ctypes.cast(placement_arr.ctypes.data if isinstance(placement_arr, np.ndarray) else placement_arr,
                               ctypes.POINTER(Placement))
    j.out_dev = out_ptr
    return j


def _build_jobs(atlas: "Atlas", canvases, placement_rows, atlas_of: Optional[Sequence[int]] = None):
    """-> (ctypes Job array with null outputs, [(W, H)...], objects to keep alive while it is used).
    atlas_of[i]: index (into the call's atlas array) of the atlas canvas i's placements refer to."""
    torch = _torch()
    n = len(canvases)
    jobs = (Job * max(n, 1))()
    sizes: List[Tuple[int, int]] = []
    keep: List[Any] = []
    for i, (cv, rows) in enumerate(zip(canvases, placement_rows)):
        if isinstance(cv, SolidCanvas):
            W, H = cv.size
            bg_ptr, rgba = None, cv  # (the colour, host or device, is filled in by the canvas itself)
            keep.append(cv)
        else:
            if cv.dtype != torch.uint8 or cv.dim() != 3 or cv.shape[2] != 4 or not cv.is_contiguous():
                raise ValueError("device canvas must be a contiguous uint8 (H, W, 4) tensor")
            if cv.device != atlas.ctx.torch_device:
                raise ValueError("canvas lives on another device than the atlas")
            H, W = int(cv.shape[0]), int(cv.shape[1])
            bg_ptr, rgba = cv.data_ptr(), (0, 0, 0, 0)
            keep.append(cv)
        parr = _fill_placements(rows, atlas_of[i] if atlas_of is not None else 0)
        jobs[i] = _make_job((W, H), bg_ptr, rgba, parr, len(rows), None)
        keep.append(parr)
        sizes.append((W, H))
    return jobs, sizes, keep


class CompositeBatch:
    """A batch of composite jobs resolved once into a persistent libmic plan (mic_plan_create):
    device layer records, resample tables and scratch live with the plan.  run() re-executes all
    the pixel work for the batch with ONE mic_plan_run call -- a small job-table upload plus the
    launches -- so a caller that re-composites the same variants (refine iterations, a timed
    loop) pays neither per-placement Python cost nor per-call table building.

    canvases[i] is a SolidCanvas or a contiguous torch uint8 (H, W, 4) tensor on the atlas'
    device; placement_rows[i] = [(object_id, x1, y1, x2, y2), ...] already coerced."""

    def __init__(self, atlas: Union[Atlas, Sequence[Atlas]], canvases: Sequence[Union[SolidCanvas, Any]],
                 placement_rows: Sequence[Sequence[Tuple[int, int, int, int, int]]], filter: int = LANCZOS,
                 atlas_of: Optional[Sequence[int]] = None):
        """atlas: one Atlas, or several (variants of different bundles in one launch) with
        atlas_of[i] = which of them canvas i's object ids refer to."""
        if filter not in _FILTERS:
            raise ValueError(f"unknown filter {filter}")
        torch = _torch()
        atlases = [atlas] if isinstance(atlas, Atlas) else list(atlas)
        if not atlases or any(a.ctx is not atlases[0].ctx for a in atlases):
            raise ValueError("the atlases of a batch must live in one context")
        self.atlas = atlases[0]
        self.atlases = atlases
        self.ctx = atlases[0].ctx
        self.filter = filter
        self.n = len(canvases)
        if len(placement_rows) != self.n or (atlas_of is not None and len(atlas_of) != self.n):
            raise ValueError("one placement list (and atlas index) per canvas")
        if atlas_of is not None and any(not 0 <= int(k) < len(atlases) for k in atlas_of):
            raise ValueError("atlas index out of range")
        jobs, self.sizes, keep = _build_jobs(atlases[0], canvases, placement_rows, atlas_of)
        self._keep: List[Any] = atlases + keep
        atl = (_P * len(atlases))(*[a.handle for a in atlases])
        h = _P()
        for a in atlases:
            a.wait_ready()
        with torch.cuda.device(self.ctx.torch_device):
            _native.check(_native.lib().mic_plan_create(self.ctx.handle, len(atlases), atl, self.n, jobs, filter,
                                                        ctypes.byref(h)))
        self.handle = h
        self._outs = (_P * max(self.n, 1))()
        self._shapes = [(H, W, 4) for (W, H) in self.sizes]

    def alloc_outputs(self):
        torch = _torch()
        return [torch.empty(shape, dtype=torch.uint8, device=self.ctx.torch_device) for shape in self._shapes]

    def stats(self) -> Dict[str, int]:
        st = _native.Stats()
        _native.check(_native.lib().mic_plan_stats(self.handle, ctypes.byref(st)))
        return st.as_dict()

    def invalidate(self) -> None:
        """The next run() resamples the plan's layers again (they are otherwise kept from the first run on)."""
        _native.check(_native.lib().mic_plan_invalidate(self.handle))

    def run(self, outs: Optional[Sequence[Any]] = None, check: bool = True):
        """Enqueue the batch on torch's current stream (not synchronised); returns the outputs.
        check=False skips the per-tensor shape/dtype validation (hot loops over known buffers)."""
        if outs is None:
            outs = self.alloc_outputs()
        if len(outs) != self.n:
            raise ValueError("one output canvas per job")
        arr = self._outs
        if check:
            torch = _torch()
            for a in self.atlases:
                a.wait_ready()
            for i, out in enumerate(outs):
                if tuple(out.shape) != self._shapes[i] or out.dtype != torch.uint8 or not out.is_contiguous() \
                        or out.device != self.ctx.torch_device:
                    raise ValueError("output canvas has the wrong shape/dtype/device")
                arr[i] = out.data_ptr()
        else:
            for i, out in enumerate(outs):
                arr[i] = out.data_ptr()
        _native.check(_native.lib().mic_plan_run(self.handle, arr, _P(self.ctx.stream_ptr())))
        return list(outs)

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                _native.lib().mic_plan_destroy(h)
            except Exception:
                pass
            self.handle = None


def composite_device(atlas: Atlas, canvases: Sequence[Union[SolidCanvas, Any]],
                     placement_rows: Sequence[Sequence[Tuple[int, int, int, int, int]]],
                     outs: Optional[Sequence[Any]] = None, filter: int = LANCZOS):
    """One-shot batch composite, everything device-resident: one mic_composite_batch call (tables go
    through the context's staging ring, resampled layers through its arena -- no allocation, no
    synchronisation).  Returns the list of output canvases (torch uint8 (H, W, 4)); work is enqueued
    on torch's current stream.  Callers that re-run the same batch should keep a CompositeBatch."""
    if filter not in _FILTERS:
        raise ValueError(f"unknown filter {filter}")
    if len(placement_rows) != len(canvases):
        raise ValueError("one placement list per canvas")
    torch = _torch()
    ctx = atlas.ctx
    jobs, sizes, keep = _build_jobs(atlas, canvases, placement_rows)
    if outs is None:
        outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=ctx.torch_device) for (W, H) in sizes]
    if len(outs) != len(sizes):
        raise ValueError("one output canvas per job")
    for i, out in enumerate(outs):
        W, H = sizes[i]
        if tuple(out.shape) != (H, W, 4) or out.dtype != torch.uint8 or not out.is_contiguous() \
                or out.device != ctx.torch_device:
            raise ValueError("output canvas has the wrong shape/dtype/device")
        jobs[i].out_dev = out.data_ptr()
    atl = (_P * 1)(atlas.handle)
    atlas.wait_ready()
    with _device_guard(ctx):
        _native.check(_native.lib().mic_composite_batch(ctx.handle, 1, atl, len(sizes), jobs, filter,
                                                        _P(ctx.stream_ptr())))
    del keep
    return list(outs)


_tls = threading.local()
_raw_stream = None
_DIRECT_HOST_MAX = int(os.environ.get("MIC_DIRECT_HOST_MAX", str(2 << 20)))  # bytes; 0: always device canvas + copy


def _stream_of(ctx: _native.Context) -> int:
    """torch's current stream on the context's device as a raw hipStream_t (the private accessor when this torch has
    it: a tenth of the cost of torch.cuda.current_stream().cuda_stream)."""
    global _raw_stream
    if _raw_stream is None:
        fn = getattr(_torch()._C, "_cuda_getCurrentRawStream", None)
        _raw_stream = fn if fn is not None else False
    return int(_raw_stream(ctx.device)) if _raw_stream else ctx.stream_ptr()


class _PendingDownload:
    __slots__ = ("ctx", "ticket", "pin", "size", "waited")

    def __init__(self, ctx, ticket, pin, size):
        self.ctx, self.ticket, self.pin, self.size, self.waited = ctx, ticket, pin, size, False

    def __del__(self):
        # dropped without image() (an exception between enqueue and wait): the writer of `pin` must have finished
        # before the block returns to the allocator
        if not self.waited:
            try:
                _native.lib().mic_download_wait(self.ctx.handle, self.ticket)
            except Exception:  # noqa: BLE001  (interpreter shutdown)
                pass

    def image(self) -> Image.Image:
        _native.check(_native.lib().mic_download_wait(self.ctx.handle, self.ticket))
        self.waited = True
        w, h = self.size
        if _RESULT_COPY:
            im = Image.new("RGBA", (w, h), None)
            if _pilmem.copy_from(im, self.pin.data_ptr()):
                return im
            return Image.frombuffer("RGBA", (w, h), self.pin.numpy(), "raw", "RGBA", 0, 1).copy()
        im = Image.frombuffer("RGBA", (w, h), self.pin.numpy(), "raw", "RGBA", 0, 1)
        im.readonly = 0  # (see _to_pil: the buffer is the image's own)
        return im


def _composite_one(atlas: Atlas, canvas, rows, filter: int, download: bool = True):
    """ONE canvas through mic_composite_batch with everything the call needs kept per thread between calls: the
    ctypes job / placement arrays, the device output canvas of that size (the result leaves as a PIL image, the device
    copy is scratch).  download=True: the device -> pinned-host copy is enqueued behind the kernel and a
    _PendingDownload is returned (its .image() waits for that copy's own event, not for the stream); otherwise the
    device tensor.  The reference-sized call (492 x 492, 4 cutouts) spends more time in Python than on the GPU:
    profiles/r03_c1_breakdown.json."""
    torch = _torch()
    ctx = atlas.ctx
    n = len(rows)
    cache = _tls.__dict__.setdefault("one", {})
    cap = 8
    while cap < n:
        cap *= 2
    ent = cache.get(cap)
    if ent is None:
        parr = (Placement * cap)()
        jobs = (Job * 1)()
        jobs[0].placements = ctypes.cast(parr, ctypes.POINTER(Placement))
        ent = cache[cap] = (jobs, parr, (_P * 1)())
    jobs, parr, atl = ent
    for i, (oid, x1, y1, x2, y2) in enumerate(rows):
        if not (_I32_MIN <= x1 <= _I32_MAX and _I32_MIN <= y1 <= _I32_MAX and
                _I32_MIN <= x2 <= _I32_MAX and _I32_MIN <= y2 <= _I32_MAX and _I32_MIN <= oid <= _I32_MAX):
            raise OverflowError("placement coordinates do not fit 32 bits")
        p = parr[i]
        p.atlas = 0
        p.object_id = oid
        b = p.box
        b[0], b[1], b[2], b[3] = x1, y1, x2, y2
    j = jobs[0]
    keep = None
    if isinstance(canvas, SolidCanvas):
        W, H = canvas.size
        canvas._job_colour(j)
        keep = canvas
    else:
        if canvas.dtype != torch.uint8 or canvas.dim() != 3 or canvas.shape[2] != 4 or not canvas.is_contiguous():
            raise ValueError("device canvas must be a contiguous uint8 (H, W, 4) tensor")
        if canvas.device != ctx.torch_device:
            raise ValueError("canvas lives on another device than the atlas")
        H, W = int(canvas.shape[0]), int(canvas.shape[1])
        j.bg_dev = canvas.data_ptr()
        j.bg_rgba_dev = None
        keep = canvas
    j.width, j.height, j.n_placements = W, H, n
    nbytes = H * W * 4
    # A small canvas that only leaves as a PIL image is written by the kernel STRAIGHT into the pinned host buffer the
    # image will show (page-locked memory is device-visible at the same address): the copy engine's start-up costs
    # more than the 0.97 MB of a 492 x 492 canvas takes over PCIe.  Big canvases keep the device canvas + DMA copy.
    direct = download and nbytes <= _DIRECT_HOST_MAX
    pin = _pinned(nbytes) if download else None
    out = None
    if not direct:
        outs = _tls.__dict__.setdefault("outs", {})
        key = (ctx.device, W, H)
        out = outs.get(key) if download else None
        if out is None:
            out = torch.empty((H, W, 4), dtype=torch.uint8, device=ctx.torch_device)
            if download:
                if len(outs) >= 4:
                    outs.pop(next(iter(outs)))
                outs[key] = out
    j.out_dev = pin.data_ptr() if direct else out.data_ptr()
    atl[0] = atlas.handle
    atlas.wait_ready()
    stream = _stream_of(ctx)
    lib = _native.lib()
    with _device_guard(ctx):
        _native.check(lib.mic_composite_batch(ctx.handle, 1, atl, 1, jobs, filter, _P(stream)))
        del keep
        if not download:
            return out
        ticket = ctypes.c_int32()
        _native.check(lib.mic_download(ctx.handle, None if direct else _P(out.data_ptr()), _P(pin.data_ptr()),
                                       0 if direct else nbytes, _P(stream), ctypes.byref(ticket)))
    return _PendingDownload(ctx, ticket.value, pin, (W, H))


def _rows_solid(table: int, W: int, H: int, rgba) -> bool:
    """Every pixel of the host image behind Pillow's row table == rgba?  (mic_host_rows_solid; big images on the
    4 copy threads: the C call releases the GIL)"""
    lib = _native.lib()
    col = (ctypes.c_uint8 * 4)(*rgba)

    def part(y0, y1):
        ok = ctypes.c_int()
        _native.check(lib.mic_host_rows_solid(_P(table), W, y0, y1, col, ctypes.byref(ok)))
        return ok.value == 1

    if W * H * 4 < (4 << 20):
        return part(0, H)
    cuts = [H * k // 4 for k in range(5)]
    return all(_pilmem._workers().map(lambda k: part(cuts[k], cuts[k + 1]), range(4)))


def _composite_pil_background(atlas: Atlas, background_img: Image.Image, rows, filter: int) -> Image.Image:
    """composite() / render() onto a PIL background.  The pipeline's backgrounds are fill_solid() canvases re-opened
    from canvas.png (macro_placement_test.py:1510): one colour, which the kernel synthesises instead of reading an
    uploaded image.  Whether the image IS one colour takes a full scan (exactness) -- as long as the GPU work itself
    at the reference's size -- so the call speculates: when the first and the last pixel agree, the composite over
    that colour and its download are enqueued FIRST and the scan runs while the GPU works; the rare image that then
    turns out not to be solid is uploaded and composited again."""
    tab = _pilmem.row_table(background_img)
    if tab is not None:
        table, W, H = tab
        first = background_img.getpixel((0, 0))
        if background_img.getpixel((W - 1, H - 1)) == first and background_img.getpixel((W // 2, H // 2)) == first:
            pending = _composite_one(atlas, SolidCanvas((W, H), first), rows, filter)
            solid = False
            try:
                solid = _rows_solid(table, W, H, first)
            finally:
                # whatever happens in the scan: the kernel (or copy) that writes into pending's pinned block has
                # finished before the block can go back to torch's host allocator (it knows nothing of that kernel)
                if not solid:
                    _native.lib().mic_download_wait(pending.ctx.handle, pending.ticket)
            if solid:
                return pending.image()
    return _composite_one(atlas, _upload(background_img, atlas.ctx), rows, filter).image()


def composite(background_img: Image.Image, object_images: Mapping[int, Image.Image],
              placements: List[Dict], *, filter: int = LANCZOS) -> Image.Image:
    """Drop-in for the reference's composite() (compositor.py:6-22).

    placements: list of {object_id, box: [x1, y1, x2, y2]}; extra keys are ignored.  Painter's
    order = list order; unknown ids are skipped; the background image is not modified."""
    rows = coerce_placements(object_images, placements)
    if not rows:
        return background_img.copy()  # compositor.py:11 with an empty loop
    if background_img.mode != "RGBA":
        raise ValueError("image has wrong mode")  # what Pillow's core.alpha_composite raises
    if filter not in _FILTERS:
        raise ValueError(f"unknown filter {filter}")
    return _composite_pil_background(_as_atlas(object_images), background_img, rows, filter)


def _layout_rows(layout_json: Any, images: Mapping[int, Any], atlas: "Atlas", size: Tuple[int, int]):
    """layout (Flex tree as dict or JSON text, {"placements": [...]}, or a list) -> coerced rows
    [(object_id, x1, y1, x2, y2)] for ids the atlas knows.  Flex trees go through the native placer
    (mic_flex_place) when it can mirror them, otherwise through flex.py -- same boxes either way,
    and flex.py is what raises the reference's errors for malformed trees."""
    native = flex.native_boxes(layout_json, images, size)
    if native is not None:
        return [r for r in native if r[0] in atlas]
    if isinstance(layout_json, (str, bytes)):
        layout_json = json.loads(layout_json)
    return coerce_placements(atlas, flex.layout_to_placements(layout_json, images, size))


def _render_native(layout_json: Any, atlas: "Atlas", canvas: Any, size: Tuple[int, int], filter: int):
    """mic_render: Flex JSON text -> boxes -> composite inside libmic.  None when the layout is not a
    Flex tree or the native placer leaves it to flex.py (which also raises the reference's errors)."""
    if isinstance(layout_json, (bytes, str)):
        text = layout_json.encode("utf-8") if isinstance(layout_json, str) else layout_json
    elif isinstance(layout_json, dict) and "root" in layout_json:
        try:
            text = json.dumps(layout_json, separators=(",", ":")).encode("utf-8")
        except (TypeError, ValueError):
            return None
    else:
        return None
    if filter not in _FILTERS:
        raise ValueError(f"unknown filter {filter}")
    torch = _torch()
    ctx = atlas.ctx
    W, H = size
    job = _tls.__dict__.get("render_job")
    if job is None:
        job = _tls.render_job = Job()
    job.width, job.height, job.n_placements = W, H, 0
    if isinstance(canvas, SolidCanvas):
        canvas._job_colour(job)
    else:
        if canvas.dtype != torch.uint8 or canvas.dim() != 3 or canvas.shape[2] != 4 or not canvas.is_contiguous():
            raise ValueError("device canvas must be a contiguous uint8 (H, W, 4) tensor")
        if canvas.device != ctx.torch_device:
            raise ValueError("canvas lives on another device than the atlas")
        job.bg_dev, job.bg_rgba_dev = canvas.data_ptr(), None
    out = torch.empty((H, W, 4), dtype=torch.uint8, device=ctx.torch_device)
    job.out_dev = out.data_ptr()
    atlas.wait_ready()
    with _device_guard(ctx):
        rc = _native.lib().mic_render_job(ctx.handle, atlas.handle, text, len(text), ctypes.byref(job), filter,
                                          _P(ctx.stream_ptr()), None)
    if rc in (_native.ERR_UNSUPPORTED, _native.ERR_FORMAT):
        return None
    _native.check(rc)
    return out


def _canvas_size(canvas) -> Tuple[int, int]:
    if isinstance(canvas, (SolidCanvas, Image.Image)):
        return canvas.size
    return int(canvas.shape[1]), int(canvas.shape[0])  # tensor (H, W, 4)


def render(layout_json: Any, objects: Mapping[int, Any], canvas: Any, *, filter: int = LANCZOS,
           as_tensor: bool = False):
    """render(layout_json, objects, canvas): Flex-DSL layout -> boxes -> composite.

    Equals composite(canvas, objects, clamp(place(layout_json))) of the reference exactly
    (macro_placement_test.py:1495-1498 + :1511).  layout_json: {"root": flex tree} (placed from
    (0,0) over the whole canvas, then clamped), or {"placements": [...]}/a list (used as-is).
    objects: the dict from load_object_images, any {id: RGBA Image}, or an Atlas.
    canvas: an RGBA PIL image, a SolidCanvas, or a device uint8 (H, W, 4) tensor."""
    atlas = _as_atlas(objects)
    size = _canvas_size(canvas)
    if not isinstance(canvas, Image.Image):
        out = _render_native(layout_json, atlas, canvas, size, filter)  # place + composite in one ABI call
        if out is not None:
            return out if as_tensor else _to_pil(out)
    rows = _layout_rows(layout_json, objects if not isinstance(objects, Atlas) else atlas, atlas, size)
    if isinstance(canvas, Image.Image):
        if not rows and not as_tensor:
            return canvas.copy()
        if canvas.mode != "RGBA":
            raise ValueError("image has wrong mode")
        if filter not in _FILTERS:
            raise ValueError(f"unknown filter {filter}")
        if not as_tensor:
            return _composite_pil_background(atlas, canvas, rows, filter)
        solid = _pilmem.solid_colour(canvas)
        canvas = SolidCanvas(canvas.size, solid) if solid is not None else _upload(canvas, atlas.ctx)
    out = composite_device(atlas, [canvas], [rows], filter=filter)[0]
    return out if as_tensor else _to_pil(out)


def render_batch(layouts: Sequence[Any], objects: Mapping[int, Any], canvases: Sequence[Any], *,
                 filter: int = LANCZOS, outs: Optional[Sequence[Any]] = None):
    """One launch for a batch of variants of one bundle (BASELINE.json configs[3]): layouts[i] onto
    canvases[i].  Returns device tensors; nothing is copied to the host."""
    atlas = _as_atlas(objects)
    canvases = list(canvases)
    if len(layouts) != len(canvases):
        raise ValueError("one layout per canvas")
    got = _render_batch_native(layouts, atlas, canvases, outs, filter)
    if got is not None:
        return got
    rows = [_layout_rows(layout, atlas, atlas, _canvas_size(cv)) for layout, cv in zip(layouts, canvases)]
    return composite_device(atlas, canvases, rows, outs=outs, filter=filter)


def _render_batch_native(layouts, atlas: "Atlas", canvases, outs, filter: int):
    """mic_render_batch: every Flex tree (JSON text, or a {"root": ...} dict serialised here) placed and composited
    inside libmic -- no placement passes through Python objects.  None: some layout is not a Flex tree, or the native
    placer leaves one of them to flex.py (the caller then takes the per-layout path, which also raises the reference's
    errors for malformed trees)."""
    n = len(layouts)
    if n == 0:
        return None
    texts = []
    for lay in layouts:
        if isinstance(lay, bytes):
            texts.append(lay)
        elif isinstance(lay, str):
            texts.append(lay.encode("utf-8"))
        elif isinstance(lay, dict) and "root" in lay:
            try:
                texts.append(json.dumps(lay, separators=(",", ":")).encode("utf-8"))
            except (TypeError, ValueError):
                return None
        else:
            return None
    if filter not in _FILTERS:
        raise ValueError(f"unknown filter {filter}")
    torch = _torch()
    ctx = atlas.ctx
    jobs = (Job * n)()
    keep = []
    sizes = []
    for i, cv in enumerate(canvases):
        j = jobs[i]
        if isinstance(cv, SolidCanvas):
            W, H = cv.size
            cv._job_colour(j)
        else:
            if cv.dtype != torch.uint8 or cv.dim() != 3 or cv.shape[2] != 4 or not cv.is_contiguous():
                raise ValueError("device canvas must be a contiguous uint8 (H, W, 4) tensor")
            if cv.device != ctx.torch_device:
                raise ValueError("canvas lives on another device than the atlas")
            H, W = int(cv.shape[0]), int(cv.shape[1])
            j.bg_dev, j.bg_rgba_dev = cv.data_ptr(), None
        keep.append(cv)
        j.width, j.height, j.n_placements = W, H, 0
        sizes.append((W, H))
    if outs is None:
        outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=ctx.torch_device) for (W, H) in sizes]
    if len(outs) != n:
        raise ValueError("one output canvas per job")
    for i, out in enumerate(outs):
        W, H = sizes[i]
        if tuple(out.shape) != (H, W, 4) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != ctx.torch_device:
            raise ValueError("output canvas has the wrong shape/dtype/device")
        jobs[i].out_dev = out.data_ptr()
    tarr = (ctypes.c_char_p * n)(*texts)
    lens = (ctypes.c_size_t * n)(*[len(t) for t in texts])
    atlas.wait_ready()
    with _device_guard(ctx):
        rc = _native.lib().mic_render_batch(ctx.handle, atlas.handle, n, tarr, lens, jobs, filter, _P(ctx.stream_ptr()), None)
    del keep
    if rc in (_native.ERR_UNSUPPORTED, _native.ERR_FORMAT):
        return None
    _native.check(rc)
    return list(outs)
