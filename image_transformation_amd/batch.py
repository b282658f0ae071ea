"""One-image-per-GPU sharding of a batch of aspect-ratio variants (BASELINE.json configs[3]).

Nothing in the reference is distributed; this is the build-side design of
section 8(e): every variant (its own canvas size + placements) is an independent image, the only
shared datum is the read-only object atlas.  One process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI):

  * variant v of the batch goes to rank v mod world_size -- no data-path collective;
  * the atlas blob is broadcast ONCE per bundle from rank 0 and stays resident on every GPU for
    all later batches / refine iterations.

broadcast_blob works on any backend/device (the CPU tests drive it over gloo).
"""
from __future__ import annotations

from typing import Any, List, Mapping, Optional, Sequence


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Indices of the variants rank `rank` renders: v -> GPU v mod G."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    return list(range(rank, n_items, world_size))


def broadcast_blob(blob, src: int = 0, device=None, group=None):
    """Broadcast a 1-D uint8 tensor whose length only `src` knows.  Non-source ranks pass None
    (or anything) and get a freshly allocated tensor on `device`.  Two collectives: the byte
    count, then the payload."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    if rank == src:
        if blob is None or blob.dtype != torch.uint8 or blob.dim() != 1:
            raise ValueError("source rank must pass a 1-D uint8 tensor")
        device = blob.device
    elif device is None:
        raise ValueError("non-source ranks must say which device receives the blob")
    n = torch.tensor([blob.numel() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    if rank != src:
        blob = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(blob, src=src, group=group)
    return blob


def broadcast_atlas(objects: Optional[Mapping[int, Any]], src: int = 0, group=None, force: bool = False):
    """Rank `src` packs + uploads the cutouts, every rank ends up with a resident Atlas.
    With a single process (no process group, or a group of one unless force=True) this is just Atlas(objects);
    force=True sends the blob through the collective even in a group of one (bench.py's one-rank RCCL rehearsal)."""
    import torch
    import torch.distributed as dist

    from . import _native
    from .compositor import Atlas, pack_blob

    ctx = _native.context()
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return Atlas(objects, ctx.device)
    rank = dist.get_rank(group)
    blob = None
    if rank == src:
        blob = pack_blob(objects, pin=True).to(ctx.torch_device)
    blob = broadcast_blob(blob, src=src, device=ctx.torch_device, group=group)
    torch.cuda.current_stream(ctx.torch_device).synchronize()
    return Atlas.from_blob(blob, ctx.device)


def render_variants(variants: Sequence[Any], atlas, rank: int = 0, world_size: int = 1, filter: int = 0):
    """Render this rank's share of `variants` = [((W, H) or canvas, layout_json), ...] in one launch.
    Returns {variant index: device tensor}."""
    from .compositor import SolidCanvas, render_batch
    from .synthetic import SOLID_BG

    mine = shard_indices(len(variants), rank, world_size)
    canvases, layouts = [], []
    for v in mine:
        cv, layout = variants[v]
        canvases.append(cv if not isinstance(cv, tuple) else SolidCanvas(cv, SOLID_BG))
        layouts.append(layout)
    outs = render_batch(layouts, atlas, canvases, filter=filter) if mine else []
    return dict(zip(mine, outs))
