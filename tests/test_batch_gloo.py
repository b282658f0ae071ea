"""N > 1 path on CPU: world_size-2 gloo processes run the sharding + atlas-blob broadcast protocol
of image_transformation_amd.batch (variant v -> rank v mod G, one broadcast of
the packed atlas per bundle, no collective on the data path).  The device half (wrapping the blob,
rendering) is covered by the -m gpu tests; here the blob stays a CPU tensor."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from image_transformation_amd import synthetic
    from image_transformation_amd.batch import broadcast_blob, shard_indices
    from image_transformation_amd.compositor import pack_blob, parse_blob_header

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        objs = synthetic.make_cutouts(5, (9, 40), (7, 30), seed=77, alpha_mode="soft")
        blob = pack_blob(objs) if rank == 0 else None
        got = broadcast_blob(blob, src=0, device=torch.device("cpu"))
        ref = pack_blob(objs)  # every rank can rebuild it from the seed: must be identical bytes
        assert got.dtype == torch.uint8 and torch.equal(got, ref)
        sizes = parse_blob_header(got.numpy())
        assert sizes == {k: (v.shape[1], v.shape[0]) for k, v in objs.items()}
        # a second bundle of a different size through the same group
        objs2 = synthetic.make_cutouts(2, (3, 5), (3, 5), seed=78)
        got2 = broadcast_blob(pack_blob(objs2) if rank == 0 else None, src=0, device=torch.device("cpu"))
        assert torch.equal(got2, pack_blob(objs2))
        # variant sharding: disjoint, complete, v mod G
        n = 13
        mine = shard_indices(n, rank, world)
        assert all(v % world == rank for v in mine)
        flags = torch.zeros(n, dtype=torch.int32)
        flags[mine] = 1
        dist.all_reduce(flags)
        assert flags.tolist() == [1] * n
        # per-rank "work done" reduces like bench.py's MAX-over-ranks timing
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _worker8(rank, world, port, tmpdir):
    """One of 8 CPU ranks: bench.py's own N > 1 protocol objects (Dist: barrier, max / min, sum, gather) and the C4
    partition, over gloo -- no GPU is touched (the pool allows at most six processes on a card, so the 8-rank case
    cannot be rehearsed on the one-GPU box)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import bench
    from image_transformation_amd import synthetic

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = bench.Dist(world, rank, torch.device("cpu"), rehearsal=True)
        assert D.active and D.backend == "gloo"
        _, variants = synthetic.c4_workload("binary", seed=4, n_variants=64) if rank == 0 else (None, None)
        sizes = [None]
        if rank == 0:
            sizes = [[list(v[0]) for v in variants]]
        dist.broadcast_object_list(sizes, src=0)  # (only rank 0 pays for generating the cutouts)
        part = bench.c4_partition(64, world, [tuple(x) for x in sizes[0]])
        mine = part[rank]
        assert len(mine["variants"]) == 8 and all(v % world == rank for v in mine["variants"])
        classes = [(2160, 3840), (2880, 2880), (3840, 2160), (4399, 1885)]
        assert mine["canvas_sizes"] == [classes[rank % 4]]  # ONE class per rank at G = 8
        D.barrier()
        # timing protocol: value = units of ALL ranks / MAX over ranks of the timed region
        px = float(sum(sizes[0][v][0] * sizes[0][v][1] for v in mine["variants"]))
        total = D.sum(px)
        assert total == float(sum(w * h for w, h in sizes[0]))
        mx, mn = D.max_min(0.001 * (rank + 1))
        assert abs(mx - 0.008) < 1e-12 and abs(mn - 0.001) < 1e-12
        got = D.gather({"rank": rank, "n": len(mine["variants"])})
        assert [g["rank"] for g in got] == list(range(world)) and sum(g["n"] for g in got) == 64
        with open(os.path.join(tmpdir, f"ok{rank}"), "w") as f:
            f.write("ok")
    finally:
        dist.destroy_process_group()


def test_c4_partition_and_protocol_8_ranks(tmp_path):
    import torch.multiprocessing as mp

    world, port = 8, _free_port()
    mp.spawn(_worker8, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(8)]


def test_shard_indices_edges():
    from image_transformation_amd.batch import shard_indices
    assert shard_indices(0, 0, 4) == []
    assert shard_indices(3, 3, 4) == []
    assert shard_indices(64, 5, 8) == list(range(5, 64, 8))
    got = sorted(v for r in range(8) for v in shard_indices(64, r, 8))
    assert got == list(range(64))
    with pytest.raises(ValueError):
        shard_indices(4, 4, 4)


def test_pack_blob_layout_guard_bands():
    """Every cutout sits 256-byte aligned between >= 16 readable guard bytes (the composite kernel's
    edge loads rely on it)."""
    sys.path.insert(0, ROOT)
    from image_transformation_amd import synthetic
    from image_transformation_amd.compositor import pack_blob, parse_blob_header
    objs = synthetic.make_cutouts(4, (1, 64), (1, 64), seed=5, alpha_mode="soft")
    blob = pack_blob(objs).numpy()
    assert bytes(blob[:4]) == b"MICA"
    n = int(np.frombuffer(blob[8:12].tobytes(), np.uint32)[0])
    tab = np.frombuffer(blob[32:32 + 32 * n].tobytes(), np.int32).reshape(n, 8)
    offs = np.frombuffer(blob[32:32 + 32 * n].tobytes(), np.uint64).reshape(n, 4)[:, 2]
    ends = []
    for (oid, w, h, *_), off in zip(tab, offs):
        off = int(off)
        assert off % 256 == 0 and off >= 32 + 32 * n + 16
        px = blob[off:off + w * h * 4].reshape(h, w, 4)
        assert np.array_equal(px, objs[int(oid)])
        ends.append((off, off + w * h * 4))
    ends.sort()
    for (a0, a1), (b0, b1) in zip(ends, ends[1:]):
        assert b0 - a1 >= 16
    assert len(blob) - ends[-1][1] >= 16
    assert parse_blob_header(blob) == {k: (v.shape[1], v.shape[0]) for k, v in objs.items()}
