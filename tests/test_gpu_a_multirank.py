"""The N > 1 path on a device.  The driver's one-GPU box cannot give every rank a GPU of its own, so the ranks
are fresh child processes that share cuda:0 and rendezvous over gloo (127.0.0.1) -- the same batch.py /
bench.py code a real node runs over RCCL: the atlas broadcast from rank 0, v -> rank v mod G sharding of the 64
C4 variants, one launch per rank, and bench.py's max-over-ranks timing protocol.  So the first 8-GPU run is not
also the first run of that code on a device.  (Sorted before the other GPU test modules on purpose: the children
are started before this process has touched the GPU.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_render_all_64_c4_variants(tmp_path, golden_dir):
    world, port = 2, str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(r), str(world), port,
                               str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()  # exactly the processes this test started
            raise
        outs.append(out.decode("utf-8", "replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    with open(os.path.join(golden_dir, "c4_hashes.json"), encoding="utf-8") as f:
        want = {r["name"]: r["sha16"] for r in json.load(f)["cases"]}
    got = {}
    for r in range(world):
        with open(tmp_path / f"rank{r}.json", encoding="utf-8") as f:
            rec = json.load(f)
        assert any(n.startswith("libmic") for n in rec["native"]), rec  # the HIP library did the work
        assert all(int(v) % world == r for v in rec["hashes"])
        got.update(rec["hashes"])
    assert sorted(int(v) for v in got) == list(range(64))
    wrong = [v for v in got if got[v] != want[f"c4_variant_{v}"]]
    assert not wrong, f"variants differing from the reference: {wrong}"


def test_bench_protocol_with_two_ranks():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, one process per rank),
    rehearsal mode: both ranks on cuda:0 over gloo.  One JSON line from rank 0, whole-job value, per-rank spread."""
    env = dict(os.environ, MIC_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--batch", "4"]
    res = subprocess.run(cmd, env=env, capture_output=True, timeout=420, cwd=ROOT)
    text = res.stdout.decode("utf-8", "replace")
    assert res.returncode == 0, text + res.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 6 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and rec["config"]["canvases_per_step_per_gpu"] == 4
    pr = rec["per_rank"]
    assert 0 < pr["timed_region_s_min"] <= pr["timed_region_s_max"]
    assert abs(rec["ms_per_step"] - pr["timed_region_s_max"] / 6 * 1e3) < 1e-3
    assert rec["roofline"]["frac"] > 0 and rec["cpu_baseline"] is None
