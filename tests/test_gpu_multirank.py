"""-m gpu: the N > 1 launch of bench.py (one process per GPU, page shards, no data-path collective) rehearsed with two ranks that share the
one card of the test box (gloo rendezvous; the driver's real runs use RCCL with one card per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scatter", [False, True])
def test_bench_two_ranks_one_card(scatter):
    """Both N > 1 set-ups of bench.py: the packed weight blob exported by rank 0, broadcast and imported by rank 1 (the RCCL path's
    code, collectives over gloo with a host hop on this one-card box), every rank rendering its shard or -- ``--scatter`` -- rank 0
    rendering all pages and dist.scatter_pages sending rank 1 its block."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(29537 + int(scatter)), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--backend", "gloo", "--batch", "8",
           "--cpu-pages", "0"] + (["--scatter"] if scatter else [])
    # a child process, never an exec: this pytest process has initialised the GPU
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 1
    assert d["config"]["batch_per_gpu"] == 8 and "cpu_baseline" not in d          # the CPU leg runs at N = 1 only
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["weights"].startswith("packed device blob") and "MB" in d["config"]["weights"]       # not the local fallback
    assert d["config"]["pages"].startswith("scattered from rank 0" if scatter else "every rank renders")
    assert d["config"]["boxes_per_step_rank0"] > 100
    # who took part, readable from the JSON alone (a SCALE run is checked for N distinct cards this way; two gloo ranks share this one)
    assert d["ranks_seen"] == 2 and d["weights_broadcast_ok"] is True
    assert [x["rank"] for x in d["devices"]] == [0, 1] and all(x["pci_bus_id"] or x["uuid"] for x in d["devices"])


def test_failed_weight_broadcast_is_fatal():
    """A broadcast that fails (here: rank 1 lays its plans out for another precision, so the blob sizes disagree) must end the run non-zero
    on every rank instead of quietly benchmarking locally built weights; --allow-local-weights turns it into a reported fallback."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", BBOCR_BENCH_INJECT="precision_mismatch")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29541",
            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--batch", "2", "--cpu-pages", "0"]
    res = subprocess.run(base, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode != 0 and b"packed weight broadcast failed" in res.stderr
    assert not [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    base[base.index("29541")] = "29542"
    res = subprocess.run(base + ["--allow-local-weights"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    d = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["weights_broadcast_ok"] is False and "FAILED" in d["config"]["weights"]


def test_weight_broadcast_failing_on_one_rank_only_ends_every_rank():
    """a rank-LOCAL failure (rank 1 cannot build its reader) must not leave rank 0 inside the size all_gather: the ranks agree on
    construction before the first collective, the healthy rank's reader is closed, the run ends non-zero on both -- or, with
    --allow-local-weights, continues on locally built weights and says so."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", BBOCR_BENCH_INJECT="construct_fail_rank1")
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29545",
            os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--batch", "2", "--cpu-pages", "0"]
    res = subprocess.run(base, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert res.returncode != 0 and b"packed weight broadcast failed" in res.stderr and b"construction failed" in res.stderr
    assert not [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    base[base.index("29545")] = "29546"
    res = subprocess.run(base + ["--allow-local-weights"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    d = json.loads([l for l in res.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["weights_broadcast_ok"] is False and "FAILED" in d["config"]["weights"] and d["ranks_seen"] == 2
