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


def test_rccl_entry_points_of_the_c_abi_with_a_one_rank_communicator(tmp_path):
    """include/bbocr.h bbocr_dist_*: the C-side RCCL path (dlopen'd librccl, ncclCommInitRank, all-gather of sizes, broadcast, grouped
    send / recv, result gather) on the one card of this box: a communicator of ONE rank (RCCL refuses two ranks on one device).  Run in a
    child process: RCCL initialises its own state and this pytest process already holds torch's.  The weight blob survives the broadcast,
    the scatter hands the rank its whole block, the gathered bytes unpack to the results that went in."""
    script = tmp_path / "one_rank.py"
    script.write_text('''
import ctypes as C, sys
sys.path.insert(0, %r)
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import _lib, synth, weights
lib = _lib.load()
r = bb_ocr_amd.Reader(["en"], gpu=True, weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)), precision="fp16")
img = synth.page(5, width=320, height=192, lines=3, margin=24)[0]
want = r.readtext(img)
uid = (C.c_char * 128)()
assert lib.bbocr_dist_unique_id(uid) == 0
r._check(lib.bbocr_dist_init(r._h, 0, 1, uid))
assert lib.bbocr_dist_init(r._h, 0, 1, uid) == -4                     # one communicator per context
r._check(lib.bbocr_bcast_weights(r._h, 0))
assert r.readtext(img) == want                                       # the weights are what they were
pages = torch.from_numpy(np.stack([img, img[::-1].copy(), img[:, ::-1].copy()])).cuda()
local = torch.zeros_like(pages)
first, count = C.c_longlong(-1), C.c_longlong(-1)
r._check(lib.bbocr_scatter_images(r._h, C.c_void_p(pages.data_ptr()), 3, img.size, 0, C.c_void_p(local.data_ptr()), C.byref(first), C.byref(count)))
assert (first.value, count.value) == (0, 3) and torch.equal(local, pages)
res = C.POINTER(_lib.bbocr_result)()
p = _lib.bbocr_params(); lib.bbocr_default_params(C.byref(p))
r._check(lib.bbocr_readtext_batch(r._h, C.c_void_p(local.data_ptr()), None, 3, img.shape[0], img.shape[1], C.byref(p), C.byref(res)))
blob, n = C.c_void_p(), C.c_size_t()
assert lib.bbocr_result_pack(res, C.byref(blob), C.byref(n)) == 0
allp, sizes = C.c_void_p(), (C.c_size_t * 1)()
r._check(lib.bbocr_gather_results(r._h, blob, n.value, 0, C.byref(allp), sizes))
assert sizes[0] == n.value and C.string_at(allp.value, n.value) == C.string_at(blob.value, n.value)
back = C.POINTER(_lib.bbocr_result)()
assert lib.bbocr_result_unpack(allp, sizes[0], C.byref(back)) == 0
assert r._collect(back)[0] == want and len(r._collect(res)) == 3
lib.bbocr_free_bytes(blob); lib.bbocr_free_bytes(allp)
r._check(lib.bbocr_dist_finalize(r._h))
r.close()
print("ONE_RANK_OK")
''' % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0 and b"ONE_RANK_OK" in res.stdout, res.stderr.decode()[-3000:]
