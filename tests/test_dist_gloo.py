"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py (weight broadcast from rank 0, page shards, result gather)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bb_ocr_amd  # noqa: F401
        from bb_ocr_amd import dist as bdist
        from bb_ocr_amd import weights

        state = weights.synthetic_crnn_state(7) if rank == 0 else None
        got = bdist.broadcast_state(state, src=0, device="cpu")
        ref = weights.synthetic_crnn_state(7)
        same = all(np.array_equal(got[k], v) for k, v in ref.items() if not k.endswith("num_batches_tracked"))
        a, b = bdist.shard_range(10, rank, world)
        local = [[([[0, 0], [1, 0], [1, 1], [0, 1]], f"page{g}", 0.5)] * (g % 3) for g in range(a, b)]
        allr = bdist.gather_results(local, dst=0)
        # scatter -> per-rank "OCR" -> gather, with uneven and empty shards (512 % 8 == 0 on the real node, 130 % 8 != 0)
        for n_pages in (13, 2 * world, 1, 0):
            shape = (6, 8, 3)
            pages = None
            if rank == 0:
                pages = torch.arange(n_pages * 6 * 8 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(n_pages, *shape)
            mine = bdist.scatter_pages(pages, n_pages, shape, src=0, device="cpu")
            sa, sb = bdist.shard_range(n_pages, rank, world)
            want = torch.arange(n_pages * 6 * 8 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(n_pages, *shape)[sa:sb]
            same = same and mine.shape == want.shape and torch.equal(mine, want)
            fake = [[([[0, 0], [1, 0], [1, 1], [0, 1]], f"sum{int(p.sum())}", 1.0)] for p in mine]      # stands in for readtext per page
            got_all = bdist.gather_results(fake, dst=0)
            if rank == 0:
                full = torch.arange(n_pages * 6 * 8 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).reshape(n_pages, *shape)
                same = same and [r[0][1] for r in got_all] == [f"sum{int(p.sum())}" for p in full]
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # the max-over-ranks timing reduction of bench.py
        ok = same and t.item() == world
        if rank == 0:
            ok = ok and len(allr) == 10 and all(len(p) == g % 3 for g, p in enumerate(allr)) and all(p[0][1] == f"page{g}" for g, p in enumerate(allr) if p)
        else:
            ok = ok and allr is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_broadcast_scatter_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == {r: True for r in range(world)}


class _FakeReader:
    """Stands in for bb_ocr_amd.Reader in dist.broadcast_packed (the blob protocol only; no GPU in this container)."""

    device, device_index = "cpu", 0

    def __init__(self, fill):
        self.blob = torch.full((4096,), fill, dtype=torch.uint8)
        self.closed = False

    def weights_blob_size(self):
        return self.blob.numel()

    def export_weights_blob(self):
        return self.blob.clone()

    def import_weights_blob(self, b):
        self.blob = b.clone()

    def close(self):
        self.closed = True


def _packed_worker(rank, world, port, q, fail_rank):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bb_ocr_amd import dist as bdist

        built = []

        def make(fill):
            def f():
                if rank == fail_rank:
                    raise MemoryError("injected")
                built.append(_FakeReader(fill))
                return built[-1]
            return f

        try:
            r = bdist.broadcast_packed(make(7), make(0), src=0, via_host=True)
            q.put((rank, "ok", bool((r.blob == 7).all()), False))
        except RuntimeError as e:
            q.put((rank, "raised", "construction failed" in str(e), all(b.closed for b in built)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1, 0])
def test_broadcast_packed_agrees_on_construction_before_the_first_collective(fail_rank):
    """a reader that cannot be built on ONE rank ends broadcast_packed with an exception on EVERY rank (nobody waits in the size
    all_gather), the readers that were built are closed; without a failure the receivers end up with the root's blob."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank < 0:
        assert res == {0: ("ok", True, False), 1: ("ok", True, False)}
    else:
        assert res == {0: ("raised", True, True), 1: ("raised", True, True)}
