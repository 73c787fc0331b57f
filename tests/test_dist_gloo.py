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


def test_broadcast_shard_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
