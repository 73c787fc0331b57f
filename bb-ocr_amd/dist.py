"""One process per GPU: page sharding, one-time weight broadcast, result gather.

Pages are independent through every stage, so the path shards embarrassingly: rank r takes the
contiguous block ``[r*B/G, (r+1)*B/G)`` and there is NO collective on the data path.  The only
exchanges are (1) the detector + recogniser parameters broadcast once from rank 0 -- the
process-per-GPU replacement of ``torch.nn.DataParallel``'s per-forward ``broadcast_coalesced`` in
``easyocr.py::get_detector/get_recognizer`` -- and (2) the gather of the (tiny, variable-length)
results to rank 0.  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition; the first ``n_items % world`` ranks get one extra item."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_state(state: dict | None, src: int = 0, device="cpu"):
    """Broadcast a state-dict from ``src`` as ONE flat fp32 buffer (+ a small pickled key/shape table)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        items = [(k, np.ascontiguousarray(v, dtype=np.float32)) for k, v in state.items() if not k.endswith("num_batches_tracked")]
        meta[0] = [(k, a.shape) for k, a in items]
    dist.broadcast_object_list(meta, src=src)
    total = int(sum(int(np.prod(s)) if len(s) else 1 for _, s in meta[0]))
    if rank == src:
        flat = torch.from_numpy(np.concatenate([a.reshape(-1) for _, a in items])).to(device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out, o = {}, 0
    for k, s in meta[0]:
        n = int(np.prod(s)) if len(s) else 1
        out[k] = host[o:o + n].reshape(s).copy()
        o += n
    return out


def gather_results(local_results: list, dst: int = 0):
    """Per-rank list of per-page results -> on ``dst`` the concatenation in global page order, else None."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local_results, bucket, dst=dst)
    if rank != dst:
        return None
    return [page for part in bucket for page in part]
