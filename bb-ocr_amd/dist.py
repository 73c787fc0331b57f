"""One process per GPU: page sharding, one-time weight broadcast, page scatter, result gather.

Pages are independent through every stage, so the path shards embarrassingly: rank r takes the
contiguous block ``[r*B/G, (r+1)*B/G)`` and there is NO collective inside the OCR path.  The exchanges are

1. the detector + recogniser parameters, broadcast ONCE from rank 0 -- the process-per-GPU replacement of
   ``torch.nn.DataParallel``'s per-forward ``broadcast_coalesced`` in ``easyocr.py::get_detector/get_recognizer``.
   ``broadcast_packed`` ships the weights as the kernels read them (BN folded, rounded to the element type, MFMA fragment order:
   one contiguous DEVICE blob, ~49 MB in bf16) device-to-device: rank 0 ``bbocr_weights_export`` -> ``ncclBroadcast`` over xGMI ->
   ``bbocr_weights_import`` on the others, no host hop, no re-packing.  ``broadcast_state`` (fp32 state-dict through the host) stays
   for CPU-side consumers such as the oracle;
2. ``scatter_pages``: the loader rank holds the decoded pages and sends every rank its block -- grouped point-to-point sends
   (``ncclGroupStart; ncclSend x (G-1); ncclGroupEnd``: all 7 xGMI links of the root in use at once), blocks may be uneven;
   when every rank can decode / render its own shard this step disappears (bench.py does that);
3. ``gather_results``: the (tiny, variable-length) per-page results to rank 0.

Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests (tests/test_dist_gloo.py).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block partition; the first ``n_items % world`` ranks get one extra item."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def broadcast_state(state: dict | None, src: int = 0, device="cpu"):
    """Broadcast a state-dict from ``src`` as ONE flat fp32 buffer (+ a small pickled key/shape table) -> numpy state-dict on every rank."""
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


def broadcast_packed(make_root_reader, make_empty_reader, src: int = 0, via_host: bool = False):
    """-> this rank's ``Reader`` with the packed weights of rank ``src``.

    ``make_root_reader()`` builds the source reader from real weights (called on ``src`` only); ``make_empty_reader()`` builds a
    ``Reader(weights="empty", precision=<same>)`` on the other ranks.  The blob travels device-to-device (``via_host=True`` only for the
    gloo rehearsal, where the collective runs on CPU tensors)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    # (1) construction is rank-local and may fail on ONE rank (out of memory, a missing checkpoint): agree on it before the first
    # collective that assumes a reader everywhere -- otherwise the healthy ranks would sit in all_gather for ever
    reader, err = None, None
    try:
        reader = make_root_reader() if rank == src else make_empty_reader()
    except Exception as e:          # noqa: BLE001 -- reported below, on every rank
        err = e
    dev = "cpu" if (via_host or reader is None) else reader.device
    if not via_host and reader is None:
        dev = f"cuda:{torch.cuda.current_device()}"
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if reader is not None:
            reader.close()
        raise RuntimeError(f"reader construction failed on {'this rank: ' + repr(err) if err else 'another rank'}")
    try:
        size = torch.tensor([reader.weights_blob_size()], dtype=torch.int64)
        sizes = [torch.zeros_like(size) for _ in range(dist.get_world_size())]
        if via_host:
            dist.all_gather(sizes, size)
        else:
            size = size.to(reader.device)
            sizes = [s.to(reader.device) for s in sizes]
            dist.all_gather(sizes, size)
        if len({int(s.item()) for s in sizes}) != 1:
            raise RuntimeError("ranks disagree on the weight blob size: same precision and networks on every rank?")
        if rank == src:
            blob = reader.export_weights_blob()
        else:
            blob = torch.empty(reader.weights_blob_size(), dtype=torch.uint8, device=reader.device)
        if via_host:
            hb = blob.cpu()
            dist.broadcast(hb, src=src)
            if rank != src:
                blob.copy_(hb)
                if blob.is_cuda:
                    torch.cuda.current_stream(reader.device_index).synchronize()
        else:
            dist.broadcast(blob, src=src)                # ncclBroadcast, device to device over xGMI
            torch.cuda.current_stream(reader.device_index).synchronize()
        if rank != src:
            reader.import_weights_blob(blob)
    except Exception:
        reader.close()              # the caller gets an exception, never a half-built reader: no second context / arena stays on the card
        raise
    return reader


def scatter_pages(pages, n_pages: int, page_shape, src: int = 0, device="cpu"):
    """The loader rank ``src`` holds ``pages`` (uint8 tensor ``[n_pages, *page_shape]``, ``None`` elsewhere); every rank returns its
    contiguous block ``shard_range(n_pages, rank, world)`` as a tensor on ``device``.  Blocks may be uneven (``n_pages % world != 0``) or
    empty.  One grouped batch of point-to-point operations: the root's sends to all peers run concurrently."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    a, b = shard_range(n_pages, rank, world)
    if rank == src:
        if pages is None or tuple(pages.shape) != (n_pages, *page_shape) or pages.dtype != torch.uint8:
            raise ValueError(f"root must hold uint8 pages [{n_pages}, {page_shape}]")
        ops, keep = [], []
        for r in range(world):
            if r == src:
                continue
            ra, rb = shard_range(n_pages, r, world)
            if rb > ra:
                block = pages[ra:rb].contiguous()
                keep.append(block)
                ops.append(dist.P2POp(dist.isend, block, r))
        local = pages[a:b].to(device).clone()
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return local
    local = torch.empty((b - a, *page_shape), dtype=torch.uint8, device=device)
    if b > a:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, local, src)]):
            req.wait()
    return local


def gather_results(local_results: list, dst: int = 0):
    """Per-rank list of per-page results -> on ``dst`` the concatenation in global page order, else None."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local_results, bucket, dst=dst)
    if rank != dst:
        return None
    return [page for part in bucket for page in part]
