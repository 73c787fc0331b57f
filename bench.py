#!/usr/bin/env python
"""Headline benchmark: book-page images/s, end-to-end detect+recognise @1280x960 (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (CRAFT detect -> boxes -> crops -> CRNN -> CTC) over one batch of
synthetic 1280x960 pages per GPU, pages already resident in HBM.  Workload = BASELINE.json configs[2]
("Full CRAFT+CRNN detect+recognize, batch=64 @1280x960, 1 MI355X"); with N GPUs every rank processes its own
64-page shard (weak scaling, no data-path collective; weights are broadcast once from rank 0 over RCCL).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     -- the dominant kernel (conv_mfma, detector launches): algorithmic FLOPs / sum of launch durations
                  measured with HIP events on the library's stream inside the timed region; peak = 2.5 PFLOP/s dense bf16.
  cpu_baseline -- the CPU oracle (restatement of the easyocr algorithm, kind "port") timed on this host, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
CRAFT_GFLOP_PER_PAGE = 874.22  # SURVEY.md section 8d, 1280x960 page


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="pages per GPU per step")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic pages rendered per rank (tiled to --batch)")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=960)
    ap.add_argument("--lines", type=int, default=20)
    ap.add_argument("--cpu-pages", type=int, default=5, help="pages for the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--det-sub-batch", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU rehearsals)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)     # rehearsal: several ranks may share one card
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import bb_ocr_amd
    from bb_ocr_amd import dist as bdist
    from bb_ocr_amd import synth, weights

    # ---- weights: built on rank 0, broadcast once (RCCL) -- the process-per-GPU form of DataParallel's replicate
    cs = rs = None
    if rank == 0:
        cs, rs = weights.designed_craft_state(0), weights.synthetic_crnn_state(0)
    if world > 1:
        bdev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"
        cs = bdist.broadcast_state(cs, 0, device=bdev)
        rs = bdist.broadcast_state(rs, 0, device=bdev)
    reader = bb_ocr_amd.Reader(["en"], gpu=True, weights=(cs, rs), device_index=local_rank, det_sub_batch=args.det_sub_batch)

    # ---- this rank's shard of the global batch (contiguous block), rendered on the host, then resident in HBM
    B = args.batch
    g0, g1 = bdist.shard_range(B * world, rank, world)
    uniq = [synth.page(1234 + (g0 + i), width=args.width, height=args.height, lines=args.lines)[0] for i in range(min(args.unique, B))]
    host = np.stack([uniq[i % len(uniq)] for i in range(g1 - g0)])
    rgb = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def step():
        return reader.readtext_device(rgb, None)

    log(f"pages resident: {tuple(rgb.shape)}; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        out = step()
        log(f"warm-up step done: {reader.stage_times()}")
    reader.set_profiling(os.environ.get("BBOCR_BENCH_NOPROF") != "1")   # NOPROF: A/B of the event-recording overhead only
    if os.environ.get("BBOCR_BENCH_NOFREEZE") != "1":
        bb_ocr_amd.freeze_gc()   # host-process hygiene of a long-running OCR worker (see freeze_gc.__doc__): full GC passes stop re-walking torch
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage = {}
    for _ in range(args.steps):
        out = step()
        for k, v in reader.stage_times().items():
            stage[k] = stage.get(k, 0.0) + v
        log("timed step done")
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    conv_ms, conv_flops, conv_launches = reader.conv_profile(0)
    n_boxes = sum(len(p) for p in out)
    n_chars = sum(len(t) for p in out for _, t, _ in p)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    pages = B * world * args.steps
    result = {
        "metric": "book-page images/sec end-to-end (detect+recognize) @1280x960",
        "value": pages / dt,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": f"synthetic ({len(uniq)} distinct seeded pages per GPU tiled to the batch; seeded designed-detector + random-recogniser weights)",
        "config": {
            "workload": "full CRAFT+CRNN detect+recognize, batch=64 @1280x960 per GPU (BASELINE.json configs[2])",
            "batch_per_gpu": B, "page_wh": [args.width, args.height], "text_lines_per_page": args.lines,
            "boxes_per_step_rank0": n_boxes, "chars_per_step_rank0": n_chars, "parallelism": f"dp{world} (page shards, no data-path collective)",
        },
        "stage_ms_per_step_rank0": {k: v / args.steps for k, v in stage.items()},
    }
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of the
    # same detector, reads = 2 x FETCH_SIZE KiB on gfx950), scaled by the pages the average launch of THIS run processed
    traffic, traffic_src = None, None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm.json")
    if os.path.exists(pmc) and conv_launches > 0 and (args.width, args.height) == (1280, 960):
        with open(pmc) as f:
            t = json.load(f)
        traffic = (t["read_MB_per_page"] + t["write_MB_per_page"]) * 1e6 * B * args.steps / conv_launches
        traffic_src = f"profiles/r01_pmc_hbm.json ({t['read_MB_per_page']:.0f} MB read + {t['write_MB_per_page']:.0f} MB written per page; bytes per average launch)"
    result["roofline"] = {
        "kernel": "conv3x3_dma_kernel + conv1x1_dma_kernel (all 27 detector launches of a pass: conv1_1+conv1_2 fused .. conv_cls.4+tail)",
        "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
        "traffic": traffic, "traffic_source": traffic_src,
        "launches": conv_launches, "avg_launch_ms": conv_ms / max(conv_launches, 1),
        "algorithmic_gflop_per_page": conv_flops / 1e9 / max(B * args.steps, 1),
        # (the recogniser's ~500 launches per step overlap on side streams and are not event-timed in the measured run:
        #  Reader.set_profiling(2) + conv_profile(1) gives their busy time)
    }
    log(f"GPU legs done: {pages / dt:.1f} images/s; CPU baseline next")
    if world == 1 and args.cpu_pages > 0:
        result["cpu_baseline"] = cpu_baseline(cs, rs, uniq, args.cpu_pages)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(cs, rs, pages, n_pages):
    """The oracle (CPU restatement of the same algorithm, batch 1 per page and per box) on this host's cores."""
    import torch

    from oracle import pipeline

    # the GPU box gives one GPU a 16-core share of the host; os.cpu_count() reports the whole machine
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("BBOCR_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    n = min(n_pages, len(pages))
    t0 = time.perf_counter()
    nb = 0
    for i in range(n):
        nb += len(ref.readtext(pages[i]))
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} of the same synthetic 1280x960 pages through oracle.pipeline.OracleReader.readtext "
                      f"(torch fp32 CPU, batch 1 per page and per box, {nb} boxes, no warm-up)",
            "seconds": dt}


if __name__ == "__main__":
    main()
