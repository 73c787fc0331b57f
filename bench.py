#!/usr/bin/env python
"""Headline benchmark: book-page images/s, end-to-end detect+recognise @1280x960 (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (CRAFT detect -> boxes -> crops -> CRNN -> CTC) over one batch of
synthetic 1280x960 pages per GPU, pages already resident in HBM.  Workload = BASELINE.json configs[2]
("Full CRAFT+CRNN detect+recognize, batch=64 @1280x960, 1 MI355X"); with N GPUs every rank processes its own
64-page shard (weak scaling, no data-path collective; weights are broadcast once from rank 0 over RCCL) -- at N = 8
that is BASELINE.json configs[3] (512 pages sharded across 8 MI355X).  `--config a4` runs configs[4]'s per-GPU share
instead: 16 dense A4@300dpi scans (2480x3504) per GPU on the fp16 MFMA path.  `--precision` selects bbocr_config::precision
(bf16 default; fp16; exact = split-fp16 recogniser whose text equals the fp32 CPU path's).

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

PEAK_BF16_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 / fp16 MFMA
CRAFT_GFLOP_PER_PAGE = {"p1": 874.22, "a4": 3322.03}  # 1280x960 page / A4@300dpi on the 2560 canvas
CONFIGS = {   # BASELINE.json configs -> (page W, H, batch per GPU, text lines, line pitch, precision, workload label)
    "p1": (1280, 960, 64, 24, 38, "bf16", "full CRAFT+CRNN detect+recognize, batch=64 @1280x960 per GPU (BASELINE.json configs[2])"),
    "a4": (2480, 3504, 16, 110, 31, "fp16", "A4@300dpi 2480x3504 dense-text scans, fp16 MFMA conv path, 16 pages per GPU "
                                           "(BASELINE.json configs[4]: batch=128 on 8 GPUs)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="p1", help="p1 = BASELINE.json configs[2]/[3] (the metric's workload), a4 = configs[4]")
    ap.add_argument("--batch", type=int, default=0, help="pages per GPU per step (0 = the config's)")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic pages rendered per rank (tiled to --batch)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--lines", type=int, default=0, help="text lines per page (0 = the config's: 24 / 110, SURVEY.md section 8d)")
    ap.add_argument("--precision", choices=("bf16", "fp16", "exact"), default=None, help="bbocr_config::precision (default: the config's)")
    ap.add_argument("--cpu-pages", type=int, default=8, help="pages for the CPU-oracle baseline after one warm-up page (0 = skip)")
    ap.add_argument("--det-sub-batch", type=int, default=0)
    ap.add_argument("--scatter", action="store_true", help="N > 1: rank 0 renders every rank's pages and scatters them (dist.scatter_pages: grouped "
                                                           "RCCL sends over xGMI) instead of every rank rendering its own shard; outside the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU rehearsals)")
    args = ap.parse_args()
    cw, ch, cb, cl, cpitch, cprec, workload = CONFIGS[args.config]
    args.width, args.height = args.width or cw, args.height or ch
    args.batch, args.lines = args.batch or cb, args.lines or cl
    args.precision = args.precision or cprec

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

    # ---- weights: built and packed on rank 0, broadcast ONCE as the packed device blob (RCCL, device to device over xGMI) -- the
    # process-per-GPU form of DataParallel's per-forward replicate; the other ranks never see an fp32 state-dict
    cs = rs = None
    if rank == 0:
        cs, rs = weights.designed_craft_state(0), weights.synthetic_crnn_state(0)
    mk = lambda w: bb_ocr_amd.Reader(["en"], gpu=True, weights=w, device_index=local_rank, det_sub_batch=args.det_sub_batch, precision=args.precision)
    weights_path = "local (single process)"
    if world > 1:
        try:
            t_b = time.perf_counter()
            reader = bdist.broadcast_packed(lambda: mk((cs, rs)), lambda: mk("empty"), src=0, via_host=(args.backend != "nccl"))
            weights_path = (f"packed device blob ({reader.weights_blob_size() / 1e6:.1f} MB) broadcast from rank 0 over "
                            f"{'RCCL, device to device' if args.backend == 'nccl' else args.backend + ' (host hop: rehearsal only)'} "
                            f"in {(time.perf_counter() - t_b) * 1e3:.0f} ms incl. packing")
        except Exception as e:      # a measurement must not die on the one-off setup collective: the weights are seeded, every rank can build them
            print(f"[bench rank {rank}] packed weight broadcast failed ({type(e).__name__}: {e}); building the seeded weights locally", file=sys.stderr, flush=True)
            reader = mk((weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
            weights_path = f"local on every rank (broadcast failed: {type(e).__name__})"
    else:
        reader = mk((cs, rs))

    # ---- this rank's shard of the global batch (contiguous block), rendered on the host, then resident in HBM
    B = args.batch
    g0, g1 = bdist.shard_range(B * world, rank, world)
    # seeded pages: every other one on tinted stock / coloured ink so that the gray plane is a real 3-channel mix
    page_kw = dict(width=args.width, height=args.height, lines=args.lines, line_pitch=cpitch, margin=24)
    if args.config == "a4":
        page_kw.update(font_size=20, word_gap=14, margin=60)
    render = lambda a, n: [synth.page(1234 + (a + i), colour=bool((a + i) & 1), **page_kw)[0] for i in range(min(args.unique, n))]
    pages_path = "every rank renders its own shard (no scatter)"
    if args.scatter and world > 1:
        # BASELINE.json configs[3] names an image scatter: the loader rank holds all pages and sends every rank its block
        allp = None
        if rank == 0:
            blocks = []
            for r in range(world):
                ra, rb = bdist.shard_range(B * world, r, world)
                u = render(ra, rb - ra)
                blocks.append(np.stack([u[i % len(u)] for i in range(rb - ra)]))
            allp = torch.from_numpy(np.concatenate(blocks))
            if args.backend == "nccl":
                allp = allp.cuda()
        t_s = time.perf_counter()
        rgb = bdist.scatter_pages(allp, B * world, (args.height, args.width, 3), src=0, device=("cuda" if args.backend == "nccl" else "cpu")).cuda()
        torch.cuda.synchronize()
        pages_path = f"scattered from rank 0 (dist.scatter_pages, {rgb.numel() / 1e6:.0f} MB per rank) in {(time.perf_counter() - t_s) * 1e3:.0f} ms"
        uniq = [rgb[i].cpu().numpy() for i in range(min(args.unique, g1 - g0))]
        del allp
    else:
        uniq = render(g0, B)
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
        "metric": "book-page images/sec end-to-end (detect+recognize) @1280x960" if args.config == "p1" else
                  "A4@300dpi page images/sec end-to-end (detect+recognize) @2480x3504, fp16 MFMA conv path (BASELINE.json configs[4], per-GPU share)",
        "value": pages / dt,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"bf16": "bf16", "fp16": "fp16", "exact": "fp16 (split hi+lo operands in the recogniser)"}[args.precision],
        "data": f"synthetic ({len(uniq)} distinct seeded pages per GPU tiled to the batch; seeded designed-detector + random-recogniser weights)",
        "config": {
            "workload": workload if not (args.config == "p1" and world == 8) else
                        "batch=512 @1280x960 sharded across 8 MI355X, 64 pages per GPU (BASELINE.json configs[3]; RCCL weight broadcast, "
                        "every rank renders its own shard)",
            "precision": args.precision,
            "batch_per_gpu": B, "page_wh": [args.width, args.height], "text_lines_per_page": args.lines,
            "boxes_per_step_rank0": n_boxes, "chars_per_step_rank0": n_chars, "parallelism": f"dp{world} (page shards, no data-path collective)",
            "weights": weights_path, "pages": pages_path,
        },
        "stage_ms_per_step_rank0": {k: v / args.steps for k, v in stage.items()},
    }
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of the
    # same detector, reads = 2 x FETCH_SIZE KiB on gfx950), scaled by the pages the average launch of THIS run processed
    traffic, traffic_src = None, None
    pmc_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    pmc = next((os.path.join(pmc_dir, n) for n in ("r02_pmc_hbm.json", "r01_pmc_hbm.json") if os.path.exists(os.path.join(pmc_dir, n))), None)
    if pmc and conv_launches > 0 and args.config == "p1" and args.precision == "bf16":
        with open(pmc) as f:
            t = json.load(f)
        traffic = (t["read_MB_per_page"] + t["write_MB_per_page"]) * 1e6 * B * args.steps / conv_launches
        traffic_src = (f"NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same detector, committed as profiles/{os.path.basename(pmc)} "
                       f"({t['read_MB_per_page']:.0f} MB read + {t['write_MB_per_page']:.0f} MB written per page), scaled to bytes per average launch")
    result["roofline"] = {
        "kernel": "conv3x3_dma_kernel + conv1x1_dma_kernel + conv3x3_up4_kernel + conv3x3_resw_kernel (every conv launch of the detector passes -- 25 per pass at this page size: conv1_1+conv1_2 fused .. upconv3.3x3 + upconv4.1x1(y) .. upconv4 fused .. conv_cls.4+tail)",
        "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
        "traffic": traffic, "traffic_source": traffic_src,
        "launches": conv_launches, "avg_launch_ms": conv_ms / max(conv_launches, 1),
        "algorithmic_gflop_per_page": conv_flops / 1e9 / max(B * args.steps, 1),
        "survey_gflop_per_page": CRAFT_GFLOP_PER_PAGE[args.config],
        # (the recogniser's ~500 launches per step overlap on side streams and are not event-timed in the measured run:
        #  Reader.set_profiling(2) + conv_profile(1) gives their busy time)
    }
    log(f"GPU legs done: {pages / dt:.1f} images/s; CPU baseline next")
    if world == 1 and args.cpu_pages > 0:
        result["cpu_baseline"] = cpu_baseline(cs, rs, uniq, args.cpu_pages, (args.width, args.height))
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cs, rs, pages, n_pages, wh):
    """The oracle (CPU restatement of the same algorithm, batch 1 per page and per box like the reference's readtext call) on this host's
    cores: one warm-up page, then n pages timed one by one; value = 1 / median seconds per page."""
    import statistics

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
    ref.readtext(pages[0])                                   # warm-up (thread pool, oneDNN primitive caches)
    per_page, nb = [], 0
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        nb += len(ref.readtext(pages[i]))
        per_page.append(time.perf_counter() - t0)
    med = statistics.median(per_page)
    return {"value": 1.0 / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model(),
            "sample": f"CPU restatement (EasyOCR-equivalent algorithm, synthetic weights): {n} of the same synthetic {wh[0]}x{wh[1]} pages through "
                      f"oracle.pipeline.OracleReader.readtext (torch fp32 CPU, batch 1 per page and per box, {nb} boxes) after 1 warm-up page; "
                      f"value = 1 / median seconds per page",
            "median_s_per_page": med, "seconds": time.perf_counter() - t_all}


if __name__ == "__main__":
    main()
