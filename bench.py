#!/usr/bin/env python
"""Headline benchmark: book-page images/s, end-to-end detect+recognise @1280x960 (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (CRAFT detect -> boxes -> crops -> CRNN -> CTC) over one batch of
synthetic 1280x960 pages per GPU, pages already resident in HBM.  Workload = BASELINE.json configs[2]
("Full CRAFT+CRNN detect+recognize, batch=64 @1280x960, 1 MI355X"); with N GPUs every rank processes its own
64-page shard (weak scaling, no data-path collective; weights are broadcast once from rank 0 over RCCL) -- at N = 8
that is BASELINE.json configs[3] (512 pages sharded across 8 MI355X).  `--config a4` runs configs[4]'s per-GPU share
instead: 16 dense A4@300dpi scans (2480x3504) per GPU on the fp16 MFMA path.  `--precision` selects bbocr_config::precision; the
default is "fp16" -- the LIBRARY's default, the one mode whose boxes and strings equalled the fp32 CPU path's on every input class
measured (synthetic pages, dense A4 scans, the reference's real images).  "mixed" (bf16 detector, 2 % faster, loses
boxes on continuous-tone inputs) runs as `legs.mixed`.  Every run re-checks identity with the CPU oracle (`parity_in_run`).

Calls in flight: the reference shares ONE Reader between two ThreadPoolExecutor workers (batch_processor_enhanced.py:215); the timed
region does the same -- `--in-flight 2` (default) worker threads call `readtext_device` on one Reader, libbbocr gives each call its own
call slot, and batch k+1's detector is on the card while batch k's host thread does box geometry, CTC read-back and result export.
Exactly K calls are timed either way; `legs.serial` is the one-call-at-a-time rate of rounds 1-3.

Weights: the designed detector (bb_ocr_amd.weights.designed_craft_state) and the recogniser checkpoint trained on these
synthetic pages (tests/golden/crnn_synth_fp16.npz, tests/golden/train_crnn.py) -- a recogniser that reads the pages has the
top-2 logit margins of a trained model, so "same text as the fp32 CPU path" is measurable (`parity_in_run`).

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      -- the dominant kernel (conv_mfma, detector launches): algorithmic FLOPs / sum of launch durations
                   measured with HIP events on the library's stream inside the timed region; peak = 2.5 PFLOP/s dense bf16.
  cpu_baseline  -- the CPU oracle (restatement of the easyocr algorithm, kind "port") timed on this host, N=1 only.
  parity_in_run -- the SAME pages the CPU oracle just read, compared with what the timed GPU step returned for them:
                   boxes identical n/m, texts identical n/m.
  legs          -- N=1 only, after the timed region, own readers: `serial` (one call in flight), `mixed`, `exact`, `a4_fp16` (configs[4] share),
                   `det_only_b32_bf16` (configs[1], own roofline), `single_page` (the reference's real call: readtext(<jpeg path>), p50 / p95
                   incl. decode + H2D), `host_pages` (64-page batches starting in pinned host memory, H2D overlapped), `lowconf` (faint-ink
                   pages on which upstream's contrast retry is live, compared with the CPU oracle box by box).
  N > 1         -- ranks_seen, devices (per-rank PCI bus id / uuid, all-gathered; must be distinct under nccl),
                   weights_broadcast_ok (a failed broadcast is fatal unless --allow-local-weights).
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
    "p1": (1280, 960, 64, 24, 38, "fp16", "full CRAFT+CRNN detect+recognize, batch=64 @1280x960 per GPU (BASELINE.json configs[2])"),
    "a4": (2480, 3504, 16, 110, 31, "fp16", "A4@300dpi 2480x3504 dense-text scans, fp16 MFMA conv path, 16 pages per GPU "
                                           "(BASELINE.json configs[4]: batch=128 on 8 GPUs)"),
}
TRAINED_CRNN = os.path.join(ROOT, "tests", "golden", "crnn_synth_fp16.npz")
METRIC = {"p1": "book-page images/sec end-to-end (detect+recognize) @1280x960",
          "a4": "A4@300dpi page images/sec end-to-end (detect+recognize) @2480x3504, fp16 MFMA conv path (BASELINE.json configs[4], per-GPU share)"}
DTYPE = {"bf16": "bf16", "fp16": "fp16", "exact": "fp16 (split hi+lo operands, three MFMA product terms per layer, in both networks)", "mixed": "bf16 (detector) + fp16 (recogniser)",
         "exact_rec": "fp16 (detector) + split fp16 hi+lo operands (recogniser)"}


def log(msg, rank=0):
    if rank == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def load_states(rec_weights):
    """(craft_state, crnn_state, label).  The trained recogniser is a test-infrastructure checkpoint; it enters like any state-dict."""
    from bb_ocr_amd import weights

    cs = weights.designed_craft_state(0)
    if rec_weights == "trained":
        if not os.path.exists(TRAINED_CRNN):
            raise SystemExit(f"{TRAINED_CRNN} is missing (tests/golden/train_crnn.py writes it); --rec-weights random runs without it")
        return cs, weights.load_npz_state(TRAINED_CRNN), "seeded designed-detector weights + the recogniser trained on these synthetic pages (tests/golden/crnn_synth_fp16.npz)"
    return cs, weights.synthetic_crnn_state(0), "seeded designed-detector + random-recogniser weights"


def page_kwargs(config, width, height, lines):
    kw = dict(width=width, height=height, lines=lines, line_pitch=CONFIGS[config][4], margin=24)
    if config == "a4":
        kw.update(font_size=20, word_gap=14, margin=60)
    return kw


def render_pages(config, width, height, lines, first, count, unique):
    """Seeded pages: every other one on tinted stock / coloured ink so that the gray plane is a real 3-channel mix."""
    from bb_ocr_amd import synth

    kw = page_kwargs(config, width, height, lines)
    return [synth.page(1234 + (first + i), colour=bool((first + i) & 1), **kw)[0] for i in range(min(unique, count))]


def timed_steps(reader, rgb, steps, warmup, dist=None, backend="nccl", rank=0, tag="", in_flight=2):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.  -> (seconds, stage sums, last result).
    `in_flight` worker threads share the Reader (the reference's ThreadPoolExecutor contract); a step = one readtext_device call."""
    from concurrent.futures import ThreadPoolExecutor

    import torch

    import bb_ocr_amd

    def one(_):
        out = reader.readtext_device(rgb, None)
        return out, reader.stage_times()          # bbocr_stage_times answers for the calling thread

    with ThreadPoolExecutor(max_workers=max(1, in_flight), thread_name_prefix="bench-call") as ex:
        for out, st in ex.map(one, range(warmup)):
            log(f"{tag}warm-up step done: {st}", rank)
        reader.set_profiling(os.environ.get("BBOCR_BENCH_NOPROF") != "1")   # NOPROF: A/B of the event-recording overhead only
        if os.environ.get("BBOCR_BENCH_NOFREEZE") != "1":
            bb_ocr_amd.freeze_gc()   # host-process hygiene of a long-running OCR worker (see freeze_gc.__doc__)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stage, out = {}, None
        for out, st in ex.map(one, range(steps)):
            for k, v in st.items():
                stage[k] = stage.get(k, 0.0) + v
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
    log(f"{tag}{steps} timed steps done", rank)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, stage, out


def roofline(reader, config, batch, steps):
    conv_ms, conv_flops, conv_launches = reader.conv_profile(0)
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    return {
        "kernel": "conv3x3_dma_kernel + conv1x1_dma_kernel + conv3x3_up4_kernel + conv3x3_resw_kernel (every conv launch of the detector passes -- 25 per pass "
                  "at 1280x960: conv1_1+conv1_2 fused .. upconv3.3x3 + upconv4.1x1(y) .. upconv4 fused .. conv_cls.4+tail)",
        "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
        # HBM bytes are not measurable from inside the process; the PMC passes of the same detector are committed under profiles/
        # (r04_pmc_hbm.json, tools/pmc_traffic.py) -- a constant copied from a file would not be a measurement of THIS run
        "traffic": None,
        "traffic_profile": "profiles/r04_pmc_hbm.json (separate --pmc FETCH_SIZE / WRITE_SIZE passes over the same 25 launches, 32 pages, bf16): "
                           "839 MB read + 499 MB written per page = 1.71 GB per launch, against ~1.1 GB per page algorithmic",
        "launches": conv_launches, "avg_launch_ms": conv_ms / max(conv_launches, 1),
        "algorithmic_gflop_per_page": conv_flops / 1e9 / max(batch * steps, 1),
        "survey_gflop_per_page": CRAFT_GFLOP_PER_PAGE[config],
    }


def device_identity(local_rank):
    """What distinguishes this rank's card: PCI bus id (hipDeviceGetPCIBusId) and, when torch exposes it, the device uuid."""
    import ctypes

    import torch

    ident = {"local_rank": local_rank, "name": torch.cuda.get_device_name(local_rank), "pci_bus_id": None, "uuid": None}
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(local_rank)) == 0:
            ident["pci_bus_id"] = buf.value.decode()
    except OSError:
        pass
    try:
        ident["uuid"] = str(torch.cuda.get_device_properties(local_rank).uuid)
    except Exception:
        pass
    return ident


def run_leg(name, config, precision, steps, states, pages=None, in_flight=2):
    """One of the other driver-visible modes on this card, after the main timed region: own reader; `pages` = the main run's rendered
    pages when the leg reads the same workload (else its own are rendered)."""
    import numpy as np
    import torch

    import bb_ocr_amd

    cw, ch, cb, cl, _, _, workload = CONFIGS[config]
    t_leg = time.perf_counter()
    reader = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision=precision)
    try:
        uniq = pages if pages is not None else render_pages(config, cw, ch, cl, 0, cb, 4 if config == "a4" else cb)
        rgb = torch.from_numpy(np.stack([uniq[i % len(uniq)] for i in range(cb)])).cuda()
        dt, stage, out = timed_steps(reader, rgb, steps, max(1, in_flight), tag=f"[leg {name}] ", in_flight=in_flight)
        rf = roofline(reader, config, cb, steps)
        return {"metric": METRIC[config], "value": cb * steps / dt, "unit": "images/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": max(1, in_flight),
                "dtype": DTYPE[precision], "config": {"workload": workload, "precision": precision, "batch_per_gpu": cb, "calls_in_flight": in_flight,
                                                      "boxes_per_step": sum(len(p) for p in out), "chars_per_step": sum(len(t) for p in out for _, t, _ in p)},
                "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms")},
                "stage_ms_per_step": {k: v / steps for k, v in stage.items()}, "leg_seconds": time.perf_counter() - t_leg}
    finally:
        reader.close()
        del reader
        torch.cuda.empty_cache()


def leg_det_only(states, steps=5, batch=32):
    """BASELINE.json configs[1]: CRAFT detection only, batch 32 synthetic 1280x960 pages, bf16, one card: pages -> heat-maps (bbocr_detect)."""
    import numpy as np
    import torch

    import bb_ocr_amd

    t_leg = time.perf_counter()
    reader = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision="bf16", recognizer=False)
    try:
        uniq = render_pages("p1", 1280, 960, 24, 0, batch, 8)
        rgb = torch.from_numpy(np.stack([uniq[i % len(uniq)] for i in range(batch)])).cuda()
        reader.heatmap_device(rgb)
        reader.set_profiling(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            heat, _ = reader.heatmap_device(rgb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rf = roofline(reader, "p1", batch, steps)
        return {"metric": "CRAFT detection only, pages/sec @1280x960 (pages -> region/affinity heat-maps)", "value": batch * steps / dt, "unit": "images/s",
                "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": 1, "dtype": "bf16",
                "config": {"workload": "CRAFT detection only, batch=32 synthetic 1280x960 pages, bf16, 1 MI355X (BASELINE.json configs[1])", "batch_per_gpu": batch},
                "roofline": {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_ms")},
                "leg_seconds": time.perf_counter() - t_leg}
    finally:
        reader.close()
        del reader
        torch.cuda.empty_cache()


def leg_single_page(reader, pages, n=32):
    """The reference's real call (enhanced_extractor.py:520): readtext(<jpeg path>, paragraph=False, batch_size=1, workers=0), one 1280x960 page per
    call, file decode + H2D + detect + recognise + result marshalling inside the clock."""
    import statistics
    import tempfile

    from PIL import Image

    t_leg = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i, pg in enumerate(pages[:8]):
            paths.append(os.path.join(d, f"page{i}.jpg"))
            Image.fromarray(pg).save(paths[-1], quality=95)
        for pth in paths[:2]:
            reader.readtext(pth, paragraph=False, batch_size=1, workers=0)
        ms, dec, nb = [], [], 0
        for i in range(n):
            t0 = time.perf_counter()
            res = reader.readtext(paths[i % len(paths)], paragraph=False, batch_size=1, workers=0)
            ms.append((time.perf_counter() - t0) * 1e3)
            nb += len(res)
        from bb_ocr_amd.reader import decode_file
        for i in range(8):
            t0 = time.perf_counter()
            decode_file(paths[i % len(paths)])
            dec.append((time.perf_counter() - t0) * 1e3)
    ms.sort()
    return {"what": "Reader.readtext(<1280x960 JPEG path>, paragraph=False, batch_size=1, workers=0): the call at enhanced_extractor.py:520, one page per call, "
                    "JPEG decode (PIL / libjpeg, host) + H2D + detect + recognise + result list inside the clock",
            "calls": n, "p50_ms": statistics.median(ms), "p95_ms": ms[min(len(ms) - 1, int(0.95 * len(ms)))], "mean_ms": sum(ms) / len(ms),
            "decode_only_p50_ms": statistics.median(dec), "boxes_per_page": nb / n, "precision": reader.precision, "leg_seconds": time.perf_counter() - t_leg}


def leg_host_pages(reader, pages, batch, steps, resident_value, in_flight=2):
    """Batches that START in pinned host memory: the H2D copy of batch k+1 (torch, own stream) runs while batch k is on the card
    (Reader.readtext_stream drains its producer one batch ahead of the worker threads)."""
    import numpy as np
    import torch

    t_leg = time.perf_counter()
    host = torch.from_numpy(np.stack([pages[i % len(pages)] for i in range(batch)])).pin_memory()
    copy_stream = torch.cuda.Stream()

    def feed(k):
        for _ in range(k):
            with torch.cuda.stream(copy_stream):
                dev = host.to("cuda", non_blocking=True)
            copy_stream.synchronize()
            yield dev

    for _ in reader.readtext_stream(feed(in_flight), in_flight=in_flight):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for res in reader.readtext_stream(feed(steps), in_flight=in_flight):
        nb += sum(len(pg) for pg in res)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    v = batch * steps / dt
    return {"what": f"{steps} batches of {batch} pages, each starting in PINNED HOST memory ({host.numel() / 1e6:.0f} MB per batch): H2D on a copy stream one batch "
                    f"ahead, {in_flight} calls in flight (Reader.readtext_stream); PCIe-inclusive, never the headline value",
            "value": v, "unit": "images/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "vs_resident": v / resident_value if resident_value else None,
            "boxes_per_step": nb / steps, "leg_seconds": time.perf_counter() - t_leg}


def leg_extractor(reader, pages, n=1024):
    """SURVEY section 8 row f3, the application's own loop: a directory of JPEG files -> extractor_batch.extract_texts -> {index: text}
    (file read + JPEG decode on the host pool + H2D + the whole OCR path + the joined strings of enhanced_extractor.py:521) -- everything
    the reference's per-page loop at :680-688 does for its OCR step, batched.  PCIe- and decode-inclusive, never the headline value."""
    import tempfile

    from PIL import Image

    from bb_ocr_amd import extractor_batch

    t_leg = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i in range(min(len(pages), 16)):
            p = os.path.join(d, f"page_{i:03d}.jpg")
            Image.fromarray(pages[i]).save(p, quality=92)
            paths.append(p)
        paths = [paths[i % len(paths)] for i in range(n)]
        extractor_batch.extract_texts(reader, paths[:128])
        t0 = time.perf_counter()
        texts = extractor_batch.extract_texts(reader, paths)
        dt = time.perf_counter() - t0
    return {"what": f"{n} JPEG files (1280x960, quality 92) -> extractor_batch.extract_texts: decode pool (one YCbCr decode per page, RGB + Y plane "
                    f"derived on the card), one upload call per batch a batch ahead, two device batches in flight, joined strings",
            "value": n / dt, "unit": "images/s", "pages": n, "non_empty_texts": sum(bool(t) for t in texts.values()), "host_cores": host_cores(),
            "leg_seconds": time.perf_counter() - t_leg}


def leg_preprocess(reader, H=4284, W=5712, n=20):
    """the reference's preprocess_for_book_cover chain on the device, one photograph-sized BGR page per call
    (the reference's largest inputs are 5712x4284 phone photographs); seeded noise, resident in HBM.  Algorithmic bytes: one read + one write
    of the plane per stage (5 B per source pixel for gray + the resize's read, 21 transfers of the 1.5x plane), DESIGN section 3."""
    import numpy as np
    import torch

    from bb_ocr_amd import preprocess as dev_pp

    t_leg = time.perf_counter()
    bgr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    for _ in range(2):
        out = dev_pp.preprocess_bgr_device(reader, bgr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = dev_pp.preprocess_bgr_device(reader, bgr)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    plane = out.shape[0] * out.shape[1]
    alg = H * W * 5 + 21 * plane      # gray 3 + 1, resize 1 + 1, blur 1 + 1, CLAHE 1 + (1 + 1), six box passes 2 each, unsharp 2 + 1 (tools/preprocess_bench.py)
    return {"what": f"bbocr_preprocess_chain (gray, cubic 1.5x, blur, contrast+brightness LUT, CLAHE, unsharp) on one {W}x{H} BGR page -> "
                    f"{out.shape[1]}x{out.shape[0]} gray plane, host-synchronous call, bit-identical to the CPU restatement (tests/test_gpu_ops.py)",
            "value": 1e3 / ms, "unit": "pages/s", "ms_per_page": ms, "pages": n, "algorithmic_MB_per_page": alg / 1e6,
            "roofline": {"bound": "hbm", "achieved": alg / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": alg / ms / 1e6 / 8000.0, "traffic": None},
            "leg_seconds": time.perf_counter() - t_leg}


def leg_lowconf(reader, states, n_gpu_pages=16, n_cpu_pages=2, steps=3):
    """Low-confidence workload: faint-ink lines (synth.page(faint=0.5)) whose first-pass confidence falls under contrast_ths = 0.1, so that
    upstream's contrast retry (recognition.get_text: adjust_contrast_grey, second prediction, keep the better) is LIVE.  Timed on the card;
    a few pages are also read by the CPU oracle: boxes / texts compared, and the retry DECISION (first-pass conf < contrast_ths) per box."""
    import numpy as np
    import torch

    from bb_ocr_amd import synth
    from oracle import imgproc, pipeline

    t_leg = time.perf_counter()
    kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24, faint=0.5)
    pages = [synth.page(9000 + i, **kw)[0] for i in range(n_gpu_pages)]
    rgb = torch.from_numpy(np.stack(pages)).cuda()
    out = reader.readtext_device(rgb, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    retry_ms = 0.0
    for _ in range(steps):
        out = reader.readtext_device(rgb, None)
        retry_ms += reader.stage_times()["contrast_retry"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    first = reader.readtext_device(rgb[:n_cpu_pages].contiguous(), None, contrast_ths=0.0)       # first-pass confidences (no retry)
    cs, rs = states
    torch.set_num_threads(host_cores())
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    boxes = same_box = same_text = low_ref = low_gpu = dec_diff = text_diff_same_decision = 0
    for i in range(n_cpu_pages):
        img, grey = imgproc.reformat_input(pages[i])
        h, f = ref.detect(img)                                             # one detector pass on the CPU, two recogniser passes
        want, want1 = ref.recognize(grey, h, f), ref.recognize(grey, h, f, contrast_ths=0.0)
        got, got1 = out[i], first[i]
        boxes += len(want)
        for w, g, w1, g1 in zip(want, got, want1, got1):
            b = np.array_equal(np.asarray(w[0], dtype=np.float64), np.asarray(g[0], dtype=np.float64))
            dw, dg = float(w1[2]) < 0.1, float(g1[2]) < 0.1
            same_box += b
            same_text += w[1] == g[1]
            low_ref += dw
            low_gpu += dg
            dec_diff += dw != dg
            text_diff_same_decision += (dw == dg) and (w[1] != g[1])
    n_all = sum(len(pg) for pg in out)
    return {"what": "faint-ink pages (synth.page(faint=0.5): half of the lines a few grey levels from the paper in cv2's gray plane): upstream's contrast retry is "
                    "live; the last timed step's result for the first pages vs oracle.pipeline.OracleReader.readtext, retry decision = first-pass conf < 0.1",
            "precision": reader.precision, "value": n_gpu_pages * steps / dt, "unit": "images/s", "batch": n_gpu_pages, "steps": steps, "ms_per_step": dt / steps * 1e3,
            "contrast_retry_ms_per_step": retry_ms / steps, "boxes_per_step": n_all,
            "oracle_pages": n_cpu_pages, "boxes": boxes, "boxes_identical": f"{same_box}/{boxes}", "texts_identical": f"{same_text}/{boxes}",
            "low_confidence_boxes_oracle": low_ref, "low_confidence_boxes_gpu": low_gpu, "retry_decisions_differing": dec_diff,
            "texts_differing_with_equal_decision": text_diff_same_decision,
            "all_identical": bool(same_box == boxes and same_text == boxes), "leg_seconds": time.perf_counter() - t_leg}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="p1", help="p1 = BASELINE.json configs[2]/[3] (the metric's workload), a4 = configs[4]")
    ap.add_argument("--batch", type=int, default=0, help="pages per GPU per step (0 = the config's)")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic pages rendered per rank, tiled to --batch (0 = --batch: every page of the batch is its own page)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--lines", type=int, default=0, help="text lines per page (0 = the config's: 24 / 110, SURVEY.md section 8d)")
    ap.add_argument("--precision", choices=("bf16", "fp16", "exact", "mixed", "exact_rec"), default=None, help="bbocr_config::precision (default: the config's)")
    ap.add_argument("--rec-weights", choices=("trained", "random"), default="trained", help="recogniser: tests/golden/crnn_synth_fp16.npz or seeded random")
    ap.add_argument("--cpu-pages", type=int, default=16, help="pages for the CPU-oracle baseline + parity_in_run after one warm-up page (0 = skip)")
    ap.add_argument("--legs", default="serial,mixed,exact,exact_rec,a4,det_only,single_page,host_pages,lowconf,preprocess,extractor",
                    help="N=1: extra legs run after the timed region (comma list of serial, fp16, mixed, bf16, exact, exact_rec, a4, det_only, single_page, host_pages, lowconf, preprocess, extractor; '' = none)")
    ap.add_argument("--leg-steps", type=int, default=6)
    ap.add_argument("--in-flight", type=int, default=2, help="calls in flight on the one Reader during the timed region (worker threads; bbocr_config::call_slots = 2)")
    ap.add_argument("--det-sub-batch", type=int, default=0)
    ap.add_argument("--scatter", action="store_true", help="N > 1: rank 0 renders every rank's pages and scatters them (dist.scatter_pages: grouped "
                                                           "RCCL sends over xGMI) instead of every rank rendering its own shard; outside the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo only for single-GPU rehearsals)")
    ap.add_argument("--allow-local-weights", action="store_true", help="N > 1: if the packed weight broadcast fails, build the seeded weights on every rank "
                                                                       "and continue (reported as weights_broadcast_ok false) instead of exiting non-zero")
    args = ap.parse_args()
    cw, ch, cb, cl, cpitch, cprec, workload = CONFIGS[args.config]
    args.width, args.height = args.width or cw, args.height or ch
    args.batch, args.lines = args.batch or cb, args.lines or cl
    args.precision = args.precision or cprec
    args.unique = args.unique or args.batch

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

    # ---- weights: built and packed on rank 0, broadcast ONCE as the packed device blob (RCCL, device to device over xGMI) -- the
    # process-per-GPU form of DataParallel's per-forward replicate; the other ranks never see an fp32 state-dict
    cs = rs = None
    weights_label = ""
    if rank == 0 or world == 1:
        cs, rs, weights_label = load_states(args.rec_weights)
    mk = lambda w: bb_ocr_amd.Reader(["en"], gpu=True, weights=w, device_index=local_rank, det_sub_batch=args.det_sub_batch, precision=args.precision)
    if os.environ.get("BBOCR_BENCH_INJECT") == "precision_mismatch" and rank == 1:     # tests/test_gpu_multirank.py: a broadcast that must fail
        mk = lambda w: bb_ocr_amd.Reader(["en"], gpu=True, weights=w, device_index=local_rank, precision="exact" if args.precision != "exact" else "bf16")
    if os.environ.get("BBOCR_BENCH_INJECT") == "construct_fail_rank1" and rank == 1:       # a failure ONE rank sees (the others must not hang in a collective)
        def mk(w):
            raise MemoryError("injected: reader construction fails on rank 1 only")
    weights_path, broadcast_ok, devices = "local (single process)", None, None
    if world > 1:
        # who is here: every rank's card, all-gathered (a SCALE run can be checked for N distinct devices from the JSON alone)
        devices = [None] * world
        dist.all_gather_object(devices, dict(device_identity(local_rank), rank=rank))
        if args.backend == "nccl":
            ids = [d["pci_bus_id"] or d["uuid"] or f"local{d['local_rank']}" for d in devices]
            if len(set(ids)) != world:
                raise SystemExit(f"ranks share a device under nccl: {ids}")
        reader, err = None, None
        t_b = time.perf_counter()
        try:
            reader = bdist.broadcast_packed(lambda: mk((cs, rs)), lambda: mk("empty"), src=0, via_host=(args.backend != "nccl"))
        except Exception as e:
            err = e
            print(f"[bench rank {rank}] packed weight broadcast failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
        # agree on the outcome before anyone goes on (a rank-local failure must not leave the others inside a later collective)
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        broadcast_ok = bool(int(flag.item()))
        if broadcast_ok:
            weights_path = (f"packed device blob ({reader.weights_blob_size() / 1e6:.1f} MB) broadcast from rank 0 over "
                            f"{'RCCL, device to device' if args.backend == 'nccl' else args.backend + ' (host hop: rehearsal only)'} "
                            f"in {(time.perf_counter() - t_b) * 1e3:.0f} ms incl. packing")
        else:
            if reader is not None:       # (broadcast_packed closes the reader it built before it raises; a reader that came back is whole)
                reader.close()
            if not args.allow_local_weights:
                dist.destroy_process_group()
                raise SystemExit("the packed weight broadcast failed on at least one rank; --allow-local-weights lets every rank build the seeded weights instead")
            cs, rs, weights_label = load_states(args.rec_weights)
            reader = bb_ocr_amd.Reader(["en"], gpu=True, weights=(cs, rs), device_index=local_rank, det_sub_batch=args.det_sub_batch, precision=args.precision)
            weights_path = "local on every rank (the packed broadcast FAILED; --allow-local-weights)"
    else:
        reader = mk((cs, rs))

    # ---- this rank's shard of the global batch (contiguous block), rendered on the host, then resident in HBM
    B = args.batch
    g0, g1 = bdist.shard_range(B * world, rank, world)
    render = lambda a, n: render_pages(args.config, args.width, args.height, args.lines, a, n, args.unique)
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

    log(f"pages resident: {tuple(rgb.shape)}; warm-up x{args.warmup}", rank)
    dt, stage, out = timed_steps(reader, rgb, args.steps, args.warmup, dist, args.backend, rank, in_flight=args.in_flight)
    rf = roofline(reader, args.config, B, args.steps)
    n_boxes = sum(len(p) for p in out)
    n_chars = sum(len(t) for p in out for _, t, _ in p)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    pages = B * world * args.steps
    result = {
        "metric": METRIC[args.config],
        "value": pages / dt,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE[args.precision],
        "data": f"synthetic ({len(uniq)} distinct seeded pages per GPU tiled to the batch; {weights_label})",
        "config": {
            "workload": workload if not (args.config == "p1" and world == 8) else
                        "batch=512 @1280x960 sharded across 8 MI355X, 64 pages per GPU (BASELINE.json configs[3]; RCCL weight broadcast, "
                        "every rank renders its own shard)",
            "precision": args.precision,
            "batch_per_gpu": B, "page_wh": [args.width, args.height], "text_lines_per_page": args.lines,
            "boxes_per_step_rank0": n_boxes, "chars_per_step_rank0": n_chars, "parallelism": f"dp{world} (page shards, no data-path collective)",
            "calls_in_flight": args.in_flight,
            "weights": weights_path, "pages": pages_path,
        },
        "stage_ms_per_step_rank0": {k: v / args.steps for k, v in stage.items()},
        "stage_ms_note": ("per CALL, from device events and host clocks inside that call; with more than one call in flight the stages of different calls overlap "
                          "(their sum exceeds ms_per_step) and a stage queued behind the other call's detector includes that wait -- "
                          "legs.serial holds the one-call-at-a-time breakdown") if args.in_flight > 1 else "one call at a time",
        "roofline": rf,
    }
    if world > 1:
        result["ranks_seen"] = len([d for d in devices if d is not None])
        result["devices"] = devices
        result["weights_broadcast_ok"] = broadcast_ok
    log(f"timed region done: {pages / dt:.1f} images/s", rank)
    if world == 1:
        legs = {}
        names = [n for n in args.legs.split(",") if n]
        same = args.config == "p1" and (args.width, args.height, args.lines, args.batch) == CONFIGS["p1"][:2] + (CONFIGS["p1"][3], CONFIGS["p1"][2])
        # legs that reuse the main reader (same weights, same precision) run before it is closed
        if "single_page" in names and args.config == "p1":
            legs["single_page"] = leg_single_page(reader, uniq)
            log(f"leg single_page: p50 {legs['single_page']['p50_ms']:.2f} ms")
        if "host_pages" in names and args.config == "p1":
            legs["host_pages"] = leg_host_pages(reader, uniq, B, max(12, args.leg_steps), pages / dt, args.in_flight)
            log(f"leg host_pages: {legs['host_pages']['value']:.1f} images/s")
        if "extractor" in names and args.config == "p1":
            legs["extractor_jpeg"] = leg_extractor(reader, uniq)
            log(f"leg extractor_jpeg: {legs['extractor_jpeg']['value']:.1f} images/s")
        if "preprocess" in names:
            legs["preprocess_f2"] = leg_preprocess(reader)
            log(f"leg preprocess_f2: {legs['preprocess_f2']['ms_per_page']:.2f} ms per page")
        if "lowconf" in names and args.rec_weights == "trained":
            legs["lowconf"] = leg_lowconf(reader, (cs, rs))
            log(f"leg lowconf: retry {legs['lowconf']['contrast_retry_ms_per_step']:.2f} ms per step, texts {legs['lowconf']['texts_identical']}")
        reader.close()
        del reader
        torch.cuda.empty_cache()
        for name in names:
            if name in ("exact", "mixed", "fp16", "bf16", "exact_rec") and not (args.config == "p1" and args.precision == name):
                legs[name] = run_leg(name, "p1", name, args.leg_steps, (cs, rs), uniq if same else None, args.in_flight)
            elif name == "serial":
                legs["serial"] = run_leg("serial", "p1", args.precision, args.leg_steps, (cs, rs), uniq if same else None, 1)
            elif name == "a4" and args.config != "a4":
                legs["a4_fp16"] = run_leg("a4_fp16", "a4", "fp16", args.leg_steps, (cs, rs), None, args.in_flight)
            elif name == "det_only":
                legs["det_only_b32_bf16"] = leg_det_only((cs, rs))
            else:
                continue
            log(f"leg {name}: {list(legs.values())[-1]['value']:.1f} images/s")
        if legs:
            result["legs"] = legs
        if args.cpu_pages > 0:
            log("CPU baseline next")
            result["cpu_baseline"], ref_pages = cpu_baseline(cs, rs, uniq, args.cpu_pages, (args.width, args.height))
            result["parity_in_run"] = parity(ref_pages, out, args.precision)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def parity(ref_pages, gpu_pages, mode):
    """The pages the CPU oracle read (indices 0..n-1 of this rank's batch) against what the last TIMED GPU step returned for them."""
    import numpy as np

    boxes = same_boxes = same_text = pages_same = 0
    worst = None
    for want, got in zip(ref_pages, gpu_pages):
        boxes += len(want)
        ok_page = len(want) == len(got)
        for w, g in zip(want, got):
            b = np.array_equal(np.asarray(w[0], dtype=np.float64), np.asarray(g[0], dtype=np.float64))
            t = w[1] == g[1]
            same_boxes += b
            same_text += t
            ok_page = ok_page and b and t
            if not t and worst is None:
                worst = {"oracle": w[1], "gpu": g[1]}
        pages_same += ok_page
    return {"pages": len(ref_pages), "mode": mode, "boxes": boxes, "boxes_identical": f"{same_boxes}/{boxes}", "texts_identical": f"{same_text}/{boxes}",
            "pages_identical": f"{pages_same}/{len(ref_pages)}", "all_identical": bool(same_boxes == boxes and same_text == boxes),
            "first_text_difference": worst,
            "what": "oracle.pipeline.OracleReader.readtext (fp32 CPU) vs the last timed GPU step's result for the same pages: box coordinates equal as numbers, decoded strings equal"}


def host_cores():
    """the GPU box gives one GPU a 16-core share of the host; os.cpu_count() reports the whole machine"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("BBOCR_CPU_THREADS", "16"))))


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
    cores: one warm-up page, then n pages timed one by one; value = 1 / median seconds per page.
    -> (cpu_baseline object, the oracle's per-page results for parity_in_run)."""
    import statistics

    import torch

    from oracle import pipeline

    torch.set_num_threads(host_cores())
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    n = min(n_pages, len(pages))
    ref.readtext(pages[0])                                   # warm-up (thread pool, oneDNN primitive caches)
    per_page, nb, results = [], 0, []
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        results.append(ref.readtext(pages[i]))
        per_page.append(time.perf_counter() - t0)
        nb += len(results[-1])
    med = statistics.median(per_page)
    return {"value": 1.0 / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "cpu": cpu_model(),
            "sample": f"CPU restatement (EasyOCR-equivalent algorithm, synthetic weights): {n} of the same synthetic {wh[0]}x{wh[1]} pages through "
                      f"oracle.pipeline.OracleReader.readtext (torch fp32 CPU, batch 1 per page and per box, {nb} boxes) after 1 warm-up page; "
                      f"value = 1 / median seconds per page",
            "median_s_per_page": med, "seconds": time.perf_counter() - t_all}, results


if __name__ == "__main__":
    main()
