#!/usr/bin/env python
"""Text / box identity of every precision mode against the fp32 CPU oracle on >= 2,000 boxes, with the oracle's
top-2 logit margin histogram.  Test infrastructure (imports oracle/).  On a GPU box:

    python tools/parity_sweep.py --pages 64 --out gpurun_out/text_parity.json

Pages: the bench's seeded 1280x960 pages, `--pages` DISTINCT ones (bench.py tiles 8); weights: designed detector + the trained
recogniser (tests/golden/crnn_synth_fp16.npz).  The oracle reads them one by one (batch 1 per page and per box, ~2.6 s per page on
16 host threads); every GPU mode reads them as one device batch, exactly like the timed bench step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pages", type=int, default=64)
    ap.add_argument("--first", type=int, default=0, help="index of the first page (seed 1234 + first); bench.py uses 0..7")
    ap.add_argument("--modes", default="bf16,mixed,fp16,exact")
    ap.add_argument("--config", default="p1", choices=("p1", "a4"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "text_parity.json"))
    args = ap.parse_args()
    import torch

    import bb_ocr_amd
    import bench
    from bb_ocr_amd import synth, weights
    from conftest import LogitTap
    from oracle import pipeline

    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cs, rs, label = bench.load_states("trained")
    cw, ch, _, cl, _, _, _ = bench.CONFIGS[args.config]
    kw = bench.page_kwargs(args.config, cw, ch, cl)
    rendered = [synth.page(1234 + args.first + i, colour=bool((args.first + i) & 1), **kw) for i in range(args.pages)]
    pages = [p[0] for p in rendered]
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    t0 = time.time()
    want, margins = [], []
    for i, img in enumerate(pages):
        with LogitTap(ref) as tap:
            want.append(ref.readtext(img))
        mm = tap.min_margins()
        margins += mm[:len(want[-1])]
        if i % 8 == 7:
            print(f"[oracle] {i + 1} pages, {sum(len(w) for w in want)} boxes, {time.time() - t0:.0f} s", flush=True)
    n_boxes = sum(len(w) for w in want)
    # does the recogniser read the page?  ground-truth words vs the oracle's joined text, per page
    gt_words = sum(len(p[1]) for p in rendered)
    read = 0
    for (img, words), res in zip(rendered, want):
        have = " ".join(t for _, t, _ in res).split()
        pool = {}
        for w in have:
            pool[w] = pool.get(w, 0) + 1
        for w in words:
            if pool.get(w[4], 0) > 0:
                pool[w[4]] -= 1
                read += 1
    mg = np.array(margins)
    hist_edges = [0, 1e-3, 2e-3, 4e-3, 8e-3, 1.6e-2, 3e-2, 6e-2, 0.12, 0.25, 1.0]
    report = {"pages": args.pages, "first_page": args.first, "config": args.config, "boxes": n_boxes, "weights": label,
              "oracle_seconds": time.time() - t0,
              "oracle_reads_ground_truth_words": f"{read}/{gt_words}",
              "oracle_confidence_quantiles": {q: float(np.quantile([float(c) for w in want for _, _, c in w], float(q))) for q in ("0.01", "0.1", "0.5")},
              "oracle_min_margin_per_box": {"what": "min over the box's time steps of (top1 - top2 logit) / max |logit| in the fp32 oracle",
                                            "quantiles": {q: float(np.quantile(mg, float(q))) for q in ("0.001", "0.01", "0.05", "0.5")},
                                            "histogram_edges": hist_edges, "histogram": np.histogram(mg, hist_edges)[0].tolist()},
              "modes": {}}
    print(f"[oracle] {n_boxes} boxes; reads {read}/{gt_words} ground-truth words; min-margin quantiles {report['oracle_min_margin_per_box']['quantiles']}", flush=True)
    rgb = torch.from_numpy(np.stack(pages)).cuda()
    for mode in args.modes.split(","):
        r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision=mode)
        try:
            got = r.readtext_device(rgb)
            t1 = time.time()
            got = r.readtext_device(rgb)
            dt = time.time() - t1
            p = bench.parity(want, got, mode)
            diffs, conf_err, conf_all = [], 0.0, []
            k = 0
            for pw, pg in zip(want, got):
                for w, g in zip(pw, pg):
                    if w[1] != g[1] and len(diffs) < 40:
                        diffs.append({"oracle": w[1], "gpu": g[1], "oracle_min_margin": margins[k] if k < len(margins) else None, "oracle_conf": float(w[2]), "gpu_conf": float(g[2])})
                    if w[1] == g[1] and np.array_equal(np.asarray(w[0], dtype=np.float64), np.asarray(g[0], dtype=np.float64)):
                        conf_all.append(abs(float(w[2]) - float(g[2])) / max(float(w[2]), 1e-3))
                        conf_err = max(conf_err, conf_all[-1])
                    k += 1
            p.update(relative_confidence_error_quantiles={q: float(np.quantile(conf_all, float(q))) for q in ("0.5", "0.9", "0.99", "1.0")} if conf_all else None,
                     ms_per_batch=dt * 1e3, differing_boxes=diffs, max_relative_confidence_error_on_equal_boxes_and_texts=conf_err,
                     box_count_equal=bool(all(len(a) == len(b) for a, b in zip(want, got))))
            report["modes"][mode] = p
            print(f"[{mode}] boxes identical {p['boxes_identical']}, texts identical {p['texts_identical']}, pages identical {p['pages_identical']}; "
                  f"{dt * 1e3:.1f} ms per {args.pages}-page batch; max rel. confidence error {conf_err:.2e}", flush=True)
            for d in diffs[:6]:
                print("     ", d, flush=True)
        finally:
            r.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
