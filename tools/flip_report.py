#!/usr/bin/env python
"""Threshold-flip report: how often the detector's reduced-precision heat-map decides `text > 0.4`, `link > 0.4` or `text >= 0.7`
differently from the fp32 CPU oracle, and what that does to the box set (SURVEY.md section 7 "measure flip rate").

Test infrastructure (imports oracle/).  Runs on a GPU box:
    python tools/flip_report.py --out gpurun_out/flip_report.json [--sets a4,photo,p1,noise] [--precisions bf16,fp16,exact]

Page sets
  a4     4 dense A4@300dpi scans (2480x3504, the configs[4] pages): anti-aliased after the 0.73x canvas resize
  photo  the reference's two real photographs (tests/golden/photos/IMG_968{4,5}.JPG, pipeline_demo/books/2a)
  p1     4 of the bench's 1280x960 pages
  noise  4 synthetic 640x480 pages through a NOISE-SENSITIVE detector: seeded random CRAFT whose last 1x1 layer is rescaled so that both
         maps span the thresholds (region 0.30 +- 0.20, affinity 0.20 +- 0.15) -- every layer of the trunk contributes to every pixel
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


def page_sets(which):
    from PIL import Image

    from bb_ocr_amd import synth

    out = {}
    if "a4" in which:
        kw = dict(width=2480, height=3504, lines=110, font_size=20, word_gap=14, line_pitch=31, margin=60)
        out["a4"] = [synth.page(1234 + i, colour=bool(i & 1), **kw)[0] for i in range(4)]
    if "photo" in which:
        d = os.path.join(ROOT, "tests", "golden", "photos")
        out["photo"] = [np.asarray(Image.open(os.path.join(d, n)).convert("RGB")) for n in ("IMG_9685.JPG", "IMG_9684.JPG")]
    if "p1" in which:
        kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24)
        out["p1"] = [synth.page(1234 + i, colour=bool(i & 1), **kw)[0] for i in range(4)]
    if "noise" in which:
        out["noise"] = [synth.page(910 + i, width=640, height=480, lines=10, margin=24, colour=bool(i & 1))[0] for i in range(4)]
    return out


def compare(h, st, sl):
    t, l = h[..., 0], h[..., 1]
    near = lambda m, c, r: (np.abs(m - c) < r)
    d = {
        "pixels": int(st.size),
        "max_err_text": float(np.abs(t - st).max()), "max_err_link": float(np.abs(l - sl).max()),
        "max_err_text_near_thresholds": float(np.abs(t - st)[near(st, 0.4, 0.1) | near(st, 0.7, 0.1)].max(initial=0.0)),
        "max_err_link_near_threshold": float(np.abs(l - sl)[near(sl, 0.4, 0.1)].max(initial=0.0)),
        "flips_text_gt_0.4": int(((t > 0.4) != (st > 0.4)).sum()),
        "flips_link_gt_0.4": int(((l > 0.4) != (sl > 0.4)).sum()),
        "flips_text_ge_0.7": int(((t >= 0.7) != (st >= 0.7)).sum()),
        "oracle_pixels_within_0.01_of_a_threshold": int((near(st, 0.4, 0.01) | near(sl, 0.4, 0.01) | near(st, 0.7, 0.01)).sum()),
        "oracle_min_distance_to_threshold": float(min(np.abs(st - 0.4).min(), np.abs(sl - 0.4).min(), np.abs(st - 0.7).min())),
    }
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="a4,photo,p1,noise")
    ap.add_argument("--precisions", default="bf16,fp16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "flip_report.json"))
    args = ap.parse_args()
    import torch

    import bb_ocr_amd
    import conftest
    from bb_ocr_amd import weights
    from oracle import boxes as obox
    from oracle import pipeline

    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sets = page_sets(args.sets.split(","))
    designed = (weights.designed_craft_state(0), weights.synthetic_crnn_state(0))
    report = {}
    for name, pages in sets.items():
        states = conftest.noise_sensitive_craft() if name == "noise" else designed
        ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in states[0].items()}, {k: torch.from_numpy(v) for k, v in states[1].items()})
        t0 = time.time()
        want = []
        for img in pages:
            st, sl, ratio = ref.heatmap(img)
            oh, of, op = obox.detect_from_heatmap(st, sl, ratio)
            want.append((st, sl, ratio, oh, of, op))
        print(f"[{name}] oracle: {len(pages)} pages in {time.time() - t0:.1f} s; boxes per page {[len(w[3]) + len(w[4]) for w in want]}", flush=True)
        report[name] = {}
        for prec in args.precisions.split(","):
            r = bb_ocr_amd.Reader(["en"], weights=states, precision=prec)
            try:
                rows = []
                for img, (st, sl, ratio, oh, of, op) in zip(pages, want):
                    heat, r2 = r.heatmap_device(torch.from_numpy(img[None]).cuda())
                    assert r2 == ratio
                    hori, free, polys = r.boxes_from_heatmap(heat, ratio)
                    d = compare(heat[0].cpu().numpy(), st, sl)
                    wh = {tuple(map(int, b)) for b in oh}
                    wf = {tuple(np.asarray(b, dtype=np.float64).reshape(-1).tolist()) for b in of}
                    wp = {tuple(map(int, p)) for p in op}
                    d.update(polys=len(polys[0]), polys_oracle=len(op), polys_identical=sum(tuple(p) in wp for p in polys[0]),
                             hori=len(hori[0]), hori_oracle=len(oh), hori_identical=sum(tuple(b) in wh for b in hori[0]),
                             free=len(free[0]), free_oracle=len(of),
                             free_identical=sum(tuple(np.asarray(b, dtype=np.float64).reshape(-1).tolist()) in wf for b in free[0]),
                             same_order=bool([list(map(int, b)) for b in oh] == hori[0]))
                    rows.append(d)
                report[name][prec] = rows
                tot = lambda k: sum(x[k] for x in rows)
                print(f"[{name}] {prec:5s}: flips text>0.4 {tot('flips_text_gt_0.4')}, link>0.4 {tot('flips_link_gt_0.4')}, text>=0.7 {tot('flips_text_ge_0.7')} "
                      f"of {tot('pixels')} px; max err near thresholds {max(x['max_err_text_near_thresholds'] for x in rows):.5f} / "
                      f"{max(x['max_err_link_near_threshold'] for x in rows):.5f} (anywhere {max(x['max_err_text'] for x in rows):.4f} / {max(x['max_err_link'] for x in rows):.4f}); "
                      f"polys identical {tot('polys_identical')}/{tot('polys_oracle')}, hori {tot('hori_identical')}/{tot('hori_oracle')}, "
                      f"free {tot('free_identical')}/{tot('free_oracle')}", flush=True)
            finally:
                r.close()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
