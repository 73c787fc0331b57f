#!/usr/bin/env python
"""Trains the recogniser the parity tests and bench.py load: tests/golden/crnn_synth_fp16.npz.

TEST INFRASTRUCTURE -- never imported by the product (bb-ocr_amd/).  Plain torch: the model is the ORACLE's CRNN
(oracle/nets.py::CRNN, i.e. easyocr/model/vgg_model.py::Model with the english_g2 sizes), the data are lines of the seeded
synthetic pages (bb-ocr_amd/synth.py) cropped exactly as the reference path crops them (oracle/recog.py::get_image_list +
align_collate_one), the loss is CTC.  Why it exists: with RANDOM recogniser weights every box has a near-zero top-2 logit
margin somewhere, so "the bf16 path decodes the same text as the fp32 CPU path" (BASELINE.json north_star; the reference
consumes only the strings, enhanced_extractor.py:520-521) cannot be measured.  A recogniser that actually reads the rendered
words has the margins of a trained model.  The real english_g2.pth is not available offline.

The checkpoint is an ordinary upstream-named state-dict (FeatureExtraction.ConvNet.N.*, SequenceModeling.{0,1}.*, Prediction.*)
stored as fp16 .npz (~7 MB) and enters both the oracle and the HIP backend through the same state-dict loaders as any
checkpoint would (bb_ocr_amd.weights.load_npz_state).

    python tests/golden/train_crnn.py --pages 2500 --budget-s 900          # on a GPU box (torch-ROCm as a trainer), 15 min
    python tests/golden/train_crnn.py --device cpu --iters 20 --pages 8   # plumbing check

Boxes are NOT taken from a detector: a line's box is its ground-truth word extent widened by the margins the oracle's detector
+ group_text_box produce on these pages (measured: left 8-13, right 6-12, top 5-18, bottom 8-14 pixels; the crop therefore shows
slivers of the neighbouring lines, as the real crops do), jittered, and lines are also cut at random word boundaries.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bb_ocr_amd  # noqa: E402,F401  (registers the hyphenated package directory)
from bb_ocr_amd import synth  # noqa: E402
from oracle import imgproc, recog  # noqa: E402

_LETTERS = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
_PUNCT = ".,:;!?-'()&"


def _random_words(rng, n):
    out = []
    for _ in range(n):
        k = int(rng.integers(1, 10))
        w = "".join(_LETTERS[int(rng.integers(0, len(_LETTERS) if rng.random() < 0.3 else 26))] for _ in range(k))
        if rng.random() < 0.2:
            w += _PUNCT[int(rng.integers(0, len(_PUNCT)))]
        out.append(w)
    return out


PAGE_KINDS = {
    # bench.py --config p1 / a4 and the test pages (tests/test_gpu_*.py): same fonts, gaps and pitches
    "p1": dict(width=1280, height=960, lines=24, line_pitch=38, margin=24),
    "a4": dict(width=2480, height=760, lines=20, font_size=20, word_gap=14, line_pitch=31, margin=60),
    "small": dict(width=512, height=320, lines=6, margin=24),
}


def page_samples(seed: int, whole_lines: bool = False):
    """-> [(crop uint8 [64, w], imgW, label)] for one seeded page."""
    rng = np.random.default_rng(seed)
    kind = ("p1", "a4", "small")[int(rng.choice(3, p=[0.6, 0.25, 0.15]))]
    kw = dict(PAGE_KINDS[kind])
    if kind == "small":
        kw["width"] = int(rng.choice([384, 448, 512]))
    vocab = None
    if not whole_lines and rng.random() < 0.5:
        vocab = list(synth._WORDS) + _random_words(rng, 80)
    img, words = synth.page(seed, colour=bool(rng.integers(0, 2)), vocab=vocab, **kw)
    _, grey = imgproc.reformat_input(img)
    lines, cur = [], []
    for w in words:                       # a new line starts where x jumps back
        if cur and w[0] < cur[-1][0]:
            lines.append(cur)
            cur = []
        cur.append(w)
    if cur:
        lines.append(cur)
    out = []
    for ln in lines:
        n = len(ln)
        if whole_lines or rng.random() < (0.2 if kind == "a4" else 0.45):     # (a whole A4 line is ~800 sequential LSTM steps: keep them rare)
            cuts = [0, n]
        else:
            k = int(rng.integers(1, min(4, n) + 1))
            cuts = sorted({0, n, *[int(c) for c in rng.integers(1, max(n, 2), k - 1)]})
            cuts = [c for c in cuts if 0 <= c <= n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            seg = ln[a:b]
            if not seg:
                continue
            gx0, gx1 = min(w[0] for w in seg), max(w[2] for w in seg)
            gy0, gy1 = min(w[1] for w in seg), max(w[3] for w in seg)
            box = [gx0 - int(rng.integers(6, 16)), gx1 + int(rng.integers(4, 15)), gy0 - int(rng.integers(3, 20)), gy1 + int(rng.integers(6, 17))]
            il, max_width = recog.get_image_list([box], [], grey, model_height=64)
            if not il:
                continue
            out.append((np.ascontiguousarray(il[0][1]), int(max_width), " ".join(w[4] for w in seg)))
    return out


def _encode(label):
    return [recog.CHARSET.index(c) + 1 for c in label]


def batches(samples, rng, col_budget):
    by_w = {}
    for i, s in enumerate(samples):
        by_w.setdefault(s[1], []).append(i)
    out = []
    for w, idx in by_w.items():
        idx = list(rng.permutation(idx))
        per = max(1, min(256, col_budget // w))
        out += [idx[i:i + per] for i in range(0, len(idx), per)]
    order = rng.permutation(len(out))
    return [out[i] for i in order]


def to_input(samples, idx):
    imgW = samples[idx[0]][1]
    return np.stack([recog.align_collate_one(samples[i][0], 64, imgW) for i in idx])


def evaluate(model, samples, device, torch, max_n=400):
    """-> (exact-match rate, character error rate, per-box min relative top-2 margins)."""
    model.eval()
    ok = n = errs = chars = 0
    margins = []
    with torch.no_grad():
        for s in samples[:max_n]:
            x = torch.from_numpy(recog.align_collate_one(s[0], 64, s[1])[None]).to(device)
            lg = model(x).float().cpu().numpy()
            text = recog.predict_from_logits(lg)[0][0]
            v = np.sort(lg[0], axis=1)
            margins.append(float(((v[:, -1] - v[:, -2]) / max(np.abs(lg).max(), 1e-30)).min()))
            ok += text == s[2]
            n += 1
            errs += _edit(text, s[2])
            chars += len(s[2])
    model.train()
    return ok / max(n, 1), errs / max(chars, 1), np.array(margins)


def _edit(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[-1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--iters", type=int, default=6000)
    ap.add_argument("--pages", type=int, default=1200, help="training pages rendered up front (each ~30-60 line crops)")
    ap.add_argument("--workers", type=int, default=12)
    ap.add_argument("--cols", type=int, default=120000, help="pixel columns per batch (batch size = cols / imgW, at most 256; the native LSTM "
                                                             "is launch-bound, so wide batches are nearly free)")
    ap.add_argument("--init", default=None, help="continue from this .npz state-dict")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--miopen", action="store_true", help="leave MIOpen enabled (measured on a fresh MI355X box: it compiles kernels per "
                                                          "shape, 21 s for the first iteration; torch's native conv / LSTM kernels need none)")
    ap.add_argument("--out", default=os.path.join(HERE, "crnn_synth_fp16.npz"))
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--budget-s", type=float, default=0.0, help="stop training after this many seconds (0 = run --iters)")
    args = ap.parse_args()

    t0 = time.time()
    seeds = [10_000_000 + args.seed * 100_000 + i for i in range(args.pages)]
    val_seeds = [20_000_000 + i for i in range(max(4, args.pages // 100))]
    if args.workers > 1:            # fork BEFORE anything touches the GPU
        import multiprocessing as mp

        with mp.get_context("fork").Pool(args.workers) as pool:
            parts = pool.map(page_samples, seeds, chunksize=4)
            vparts = pool.starmap(page_samples, [(s, True) for s in val_seeds])
    else:
        parts = [page_samples(s) for s in seeds]
        vparts = [page_samples(s, True) for s in val_seeds]
    samples = [s for p in parts for s in p]
    val = [s for p in vparts for s in p]
    print(f"[{time.time() - t0:6.1f}s] {len(samples)} training crops from {len(seeds)} pages, {len(val)} validation lines; "
          f"imgW {min(s[1] for s in samples)}..{max(s[1] for s in samples)}", flush=True)

    import torch
    import torch.nn.functional as F

    from oracle import nets

    if not args.miopen:
        torch.backends.cudnn.enabled = False
    torch.manual_seed(args.seed)
    device = torch.device(args.device)
    model = nets.CRNN()
    if args.init:
        with np.load(args.init) as z:
            nets.load_state_dict_any(model, {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files})
    model = model.to(device).train()

    def save(path):
        sd = {k: v.detach().float().cpu().numpy().astype(np.float16) for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez_compressed(path, **sd)
        return len(sd)

    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=1e-5)
    warm = min(200, max(1, args.iters // 10))

    def lr_at(it, elapsed):      # linear warm-up, then cosine to 2 % over --iters or --budget-s, whichever ends first
        if it < warm:
            return args.lr * (it + 1) / warm
        prog = (it - warm) / max(1, args.iters - warm)
        if args.budget_s:
            prog = max(prog, elapsed / args.budget_s)
        return args.lr * (0.02 + 0.98 * 0.5 * (1 + np.cos(np.pi * min(1.0, prog))))

    rng = np.random.default_rng(args.seed)
    it, t_train, run_loss = 0, time.time(), None
    done = False
    while not done:
        for idx in batches(samples, rng, args.cols):
            x = torch.from_numpy(to_input(samples, idx)).to(device)
            tgt = [_encode(samples[i][2]) for i in idx]
            logits = model(x)                                      # [B, T, 97]
            lp = F.log_softmax(logits.float(), dim=2).permute(1, 0, 2)
            T = lp.shape[0]
            loss = F.ctc_loss(lp, torch.tensor([c for t in tgt for c in t], dtype=torch.long), torch.full((len(idx),), T, dtype=torch.long),
                              torch.tensor([len(t) for t in tgt], dtype=torch.long), blank=0, zero_infinity=True)
            lr = lr_at(it, time.time() - t_train)
            for gparam in opt.param_groups:
                gparam["lr"] = lr
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
            opt.step()
            lv = float(loss.item())
            run_loss = lv if run_loss is None else 0.95 * run_loss + 0.05 * lv
            it += 1
            if it % args.log_every == 0 or it == 1:
                print(f"[{time.time() - t0:6.1f}s] it {it:5d}  loss {lv:8.4f}  (ema {run_loss:8.4f})  lr {lr:.2e}  "
                      f"B {len(idx)} x W {samples[idx[0]][1]}  {(time.time() - t_train) / it * 1e3:.0f} ms/it", flush=True)
            if it % 500 == 0:
                save(args.out)
                acc, cer, mg = evaluate(model, val, device, torch, 100)
                print(f"[{time.time() - t0:6.1f}s]   val: exact lines {acc:.3f}  CER {cer:.4f}  min-margin median {np.median(mg):.3f} "
                      f"p05 {np.quantile(mg, 0.05):.4f}", flush=True)
            if it >= args.iters or (args.budget_s and time.time() - t_train > args.budget_s):
                done = True
                break
    acc, cer, mg = evaluate(model, val, device, torch, 400)
    print(f"[{time.time() - t0:6.1f}s] final val ({min(len(val), 400)} lines): exact {acc:.4f}  CER {cer:.5f}  min-margin quantiles "
          f"p01 {np.quantile(mg, 0.01):.4f} p05 {np.quantile(mg, 0.05):.4f} p50 {np.median(mg):.4f}; boxes with margin < 6e-2: {np.mean(mg < 6e-2):.3f}, "
          f"< 8e-3: {np.mean(mg < 8e-3):.3f}", flush=True)
    n = save(args.out)
    print(f"[{time.time() - t0:6.1f}s] wrote {args.out} ({os.path.getsize(args.out) / 1e6:.2f} MB, {n} tensors, fp16)", flush=True)


if __name__ == "__main__":
    main()
