#!/usr/bin/env python
"""Regenerates the committed golden vectors under tests/golden/ (run from the repo root, CPU only).

Two kinds of data:
  * reference_pairs.json -- the (input image -> joined EasyOCR text) pairs the REFERENCE repo holds for this path
    (pipeline_components/img_to_json/ocr_testing/results/json/ocr_comparison_*.json:4-8).  They need
    the real craft_mlt_25k.pth / english_g2.pth and are replayed by tests/test_golden_replay.py when BBOCR_WEIGHTS_DIR is
    set.  The five pre-processed input images are committed under ref_images/ (data files of the reference's own tests).
  * legacy_preprocess/book{1,2,4,5,6}.png -- the reference's pipeline_components/books/dataset/*.png, i.e. the INPUTS whose
    stored outputs are ref_images/book*_preprocessed.png (legacy preprocess_for_book_cover, image_preprocessor.py:221-252):
    five (input, output) vectors that pin the oracle's pre-processing stages (tests/test_oracle_cpu.py::test_legacy_preprocess_fixtures).
    Copied, not generated: `cp` from /root/reference (see copy_reference_images below).
  * photos/IMG_968{4,5}.JPG -- pipeline_demo/books/2a/*.JPG, the inputs of ocr_comparison_IMG_968{4,5}.json (copied as data).
  * crnn_synth_fp16.npz -- NOT made here: tests/golden/train_crnn.py (recogniser trained on the synthetic pages, test infrastructure).
  * oracle_*.npz -- seeded input/output vectors of the CPU oracle (oracle/), so that (a) the oracle cannot drift
    silently and (b) the GPU tests can compare against fixed numbers.  PARITY UNPINNED against real EasyOCR: neither the
    package nor its weights exist offline.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

REF_JSON = "/root/reference/pipeline_components/img_to_json/ocr_testing/results/json"
PAIRS = {  # json name -> (input image relative to the reference root, committed copy or None)
    "ocr_comparison_IMG_9684.json": ("pipeline_demo/books/2a/IMG_9684.JPG", None),
    "ocr_comparison_IMG_9685.json": ("pipeline_demo/books/2a/IMG_9685.JPG", None),
    "ocr_comparison_book1.json": ("pipeline_components/img_to_json/ocr_testing/results/images/book1_preprocessed.png", "ref_images/book1_preprocessed.png"),
    "ocr_comparison_book2.json": ("pipeline_components/img_to_json/ocr_testing/results/images/book2_preprocessed.png", "ref_images/book2_preprocessed.png"),
    "ocr_comparison_book4.json": ("pipeline_components/img_to_json/ocr_testing/results/images/book4_preprocessed.png", "ref_images/book4_preprocessed.png"),
    "ocr_comparison_book5.json": ("pipeline_components/img_to_json/ocr_testing/results/images/book5_preprocessed.png", "ref_images/book5_preprocessed.png"),
    "ocr_comparison_book6.json": ("pipeline_components/img_to_json/ocr_testing/results/images/book6_preprocessed.png", "ref_images/book6_preprocessed.png"),
}


def copy_reference_images():
    """Data files of the reference's own manual tests (inputs and stored outputs), copied verbatim."""
    import shutil

    root = "/root/reference/pipeline_components"
    os.makedirs(os.path.join(HERE, "legacy_preprocess"), exist_ok=True)
    for n in (1, 2, 4, 5, 6):
        shutil.copyfile(f"{root}/books/dataset/book{n}.png", os.path.join(HERE, "legacy_preprocess", f"book{n}.png"))
        shutil.copyfile(f"{root}/img_to_json/ocr_testing/results/images/book{n}_preprocessed.png",
                        os.path.join(HERE, "ref_images", f"book{n}_preprocessed.png"))
    # the two photographs behind ocr_comparison_IMG_968{4,5}.json (BASELINE.json configs[0] names a single book-cover JPEG):
    # real ragged / slanted / huge / tiny components for the box stages, the off-grid 1014x971 canvas, libjpeg's Y plane
    os.makedirs(os.path.join(HERE, "photos"), exist_ok=True)
    for n in ("IMG_9684.JPG", "IMG_9685.JPG"):
        shutil.copyfile(f"/root/reference/pipeline_demo/books/2a/{n}", os.path.join(HERE, "photos", n))


def reference_pairs():
    out = []
    for name, (rel, local) in PAIRS.items():
        with open(os.path.join(REF_JSON, name)) as f:
            d = json.load(f)
        out.append({"source": f"pipeline_components/img_to_json/ocr_testing/results/json/{name}", "image": rel, "committed_copy": local,
                    "easyocr_text": d["easyocr"]["text"], "text_length": d["easyocr"]["text_length"]})
    with open(os.path.join(HERE, "reference_pairs.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)


def oracle_vectors():
    import torch

    import bb_ocr_amd  # noqa: F401
    from bb_ocr_amd import synth, weights
    from oracle import boxes as obox
    from oracle import imgproc, pipeline, recog

    torch.set_num_threads(8)
    rng = np.random.default_rng(2024)
    # --- byte-exact image ops
    src = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    src3 = rng.integers(0, 256, (24, 31, 3), dtype=np.uint8)
    tall = rng.integers(0, 256, (150, 64), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "oracle_imgops.npz"),
                        src=src, lin_64x91=imgproc.resize_linear_u8(src, (91, 64)), lin_18x26=imgproc.resize_linear_u8(src, (26, 18)),
                        src3=src3, lin3_40x50=imgproc.resize_linear_u8(src3, (50, 40)), area_12x15=imgproc.resize_linear_u8(src3[:, :30], (15, 12)),
                        tall=tall, bicubic_28x64=imgproc.pil_resize_bicubic_u8(tall, (28, 64)),
                        gray_bgr=imgproc.gray_from_3ch(src3, "bgr"))
    # --- box extraction on a fixed heat-map
    yy, xx = np.mgrid[0:64, 0:96].astype(np.float32)
    text = np.zeros((64, 96), np.float32)
    link = np.zeros((64, 96), np.float32)
    for (cx, cy, lw, lh, ang) in [(20, 10, 12, 3, 0.0), (60, 12, 18, 3, 0.0), (30, 34, 14, 3, 0.35), (70, 40, 10, 4, -0.5), (48, 56, 30, 3, 0.0)]:
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        text = np.maximum(text, np.clip(1.5 - np.maximum(np.abs(u) / lw, np.abs(v) / lh), 0, 1))
    link[9:12, 30:44] = 0.8
    h, f, polys = obox.detect_from_heatmap(text, link, 1.0)
    np.savez_compressed(os.path.join(HERE, "oracle_boxes.npz"), text=text, link=link, polys=np.array(polys, dtype=np.int32),
                        hori=np.array(h, dtype=np.int64).reshape(-1, 4), free=np.array(f, dtype=np.float64).reshape(-1, 4, 2))
    # --- CTC
    logits = (rng.standard_normal((4, 31, 97)) * 5).astype(np.float32)
    logits[0, :, 0] += 40
    logits[1, 5:12, 17] += 40
    res = recog.predict_from_logits(logits)
    np.savez_compressed(os.path.join(HERE, "oracle_ctc.npz"), logits=logits, texts=np.array([r[0] for r in res]),
                        conf=np.array([float(r[1]) for r in res], dtype=np.float64))
    # --- networks (seeded weights are regenerated from the seed, not stored) and the whole path on one small page
    cs, rs = weights.designed_craft_state(0), weights.synthetic_crnn_state(0)
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    img, words = synth.page(77, width=256, height=128, lines=2, margin=16)
    st, sl, ratio = ref.heatmap(img)
    out = ref.readtext(img)
    x = ((rng.integers(0, 256, (1, 1, 64, 128)).astype(np.float32) / 255.0) - 0.5) / 0.5
    lg = ref._logits(x)
    np.savez_compressed(os.path.join(HERE, "oracle_e2e.npz"), page=img[..., 0], n_words=len(words), heat_text=st.astype(np.float16),
                        heat_link=sl.astype(np.float16), ratio=ratio, boxes=np.array([b for b, _, _ in out], dtype=np.int64),
                        texts=np.array([t for _, t, _ in out]), conf=np.array([c for _, _, c in out], dtype=np.float64),
                        crnn_in=x.astype(np.float16), crnn_logits=lg.astype(np.float32))


if __name__ == "__main__":
    if os.path.isdir(REF_JSON):
        copy_reference_images()
        reference_pairs()
    oracle_vectors()
    print("golden vectors written to", HERE)
