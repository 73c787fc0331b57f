"""Text / box identity against the fp32 CPU oracle with a recogniser that READS the pages (tests/golden/crnn_synth_fp16.npz, trained by
tests/golden/train_crnn.py on the synthetic pages): hard mismatch counts per precision mode, no margin escape hatch at box level.

north_star: "decoded text strings and box indices bit-identical to the reference CPU EasyOCR path"; the reference consumes only the
strings (enhanced_extractor.py:520-521).  The >= 2,000-box measurement behind the bounds is tools/parity_sweep.py
(profiles/r03_text_parity.json)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import TRAINED_CRNN

# boxes (of the full-page test below) whose decoded text may differ from the oracle's, per mode.  0 for the benchmarked default and for
# `exact`; tools/parity_sweep.py measured the others on 2,000+ boxes (profiles/r03_text_parity.json).
MAX_TEXT_MISMATCH = {"bf16": 1, "mixed": 0, "fp16": 0, "exact": 0, "exact_rec": 0}      # sweep: bf16 1 of 2,051; mixed / fp16 / exact 0 of 2,051
# relative confidence error on boxes whose text agrees (max, median).  The confidence is custom_mean = prod(p_t over the non-blank steps)
# ** (2 / sqrt(n)): ONE arg-max flip at a blank <-> character transition (p ~ 0.5 on both sides; the text does not change) adds or
# removes a factor ~0.5 ** (2 / sqrt(n)) -- 18 % on a 50-step line, more on short words -- so only `exact` reproduces it closely
CONF_BOUND = {"bf16": (1.0, 0.02), "mixed": (0.3, 0.004), "fp16": (0.3, 0.004), "exact": (1e-3, 1e-4), "exact_rec": (1e-3, 1e-4)}
# per time step: an arg-max may differ from the oracle's only where the ORACLE's own top-2 margin at THAT step (relative to the largest
# |logit| of the crop) is below the mode's logit noise (compare per time step, not per box)
STEP_MARGIN_BOUND = {"bf16": 6e-2, "mixed": 8e-3, "fp16": 8e-3, "exact": 1e-4, "exact_rec": 1e-4}


def _bench_pages(n, first=0):
    from bb_ocr_amd import synth

    kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24)
    return [synth.page(1234 + first + i, colour=bool((first + i) & 1), **kw) for i in range(n)]


def _same_box(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))


def test_trained_checkpoint_reads_the_synthetic_page(oracle_trained):
    """CPU (no GPU needed): the committed checkpoint is a recogniser, not noise -- through the ORACLE it reads the words rendered on a small
    page, with confidences far above the contrast-retry threshold.  Pins the fixture the GPU parity tests rest on."""
    from bb_ocr_amd import synth

    assert os.path.getsize(TRAINED_CRNN) < 9e6
    img, words = synth.page(4242, width=512, height=320, lines=6, margin=24)
    res = oracle_trained.readtext(img)
    have = " ".join(t for _, t, _ in res).split()
    hit = sum(w[4] in have for w in words)
    assert len(res) >= 4 and hit >= 0.9 * len(words), (hit, len(words), have)
    assert min(float(c) for _, _, c in res) > 0.3


@pytest.mark.gpu
def test_text_and_boxes_identical_on_full_size_pages(readers_trained, oracle_trained):
    """Four full-size 1280x960 bench pages (the oracle needs ~2.6 s each): every mode returns the oracle's boxes, exactly; decoded texts
    differ on at most MAX_TEXT_MISMATCH[mode] boxes -- 0 in the benchmarked default mode (`fp16`), in `mixed` and in `exact`; confidences
    within CONF_BOUND."""
    pages = _bench_pages(4)
    want = [oracle_trained.readtext(p[0]) for p in pages]
    n_boxes = sum(len(w) for w in want)
    assert n_boxes >= 100
    rgb = torch.from_numpy(np.stack([p[0] for p in pages])).cuda()
    report = {}
    for mode in ("bf16", "mixed", "fp16", "exact", "exact_rec"):      # exact_rec: fp16 detector + split-fp16 recogniser (confidences as exact's)
        got = readers_trained[mode].readtext_device(rgb)
        bad_text, conf_err = [], []
        for pw, pg in zip(want, got):
            assert len(pw) == len(pg), mode
            for w, g in zip(pw, pg):
                assert _same_box(w[0], g[0]), (mode, w[0], g[0])
                if w[1] != g[1]:
                    bad_text.append((w[1], g[1]))
                else:
                    conf_err.append(abs(float(w[2]) - g[2]) / max(float(w[2]), 1e-3))
        report[mode] = (len(bad_text), float(np.max(conf_err)), float(np.median(conf_err)))
        assert len(bad_text) <= MAX_TEXT_MISMATCH[mode], (mode, bad_text[:4])
        assert np.max(conf_err) <= CONF_BOUND[mode][0] and np.median(conf_err) <= CONF_BOUND[mode][1], (mode, report[mode])
    print(f"{n_boxes} boxes; per mode (boxes whose text differs from the fp32 oracle, max / median relative confidence error): {report}")


@pytest.mark.gpu
def test_argmax_flips_only_where_the_oracle_margin_is_below_the_noise(readers_trained, oracle_trained):
    """Recogniser alone on the crops of one page, per TIME STEP: wherever a mode's arg-max differs from the fp32 oracle's, the oracle's own
    top-2 margin at that step must be below the mode's logit noise bound -- a flip anywhere else is a kernel bug, not rounding."""
    from oracle import imgproc, recog

    img = _bench_pages(1, first=9)[0][0]
    _, grey = imgproc.reformat_input(img)
    hori, free = oracle_trained.detect(img)
    crops = {}
    for b in hori:
        il, mw = recog.get_image_list([b], [], grey, model_height=64)
        if il:
            crops.setdefault(int(mw), []).append(il[0][1])
    flips = {m: 0 for m in STEP_MARGIN_BOUND}
    steps = 0
    for W, lst in sorted(crops.items()):
        x = np.stack([recog.align_collate_one(c, 64, W)[0] for c in lst])              # [n, 64, W] fp32 in [-1, 1]
        ref = oracle_trained._logits(x[:, None])
        srt = np.sort(ref, axis=2)
        margin = (srt[..., -1] - srt[..., -2]) / np.abs(ref).max(axis=(1, 2), keepdims=True)[..., 0]
        steps += margin.size
        g = np.rint((x * 0.5 + 0.5) * 255.0).astype(np.int32)                            # the uint8 levels back (exact: x came from them)
        inputs = {"bf16": torch.from_numpy(x).to(torch.bfloat16), "mixed": torch.from_numpy(x).to(torch.float16),
                  "fp16": torch.from_numpy(x).to(torch.float16), "exact": torch.from_numpy((g + 1).astype(np.int16)),
                  "exact_rec": torch.from_numpy((g + 1).astype(np.int16))}
        for mode, t in inputs.items():
            r = readers_trained[mode]
            T = W // 4 - 1
            out = torch.zeros((len(lst), T, 112), dtype=torch.float32, device="cuda")
            dev = t.contiguous().cuda()
            torch.cuda.synchronize()
            r._check(r._lib.bbocr_crnn_logits(r._h, C.c_void_p(dev.data_ptr()), len(lst), W, C.c_void_p(out.data_ptr())))
            got = out.cpu().numpy()[:, :, :97]
            diff = got.argmax(-1) != ref.argmax(-1)
            flips[mode] += int(diff.sum())
            assert not (diff & (margin >= STEP_MARGIN_BOUND[mode])).any(), \
                f"{mode}: arg-max differs at a step whose oracle margin is {margin[diff].max():.3e} >= {STEP_MARGIN_BOUND[mode]:.0e}"
    print(f"{steps} time steps; arg-max flips against the fp32 oracle per mode: {flips}")
    assert flips["exact"] == 0 and flips["exact_rec"] == 0


@pytest.mark.gpu
def test_batched_entry_equals_single_pages_in_every_mode(readers_trained):
    """the batched entry (early/resume recogniser path, wide-image gather, pooled sequence stage) must return exactly what the
    single-page calls return -- in the default bf16 mode too, not only in exact mode."""
    pages = [p[0] for p in _bench_pages(3, first=20)]
    from bb_ocr_amd import synth

    small = [synth.page(300 + i, width=448, height=288, lines=6, margin=24, colour=bool(i & 1))[0] for i in range(3)]
    for mode in ("bf16", "mixed"):
        r = readers_trained[mode]
        for group in (pages, small):
            single = [r.readtext(im) for im in group]
            assert r.readtext_batched(group) == single, mode
            assert all(len(s) >= 4 for s in single)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REAL_IMAGES = ["photos/IMG_9685.JPG", "photos/IMG_9684.JPG",                      # the inputs of ocr_comparison_IMG_968{4,5}.json
               "ref_images/book1_preprocessed.png", "ref_images/book2_preprocessed.png", "ref_images/book4_preprocessed.png",
               "ref_images/book5_preprocessed.png", "ref_images/book6_preprocessed.png"]   # ... of ocr_comparison_book{1,2,4,5,6}.json (grey PNGs)


@pytest.fixture(scope="module")
def hook_readers(states_trained, tmp_path_factory):
    """The reference's own construction lines, once per precision: `import easyocr` (enhanced_extractor.py:19) after bb_ocr_amd.install(),
    `easyocr.Reader(["en"], gpu=use_gpu)` (:153) with the weights found as checkpoint files where upstream keeps them (BBOCR_WEIGHTS_DIR)
    and the precision taken from the environment -- the call site passes neither."""
    import sys

    import bb_ocr_amd

    d = tmp_path_factory.mktemp("weights")
    cs, rs = states_trained
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in cs.items()}, os.path.join(str(d), "craft_mlt_25k.pth"))
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}, os.path.join(str(d), "english_g2.pth"))
    prev = sys.modules.get("easyocr")
    saved = {k: os.environ.get(k) for k in ("BBOCR_WEIGHTS_DIR", "BBOCR_PRECISION")}
    readers = {}
    try:
        os.environ["BBOCR_WEIGHTS_DIR"] = str(d)
        bb_ocr_amd.install()
        import easyocr                                                       # enhanced_extractor.py:19

        for precision in ("fp16", "exact", "mixed"):
            os.environ["BBOCR_PRECISION"] = precision
            readers[precision] = easyocr.Reader(["en"], gpu=True)           # :153
        yield readers
    finally:
        for r in readers.values():
            r.close()
        bb_ocr_amd.uninstall()
        assert sys.modules.get("easyocr") is prev
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("name", REAL_IMAGES)
def test_reference_photographs_through_the_zero_edit_hook(name, hook_readers, oracle_trained):
    """BASELINE.json configs[0]: a real book-cover JPEG through the reference's OWN lines -- `import easyocr` (enhanced_extractor.py:19),
    `easyocr.Reader(["en"], gpu=use_gpu)` (:153, the `hook_readers` fixture), `reader.readtext(path, paragraph=False, batch_size=1,
    workers=0)` (:520), `" ".join(r[1] for r in results)` (:521) -- after bb_ocr_amd.install().  The photographs (pipeline_demo/books/2a,
    inputs of ocr_comparison_IMG_968{4,5}.json) give the box stages what the PIL-font pages never do: ragged / slanted / huge / tiny
    components, free (rotated) boxes, the off-grid 1014x971 canvas, libjpeg's Y plane as the grey image; the five pre-processed book covers
    are the images the reference itself fed to EasyOCR for ocr_comparison_book*.json (single-channel PNGs, 322x439 .. 1050x1312:
    CLAHE-sharpened real print, every size off the 32-pixel grid).  Against OracleReader.readtext(path): boxes identical (grouped and free),
    texts identical, joined strings equal in the fp16 and exact modes (fp16 detector); the mixed mode (bf16 detector) is measured and bounded."""
    path = os.path.join(GOLDEN, name)
    want = oracle_trained.readtext(path)
    n_free = sum(not isinstance(w[0][0][0], (int, np.integer)) for w in want)          # free boxes keep upstream's float corners
    assert len(want) >= 3
    report = {}
    for precision in ("fp16", "exact", "mixed"):
        results = hook_readers[precision].readtext(path, paragraph=False, batch_size=1, workers=0)      # :520
        text = " ".join(r[1] for r in results)                      # :521
        if precision == "mixed":
            # bf16 DETECTOR on continuous-tone content: a few of the ~1e6 threshold decisions sit inside bf16's 0.008 heat-map error
            # (tools/flip_report.py: 8 flips on the two photographs), so a component can gain / lose a pixel row or drop below the 0.7
            # peak test.  Measured: 6 of these 7 images identical, 3 boxes of 37 differ on book6.  Reported, bounded, not hidden.
            wb = {tuple(np.asarray(w[0], dtype=np.float64).reshape(-1).tolist()): w[1] for w in want}
            same = sum(wb.get(tuple(np.asarray(g[0], dtype=np.float64).reshape(-1).tolist())) == g[1] for g in results)
            report[precision] = f"{same}/{len(want)} boxes with identical coordinates and text"
            assert abs(len(results) - len(want)) <= 2 and same >= 0.9 * len(want), (precision, report)
            continue
        assert len(results) == len(want), precision
        for g, w in zip(results, want):
            assert _same_box(g[0], w[0]), (precision, g[0], w[0])
        assert [g[1] for g in results] == [w[1] for w in want], precision
        assert text == " ".join(w[1] for w in want)
        if precision == "exact":
            assert all(abs(g[2] - float(w[2])) <= 1e-3 * max(float(w[2]), 1e-3) for g, w in zip(results, want))
        report[precision] = "identical"
    print(f"{name}: {len(want)} boxes ({n_free} free / rotated) vs the oracle: {report}")


@pytest.mark.gpu
def test_low_confidence_pages_with_the_contrast_retry_live(readers_trained, oracle_trained):
    """upstream's retry (recognition.get_text: boxes with conf < contrast_ths = 0.1 are contrast-stretched, read again, the
    better reading kept) reads the CONFIDENCE.  On faint-ink pages (synth.page(faint=0.5): half of the lines a few grey levels from the paper
    in the gray plane) a third of the boxes take that branch.  `exact` reproduces the oracle completely -- boxes, retry decisions, final texts,
    confidences; the default `fp16` mode returns the oracle's boxes, its retry DECISIONS may differ only where the oracle's first-pass
    confidence lies within the mode's confidence error of 0.1, and its texts differ on at most a few of the (unreadable) faint boxes."""
    from bb_ocr_amd import synth
    from oracle import imgproc

    kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24, faint=0.5)
    pages = [synth.page(9000 + i, **kw)[0] for i in range(2)]
    rgb = torch.from_numpy(np.stack(pages)).cuda()
    want, want1 = [], []
    for p in pages:
        img, grey = imgproc.reformat_input(p)
        h, f = oracle_trained.detect(img)                                       # one detector pass on the CPU, two recogniser passes
        want.append(oracle_trained.recognize(grey, h, f))
        want1.append(oracle_trained.recognize(grey, h, f, contrast_ths=0.0))    # first-pass confidences: the retry disabled
    n = sum(len(w) for w in want)
    low = sum(float(b[2]) < 0.1 for w in want1 for b in w)
    assert n >= 50 and low >= 0.2 * n, (n, low)                                 # the retry branch is live on >= 20 % of the boxes
    report = {}
    for mode in ("fp16", "exact"):
        r = readers_trained[mode]
        got = r.readtext_device(rgb)
        assert r.stage_times()["contrast_retry"] > 0
        got1 = r.readtext_device(rgb, contrast_ths=0.0)
        dec_diff = text_diff = 0
        for pw, pw1, pg, pg1 in zip(want, want1, got, got1):
            assert len(pw) == len(pg) == len(pg1), mode
            for w, w1, g, g1 in zip(pw, pw1, pg, pg1):
                assert _same_box(w[0], g[0]), mode
                cw, cg = float(w1[2]), float(g1[2])
                if (cw < 0.1) != (cg < 0.1):
                    dec_diff += 1
                    assert mode != "exact" and abs(cw - 0.1) <= 0.3 * 0.1, (mode, cw, cg)    # only where the oracle itself sits at the threshold (CONF_BOUND)
                text_diff += w[1] != g[1]
                if mode == "exact":
                    assert w[1] == g[1] and abs(float(w[2]) - g[2]) <= 1e-3 * max(float(w[2]), 1e-3)
        report[mode] = {"retry_decisions_differing": dec_diff, "texts_differing": text_diff}
        if mode == "fp16":
            assert dec_diff <= max(1, n // 50) and text_diff <= max(2, n // 25), report
    print(f"{n} boxes, {low} under contrast_ths on the first pass: {report}")


@pytest.mark.gpu
def test_default_mode_equals_the_oracle_on_sixteen_more_pages(readers_trained, oracle_trained):
    """a wider driver-visible comparison of the LIBRARY DEFAULT (fp16) with the fp32 CPU oracle: 16 further full-size bench
    pages (disjoint from the four above and from bench.py's parity_in_run pages), ~45 s of oracle time: every box and every decoded string
    identical, zero tolerance."""
    pages = _bench_pages(16, first=100)
    rgb = torch.from_numpy(np.stack([p[0] for p in pages])).cuda()
    got = readers_trained["fp16"].readtext_device(rgb)
    n = bad_box = bad_text = 0
    for (img, _), pg in zip(pages, got):
        pw = oracle_trained.readtext(img)
        assert len(pw) == len(pg)
        n += len(pw)
        bad_box += sum(not _same_box(w[0], g[0]) for w, g in zip(pw, pg))
        bad_text += sum(w[1] != g[1] for w, g in zip(pw, pg))
    print(f"fp16 (library default) vs the fp32 oracle on 16 pages: {n} boxes, {bad_box} boxes / {bad_text} texts differ")
    assert n > 400 and bad_box == 0 and bad_text == 0
