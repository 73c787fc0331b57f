"""-m gpu: stage and end-to-end parity of the HIP path (through the C ABI) against the oracle."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HEAT_TOL = 0.03   # max |heat - oracle fp32| on designed-weight pages (bf16 storage, fp32 accumulate; values span 0..6)


def _synth_heat(rng, h, w, nblobs=26):
    """Region/affinity maps with axis-aligned and rotated strokes, touching image borders, tiny specks and link bridges."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    text = np.zeros((h, w), np.float32)
    link = np.zeros((h, w), np.float32)
    for _ in range(nblobs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        lw, lh = rng.uniform(3, 20), rng.uniform(2, 5)
        ang = rng.choice([0.0, 0.0, 0.0, rng.uniform(-0.6, 0.6)])
        ca, sa = math.cos(ang), math.sin(ang)
        u = (xx - cx) * ca + (yy - cy) * sa
        v = -(xx - cx) * sa + (yy - cy) * ca
        d = np.maximum(np.abs(u) / lw, np.abs(v) / lh)
        amp = rng.uniform(0.75, 1.3)
        text = np.maximum(text, amp * np.clip(1.4 - d, 0, 1))
        if rng.random() < 0.5:
            link = np.maximum(link, 0.9 * np.clip(1.6 - np.maximum(np.abs(u - lw) / (0.8 * lw), np.abs(v) / lh), 0, 1))
    text += rng.normal(0, 0.01, text.shape).astype(np.float32)
    return text.astype(np.float32), link.astype(np.float32)


def test_boxes_bit_exact_given_heatmap(reader):
    from oracle import boxes as obox

    rng = np.random.default_rng(11)
    B, h, w = 3, 160, 288
    heat = np.zeros((B, h, w, 2), np.float32)
    for b in range(B):
        heat[b, ..., 0], heat[b, ..., 1] = _synth_heat(rng, h, w)
    heat[2] = 0.0                                    # empty page
    d = torch.from_numpy(heat).cuda()
    for ratio, kw in [(1.0, {}), (0.7306, {"min_size": 10, "add_margin": 0.2})]:
        hori, free, polys = reader.boxes_from_heatmap(d, ratio, **kw)
        nfree = 0
        for b in range(B):
            oh, of, op = obox.detect_from_heatmap(heat[b, ..., 0], heat[b, ..., 1], ratio, **kw)
            assert [list(map(int, p)) for p in op] == polys[b]
            assert [list(map(int, x)) for x in oh] == hori[b]
            assert len(of) == len(free[b])
            for a, g in zip(of, free[b]):
                assert np.allclose(np.array(a, dtype=np.float64), np.array(g), rtol=0, atol=1e-9)
            nfree += len(of)
        assert len(polys[0]) > 5 and polys[2] == []
    assert nfree > 0, "the fixture must exercise the free (slanted) box branch"


def test_boxes_pathological_heatmaps(reader):
    """Edge cases of S4/S5: a lattice of minimum-size components (more components than any page of text has: the buffers are sized for the
    theoretical maximum h*w/10), one-pixel-wide strokes (one row extent per pixel), salt-and-pepper noise (thousands of rejected
    sub-10-pixel components) and a fully saturated map (ONE component covering the page)."""
    from oracle import boxes as obox

    rng = np.random.default_rng(23)
    h, w = 96, 128
    heat = np.zeros((4, h, w, 2), np.float32)
    for y in range(0, h - 3, 4):                      # 4x3-pixel blocks, one pixel apart: 12 pixels each
        for x in range(0, w - 4, 5):
            heat[0, y:y + 3, x:x + 4, 0] = 0.9
    heat[1, 4:h - 4, 3:w - 3:3, 0] = 0.95             # vertical hairlines
    heat[2, ..., 0] = (rng.random((h, w)) > 0.6) * 0.9
    heat[2, ..., 1] = (rng.random((h, w)) > 0.9) * 0.9
    heat[3, ..., 0] = 1.0
    d = torch.from_numpy(heat).cuda()
    hori, free, polys = reader.boxes_from_heatmap(d, 1.0)
    for b in range(4):
        oh, of, op = obox.detect_from_heatmap(heat[b, ..., 0], heat[b, ..., 1], 1.0)
        assert [list(map(int, p)) for p in op] == polys[b], b
        assert [list(map(int, x)) for x in oh] == hori[b], b
        assert len(of) == len(free[b])
    assert len(polys[0]) == (h // 4) * len(range(0, w - 4, 5)) and len(polys[0]) > h * w // 64    # beyond the old h*w/64 sizing
    assert len(polys[1]) == len(range(3, w - 3, 3)) and len(polys[3]) == 1


def test_heatmap_within_tolerance(reader, oracle_reader):
    from bb_ocr_amd import synth

    imgs = np.stack([synth.page(21 + i, width=384, height=256, lines=5, margin=24)[0] for i in range(3)])
    heat, ratio = reader.heatmap_device(torch.from_numpy(imgs).cuda())
    got = heat.cpu().numpy()
    for i in range(3):
        st, sl, r2 = oracle_reader.heatmap(imgs[i])
        assert ratio == r2
        assert np.abs(got[i, ..., 0] - st).max() <= HEAT_TOL
        assert np.abs(got[i, ..., 1] - sl).max() <= HEAT_TOL * 2
        assert st.max() > 0.7


def test_heatmap_random_weights_relative(reader):
    """Fully random detector (no designed channels): relative L2 error of the heat-map vs torch fp32."""
    import bb_ocr_amd
    from bb_ocr_amd import synth, weights
    from oracle import pipeline

    cs, rs = weights.synthetic_craft_state(3), weights.synthetic_crnn_state(3)
    r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision="bf16")
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    img = synth.page(5, width=352, height=224, lines=4, margin=24)[0]       # 224 = 7*32: odd tile counts at every level
    heat, _ = r.heatmap_device(torch.from_numpy(img[None]).cuda())
    st, sl, _ = ref.heatmap(img)
    got = heat[0].cpu().numpy()
    want = np.stack([st, sl], -1)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 4e-2, rel      # 27 bf16-stored layers (each ~2^-9 relative rounding), fp32 accumulate
    r.close()


def test_colour_inputs_gray_plane_and_every_input_form(reader, oracle_reader, tmp_path):
    """a2 on the device: (1) gray_kernel on genuinely coloured pixels, bit-exact against the oracle's cv2 BGR2GRAY (15-bit, channels as
    given); (2) every input form reformat_input accepts -- RGB / RGBA / gray arrays, encoded bytes, a PIL image, JPEG and PNG paths --
    gives exactly what the same page gives when the oracle's (colour array, gray plane) pair is handed to the device directly, i.e.
    decode rule + device gray plane == oracle's reformat_input; (3) a coloured page end to end: boxes identical to the oracle's."""
    import io

    from PIL import Image

    from bb_ocr_amd import synth
    from oracle import imgproc

    rng = np.random.default_rng(31)
    for shape in ((37, 53, 3), (240, 321, 3), (1, 1, 3)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        src = torch.from_numpy(a).cuda()
        dst = torch.empty(shape[:2], dtype=torch.uint8, device="cuda")
        reader._check(reader._lib.bbocr_op_preprocess_stage(reader._h, 6, C.c_void_p(src.data_ptr()), shape[0], shape[1],
                                                            C.c_void_p(dst.data_ptr()), shape[0], shape[1], 0.0))
        assert np.array_equal(dst.cpu().numpy(), imgproc.gray_from_3ch(a, "bgr"))
    page = synth.page(61, width=448, height=256, lines=5, margin=24, colour=True)[0]
    assert (page[..., 0] != page[..., 1]).mean() > 0.9 and (page[..., 2] != page[..., 1]).mean() > 0.9
    buf = io.BytesIO()
    Image.fromarray(page).save(buf, format="PNG")
    jpg, png = str(tmp_path / "p.jpg"), str(tmp_path / "p.png")
    Image.fromarray(page).save(jpg, quality=95)
    Image.fromarray(page).save(png)
    rgba = np.dstack([page, np.full(page.shape[:2], 255, np.uint8)])
    forms = {"rgb": page, "rgba": rgba, "gray": np.ascontiguousarray(page[..., 1]), "bytes": buf.getvalue(), "pil": Image.fromarray(page),
             "jpeg path": jpg, "png path": png}
    for name, x in forms.items():
        ea, eg = imgproc.reformat_input(x)
        want = reader.readtext_arrays(np.ascontiguousarray(ea)[None], np.ascontiguousarray(eg)[None])[0]
        got = reader.readtext(x)
        assert got == want, name
        assert len(got) >= 4, name
    got = reader.readtext(page)
    want = oracle_reader.readtext(page)
    assert [g[0] for g in got] == [[list(map(int, p)) for p in w[0]] for w in want]


def test_reader_from_checkpoint_directory(states, reader, tmp_path):
    """f1: Reader(model_storage_directory=...) reads craft_mlt_25k.pth / english_g2.pth (DataParallel ``module.`` prefixes and
    num_batches_tracked entries; ``{"state_dict": ...}`` wrapping) and returns exactly what Reader(weights=(cs, rs)) returns."""
    import os

    import bb_ocr_amd
    from bb_ocr_amd import synth

    cs, rs = states
    torch.save({("module." + k): torch.from_numpy(np.asarray(v)) for k, v in cs.items()}, os.path.join(str(tmp_path), "craft_mlt_25k.pth"))
    torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}}, os.path.join(str(tmp_path), "english_g2.pth"))
    r2 = bb_ocr_amd.Reader(["en"], gpu=True, model_storage_directory=str(tmp_path), precision="bf16")
    try:
        for seed in (5, 6):
            img = synth.page(seed, width=384, height=256, lines=5, margin=24, colour=bool(seed & 1))[0]
            assert r2.readtext(img) == reader.readtext(img)
    finally:
        r2.close()
    with pytest.raises(FileNotFoundError):
        bb_ocr_amd.Reader(["en"], gpu=True, model_storage_directory=str(tmp_path / "missing"))


def test_detector_resize_path(reader, oracle_reader):
    """Page larger than canvas_size: cv2-style resize + zero canvas padding before the network."""
    from bb_ocr_amd import synth

    img = synth.page(9, width=500, height=300, lines=5, margin=24)[0]
    heat, ratio = reader.heatmap_device(torch.from_numpy(img[None]).cuda(), canvas_size=320)
    st, sl, r2 = oracle_reader.heatmap(img, canvas_size=320)
    assert ratio == r2 and heat.shape[1:3] == st.shape
    got = heat[0].cpu().numpy()
    assert np.abs(got[..., 0] - st).max() <= HEAT_TOL * 2


def _crops_case(reader, grey, hori, free, contrast):
    from oracle import recog

    H, W = grey.shape
    d = torch.from_numpy(grey).cuda()
    items = []
    for b in hori:
        il, mw = recog.get_image_list([b], [], grey)
        if il:
            items.append((il[0][1], int(mw)))
    for f in free:
        il, mw = recog.get_image_list([], [f], grey)
        if il:
            items.append((il[0][1], int(mw)))
    widths = sorted({mw for _, mw in items})
    harr = (C.c_int * max(1, 4 * len(hori)))(*[int(v) for b in hori for v in b])
    farr = (C.c_double * max(1, 8 * len(free)))(*[float(v) for f in free for p in f for v in p])
    total = 0
    for mw in widths:
        want = [recog.align_collate_one(c, 64, mw, adjust_contrast=contrast) for c, m in items if m == mw]
        out = torch.zeros((len(want), 64, mw), dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the library on its own non-blocking one
        n_out = C.c_int()
        reader._check(reader._lib.bbocr_op_crops(reader._h, C.c_void_p(d.data_ptr()), H, W, harr, len(hori), farr, len(free), mw, float(contrast),
                                                 C.c_void_p(out.data_ptr()), C.byref(n_out), 0))
        assert n_out.value == len(want)
        got = out.cpu()
        ref = torch.from_numpy(np.concatenate(want, 0)).to(torch.bfloat16)
        assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"crop bucket {mw} differs"
        total += len(want)
    return total


def test_crops_bit_exact(reader):
    from bb_ocr_amd import synth

    img = synth.page(33, width=640, height=400, lines=8, margin=30)[0]
    grey = np.ascontiguousarray(img[..., 0])
    hori = [[40, 300, 30, 62], [-5, 200, 70, 100], [300, 660, 380, 410], [100, 420, 150, 171], [50, 62, 40, 120], [200, 232, 100, 300],
            [10, 74, 200, 264], [320, 600, 200, 216]]
    free = [[[100.0, 50.0], [400.0, 80.0], [395.0, 120.0], [95.0, 90.0]], [[300.5, 200.2], [340.0, 190.0], [350.0, 300.0], [310.0, 310.0]]]
    assert _crops_case(reader, grey, hori, free, 0.0) == len(hori) + len(free)
    assert _crops_case(reader, grey, hori, free, 0.5) == len(hori) + len(free)


def test_rotated_crops_bit_exact(reader):
    """rotation_info (f4): the batched branch's recogniser inputs -- every crop at the page's max_width, as is and as np.rot90 copies
    (make_rotated_img_list), through AlignCollate's general PIL bicubic resize (a rotated line shrinks hundreds of rows to 64)."""
    from bb_ocr_amd import synth
    from oracle import recog

    img = synth.page(34, width=900, height=300, lines=5, margin=30)[0]
    grey = np.ascontiguousarray(img[..., 0])
    H, W = grey.shape
    hori = [[40, 300, 30, 62], [30, 870, 70, 100], [100, 420, 150, 171], [50, 62, 40, 120], [10, 74, 200, 264]]
    free = [[[100.0, 50.0], [400.0, 80.0], [395.0, 120.0], [95.0, 90.0]]]
    image_list, max_width = recog.get_image_list(hori, free, grey, sort_output=False)       # free boxes first, then horizontal
    assert len(image_list) == 6 and max_width >= 1792
    d = torch.from_numpy(grey).cuda()
    harr = (C.c_int * (4 * len(hori)))(*[int(v) for b in hori for v in b])
    farr = (C.c_double * (8 * len(free)))(*[float(v) for f in free for p in f for v in p])
    for contrast in (0.0, 0.5):
        for k in range(4):
            crops = [c for _, c in image_list[1:]] + [image_list[0][1]]                          # the op takes horizontal boxes first
            want = [recog.align_collate_one(np.ascontiguousarray(np.rot90(c, k)), 64, max_width, adjust_contrast=contrast) for c in crops]
            out = torch.zeros((len(want), 64, max_width), dtype=torch.bfloat16, device="cuda")
            torch.cuda.synchronize()
            n_out = C.c_int()
            reader._check(reader._lib.bbocr_op_crops(reader._h, C.c_void_p(d.data_ptr()), H, W, harr, len(hori), farr, len(free), max_width,
                                                     float(contrast), C.c_void_p(out.data_ptr()), C.byref(n_out), 1 + k))
            assert n_out.value == len(want)
            ref = torch.from_numpy(np.concatenate(want, 0)).to(torch.bfloat16)
            got = out.cpu()
            for i in range(len(want)):
                assert torch.equal(got[i].view(torch.int16), ref[i].view(torch.int16)), f"rot90 x{k}, contrast {contrast}: crop {i} differs"


def test_crnn_logits_within_tolerance(reader, oracle_reader):
    rng = np.random.default_rng(17)
    for n, W in [(3, 128), (2, 320)]:
        x = (rng.integers(0, 256, (n, 1, 64, W)).astype(np.float32) / 255.0 - 0.5) / 0.5
        # smooth it a little so it looks like text strokes rather than white noise
        x = (x + np.roll(x, 1, 3) + np.roll(x, 1, 2)) / 3.0
        xb = torch.from_numpy(x).to(torch.bfloat16)
        ref = oracle_reader._logits(xb.float().numpy())
        T = W // 4 - 1
        out = torch.zeros((n, T, 112), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the library on its own non-blocking one
        reader._check(reader._lib.bbocr_crnn_logits(reader._h, C.c_void_p(xb[:, 0].contiguous().cuda().data_ptr()), n, W, C.c_void_p(out.data_ptr())))
        got = out.cpu().numpy()[:, :, :97]
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert rel < 3e-2, rel


def test_readtext_end_to_end(reader, oracle_reader):
    """Whole path on designed-weight pages: boxes identical to the oracle's, one result per box, confidences are probabilities.  (The recogniser
    of this fixture has RANDOM weights: what its text must satisfy is asserted per time step, against the oracle's own top-2 margins, in
    tests/test_gpu_precision.py::test_readtext_text_identity_by_mode; text identity with hard counts is tests/test_gpu_parity_trained.py.)"""
    from bb_ocr_amd import synth

    for seed in (101, 102):
        img = synth.page(seed, width=512, height=320, lines=6, margin=24)[0]
        got = reader.readtext(img)
        want = oracle_reader.readtext(img)
        assert [g[0] for g in got] == [[list(map(int, p)) for p in w[0]] for w in want]
        assert len(got) >= 4 and all(isinstance(t, str) and 0.0 <= c <= 1.0 for _, t, c in got)
        # same number of decoded characters give or take the arg-max flips of a margin-less recogniser: the sequence lengths (T per box) agree
        assert all(abs(len(tg) - len(tw)) <= max(4, len(tw) // 2) for (_, tg, _), (_, tw, _) in zip(got, want))


def test_readtext_edge_pages(reader, oracle_reader):
    """Degenerate page shapes through the whole path: smaller than one 32-pixel canvas cell, one pixel, long strips in both directions,
    a black page, a page whose side exceeds canvas_size (down-scaled), and a single word that fills the page edge to edge."""
    from bb_ocr_amd import synth

    word = synth.page(77, width=256, height=64, lines=1, margin=4)[0]
    cases = [np.full((17, 33, 3), 200, np.uint8), np.full((1, 1, 3), 90, np.uint8), np.full((40, 1600, 3), 235, np.uint8),
             np.full((900, 24, 3), 235, np.uint8), np.zeros((96, 128, 3), np.uint8), word,
             np.ascontiguousarray(np.tile(word, (1, 12, 1))[:, :2720])]          # 2720 wide > canvas_size 2560
    for img in cases:
        got = reader.readtext(img)
        want = oracle_reader.readtext(img)
        assert [g[0] for g in got] == [[list(map(int, p)) for p in w[0]] for w in want], img.shape
    assert reader.readtext(word) != []


def test_rotation_info_matches_oracle(reader, reader_exact, oracle_reader):
    """rotation_info (f4) end to end: the same boxes in the same (top-y sorted) order as the oracle's batched branch.  In the default
    bf16 mode which variant wins a box depends on confidences that differ by bf16-vs-fp32 noise with random recogniser weights (the
    inputs of every variant are bit-exact, test_rotated_crops_bit_exact), so there the selection is checked as a property of the
    product: the result for [90, 180, 270] is, box by box, the most confident of the three single-angle results.  The exact mode must
    reproduce the oracle's texts and confidences outright."""
    from bb_ocr_amd import synth

    img = synth.page(322, width=512, height=256, lines=3, margin=24)[0]
    single = {}
    for rot in ([90], [180], [270], [90, 180, 270]):
        got = reader.readtext(img, rotation_info=rot)
        want = oracle_reader.readtext(img, rotation_info=rot)
        assert [g[0] for g in got] == [[list(map(int, p)) for p in w[0]] for w in want]
        single[tuple(rot)] = got
        # the exact recogniser mode picks the same variant as the oracle: same text, confidence to 1e-3
        ex = reader_exact.readtext(img, rotation_info=rot)
        assert [g[0] for g in ex] == [g[0] for g in got] and [g[1] for g in ex] == [w[1] for w in want]
        assert all(abs(g[2] - float(w[2])) <= 1e-3 * max(float(w[2]), 1e-3) for g, w in zip(ex, want))
    for i, box in enumerate(single[(90, 180, 270)]):
        cands = [single[(90,)][i], single[(180,)][i], single[(270,)][i]]
        best = max(cands, key=lambda r: r[2])
        assert box[2] == best[2] and box[1] == best[1]
    plain = reader.readtext(img)
    assert sorted(str(g[0]) for g in single[(180,)]) == sorted(str(p[0]) for p in plain)
    # a page turned upside down: with rotation_info=[180] no box gets a lower confidence than without (the variants only add candidates
    # -- but the batched branch pads to the page width, so compare within that branch: [180] vs [90])
    assert reader.readtext(img[::-1, ::-1].copy(), rotation_info=[180]) is not None


def test_readtext_batched_matches_single(reader):
    from bb_ocr_amd import synth

    imgs = [synth.page(200 + i, width=384, height=256, lines=5, margin=24)[0] for i in range(3)]
    single = [reader.readtext(im) for im in imgs]
    batched = reader.readtext_batched(imgs)
    assert batched == single


def test_error_paths_raise(reader):
    with pytest.raises(ValueError):
        reader.readtext(np.zeros((4, 4), dtype=np.float32))
    with pytest.raises(NotImplementedError):
        reader.readtext(np.zeros((64, 64, 3), dtype=np.uint8), decoder="wordbeamsearch")
    with pytest.raises(ValueError):
        reader.readtext(np.zeros((64, 64, 3), dtype=np.uint8), rotation_info=[45])
    assert reader.readtext(np.full((64, 96, 3), 235, dtype=np.uint8)) == []      # blank page: no boxes, no error
    # tensors that would become out-of-bounds device accesses behind the C ABI are refused in Python (never a GPU fault)
    ok = torch.full((1, 64, 96, 3), 235, dtype=torch.uint8, device="cuda")
    bad = [torch.zeros((1, 64, 96, 3), dtype=torch.uint8),                        # host tensor
           torch.zeros((1, 64, 96, 3), dtype=torch.float32, device="cuda"),      # wrong dtype
           torch.zeros((1, 64, 96, 4), dtype=torch.uint8, device="cuda"),        # wrong channel count
           torch.zeros((1, 64, 192, 3), dtype=torch.uint8, device="cuda")[:, :, ::2],   # not contiguous
           torch.zeros((64, 96, 3), dtype=torch.uint8, device="cuda"), np.zeros((1, 64, 96, 3), np.uint8)]
    for t in bad:
        with pytest.raises(ValueError):
            reader.readtext_device(t)
        with pytest.raises(ValueError):
            reader.heatmap_device(t)
    with pytest.raises(ValueError):
        reader.readtext_device(ok, torch.zeros((1, 64, 95), dtype=torch.uint8, device="cuda"))   # gray plane of another shape
    with pytest.raises(ValueError):
        reader.recognize_device(torch.zeros((1, 64, 96), dtype=torch.uint8), [[]], [[]])
    with pytest.raises(ValueError):
        reader.boxes_from_heatmap(torch.zeros((1, 32, 48, 2), dtype=torch.float16, device="cuda"), 1.0)
    assert reader.readtext_device(ok) == [[]]
    # a context's packed weights (and with them the blob layout its peers import) are fixed: a second load is refused, not appended
    from bb_ocr_amd import weights

    arr, keep = weights.to_descs(weights.synthetic_crnn_state(5))
    with pytest.raises(RuntimeError, match="already loaded"):
        reader._check(reader._lib.bbocr_load_weights(reader._h, 1, arr, len(arr)))
    with pytest.raises(RuntimeError, match="already"):
        reader._check(reader._lib.bbocr_alloc_weights(reader._h, 0))
    with pytest.raises(ValueError):
        __import__("bb_ocr_amd").Reader(["en"], weights="empty", precision="fp8")


def test_hip_path_against_committed_golden(reader):
    """Fixed numbers from tests/golden (written by the oracle): heat-map tolerance, boxes exact, logits tolerance."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_e2e.npz"))
    img = np.repeat(g["page"][:, :, None], 3, 2)
    heat, ratio = reader.heatmap_device(torch.from_numpy(img[None]).cuda())
    got = heat[0].cpu().numpy()
    assert ratio == float(g["ratio"])
    assert np.abs(got[..., 0] - g["heat_text"].astype(np.float32)).max() <= HEAT_TOL
    assert np.abs(got[..., 1] - g["heat_link"].astype(np.float32)).max() <= HEAT_TOL * 2
    out = reader.readtext(img)
    assert np.array_equal(np.array([b for b, _, _ in out], dtype=np.int64), g["boxes"])
    x = torch.from_numpy(g["crnn_in"].astype(np.float32)).to(torch.bfloat16)
    lg = torch.zeros((1, 31, 112), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # the fill runs on torch's stream, the library on its own non-blocking one
    reader._check(reader._lib.bbocr_crnn_logits(reader._h, C.c_void_p(x[:, 0].contiguous().cuda().data_ptr()), 1, 128, C.c_void_p(lg.data_ptr())))
    ref = g["crnn_logits"]
    rel = np.linalg.norm(lg.cpu().numpy()[:, :, :97] - ref) / np.linalg.norm(ref)
    assert rel < 3e-2, rel
    gb = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_boxes.npz"))
    heat2 = torch.from_numpy(np.stack([gb["text"], gb["link"]], -1)[None].astype(np.float32)).cuda()
    hori, free, polys = reader.boxes_from_heatmap(heat2, 1.0)
    assert np.array_equal(np.array(polys[0], dtype=np.int32), gb["polys"]) and np.array_equal(np.array(hori[0], dtype=np.int64).reshape(-1, 4), gb["hori"])


def test_full_size_batch_properties(reader):
    """BASELINE.json's measured configuration (64 pages of 1280x960; 8 distinct pages tiled like bench.py), checked through
    size-independent properties: copies of a page give identical results, a page's result does not depend on the batch it
    travels in, the detector's pass schedule ([56, 8] auto vs uniform passes of 8) does not change a single heat-map bit, and
    a second run reproduces the first."""
    import bb_ocr_amd
    from bb_ocr_amd import synth, weights

    uniq = [synth.page(1000 + i)[0] for i in range(8)]
    assert uniq[0].shape == (960, 1280, 3)
    rgb = torch.from_numpy(np.stack([uniq[i % 8] for i in range(64)])).cuda()
    out = reader.readtext_device(rgb)
    assert len(out) == 64 and all(len(p) > 10 for p in out)
    for i in range(8, 64):
        assert out[i] == out[i % 8], f"copy {i} of page {i % 8} differs"
    assert reader.readtext_device(rgb) == out                                   # run-to-run
    for i in range(8):
        assert reader.readtext_device(rgb[i:i + 1])[0] == out[i], f"page {i}: batch of 1 differs from batch of 64"
    assert reader.readtext_device(rgb[5:30]) == out[5:30]                        # 25 pages: schedule [17, 8]
    heat, ratio = reader.heatmap_device(rgb)
    other = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)), det_sub_batch=8, precision="bf16")
    heat8, ratio8 = other.heatmap_device(rgb)
    assert ratio == ratio8 and torch.equal(heat, heat8)
    # every word the renderer drew is found exactly once on every page (boxes are word-level for this detector)
    words = [len(synth.page(1000 + i)[1]) for i in range(8)]
    polys = reader.boxes_from_heatmap(heat, ratio)[2]
    assert [len(p) for p in polys[:8]] == words


def test_a4_page_batch_properties(reader):
    """BASELINE.json configs[4] shape (dense A4 @300 dpi, 2480x3504 -> canvas 1824x2560, the resize path at full size), through the
    same size-independent properties: the canvas geometry, copies agree, batch of 1 == batch of 4, the drawn words are detected."""
    from bb_ocr_amd import synth

    pages, words = zip(*[synth.page(4242 + i, width=2480, height=3504, lines=80, font_size=26, line_pitch=42, margin=60) for i in range(2)])
    H32, W32, hh, hw, ratio = reader.detect_dims(3504, 2480)
    assert (H32, W32) == (2560, 1824) and (hh, hw) == (1280, 912) and ratio == 2560 / 3504
    rgb = torch.from_numpy(np.stack([pages[0], pages[1], pages[0], pages[1]])).cuda()
    out = reader.readtext_device(rgb)
    assert out[2] == out[0] and out[3] == out[1] and all(len(p) > 60 for p in out)
    assert reader.readtext_device(rgb[1:2])[0] == out[1]
    heat, r2 = reader.heatmap_device(rgb[:2])
    assert tuple(heat.shape) == (2, 1280, 912, 2) and r2 == ratio
    polys = reader.boxes_from_heatmap(heat, r2)[2]
    # at 0.73x the narrowest word gaps close: a handful of the ~1700 words per page merge with a neighbour (the same pairs on every copy)
    assert all(0 <= len(w) - len(p) <= 0.005 * len(w) for p, w in zip(polys, words)), ([len(p) for p in polys], [len(w) for w in words])


def test_extractor_batching_and_tesseract_shim(reader, tmp_path):
    """f3 / a12 on the GPU: batched extraction returns exactly the per-page ``" ".join`` of ``readtext``; the pytesseract shim
    returns one line of text per rendered text line."""
    from PIL import Image

    from bb_ocr_amd import extractor_batch as eb, synth, tesseract_shim as ts

    paths = []
    for i in range(3):
        img, _ = synth.page(300 + i, width=640, height=384, lines=5, margin=24)
        p = tmp_path / f"p{i}.png"
        Image.fromarray(img).save(p)
        paths.append(p)
    for i in range(3):                                  # JPEG pages: decoded ONCE by the batching loop (YCbCr triples, RGB + Y plane derived on the
        img, _ = synth.page(310 + i, width=640, height=384, lines=5, margin=24)       # card), twice by readtext(path) -- same strings
        p = tmp_path / f"j{i}.jpg"
        Image.fromarray(img).save(p, quality=(92, 75, 100)[i], subsampling=(2, 1, 0)[i])
        paths.append(p)
    assert eb._ocr_input(paths[3], 3)[0] == "ycc" and eb._ocr_input(paths[0], 0)[0] == "rgb"
    texts = eb.extract_texts(reader, paths)
    for i, p in enumerate(paths):
        assert texts[i] == " ".join(r[1] for r in reader.readtext(str(p))) and texts[i]
    # Reader.readtext_files: the same pipeline without the extractor's thumbnail rule -- readtext(path)'s result per file, in order
    files = [str(p) for p in paths] + [str(tmp_path / "missing.jpg")]
    per_file = reader.readtext_files(files)
    assert per_file[:-1] == [reader.readtext(f) for f in files[:-1]] and per_file[-1] == []
    from bb_ocr_amd.reader import decode_file, decode_file_ycc
    a, g = decode_file(str(paths[4]))
    want = reader.readtext_arrays(a[None], g[None])
    assert reader.readtext_ycc_arrays(decode_file_ycc(str(paths[4]))[None]) == want       # boxes, strings and confidences
    s = ts.image_to_string(Image.open(paths[0]), reader=reader)
    assert s.endswith("\n") and len(s.strip().split("\n")) == 5


def test_paragraph_and_allowlist_modes(reader):
    """f4 on the GPU path: paragraph=True returns [box, text] groups covering the same words; an allowlist restricts the alphabet."""
    from bb_ocr_amd import synth

    img = synth.page(321, width=640, height=384, lines=5, margin=24)[0]
    plain = reader.readtext(img)
    para = reader.readtext(img, paragraph=True)
    assert len(para) >= 1 and all(len(p) == 2 for p in para)
    assert sorted(" ".join(p[1] for p in para).split()) == sorted(" ".join(r[1] for r in plain).split())
    assert reader.readtext(img, paragraph=True, detail=0) == [p[1] for p in para]
    # Reader.recognize(paragraph=True) on the detector's own boxes: the same paragraphs (upstream applies get_paragraph inside recognize)
    from bb_ocr_amd.reader import reformat_input
    _, grey = reformat_input(img)
    hl, fl = reader.detect(img)
    assert reader.recognize(grey, hl[0], fl[0], paragraph=True, reformat=False) == para
    assert reader.recognize(grey, hl[0], fl[0], paragraph=True, detail=0, reformat=False) == [p[1] for p in para]
    digits = reader.readtext(img, allowlist="0123456789")
    assert [d[0] for d in digits] == [r[0] for r in plain]                        # same boxes
    assert all(set(d[1]) <= set("0123456789") for d in digits) and any(d[1] for d in digits)
    # decoder='beamsearch': same boxes and confidences (upstream scores the greedy path for every decoder), strings from the search
    beam = reader.readtext(img, decoder="beamsearch", beamWidth=5)
    assert [b[0] for b in beam] == [r[0] for r in plain] and [b[2] for b in beam] == [r[2] for r in plain]
    assert reader.readtext(img, decoder="beamsearch", beamWidth=1) is not None
    with pytest.raises(NotImplementedError):
        reader.readtext(img, decoder="wordbeamsearch")


def test_one_reader_called_from_worker_threads(reader):
    """SURVEY 8(b) threading: the reference shares ONE Reader between ThreadPoolExecutor workers (batch_processor_enhanced.py:215) and calls
    it from non-main threads (i2j_ui/app/main.py:757-771): concurrent readtext calls on one context serialise inside the library and
    return what a sequential run returns."""
    from concurrent.futures import ThreadPoolExecutor

    from bb_ocr_amd import synth

    imgs = [synth.page(500 + i, width=512 + 32 * (i % 3), height=320, lines=4, margin=20)[0] for i in range(6)]
    seq = [reader.readtext(im) for im in imgs]
    assert any(seq)
    with ThreadPoolExecutor(max_workers=3) as ex:
        par = list(ex.map(lambda im: reader.readtext(im, paragraph=False, batch_size=1, workers=0), imgs * 2))
    assert par == seq + seq


def test_two_calls_in_flight_equal_the_serial_path(states):
    """bbocr_config::call_slots: two worker threads share ONE Reader (batch_processor_enhanced.py:215) on full-size batches -- 24 pages of
    1280x960 per call, detector passes [16, 8] with the early recogniser part -- and every call returns exactly what the same call returns
    alone, whichever slot it ran in; distinct batches so that a buffer shared by mistake would show; stage times answer per thread; a
    context created with call_slots = 1 (calls serialise, as in rounds 1-3) returns the same."""
    from concurrent.futures import ThreadPoolExecutor

    import bb_ocr_amd
    from bb_ocr_amd import synth

    r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision="fp16")
    batches = [torch.from_numpy(np.stack([synth.page(40_000 + 100 * k + i, lines=6 + 4 * k + (i % 5), line_pitch=38, margin=24, colour=bool(i & 1))[0]
                                          for i in range(24)])).cuda() for k in range(3)]
    serial = [r.readtext_device(b) for b in batches]
    assert all(sum(len(p) for p in s) > 100 for s in serial)
    assert serial[0] != serial[1]

    def call(k):
        out = r.readtext_device(batches[k % 3])
        return out, r.stage_times()

    with ThreadPoolExecutor(max_workers=2) as ex:
        par = list(ex.map(call, range(8)))
    for k, (out, st) in enumerate(par):
        assert out == serial[k % 3], k
        assert st["total"] > 0 and st["detector_net"] > 0
    assert list(r.readtext_stream(iter(batches * 2))) == serial * 2                    # the streaming form: ordered results
    # three callers on two slots: the third waits for a free slot
    with ThreadPoolExecutor(max_workers=3) as ex:
        assert [o for o, _ in ex.map(call, range(6))] == (serial * 2)
    r.close()
    one = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision="fp16", call_slots=1)
    with ThreadPoolExecutor(max_workers=2) as ex:
        assert list(ex.map(lambda k: one.readtext_device(batches[k]), range(3))) == serial
    one.close()


def test_readtext_batched_streams_large_groups(reader, monkeypatch):
    """Reader.readtext_batched splits a shape group larger than BBOCR_MAX_DEVICE_BATCH into device batches and streams them with two calls in
    flight (readtext_stream): same results, same order, as one device batch -- mixed shapes included."""
    from bb_ocr_amd import synth

    imgs = [synth.page(72_000 + i, width=512 if i % 3 else 448, height=320, lines=3 + i % 4, margin=20)[0] for i in range(11)]
    want = reader.readtext_batched(imgs)
    assert any(want) and want[0] == reader.readtext(imgs[0])
    monkeypatch.setenv("BBOCR_MAX_DEVICE_BATCH", "3")
    assert reader.readtext_batched(imgs) == want
    assert reader.readtext_batched(imgs, detail=0) == [[t for _, t, _ in page] for page in want]


def test_mixed_entry_points_from_three_threads(states):
    """Call slots under a mix of entry points: three threads hammer ONE Reader with different calls at once -- whole readtext on a batch,
    detector only, boxes from a heat-map, recognise explicit boxes, a single page from an array, the pre-processing chain -- for a few
    hundred calls; every result equals what the same call returned alone.  (Two slots, so the third thread always waits for one; a work
    buffer shared between slots by mistake, a stream wait on the wrong event or an error string crossing threads would show.)"""
    import random
    from concurrent.futures import ThreadPoolExecutor

    import bb_ocr_amd
    from bb_ocr_amd import preprocess as dev_pp
    from bb_ocr_amd import synth

    r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision="fp16")
    pages = [synth.page(61_000 + i, width=640, height=384, lines=4 + i % 4, margin=24, colour=bool(i & 1))[0] for i in range(6)]
    rgb = torch.from_numpy(np.stack(pages)).cuda()
    gray = torch.from_numpy(np.stack([p[..., 1] for p in pages])).cuda()
    heat, ratio = r.heatmap_device(rgb)
    hori, free, _ = r.boxes_from_heatmap(heat, ratio)
    bgr = torch.from_numpy(np.ascontiguousarray(pages[0][:, :, ::-1])).cuda()
    jobs = {
        "readtext_batch": lambda: r.readtext_device(rgb),
        "detect": lambda: r.heatmap_device(rgb[:3])[0].cpu().numpy().tobytes(),
        "boxes": lambda: r.boxes_from_heatmap(heat, ratio),
        "recognize": lambda: r.recognize_device(gray, hori, free),
        "single": lambda: r.readtext(pages[2]),
        "beam": lambda: r.readtext(pages[3], decoder="beamsearch"),
        "preprocess": lambda: dev_pp.preprocess_bgr_device(r, bgr).cpu().numpy().tobytes(),
    }
    want = {k: f() for k, f in jobs.items()}
    assert want["readtext_batch"][2] == want["single"] and sum(len(p) for p in want["recognize"]) > 10
    order = [k for k in jobs for _ in range(24)]
    random.Random(7).shuffle(order)

    def run(k):
        return k, jobs[k]()

    with ThreadPoolExecutor(max_workers=3) as ex:
        for k, got in ex.map(run, order):
            assert got == want[k], k
    with pytest.raises(ValueError):                  # a failing call on one thread leaves the others' slots usable
        r.readtext(np.zeros((4, 4), dtype=np.float32))
    assert r.readtext_device(rgb) == want["readtext_batch"]
    r.close()


def test_context_teardown_returns_device_memory(states):
    """Ownership: the context owns weights, work buffers, streams; close() gives all of it back.  Three create / use / close
    cycles (every optional buffer exercised: beam-search probabilities, rotation variants, the pre-processing planes) leave the free
    device memory where it was."""
    import bb_ocr_amd
    from bb_ocr_amd import synth

    img = synth.page(901, width=640, height=384, lines=5, margin=24)[0]

    def cycle():
        r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision="bf16")
        assert r.readtext(img) and r.readtext(img, decoder="beamsearch") and r.readtext(img, rotation_info=[90, 180, 270])
        r.close()

    cycle()                                            # first use also initialises HIP / loads code objects
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < (32 << 20)


def test_sequence_pass_budgets_do_not_change_results(states, reader):
    """bbocr_config::rec_max_cols bounds the pooled time steps of one sequence pass.  Small budgets split a recognition pass into several
    runs, refuse the early feature part (first detector pass's pages) or make it finish on its own before the rest is recognised: every
    such schedule returns exactly what the default one returns.  24 pages = detector passes [16, 8], the last 8 pages text-heavy."""
    import bb_ocr_amd
    from bb_ocr_amd import synth

    pages = [synth.page(700 + i, width=512, height=320, lines=(2 if i < 16 else 7), margin=20)[0] for i in range(24)]
    rgb = torch.from_numpy(np.stack(pages)).cuda()
    want = reader.readtext_device(rgb)
    assert sum(len(p) for p in want[16:]) > sum(len(p) for p in want[:16])
    for cols in (1500, 12000, 40000, 90000, 105000, 118000, 130000, 160000):
        r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, rec_max_cols=cols, precision="bf16")
        assert r.readtext_device(rgb) == want, f"rec_max_cols={cols}"
        r.close()


def test_random_batches_match_single_pages(reader):
    """Seeded sweep over batch sizes (1 .. 40 pages: detector pass schedules [B], [B-8, 8], with and without the early recogniser part), page
    sizes (on and off the 32-pixel grid, i.e. with and without the canvas resize) and text densities: the batch result equals the
    page-by-page results, page for page."""
    from bb_ocr_amd import synth

    rng = np.random.default_rng(2024)
    for case in range(6):
        B = int(rng.choice([1, 3, 9, 24, 31, 40]))
        W = int(rng.choice([320, 416, 500, 640]))
        H = int(rng.choice([192, 250, 288]))
        pages = [synth.page(9000 + 50 * case + i, width=W, height=H, lines=int(rng.integers(0, 6)), margin=16)[0] for i in range(B)]
        rgb = torch.from_numpy(np.stack(pages)).cuda()
        got = reader.readtext_device(rgb)
        idx = sorted(set(rng.integers(0, B, size=min(B, 5)).tolist()) | {0, B - 1})
        for i in idx:
            assert reader.readtext_device(rgb[i:i + 1])[0] == got[i], (case, B, W, H, i)


def test_boxes_property_random_blob_maps(reader):
    """Property test (hypothesis) of the whole box stage on the device -- threshold, CCL, statistics, row extremes, host geometry,
    grouping -- against the oracle on the SAME maps: rotated boxes / ellipses, slivers, border contact, overlaps, link bridges, ragged
    map sizes.  Polygons, horizontal boxes and free boxes must be identical, in order."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    from oracle import boxes as obox
    from test_host_properties_cpu import _blob_maps

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
    @given(maps=_blob_maps(), ratio=st.sampled_from([1.0, 0.7306]), low_text=st.sampled_from([0.4, 0.2]), min_size=st.sampled_from([20, 3]))
    def check(maps, ratio, low_text, min_size):
        text, link = maps
        heat = torch.from_numpy(np.stack([text, link], -1)[None].astype(np.float32)).cuda()
        hori, free, polys = reader.boxes_from_heatmap(heat, ratio, low_text=low_text, min_size=min_size)
        oh, of, op = obox.detect_from_heatmap(text, link, ratio, low_text=low_text, min_size=min_size)
        assert polys[0] == [list(map(int, p)) for p in op]
        assert hori[0] == [list(map(int, b)) for b in oh]
        assert np.allclose(np.array(free[0], dtype=np.float64).reshape(-1, 4, 2), np.array(of, dtype=np.float64).reshape(-1, 4, 2), rtol=0, atol=1e-9)

    check()
