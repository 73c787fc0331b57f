"""CPU: the C-ABI library loads and exports what include/bbocr.h declares; host-side logic (box geometry in C++,
Python Reader plumbing, weight loading) -- no GPU compute calls."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from bb_ocr_amd import _lib

    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from bb_ocr_amd import _lib

    header = open(os.path.join(ROOT, "include", "bbocr.h")).read()
    declared = set(re.findall(r"\b(bbocr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bbocr.h but not exported"
    assert declared == set(_lib.PROTOTYPES), "ctypes prototype table out of sync with the header"


def test_default_params_mirror_readtext_defaults(lib):
    from bb_ocr_amd import _lib

    p = _lib.bbocr_params()
    lib.bbocr_default_params(C.byref(p))
    assert (p.text_threshold, p.low_text, p.link_threshold, p.canvas_size, p.mag_ratio) == (0.7, 0.4, 0.4, 2560, 1.0)
    assert (p.slope_ths, p.ycenter_ths, p.height_ths, p.width_ths, p.add_margin, p.min_size) == (0.1, 0.5, 0.5, 0.5, 0.1, 20)
    assert (p.contrast_ths, p.adjust_contrast) == (0.1, 0.5)


@pytest.mark.parametrize("H,W,canvas", [(960, 1280, 2560), (3504, 2480, 2560), (971, 1014, 2560), (300, 500, 320), (64, 96, 2560)])
def test_detect_dims_match_resize_aspect_ratio(lib, H, W, canvas):
    from oracle import imgproc

    v = [C.c_int() for _ in range(4)]
    r = C.c_double()
    assert lib.bbocr_detect_dims(H, W, canvas, 1.0, *[C.byref(x) for x in v], C.byref(r)) == 0
    canvas_img, ratio, size_heatmap = imgproc.resize_aspect_ratio(np.zeros((H, W, 3), np.uint8), canvas, 1.0)
    assert (v[0].value, v[1].value) == canvas_img.shape[:2]
    assert (v[3].value, v[2].value) == size_heatmap and r.value == ratio
    if (H, W) == (3504, 2480):
        assert (v[0].value, v[1].value) == (2560, 1824)        # A4 @300 dpi


def test_errors_are_status_codes_not_aborts(lib):
    from bb_ocr_amd import _lib

    assert lib.bbocr_detect_dims(0, 10, 2560, 1.0, None, None, None, None, None) == -1
    assert lib.bbocr_stage_times(None, None, 0) == -1
    assert lib.bbocr_last_error(None) == b"null context"
    lib.bbocr_destroy(None)
    lib.bbocr_free_result(None)
    lib.bbocr_free_boxlist(None)
    import torch

    if not torch.cuda.is_available():
        h = C.c_void_p()
        cfg = _lib.bbocr_config(device=0)
        assert lib.bbocr_create(C.byref(cfg), C.byref(h)) == -2 and not h.value       # no HIP device: status, no crash
    # an unknown precision must not silently mean one of the modes: refused before anything touches a device
    for bad in (-1, 5, 99):
        h = C.c_void_p()
        cfg = _lib.bbocr_config(device=0, precision=bad)
        assert lib.bbocr_create(C.byref(cfg), C.byref(h)) == -1 and not h.value
    assert sorted(_lib.PRECISIONS.values()) == [0, 1, 2, 3, 4]                       # bf16, fp16, exact, mixed, exact_rec (include/bbocr.h)


def _components_from_heat(text, link, low_text=0.4, link_thr=0.4, text_thr=0.7):
    """numpy stand-in for what ccl.hip emits: accepted components (raster order) + per-row text extremes."""
    from oracle import boxes as obox

    ts, ls = text > np.float32(low_text), link > np.float32(link_thr)
    n, labels, stats = obox.connected_components_4((ts | ls).astype(np.uint8))
    comps, rows = [], []
    for k in range(1, n):
        x, y, w, h, area = (int(v) for v in stats[k])
        m = labels == k
        if area < 10 or float(text[m].max()) < text_thr:
            continue
        off = len(rows) // 2
        ys, xs = np.nonzero(m)
        comps.append([int(ys.min() * text.shape[1] + xs[ys == ys.min()].min()), x, y, x + w - 1, y + h - 1, area, off])
        for yy in range(y, y + h):
            sel = m[yy] & ts[yy]
            xs2 = np.nonzero(sel)[0]
            rows += [int(xs2.min()), int(xs2.max())] if len(xs2) else [0x7FFFFFFF, -1]
    return np.array(comps, dtype=np.int32).reshape(-1, 7), np.array(rows, dtype=np.int32)


def test_host_box_geometry_matches_oracle(lib):
    """boxpost.cpp (dilation of row extremes, hull, rotating calipers, boxPoints, diamond fix, int cast) == oracle."""
    from bb_ocr_amd import _lib
    from oracle import boxes as obox

    gold = np.load(os.path.join(ROOT, "tests", "golden", "oracle_boxes.npz"))
    rng = np.random.default_rng(4)
    cases = [(gold["text"], gold["link"])]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_pipeline as tp

    for _ in range(3):
        cases.append(tp._synth_heat(rng, 120, 200))
    for text, link in cases:
        for ratio in (1.0, 0.7306):
            comps, rows = _components_from_heat(text, link)
            det, _, _ = obox.get_det_boxes_core(text, link)
            want = obox.boxes_to_int_polys(obox.adjust_result_coordinates(det, 1 / ratio, 1 / ratio))
            assert len(want) == len(comps) > 0
            out = np.zeros((len(comps), 8), np.int32)
            rc = lib.bbocr_host_component_polys(comps.ctypes.data_as(C.POINTER(C.c_int)), rows.ctypes.data_as(C.POINTER(C.c_int)), len(comps),
                                                text.shape[1], text.shape[0], ratio, out.ctypes.data_as(C.POINTER(C.c_int)))
            assert rc == 0
            assert np.array_equal(out, np.array(want, dtype=np.int32))
            # grouping
            for kw in ({}, {"add_margin": 0.2, "min_size": 5, "width_ths": 1.0}):
                p = _lib.bbocr_params()
                lib.bbocr_default_params(C.byref(p))
                for k, v in kw.items():
                    setattr(p, k, v)
                bl = C.POINTER(_lib.bbocr_boxlist)()
                assert lib.bbocr_host_group_boxes(out.ctypes.data_as(C.POINTER(C.c_int)), len(out), C.byref(p), C.byref(bl)) == 0
                b = bl.contents
                hori = [[b.hori[i * 4 + k] for k in range(4)] for i in range(b.hori_off[1])]
                free = [[b.free_q[i * 8 + k] for k in range(8)] for i in range(b.free_off[1])]
                lib.bbocr_free_boxlist(bl)
                oh, of = obox.group_text_box(want, 0.1, 0.5, 0.5, kw.get("width_ths", 0.5), kw.get("add_margin", 0.1))
                ms = kw.get("min_size", 20)
                oh = [i for i in oh if max(i[1] - i[0], i[3] - i[2]) > ms]
                of = [i for i in of if max(max(c[0] for c in i) - min(c[0] for c in i), max(c[1] for c in i) - min(c[1] for c in i)) > ms]
                assert hori == [list(map(int, x)) for x in oh]
                assert np.allclose(np.array(free).reshape(-1, 4, 2), np.array(of, dtype=np.float64).reshape(-1, 4, 2), rtol=0, atol=1e-9)


def test_host_group_boxes_empty(lib):
    from bb_ocr_amd import _lib

    bl = C.POINTER(_lib.bbocr_boxlist)()
    assert lib.bbocr_host_group_boxes(None, 0, None, C.byref(bl)) == 0
    assert bl.contents.hori_off[1] == 0 and bl.contents.free_off[1] == 0
    lib.bbocr_free_boxlist(bl)


def test_reformat_input_mirrors_upstream_rules():
    import bb_ocr_amd
    from oracle import imgproc

    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (12, 17, 3), dtype=np.uint8)
    for inp in (rgb, rgb[..., 0], rgb[..., :1], np.dstack([rgb, rgb[..., :1]])):
        a, g = bb_ocr_amd.reformat_input(inp)
        ea, eg = imgproc.reformat_input(inp)
        assert np.array_equal(a, ea) and np.array_equal(g, eg) and a.dtype == np.uint8 and a.shape[2] == 3
    from PIL import Image
    import io

    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="PNG")
    a, g = bb_ocr_amd.reformat_input(buf.getvalue())
    assert np.array_equal(a, rgb) and np.array_equal(g, imgproc.gray_from_3ch(rgb, "bgr"))
    with pytest.raises(ValueError):
        bb_ocr_amd.reformat_input(3.14)


def test_reader_requires_gpu_and_never_falls_back():
    import torch

    import bb_ocr_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="HIP device"):
        bb_ocr_amd.Reader(["en"], gpu=True, weights="synthetic")
    with pytest.raises(ValueError):
        bb_ocr_amd.Reader(["fr"], weights="synthetic")


def test_install_hook_registers_easyocr_module():
    import bb_ocr_amd

    prev = sys.modules.get("easyocr")
    try:
        m = bb_ocr_amd.install()
        import easyocr

        assert easyocr is m and easyocr.Reader is bb_ocr_amd.Reader
    finally:
        bb_ocr_amd.uninstall()
        assert sys.modules.get("easyocr") is prev


def test_synthetic_states_have_upstream_names_and_shapes(states):
    import torch

    from oracle import nets

    cs, rs = states
    nets.load_state_dict_any(nets.CRAFT(), {("module." + k): torch.from_numpy(v) for k, v in cs.items()})     # DataParallel prefix accepted
    nets.load_state_dict_any(nets.CRNN(), {k: torch.from_numpy(v) for k, v in rs.items()})
    n_conv = sum(v.size for k, v in cs.items() if k.endswith(".weight") and v.ndim == 4)
    assert abs(n_conv - 20.75e6) < 0.05e6                       # 20.75 M conv parameters
    assert rs["Prediction.weight"].shape == (97, 256)


def test_weight_descriptor_table(states):
    from bb_ocr_amd import weights

    arr, keep = weights.to_descs(states[1])
    names = {arr[i].name.decode() for i in range(len(arr))}
    assert "SequenceModeling.0.rnn.weight_hh_l0_reverse" in names and not any(n.endswith("num_batches_tracked") for n in names)
    i = [arr[k].name.decode() for k in range(len(arr))].index("FeatureExtraction.ConvNet.18.weight")
    assert list(arr[i].shape) == [256, 256, 2, 2] and arr[i].ndim == 4


def _save_upstream_checkpoints(directory, cs, rs):
    """The two files easyocr downloads, in the two wrappings upstream checkpoints come in: the detector as a DataParallel
    state-dict (``module.`` prefixes, ``num_batches_tracked`` entries), the recogniser wrapped in ``{"state_dict": ...}``."""
    import torch

    torch.save({("module." + k): torch.from_numpy(np.asarray(v)) for k, v in cs.items()}, os.path.join(directory, "craft_mlt_25k.pth"))
    torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in rs.items()}}, os.path.join(directory, "english_g2.pth"))


def test_checkpoint_loader_round_trip(states, tmp_path):
    """f1: ``Reader(model_storage_directory=...)``'s loader (weights.load_checkpoint_dir -> to_descs) on files with the upstream names:
    same tensors as the in-memory states, prefix stripped by the library's table (api.cpp TensorMap), missing file -> FileNotFoundError."""
    from bb_ocr_amd import weights

    cs, rs = states
    assert any(k.endswith("num_batches_tracked") for k in cs)
    _save_upstream_checkpoints(str(tmp_path), cs, rs)
    lc, lr = weights.load_checkpoint_dir(str(tmp_path))
    assert set(lc) == {"module." + k for k in cs} and set(lr) == set(rs)
    assert all(np.array_equal(lc["module." + k], v) for k, v in cs.items()) and all(np.array_equal(lr[k], v) for k, v in rs.items())
    arr, keep = weights.to_descs(lc)
    assert len(arr) == sum(not k.endswith("num_batches_tracked") for k in cs)
    os.remove(os.path.join(str(tmp_path), "english_g2.pth"))
    with pytest.raises(FileNotFoundError):
        weights.load_checkpoint_dir(str(tmp_path))
    import bb_ocr_amd

    with pytest.raises(FileNotFoundError):
        bb_ocr_amd.Reader._resolve_weights(None, str(tmp_path / "nowhere"))


def test_path_inputs_follow_the_stated_gray_rule(tmp_path):
    """a2 on the reference's only call pattern (a file path, enhanced_extractor.py:520): JPEG -> libjpeg's own Y plane, gray files ->
    stored samples, colour PNG -> libpng's truncating 15-bit sum; product == oracle, and the colour array is the RGB decode."""
    from PIL import Image

    import bb_ocr_amd
    from oracle import imgproc

    rng = np.random.default_rng(11)
    base = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    rgb = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))            # 48 x 64, blocky so that JPEG keeps colours apart
    jpg, png, gpng = (str(tmp_path / n) for n in ("c.jpg", "c.png", "g.png"))
    Image.fromarray(rgb).save(jpg, quality=92)
    Image.fromarray(rgb).save(png)
    Image.fromarray(rgb[..., 1]).save(gpng)
    for path in (jpg, png, gpng):
        a, g = bb_ocr_amd.reformat_input(path)
        ea, eg = imgproc.reformat_input(path)
        assert np.array_equal(a, ea) and np.array_equal(g, eg) and g.shape == a.shape[:2]
    a, g = bb_ocr_amd.reformat_input(jpg)
    y = Image.open(jpg)
    y.draft("L", y.size)
    assert np.array_equal(g, np.asarray(y)) and np.array_equal(a, np.asarray(Image.open(jpg).convert("RGB")))
    assert np.abs(g.astype(int) - imgproc.gray_from_3ch(a, "rgb").astype(int)).mean() < 1.0    # Y plane ~ luma of the decoded RGB (chroma clipping aside)
    a, g = bb_ocr_amd.reformat_input(png)
    r_, g_, b_ = (rgb[..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(a, rgb) and np.array_equal(g, ((r_ * 9797 + g_ * 19234 + b_ * 3737) >> 15).astype(np.uint8))
    a, g = bb_ocr_amd.reformat_input(gpng)
    assert np.array_equal(g, rgb[..., 1]) and np.array_equal(a, np.repeat(rgb[..., 1:2], 3, 2))
    # OpenCV 4's BGR2GRAY (15-bit) on arrays: known answers
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [12, 200, 77], [255, 255, 255]]], dtype=np.uint8)   # channel 0 takes the "B" weight
    assert imgproc.gray_from_3ch(px, "bgr").tolist() == [[29, 150, 76, 142, 255]]
    assert bb_ocr_amd.reformat_input(px)[1].tolist() == [[29, 150, 76, 142, 255]]


def test_one_ycbcr_decode_holds_both_planes_of_a_jpeg(tmp_path):
    """a2, decode once: libjpeg's YCbCr triples of a JFIF file give, bit for bit, the RGB decode (through jdcolor.c's integer conversion,
    restated in oracle/imgproc.py::jpeg_ycc_to_rgb and run on the card by bbocr_op_ycc_to_rgb) and the grayscale decode (their Y channel).
    Pinned against the decoder itself: every subsampling / quality / progressive variant, and the reference's own photographs."""
    import glob
    import io
    import os

    from PIL import Image

    from bb_ocr_amd.reader import decode_file, decode_file_ycc
    from oracle import imgproc

    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (120, 168, 3), dtype=np.uint8)
    ramp = (np.add.outer(np.arange(120), np.arange(168))[..., None] * np.array([1, 2, 3]) % 256).astype(np.uint8)
    sat = np.kron(rng.integers(0, 2, (15, 21, 3), dtype=np.uint8) * 255, np.ones((8, 8, 1), dtype=np.uint8))    # saturated colours: the clamps
    n = 0
    for img in (noise, ramp, sat):
        for kw in (dict(quality=92), dict(quality=100, subsampling=0), dict(quality=75, subsampling=1), dict(quality=95, subsampling=2),
                   dict(quality=90, progressive=True), dict(quality=30, optimize=True)):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", **kw)
            ycc = decode_file_ycc(buf.getvalue())
            rgb, grey = decode_file(buf.getvalue())
            assert ycc is not None and np.array_equal(imgproc.jpeg_ycc_to_rgb(ycc), rgb) and np.array_equal(ycc[..., 0], grey), kw
            n += 1
            pad = decode_file_ycc(buf.getvalue(), padded=True)       # Pillow's own 4-byte pixels when it can export them without a copy
            assert pad.shape[:2] == ycc.shape[:2] and pad.shape[2] in (3, 4) and np.array_equal(pad[..., :3], ycc)
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photos")
    for path in sorted(glob.glob(os.path.join(here, "*.JPG")))[:2]:
        ycc = decode_file_ycc(path)
        rgb, grey = decode_file(path)
        assert ycc is not None and np.array_equal(imgproc.jpeg_ycc_to_rgb(ycc), rgb) and np.array_equal(ycc[..., 0], grey)
    # files the single decode does not take: other containers, greyscale and CMYK JPEGs (callers fall back to decode_file)
    png, gj, cj = (str(tmp_path / f) for f in ("a.png", "g.jpg", "c.jpg"))
    Image.fromarray(noise).save(png)
    Image.fromarray(noise[..., 0]).save(gj)
    Image.fromarray(noise).convert("CMYK").save(cj)
    assert decode_file_ycc(png) is None and decode_file_ycc(gj) is None and decode_file_ycc(cj) is None
    assert decode_file_ycc(str(tmp_path / "missing.jpg")) is None and decode_file_ycc(b"not a jpeg") is None
    # known answers of the conversion: grey stays grey, the primaries' chroma clamps
    px = np.array([[[128, 128, 128], [0, 128, 128], [255, 128, 128], [76, 85, 255], [150, 44, 21], [29, 255, 107]]], dtype=np.uint8)
    assert imgproc.jpeg_ycc_to_rgb(px).tolist() == [[[128, 128, 128], [0, 0, 0], [255, 255, 255], [254, 0, 0], [0, 255, 1], [0, 0, 254]]]


def test_shard_range_partitions_exactly():
    from bb_ocr_amd import dist

    for n, w in [(512, 8), (128, 8), (10, 3), (2, 4), (0, 2)]:
        parts = [dist.shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1
    assert dist.shard_range(512, 3, 8) == (192, 256)


def test_tesseract_shim_reading_order_and_install():
    """a12: rows by vertical overlap, left to right, newline-joined; the module registers under the name scripts import."""
    import sys

    from bb_ocr_amd import tesseract_shim as ts

    def box(x0, y0, x1, y1):
        return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]

    res = [(box(200, 12, 260, 30), "world", 0.9), (box(10, 10, 80, 32), "hello", 0.9), (box(12, 60, 90, 80), "second", 0.8),
           (box(100, 58, 150, 82), "row", 0.7), (box(10, 120, 40, 140), "x", 0.5)]
    rows = ts.lines_from_results(res)
    assert [[it[1] for it in r] for r in rows] == [["hello", "world"], ["second", "row"], ["x"]]

    class FakeReader:
        def readtext(self, image, **kw):
            return res

    assert ts.image_to_string(np.zeros((8, 8, 3), np.uint8), reader=FakeReader()) == "hello world\nsecond row\nx\n"
    prev = sys.modules.get("pytesseract")
    try:
        m = ts.install(reader=FakeReader())
        import pytesseract

        assert pytesseract is m and pytesseract.image_to_string(np.zeros((8, 8, 3), np.uint8)) == "hello world\nsecond row\nx\n"
    finally:
        ts.uninstall()
        ts.set_reader(None)
    assert sys.modules.get("pytesseract") is prev


def test_extractor_downscale_rule_and_batching(tmp_path):
    """f3: the reference's OCR-input rule (enhanced_extractor.py:486-512) and the batched text assembly (:521, :529-531)."""
    from PIL import Image

    from bb_ocr_amd import extractor_batch as eb

    rng = np.random.default_rng(0)
    big = tmp_path / "big.png"
    small = tmp_path / "small.png"
    Image.fromarray(rng.integers(0, 256, (1000, 3000, 3), dtype=np.uint8)).save(big)
    Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)).save(small)
    rgb0, g0 = eb.ocr_input_image(big, 0)          # cover: <= 1600
    rgb1, g1 = eb.ocr_input_image(big, 3)          # other pages: <= 2400
    assert rgb0.shape == (533, 1600, 3) and g0.shape == (533, 1600)
    assert rgb1.shape == (800, 2400, 3)
    rs, gs = eb.ocr_input_image(small, 0)
    assert rs.shape == (300, 400, 3) and np.array_equal(rs, np.asarray(Image.open(small).convert("RGB")))
    # the reference's own sequence for the cover page
    import io
    im = Image.open(big).convert("RGB")
    im.thumbnail((1600, 1600))
    buf = io.BytesIO()
    im.save(buf, format="JPEG", quality=90)
    assert np.array_equal(rgb0, np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB")))

    good = tmp_path / "good.png"
    arr = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    arr[0, 0] = (1, 2, 3)
    Image.fromarray(arr).save(good)
    poison = np.asarray(Image.open(small).convert("RGB"))[0, 0].tolist()

    class FakeReader:
        def __init__(self):
            self.batches = []

        def readtext_arrays(self, rgb, gray=None, **kw):
            rgb = np.stack(rgb)                        # the batching loop hands over LISTS of equal-shape pages
            self.batches.append(rgb.shape)
            if any(p[0, 0].tolist() == poison for p in rgb):
                raise RuntimeError("boom")             # a batch holding the bad page fails as a whole ...
            return [[(None, f"w{rgb.shape[2]}", 0.9), (None, "x", 0.5)] for _ in range(rgb.shape[0])]

        def readtext_ycc_arrays(self, ycc, **kw):      # thumbnails are JPEGs: decoded once, colour conversion left to the card
            self.ycc_pages = getattr(self, "ycc_pages", 0) + len(ycc)
            return self.readtext_arrays(ycc, None, **kw)

    fr = FakeReader()
    texts = eb.extract_texts(fr, [big, small, big, tmp_path / "missing.png", good], [1, 2, 0, 3, 7, 4])
    # ... and is retried page by page: only the bad page maps to empty text, like the reference's per-page except (:529-531)
    assert texts == {1: "", 2: "w2400 x", 0: "w1600 x", 3: "", 4: "w400 x"}
    assert sorted(b[:3] for b in fr.batches) == sorted([(2, 300, 400), (1, 300, 400), (1, 300, 400), (1, 800, 2400), (1, 533, 1600)])   # (ycc pages: 3 or 4 bytes per pixel)
    assert fr.ycc_pages == 2                           # the two thumbnails; the PNG pages travel as (rgb, gray)
    kind, ycc0, none = eb._ocr_input(big, 0)
    from oracle import imgproc
    assert kind == "ycc" and none is None and np.array_equal(imgproc.jpeg_ycc_to_rgb(ycc0), rgb0) and np.array_equal(ycc0[..., 0], g0)
    assert eb._ocr_input(small, 0)[0] == "rgb"
    # back-pressure: many files, tiny batches -- every page still comes out once, in any order
    many = [good] * 37
    fr2 = FakeReader()
    t2 = eb.extract_texts(fr2, many, max_batch=4, decode_workers=3)
    assert t2 == {i: "w400 x" for i in range(37)} and sum(b[0] for b in fr2.batches) == 37 and max(b[0] for b in fr2.batches) <= 4
    # ... and a full group is what travels: pages that are still being decoded do not count as "held in partial groups" (round 3's
    # assembler cut batches into single pages whenever the decode pool ran ahead: 387 device calls for 512 pages)
    assert sorted(b[0] for b in fr2.batches) == [1] + [4] * 9
    fr3 = FakeReader()
    t3 = eb.extract_texts(fr3, [good] * 64, max_batch=8, decode_workers=8)
    assert len(t3) == 64 and sorted(b[0] for b in fr3.batches) == [8] * 8


def test_paragraph_and_ignore_mask_rules():
    """f4: easyocr/utils.py::get_paragraph (grouping + reading order) and Reader.recognize's ignore_char rule."""
    import bb_ocr_amd
    from bb_ocr_amd.reader import get_paragraph, ignore_mask

    def box(x0, y0, x1, y1):
        return [[x0, y0], [x1, y0], [x1, y1], [x0, y1]]

    res = [(box(10, 10, 60, 30), "Hello", 0.9), (box(70, 12, 130, 31), "world", 0.8),      # line 1
           (box(12, 36, 90, 56), "second", 0.9), (box(100, 38, 140, 57), "line", 0.7),      # line 2, same paragraph
           (box(400, 300, 460, 320), "far", 0.9), (box(470, 301, 520, 321), "away", 0.9)]   # another paragraph
    out = get_paragraph(res)
    assert [o[1] for o in out] == ["Hello world second line", "far away"]
    assert out[0][0] == [[10, 10], [140, 10], [140, 57], [10, 57]] and len(out[0]) == 2
    assert [o[1] for o in get_paragraph(res, x_ths=0.1, y_ths=0.1)] == ["Hello", "world", "second", "line", "far", "away"]
    ch, lang = bb_ocr_amd.CHARACTER, list(bb_ocr_amd.CHARSET)
    assert ignore_mask(ch, lang) == [0, 0, 0, 0]
    w = ignore_mask(ch, lang, allowlist="ab")
    kept = [ch[i] for i in range(1, len(ch)) if not (w[i >> 5] >> (i & 31)) & 1]
    assert kept == sorted(kept, key=ch.index) and set(kept) == {"a", "b"} and not (w[0] & 1)
    w = ignore_mask(ch, lang, blocklist="xyz")
    assert {ch[i] for i in range(len(ch)) if (w[i >> 5] >> (i & 31)) & 1} == {"x", "y", "z"}


@pytest.mark.parametrize("scale", [0.5, 2.0, 6.0])
def test_host_ctc_beam_matches_oracle(lib, scale):
    """f4: the C++ ctcBeamSearch (host half of decoder='beamsearch') against the oracle, flat to peaked distributions."""
    from oracle import recog

    rng = np.random.default_rng(int(scale * 10))
    n, T, Cn, cs = 12, 29, 97, 112
    lg = (rng.standard_normal((n, T, Cn)) * scale).astype(np.float32)
    lg[:, :, 0] += rng.uniform(0, 3 * scale, size=(n, 1)).astype(np.float32)      # blank-heavy rows, as a recogniser produces
    lg[0, 5:9, 17] += 20.0                                                         # a repeated symbol
    pr = recog.softmax_f32(lg)
    full = np.zeros((n, T, cs), np.float32)
    full[:, :, :Cn] = pr
    off = (C.c_int * (n + 1))()
    idx = (C.c_int * (n * T))()
    for bw in (1, 5, 9):
        assert lib.bbocr_host_ctc_beam(full.ctypes.data_as(C.POINTER(C.c_float)), n, T, Cn, cs, bw, off, idx) == 0
        for i in range(n):
            assert [idx[k] for k in range(off[i], off[i + 1])] == recog.ctc_beam_search(pr[i], bw), (scale, bw, i)
    assert lib.bbocr_host_ctc_beam(full.ctypes.data_as(C.POINTER(C.c_float)), n, T, Cn, cs, 0, off, idx) == -1     # BBOCR_ERR_ARG
    assert lib.bbocr_host_ctc_beam(None, n, T, Cn, cs, 5, off, idx) == -1


def test_output_formats():
    """f4: readtext's output_format tail ('dict' / 'json'; detail=0 wins; upstream's key is 'confident')."""
    import json

    from bb_ocr_amd.reader import format_output

    res = [([[1, 2], [30, 2], [30, 12], [1, 12]], "ab", 0.5), ([[1.5, 20.0], [30.0, 21.0], [30.0, 31.0], [1.0, 30.0]], "c\u20ac", 0.25)]
    assert format_output(res) is res and format_output(["ab"], "dict", detail=0) == ["ab"]
    d = format_output(res, "dict")
    assert d[0] == {"boxes": res[0][0], "text": "ab", "confident": 0.5}
    j = [json.loads(x) for x in format_output(res, "json")]
    assert j[1] == {"boxes": [[1, 20], [30, 21], [30, 31], [1, 30]], "text": "c\u20ac", "confident": 0.25} and "\u20ac" in format_output(res, "json")[1]
    para = [[r[0], r[1]] for r in res]
    assert format_output(para, "dict", paragraph=True)[0] == {"boxes": res[0][0], "text": "ab"}
    with pytest.raises(NotImplementedError):
        format_output(res, "free_merge")


def test_result_pack_unpack_round_trip_and_dist_entry_points_without_a_communicator(lib):
    """The bytes bbocr_gather_results moves between ranks: bbocr_result -> flat block -> bbocr_result, field for field; a block whose header
    does not match its length or its offsets is refused; the RCCL entry points are status codes, not crashes, on a null context."""
    from bb_ocr_amd import _lib

    box_off = (C.c_int * 3)(0, 2, 3)
    quads = (C.c_double * 24)(*[float(i) * 0.5 for i in range(24)])
    is_free = (C.c_int * 3)(0, 1, 0)
    text_off = (C.c_int * 4)(0, 2, 2, 5)
    text_idx = (C.c_int * 5)(11, 12, 40, 41, 96)
    conf = (C.c_double * 3)(0.9, 0.0, 0.25)
    r = _lib.bbocr_result(n_images=2, box_off=box_off, quads=quads, is_free=is_free, text_off=text_off, text_idx=text_idx, conf=conf)
    blob, n = C.c_void_p(), C.c_size_t()
    assert lib.bbocr_result_pack(C.byref(r), C.byref(blob), C.byref(n)) == 0 and n.value > 0
    out = C.POINTER(_lib.bbocr_result)()
    assert lib.bbocr_result_unpack(blob, n.value, C.byref(out)) == 0
    o = out.contents
    assert o.n_images == 2 and [o.box_off[i] for i in range(3)] == [0, 2, 3]
    assert [o.quads[i] for i in range(24)] == [float(i) * 0.5 for i in range(24)]
    assert [o.is_free[i] for i in range(3)] == [0, 1, 0] and [o.text_off[i] for i in range(4)] == [0, 2, 2, 5]
    assert [o.text_idx[i] for i in range(5)] == [11, 12, 40, 41, 96] and [o.conf[i] for i in range(3)] == [0.9, 0.0, 0.25]
    lib.bbocr_free_result(out)
    bad = C.POINTER(_lib.bbocr_result)()
    assert lib.bbocr_result_unpack(blob, n.value - 4, C.byref(bad)) == -1                      # truncated
    raw = (C.c_char * n.value).from_address(blob.value)
    raw[8] = 7                                                                            # header says 7 boxes, offsets say 3
    assert lib.bbocr_result_unpack(blob, n.value, C.byref(bad)) == -1
    lib.bbocr_free_bytes(blob)
    assert lib.bbocr_dist_init(None, 0, 1, None) == -1 and lib.bbocr_bcast_weights(None, 0) == -1
    assert lib.bbocr_gather_results(None, None, 0, 0, None, None) == -1 and lib.bbocr_dist_finalize(None) == -1


def test_host_pools_are_sized_from_the_process_share_not_the_machine(lib):
    """eight ranks on one host must not each claim every core.  The library sizes its per-slot host pools from the CPUs the
    PROCESS may use (affinity mask / cgroup quota); the Python host divides an un-pinned rank's share by LOCAL_WORLD_SIZE."""
    import subprocess

    from bb_ocr_amd.reader import auto_host_threads

    here = len(os.sched_getaffinity(0))
    assert 1 <= lib.bbocr_host_cpu_share() <= here
    if here >= 2:      # a child pinned to two CPUs sees a share of two, whatever the machine has
        code = ("import os, sys; os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:2]); sys.path.insert(0, %r); "
                "from bb_ocr_amd import _lib; print(_lib.load().bbocr_host_cpu_share())" % ROOT)
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert out.returncode == 0 and out.stdout.strip() == b"2", out.stderr.decode()[-500:]
    assert auto_host_threads(1, 128) == 0                       # single process: the library's own rule
    assert auto_host_threads(8, 128) == 16 and auto_host_threads(8, 64) == 8 and auto_host_threads(8, 4) == 1
    assert sum(auto_host_threads(8, 96) for _ in range(8)) <= 96       # eight ranks together stay within the host
