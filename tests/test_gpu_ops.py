"""-m gpu: single kernels through the C ABI against the oracle / torch fp32 on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _conv_case(reader, N, H, W, Cin, Cout, K, pad, dil, relu_in, relu_out, out_f32, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = _bf16(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, K, K, generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wq = _bf16(w)
    xin = F.relu(x) if relu_in else x
    ref = F.conv2d(xin.double(), wq.double(), b.double(), padding=pad, dilation=dil)
    if relu_out:
        ref = F.relu(ref)
    ref = ref.permute(0, 2, 3, 1).float()
    OH, OW = ref.shape[1:3]
    store = (Cout + 15) // 16 * 16
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    out = torch.full((N, OH, OW, store), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device="cuda")
    wn = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    bn = np.ascontiguousarray(b.numpy(), dtype=np.float32)
    rc = reader._lib.bbocr_op_conv2d(reader._h, C.c_void_p(xd.data_ptr()), N, H, W, Cin, wn.ctypes.data_as(C.POINTER(C.c_float)),
                                     bn.ctypes.data_as(C.POINTER(C.c_float)), Cout, K, K, pad, dil, int(relu_in), int(relu_out), int(out_f32),
                                     C.c_void_p(out.data_ptr()), 0, 0, None)
    reader._check(rc)
    got = out.float().cpu()[..., :Cout]
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    tol = (2e-5 if out_f32 else 6e-3) * max(scale, 1.0)     # fp32 accumulate; bf16 output rounding 2^-8 relative
    err = (got - ref).abs().max().item()
    assert err <= tol, f"conv err {err} > {tol} (scale {scale})"
    if store > Cout:
        pad_part = out.float().cpu()[..., Cout:]
        assert torch.isfinite(pad_part).all()


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, K, pad, dil, relu_in, relu_out, out_f32
    (1, 16, 16, 32, 64, 3, 1, 1, 0, 1, 0),      # BN=64 tile, exact tile
    (2, 19, 37, 64, 64, 3, 1, 1, 0, 1, 0),      # ragged edges, 2 chunks
    (1, 24, 40, 64, 128, 3, 1, 1, 1, 0, 0),     # BN=128, ReLU on load
    (1, 20, 28, 128, 256, 3, 1, 1, 0, 1, 0),    # BN=256
    (1, 17, 23, 256, 512, 3, 1, 1, 0, 0, 0),    # two cout tiles
    (1, 15, 20, 64, 256, 3, 6, 6, 0, 0, 0),     # dilation 6 (fc6)
    (2, 12, 20, 96, 256, 1, 0, 1, 0, 1, 0),     # 1x1
    (1, 4, 70, 64, 256, 2, 0, 1, 0, 1, 0),      # 2x2 valid (CRNN last conv), short tile rows
    (1, 8, 50, 128, 256, 3, 1, 1, 0, 1, 0),     # H = 8 tile
    (1, 9, 33, 32, 32, 3, 1, 1, 0, 1, 0),       # Cout 32 (store 32 of a 64 tile)
    (1, 9, 33, 32, 16, 3, 1, 1, 0, 1, 0),       # Cout 16
    (1, 5, 40, 256, 97, 1, 0, 1, 0, 0, 1),      # prediction layer, fp32 out, Cout 97 -> 112
    (1, 30, 45, 256, 2048, 1, 0, 1, 0, 0, 0),   # LSTM input projection shape
    (1, 18, 21, 32, 64, 1, 0, 1, 0, 0, 0),      # single k-step (1x1, one chunk)
    (1, 18, 21, 64, 256, 1, 0, 1, 1, 1, 0),     # two k-steps
    (1, 33, 47, 128, 128, 3, 1, 1, 0, 1, 0),    # BN=128, 4 chunks, two-part patch staging
    (3, 40, 52, 512, 512, 3, 1, 1, 0, 1, 0),    # long K loop (144 k-steps), batch 3
])
def test_conv_mfma_vs_fp64(reader, cfg):
    _conv_case(reader, *cfg)


@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, mode, relu_out, pool_relu, store_full
    (2, 32, 48, 64, 64, 1, 1, 0, 0),       # conv1_2-like: pooled output only
    (1, 30, 44, 64, 128, 1, 0, 1, 1),      # conv2_2-like: skip tensor (no ReLU) + ReLU'd pooled tensor
    (1, 22, 38, 128, 256, 1, 1, 0, 0),     # ragged tiles, odd tile counts
    (2, 16, 70, 128, 128, 2, 1, 0, 0),     # CRNN (2,1) pool
    (1, 8, 90, 256, 256, 2, 1, 0, 0),      # H = 8 tile rows
    (1, 64, 40, 32, 64, 1, 1, 0, 0),       # BN=64 config
])
def test_conv_fused_maxpool(reader, cfg):
    N, H, W, Cin, Cout, mode, relu_out, pool_relu, store_full = cfg
    g = torch.Generator().manual_seed(7)
    x = _bf16(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    y = F.conv2d(x.double(), _bf16(w).double(), b.double(), padding=1)
    full_ref = (F.relu(y) if relu_out else y)
    pin = F.relu(full_ref) if pool_relu else full_ref
    pooled_ref = F.max_pool2d(pin, (2, 2) if mode == 1 else (2, 1)).permute(0, 2, 3, 1).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    full = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    pooled = torch.full(tuple(pooled_ref.shape), float("nan"), dtype=torch.bfloat16, device="cuda")
    wn = np.ascontiguousarray(w.numpy(), dtype=np.float32)
    bn = np.ascontiguousarray(b.numpy(), dtype=np.float32)
    rc = reader._lib.bbocr_op_conv2d(reader._h, C.c_void_p(xd.data_ptr()), N, H, W, Cin, wn.ctypes.data_as(C.POINTER(C.c_float)),
                                     bn.ctypes.data_as(C.POINTER(C.c_float)), Cout, 3, 3, 1, 1, 0, int(relu_out), 0,
                                     C.c_void_p(full.data_ptr()) if store_full else None, mode, int(pool_relu), C.c_void_p(pooled.data_ptr()))
    reader._check(rc)
    tol = 6e-3 * max(y.abs().max().item(), 1.0)
    got = pooled.float().cpu()
    assert torch.isfinite(got).all() and (got - pooled_ref).abs().max().item() <= tol
    if store_full:
        gf = full.float().cpu()
        assert torch.isfinite(gf).all() and (gf - full_ref.permute(0, 2, 3, 1).float()).abs().max().item() <= tol


def test_resize_u8_bit_exact(reader):
    from oracle import imgproc

    rng = np.random.default_rng(3)
    for (sh, sw, dh, dw, c) in [(37, 53, 64, 91, 1), (120, 200, 64, 107, 1), (64, 64, 32, 32, 1), (50, 70, 50, 70, 3), (91, 47, 40, 33, 3),
                                (13, 300, 64, 1477, 1)]:
        src = rng.integers(0, 256, (2, sh, sw, c), dtype=np.uint8)
        d = torch.from_numpy(src).cuda()
        out = torch.zeros((2, dh, dw, c), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # the fill runs on torch's stream, the library on its own non-blocking one
        reader._check(reader._lib.bbocr_op_resize_u8(reader._h, C.c_void_p(d.data_ptr()), 2, sh, sw, c, C.c_void_p(out.data_ptr()), dh, dw))
        got = out.cpu().numpy()
        for n in range(2):
            ref = imgproc.resize_linear_u8(src[n] if c > 1 else src[n, :, :, 0], (dw, dh))
            assert np.array_equal(got[n] if c > 1 else got[n, :, :, 0], ref), (sh, sw, dh, dw, c)


def test_jpeg_colour_conversion_on_device_is_libjpegs(reader):
    """a2, decode once: bbocr_op_ycc_to_rgb on ALL 2^24 (Y, Cb, Cr) triples against the restatement of jdcolor.c (which the CPU suite pins
    against the decoder): RGB bit for bit, gray = the Y channel."""
    from oracle import imgproc

    v = np.arange(256, dtype=np.uint8)
    ycc = np.stack(np.meshgrid(v, v, v, indexing="ij"), axis=-1).reshape(1, 4096, 4096, 3)
    rgb, gray = reader.pages_from_ycc(torch.from_numpy(np.ascontiguousarray(ycc)).cuda())
    assert np.array_equal(gray.cpu().numpy(), ycc[..., 0])
    assert np.array_equal(rgb.cpu().numpy(), imgproc.jpeg_ycc_to_rgb(ycc))
    # Pillow's padded pixels (Y Cb Cr x): the fourth byte is ignored; pages handed over as a LIST reach the card through bbocr_upload_pages
    rng = np.random.default_rng(2)
    tight = rng.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    padded = np.concatenate([tight, rng.integers(0, 256, (3, 37, 53, 1), dtype=np.uint8)], axis=-1)
    rgb4, gray4 = reader.pages_from_ycc(reader._to_dev([padded[k] for k in range(3)]))
    assert np.array_equal(rgb4.cpu().numpy(), imgproc.jpeg_ycc_to_rgb(tight)) and np.array_equal(gray4.cpu().numpy(), tight[..., 0])
    ro = [np.frombuffer(tight[k].tobytes(), dtype=np.uint8).reshape(37, 53, 3) for k in range(3)]          # read-only pages, as PIL hands them out
    assert np.array_equal(reader._to_dev(ro).cpu().numpy(), tight)
    with pytest.raises(ValueError):
        reader.pages_from_ycc(torch.zeros((1, 4, 4, 5), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        reader._to_dev([tight[0], tight[1, :20]])
    with pytest.raises(ValueError):
        reader._to_dev([])
    import ctypes as C_
    dst = torch.empty((2, 16), dtype=torch.uint8, device="cuda")
    ptrs = (C_.c_void_p * 2)(tight[0].ctypes.data, None)                       # a null page: status code and message, no fault
    assert reader._lib.bbocr_upload_pages(reader._h, ptrs, 2, 16, C_.c_void_p(dst.data_ptr())) == -1
    assert b"null page" in reader._lib.bbocr_last_error(reader._h)


def test_ctc_matches_oracle(reader):
    from oracle import recog

    rng = np.random.default_rng(5)
    n, T, Cn, cs = 7, 83, 97, 112
    logits = (rng.standard_normal((n, T, cs)) * 4).astype(np.float32)
    # force repeats, blanks and an all-blank row
    logits[0, :, 0] += 30.0
    logits[1, 10:30, 5] += 30.0
    logits[2, ::2, 0] += 30.0
    d = torch.from_numpy(logits).cuda()
    off = (C.c_int * (n + 1))()
    idx = (C.c_int * (n * T))()
    conf = (C.c_double * n)()
    reader._check(reader._lib.bbocr_op_ctc(reader._h, C.c_void_p(d.data_ptr()), n, T, Cn, cs, off, idx, conf, None, 0))
    ref = recog.predict_from_logits(logits[:, :, :Cn])
    for i in range(n):
        text = "".join(recog.CHARACTER[idx[k]] for k in range(off[i], off[i + 1]))
        assert text == ref[i][0]
        assert conf[i] == pytest.approx(float(ref[i][1]), rel=2e-5, abs=1e-12)
    assert off[1] - off[0] == 0 and conf[0] == 0.0
    # f4: allowlist / blocklist = recognizer_predict's ignore_idx (zeroed classes, renormalised rows)
    import bb_ocr_amd
    from bb_ocr_amd.reader import ignore_mask

    for kw in ({"allowlist": "0123456789"}, {"blocklist": "aeiouAEIOU -"}):
        words = ignore_mask(bb_ocr_amd.CHARACTER, list(bb_ocr_amd.CHARSET), **kw)
        ign = [i for i in range(Cn) if (words[i >> 5] >> (i & 31)) & 1]
        assert 0 not in ign and len(ign) > 0
        mask = (C.c_uint * 4)(*words)
        reader._check(reader._lib.bbocr_op_ctc(reader._h, C.c_void_p(d.data_ptr()), n, T, Cn, cs, off, idx, conf, mask, 0))
        ref = recog.predict_from_logits(logits[:, :, :Cn], ignore_idx=ign)
        for i in range(n):
            text = "".join(recog.CHARACTER[idx[k]] for k in range(off[i], off[i + 1]))
            assert text == ref[i][0]
            assert conf[i] == pytest.approx(float(ref[i][1]), rel=2e-5, abs=1e-12)
            if "allowlist" in kw:
                assert set(text) <= set(kw["allowlist"])
            else:
                assert not (set(text) & set(kw["blocklist"]))
    # f4: decoder='beamsearch' -- device probabilities (softmax + mask + renormalisation) into the host ctcBeamSearch; the text must equal
    # the oracle's search on ITS float32 probabilities (ties in the ranking would expose a 1-ulp difference), the confidence is the greedy one
    soft = (rng.standard_normal((n, T, cs)) * 1.5).astype(np.float32)
    soft[:, :, 0] += 2.0
    d2 = torch.from_numpy(soft).cuda()
    for bw, mask, ign in ((5, None, ()), (3, (C.c_uint * 4)(*words), ign)):
        reader._check(reader._lib.bbocr_op_ctc(reader._h, C.c_void_p(d2.data_ptr()), n, T, Cn, cs, off, idx, conf, mask, bw))
        ref = recog.predict_from_logits(soft[:, :, :Cn], ignore_idx=ign, decoder="beamsearch", beam_width=bw)
        greedy = recog.predict_from_logits(soft[:, :, :Cn], ignore_idx=ign)
        for i in range(n):
            text = "".join(recog.CHARACTER[idx[k]] for k in range(off[i], off[i + 1]))
            assert text == ref[i][0], (bw, i)
            assert conf[i] == pytest.approx(float(greedy[i][1]), rel=2e-5, abs=1e-12)
        assert any(r[0] != g[0] for r, g in zip(ref, greedy))      # the case is not degenerate: the search changes some strings


def test_preprocess_chain_bit_exact_vs_oracle(reader):
    """f2: every stage of preprocess_for_book_cover and the whole chain, bit for bit against oracle/preprocess.py (whose PIL
    stages are pinned against Pillow in the CPU suite)."""
    import ctypes as C

    from bb_ocr_amd import preprocess as dev_pp, synth
    from oracle import preprocess as pp

    rng = np.random.default_rng(11)

    def stage(k, a, param, dh=None, dw=None):
        src = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        dh, dw = dh or a.shape[0], dw or a.shape[1]
        dst = torch.empty((dh, dw), dtype=torch.uint8, device="cuda")
        reader._check(reader._lib.bbocr_op_preprocess_stage(reader._h, k, C.c_void_p(src.data_ptr()), a.shape[0], a.shape[1],
                                                            C.c_void_p(dst.data_ptr()), dh, dw, float(param)))
        return dst.cpu().numpy()

    for shape in ((67, 91), (128, 200), (301, 257), (184, 316), (90, 128)):     # incl. one axis only a multiple of the CLAHE grid
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        b = rng.normal(190, 35, shape).clip(0, 255).astype(np.uint8)
        for img in (a, b):
            dh, dw = int(shape[0] * 1.5), int(shape[1] * 1.5)
            assert np.array_equal(stage(0, img, 0, dh, dw), pp.resize_cubic_u8(img, dw, dh))
            assert np.array_equal(stage(1, img, 3.0), pp.gaussian_blur3_u8(img, 3.0))
            assert np.array_equal(stage(2, img, 1.9), pp.pil_contrast_L(img, 1.9))
            assert np.array_equal(stage(3, img, 1.2), pp.pil_brightness_L(img, 1.2))
            assert np.array_equal(stage(4, img, 2.5), pp.clahe_u8(img, 2.5, (8, 8)))
            assert np.array_equal(stage(5, img, 1.0), pp.pil_unsharp_L(img, 1.0, 30, 3))
            assert np.array_equal(stage(7, img, 20), pp.pil_unsharp_L(img, 1.0, 20, 3))
            assert np.array_equal(stage(1, img, 5.0), pp.gaussian_blur3_u8(img, 5.0))
        # exact ties of the cubic resize (flat columns a, a, a+1, a+1 under the 1/2 phase): decided in 128-bit integers, half to even
        ties = np.repeat(np.tile(np.array([30, 30, 30, 31, 31, 31], dtype=np.uint8), 20)[None, :], shape[0], axis=0)
        assert pp.resize_cubic_near_ties(ties, 180, int(shape[0] * 1.5), 1e-9).mean() > 0.05
        assert np.array_equal(stage(0, ties, 0, int(shape[0] * 1.5), 180), pp.resize_cubic_u8(ties, 180, int(shape[0] * 1.5)))
        odd = rng.integers(0, 256, (shape[0], 97), dtype=np.uint8)                 # a size ratio that is not 3/2: general rational phases
        assert np.array_equal(stage(0, odd, 0, shape[0] + 13, 131), pp.resize_cubic_u8(odd, 131, shape[0] + 13))
        # downscales: 0.7x stays on the LDS-tiled kernel (larger source windows), 0.3x exceeds its window and takes the per-pixel kernel
        for f in (0.7, 0.3):
            sh_, sw_ = max(int(shape[0] * f), 8), max(int(shape[1] * f), 8)
            assert np.array_equal(stage(0, a, 0, sh_, sw_), pp.resize_cubic_u8(a, sw_, sh_)), (shape, f)
    # planes on the dword kernels (W % 4 == 0) whose sizes are NOT multiples of the CLAHE grid / of the resize and blur tiles, a narrow one
    # (W = 8: every thread of the fused row passes is both the first and the last of its row) and one shorter than a column strip
    for shape in ((250, 332), (403, 260), (37, 8), (9, 64), (522, 1028)):
        img = rng.normal(150, 60, shape).clip(0, 255).astype(np.uint8)
        if shape[1] >= 64:
            assert np.array_equal(stage(4, img, 2.5), pp.clahe_u8(img, 2.5, (8, 8))), shape
            assert np.array_equal(stage(4, img, 40.0), pp.clahe_u8(img, 40.0, (8, 8))), shape
            dh, dw = int(shape[0] * 1.5), int(shape[1] * 1.5)
            assert np.array_equal(stage(0, img, 0, dh, dw), pp.resize_cubic_u8(img, dw, dh)), shape
        assert np.array_equal(stage(5, img, 1.0), pp.pil_unsharp_L(img, 1.0, 30, 3)), shape
        assert np.array_equal(stage(7, img, 60), pp.pil_unsharp_L(img, 1.0, 60, 3)), shape
        assert np.array_equal(stage(5, img, 2.0), pp.pil_unsharp_L(img, 2.0, 30, 3)), shape        # box radius > 0: the per-pass kernels
    # the whole chain on a rendered page (BGR) and on noise
    page = synth.page(77, width=640, height=400, lines=8, margin=24)[0][:, :, ::-1]
    for bgr in (np.ascontiguousarray(page), rng.integers(0, 256, (123, 211, 3), dtype=np.uint8)):
        got, path, steps = dev_pp.preprocess_for_book_cover(bgr, reader=reader)
        assert path is None and steps == pp.STEPS
        assert np.array_equal(got, pp.preprocess_for_book_cover(bgr))
        got, path, steps = dev_pp.preprocess_for_book_cover(bgr, reader=reader, legacy=True)
        assert steps == dev_pp.LEGACY_STEPS and np.array_equal(got, pp.preprocess_for_book_cover_legacy(bgr))
    # stages skipped through bbocr_preproc_params (parameter 0): every subset the chain's branches distinguish
    bgr = rng.integers(0, 256, (96, 144, 3), dtype=np.uint8)
    dev = torch.from_numpy(bgr).cuda()
    for kw in (dict(scale=0.0), dict(blur_sigma=0.0), dict(contrast=0.0, brightness=0.0), dict(clahe_clip=0.0), dict(unsharp_percent=0),
               dict(blur_sigma=0.0, clahe_clip=0.0), dict(scale=0.0, blur_sigma=0.0, contrast=0.0, brightness=0.0, clahe_clip=0.0, unsharp_percent=0),
               # other scales: 1.02 / 0.8 -> 16-row resize tiles with the gray conversion in their window load, 0.3 -> gray plane + per-pixel resize
               dict(scale=1.02), dict(scale=0.8), dict(scale=0.3, clahe_clip=0.0)):
        ref_kw = dict(scale=1.5, blur_sigma=3.0, contrast=1.9, brightness=1.2, clahe_clip=2.5, unsharp_percent=30)
        ref_kw.update(kw)
        assert np.array_equal(dev_pp.preprocess_bgr_device(reader, dev, **kw).cpu().numpy(), pp.preprocess_chain(bgr, **ref_kw)), kw


@pytest.mark.parametrize("n", [2, 4, 5, 6, 1])
def test_legacy_preprocess_fixtures_on_device(reader, n):
    """f2 against the reference's own vectors (the -m gpu twin of tests/test_oracle_cpu.py::test_legacy_preprocess_fixtures): the device
    chain with the legacy parameters is bit-identical to the oracle on the reference's five inputs, hence book2 is bit-exact against the
    reference's stored output and the others differ from it on the same <= 64 near-tie pixels (Intel IPP's float32 cubic resize)."""
    import os

    from PIL import Image

    from bb_ocr_amd import preprocess as dev_pp
    from oracle import preprocess as pp
    from test_oracle_cpu import LEGACY_RESIDUAL

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rgba = np.array(Image.open(os.path.join(here, "legacy_preprocess", f"book{n}.png")))
    want = np.array(Image.open(os.path.join(here, "ref_images", f"book{n}_preprocessed.png")))
    bgr = np.ascontiguousarray(rgba[..., 2::-1])
    got = dev_pp.preprocess_bgr_device(reader, torch.from_numpy(bgr).cuda(), legacy=True).cpu().numpy()
    assert np.array_equal(got, pp.preprocess_for_book_cover_legacy(bgr))
    assert int((got != want).sum()) == LEGACY_RESIDUAL[n]


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [
    # N, H, W, Cin, Cout, relu_out, pool_mode, pool_relu        (3x3 / pad 1: the layers launch_dma_one's `lean` rule picks)
    (2, 32, 48, 64, 64, 1, 0, 0),        # 64-cout tile, interior tiles only: conv_epilogue_plain_lean<WHOLE>, `inside` fast path
    (1, 19, 37, 64, 128, 1, 0, 0),       # 128-cout tile, ragged right / bottom edges (per-lane bounds tests)
    (1, 21, 50, 128, 256, 0, 0, 0),      # two cout tiles, no ReLU: negative values stored as they are
    (2, 32, 32, 128, 128, 1, 1, 0),      # conv_epilogue_pool2x2_lean on a 128-cout layer (conv3_3 / conv4_3 shape): pooled tensor only
    (1, 32, 48, 64, 128, 0, 1, 1),       # pooled, pool_relu without relu_out (conv2_2's shape when the skip tensor is not kept)
    (1, 18, 30, 64, 64, 1, 1, 0),        # pooled 64-cout, ragged
])
def test_lean_epilogues_equal_the_shared_epilogue_bit_for_bit(states, prec, cfg):
    """the lean epilogues (packed-int16 ReLU on the rounded pair, max-before-bias pooling, whole-line regrouped stores) replace the
    shared epilogue on most trunk layers.  The shared epilogue is what the SAME launch runs when it writes fp32 (out_f32 disables the lean
    rule): its fp32 values, rounded to the element type with RNE (and max-pooled), must equal the lean launch's 16-bit output BIT FOR BIT --
    both element types, ragged tiles, negative pre-ReLU values, pool_relu without relu_out."""
    import bb_ocr_amd

    N, H, W, Cin, Cout, relu_out, pool_mode, pool_relu = cfg
    dtype = torch.bfloat16 if prec == "bf16" else torch.float16
    r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states, precision=prec, recognizer=False)
    try:
        g = torch.Generator().manual_seed(1234 + Cout + H)
        x = torch.randn(N, H, W, Cin, generator=g).to(dtype).cuda()
        w = np.ascontiguousarray((torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)).numpy(), dtype=np.float32)
        b = np.ascontiguousarray((torch.randn(Cout, generator=g) * 0.3).numpy(), dtype=np.float32)

        def run(out_f32, pool):
            full = torch.full((N, H, W, Cout), float("nan"), dtype=torch.float32 if out_f32 else dtype, device="cuda")
            pooled = torch.full((N, H // 2, W // 2, Cout), float("nan"), dtype=dtype, device="cuda") if pool else None
            rc = r._lib.bbocr_op_conv2d(r._h, C.c_void_p(x.data_ptr()), N, H, W, Cin, w.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                        Cout, 3, 3, 1, 1, 0, relu_out, int(out_f32), None if pool else C.c_void_p(full.data_ptr()), 1 if pool else 0, pool_relu,
                                        C.c_void_p(pooled.data_ptr()) if pool else None)
            r._check(rc)
            return pooled if pool else full

        ref32 = run(True, False)                                   # shared epilogue, fp32: bias (+ ReLU) on the raw accumulators
        assert torch.isfinite(ref32).all() and (ref32 < 0).any() == (not relu_out)
        if pool_mode:
            want = F.max_pool2d(ref32.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            if pool_relu:
                want = F.relu(want)
            want = want.contiguous().to(dtype)                     # RNE; max and ReLU commute with the (monotone) rounding
            got = run(False, True)
        else:
            want = ref32.to(dtype)
            got = run(False, False)
        assert got.shape == want.shape
        gi, wi = got.view(torch.int16), want.view(torch.int16)
        zero_sign = (gi != wi) & ((gi & 0x7fff) == 0) & ((wi & 0x7fff) == 0)       # +0 vs -0 cannot arise (acc + bias == -0 needs both -0): counted to be sure
        assert int(zero_sign.sum()) == 0
        assert torch.equal(gi, wi), f"{int((gi != wi).sum())} of {gi.numel()} values differ"
    finally:
        r.close()


def test_fused_upconv4_kernel_equals_the_two_launch_path_bit_for_bit(tmp_path):
    """Round 4: conv3x3_up4_kernel (CRAFT upconv4 as one launch) against conv1x1<ADDUP> + 3x3 on random fp16 tensors, 4 pages of 240 x 320
    (1,200 tiles on 512 persistent workgroups: second and third tiles per workgroup), eight runs: 0 differing values, every run.  The
    harness (tools/micro/up4_check.hip) is what found the two defects of the round-2 kernel: a bare s_barrier passed with an LDS read still
    queued, and packed-fp32 VALU writes into registers that MFMAs still in flight read as srcC.  Compiled here with the box's hipcc against
    the in-tree library."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "up4_check")
    lib = os.path.join(root, "bb-ocr_amd")
    cc = subprocess.run([hipcc, "-O2", "-std=c++20", "--offload-arch=gfx950", "-I" + os.path.join(lib, "csrc"), "-I" + os.path.join(root, "include"),
                         os.path.join(root, "tools", "micro", "up4_check.hip"), "-L" + lib, "-lbbocr", "-Wl,-rpath," + lib, "-o", exe],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert cc.returncode == 0, cc.stdout.decode()[-2000:]
    run = subprocess.run([exe, "4", "240", "320"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)       # a child process: own HIP context
    out = run.stdout.decode()
    lines = [l for l in out.splitlines() if l.startswith("run ")]
    assert run.returncode == 0 and len(lines) == 8, out[-2000:]
    assert all(l.split(":")[1].strip().startswith("0 values differ from the two-launch path, 0 from run 0") for l in lines), out[-2000:]
