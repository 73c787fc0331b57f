"""Oracle (test infrastructure): image input, resize, normalisation.

Restates, on numpy, the OpenCV / EasyOCR steps that sit in front of the CRAFT
detector.  Upstream modules followed (easyocr==1.7.2, un-vendored, see
oracle/__init__.py): ``easyocr/utils.py::reformat_input``,
``easyocr/imgproc.py::{loadImage,resize_aspect_ratio,normalizeMeanVariance}``;
OpenCV 4.10 ``imgproc/src/resize.cpp`` (8-bit bilinear, fixed point) and
``color_rgb.simd.hpp`` (RGB->gray: the 15-bit formula, PINNED by the reference's pre-processing vectors, see
oracle/__init__.py).  The resize / normalisation steps stay PARITY UNPINNED (no cv2 in this image).
"""
from __future__ import annotations

import io
import math
import os

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS

# cv2 fixed-point luma weights, OpenCV 4 (gray_shift = 15; pinned by tests/golden/legacy_preprocess)
_R2Y, _G2Y, _B2Y, _YUV_SHIFT = 9798, 19235, 3735, 15


def gray_from_3ch(img: np.ndarray, order: str = "bgr") -> np.ndarray:
    """cv2.cvtColor(img, COLOR_BGR2GRAY / COLOR_RGB2GRAY) for uint8 HWC."""
    a = img.astype(np.int32)
    if order == "bgr":
        b, g, r = a[..., 0], a[..., 1], a[..., 2]
    else:
        r, g, b = a[..., 0], a[..., 1], a[..., 2]
    y = (r * _R2Y + g * _G2Y + b * _B2Y + (1 << (_YUV_SHIFT - 1))) >> _YUV_SHIFT
    return y.astype(np.uint8)


def decode_file(path):
    """cv2.imread(path, IMREAD_GRAYSCALE) + loadImage(path) restated on PIL, ONE rule per container (re-verify list):
    JPEG -> libjpeg's own Y plane (grfmt_jpeg.cpp: out_color_space = JCS_GRAYSCALE; PIL's draft('L') requests the same);
    single-channel files -> stored samples; other colour files -> libpng's rgb_to_gray as OpenCV configures it
    (grfmt_png.cpp: png_set_rgb_to_gray(1, 0.299, 0.587) -> coefficients 9797 / 19234 / 3737, truncating >> 15)."""
    from PIL import Image

    pil = Image.open(path)
    rgb = np.asarray(pil.convert("RGB"))
    if pil.format in ("JPEG", "MPO") and pil.mode in ("RGB", "YCbCr"):
        y = Image.open(path)
        y.draft("L", y.size)
        grey = np.asarray(y.convert("L"))
        return rgb, (grey if grey.shape == rgb.shape[:2] else np.asarray(pil.convert("L")))
    if pil.mode in ("L", "1"):
        return rgb, np.asarray(pil.convert("L"))
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return rgb, ((r * 9797 + g * 19234 + b * 3737) >> 15).astype(np.uint8)


def jpeg_ycc_to_rgb(ycc: np.ndarray) -> np.ndarray:
    """libjpeg jdcolor.c::ycc_rgb_convert + build_ycc_rgb_table (libjpeg 6b API, shipped as libjpeg-turbo with Pillow / OpenCV; its SIMD
    paths are bit-exact with this C code): uint8 [..., 3] YCbCr -> uint8 [..., 3] RGB.  SCALEBITS = 16, FIX(x) = (int)(x * 65536 + 0.5),
    Cr_r = (FIX(1.40200) x + 2^15) >> 16, Cb_b = (FIX(1.77200) x + 2^15) >> 16, Cr_g = -FIX(0.71414) x, Cb_g = -FIX(0.34414) x + 2^15
    (x = sample - 128, arithmetic shifts), G = y + ((Cb_g + Cr_g) >> 16), range-limited.  Pinned against the decoder itself in
    tests/test_oracle_cpu.py (one YCbCr decode == the RGB decode and the Y-plane decode of the same file)."""
    fix = lambda v: int(v * 65536 + 0.5)
    x = np.arange(256, dtype=np.int64) - 128
    cr_r = (fix(1.40200) * x + 32768) >> 16
    cb_b = (fix(1.77200) * x + 32768) >> 16
    cr_g = -fix(0.71414) * x
    cb_g = -fix(0.34414) * x + 32768
    y = ycc[..., 0].astype(np.int64)
    cb, cr = ycc[..., 1], ycc[..., 2]
    out = np.stack([y + cr_r[cr], y + ((cb_g[cb] + cr_g[cr]) >> 16), y + cb_b[cb]], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def reformat_input(image):
    """easyocr/utils.py::reformat_input -> (img RGB uint8 HWC, img_cv_grey uint8 HW).

    File paths / bytes are decoded with PIL (cv2/skimage are absent); the gray plane of
    a path follows ``decode_file``'s per-container rule.  ndarray inputs follow upstream's
    channel rules exactly, including the BGR2GRAY-on-whatever-you-passed quirk.
    """
    from PIL import Image

    if isinstance(image, (str, os.PathLike)):
        return decode_file(os.path.expanduser(str(image)))
    if isinstance(image, (bytes, bytearray)):
        pil = Image.open(io.BytesIO(bytes(image)))
        img = np.asarray(pil.convert("RGB"))
        # upstream: imdecode -> BGR, BGR2RGB, then BGR2GRAY applied to the RGB array
        return img, gray_from_3ch(img, "bgr")
    if isinstance(image, np.ndarray):
        if image.ndim == 2:
            return np.repeat(image[:, :, None], 3, axis=2), image
        if image.ndim == 3 and image.shape[2] == 1:
            g = image[:, :, 0]
            return np.repeat(g[:, :, None], 3, axis=2), g
        if image.ndim == 3 and image.shape[2] == 3:
            return image, gray_from_3ch(image, "bgr")
        if image.ndim == 3 and image.shape[2] == 4:
            img = image[:, :, :3][:, :, ::-1]
            return np.ascontiguousarray(img), gray_from_3ch(img, "bgr")
        raise ValueError("Invalid input type. Supporting format = string(file path or url), bytes, numpy array")
    if hasattr(image, "convert"):  # PIL image (upstream: JpegImageFile)
        arr = np.asarray(image.convert("RGB"))
        return np.ascontiguousarray(arr[:, :, ::-1]), gray_from_3ch(arr, "bgr")
    raise ValueError("Invalid input type. Supporting format = string(file path or url), bytes, numpy array")


def _cv_round_f32(x: np.ndarray) -> np.ndarray:
    """cvRound on float32 values (round half to even)."""
    return np.rint(x.astype(np.float32)).astype(np.int32)


def linear_coeffs(ssize: int, dsize: int):
    """Per-destination source index + fixed-point (alpha0, alpha1) of cv::resize INTER_LINEAR."""
    inv_scale = float(dsize) / float(ssize)
    scale = 1.0 / inv_scale
    d = np.arange(dsize, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int32)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo = sx < 0
    fx[lo] = 0.0
    sx[lo] = 0
    hi = sx >= ssize - 1
    fx[hi] = 0.0
    sx[hi] = ssize - 1
    a0 = _cv_round_f32((np.float32(1.0) - fx) * np.float32(INTER_RESIZE_COEF_SCALE))
    a1 = _cv_round_f32(fx * np.float32(INTER_RESIZE_COEF_SCALE))
    sx1 = np.minimum(sx + 1, ssize - 1)
    return sx, sx1, a0, a1


def resize_linear_u8(src: np.ndarray, dsize_wh) -> np.ndarray:
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HW or HWC."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if dw <= 0 or dh <= 0:
        raise ValueError("resize: empty destination")
    squeeze = src.ndim == 2
    s = src[:, :, None] if squeeze else src
    sh, sw = s.shape[:2]
    if (sw, sh) == (dw, dh):
        out = s.copy()
    elif sw == 2 * dw and sh == 2 * dh:
        # cv::resize swaps INTER_LINEAR for the fast INTER_AREA path at exact 2x decimation
        a = s.astype(np.int32)
        out = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        x0, x1, a0, a1 = linear_coeffs(sw, dw)
        y0, y1, b0, b1 = linear_coeffs(sh, dh)
        a = s.astype(np.int32)
        rows = a[:, x0, :] * a0[None, :, None] + a[:, x1, :] * a1[None, :, None]  # [sh, dw, c], scaled 2^11
        r0 = rows[y0]
        r1 = rows[y1]
        v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
        out = np.clip(v, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_aspect_ratio(img: np.ndarray, square_size: int, mag_ratio: float = 1.0):
    """easyocr/imgproc.py::resize_aspect_ratio -> (float32 canvas HWC padded to x32, ratio, size_heatmap)."""
    height, width, channel = img.shape
    target_size = mag_ratio * max(height, width)
    if target_size > square_size:
        target_size = square_size
    ratio = target_size / max(height, width)
    target_h, target_w = int(height * ratio), int(width * ratio)
    proc = resize_linear_u8(img, (target_w, target_h))
    target_h32, target_w32 = target_h, target_w
    if target_h % 32 != 0:
        target_h32 = target_h + (32 - target_h % 32)
    if target_w % 32 != 0:
        target_w32 = target_w + (32 - target_w % 32)
    resized = np.zeros((target_h32, target_w32, channel), dtype=np.float32)
    resized[0:target_h, 0:target_w, :] = proc
    size_heatmap = (int(target_w32 / 2), int(target_h32 / 2))
    return resized, ratio, size_heatmap


MEAN = (0.485, 0.456, 0.406)
VARIANCE = (0.229, 0.224, 0.225)


def normalize_mean_variance(in_img: np.ndarray) -> np.ndarray:
    """easyocr/imgproc.py::normalizeMeanVariance (RGB order, float32)."""
    img = in_img.copy().astype(np.float32)
    img -= np.array([MEAN[0] * 255.0, MEAN[1] * 255.0, MEAN[2] * 255.0], dtype=np.float32)
    img /= np.array([VARIANCE[0] * 255.0, VARIANCE[1] * 255.0, VARIANCE[2] * 255.0], dtype=np.float32)
    return img


def detector_input(img_rgb: np.ndarray, canvas_size: int = 2560, mag_ratio: float = 1.0):
    """detection.py::test_net preprocessing for one image -> (x [3,H32,W32] f32, ratio)."""
    resized, ratio, _ = resize_aspect_ratio(img_rgb, canvas_size, mag_ratio)
    x = np.transpose(normalize_mean_variance(resized), (2, 0, 1))
    return np.ascontiguousarray(x), ratio


# ---------------------------------------------------------------------------------
# PIL bicubic (Pillow Resample.c, 8 bpc) restated on numpy so the GPU kernel has an
# integer-exact spec that does not depend on the installed Pillow version.  The
# tests also check it against PIL itself.
# ---------------------------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2


def _bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc -> (bounds [out,2], kk int32 [out,ksize])."""
    support0 = 2.0
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> _PRECISION_BITS, 0, 255).astype(np.uint8)


def pil_resize_bicubic_u8(src: np.ndarray, dsize_wh) -> np.ndarray:
    """PIL ``Image.resize((w, h), Image.BICUBIC)`` for mode 'L' (horizontal pass, then vertical)."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = src.shape
    if (sw, sh) == (dw, dh):
        return src.copy()
    cur = src
    if dw != sw:
        bounds, kk = pil_bicubic_coeffs(sw, dw)
        out = np.zeros((cur.shape[0], dw), dtype=np.uint8)
        a = cur.astype(np.int64)
        for xx in range(dw):
            xmin, n = bounds[xx]
            acc = (a[:, xmin:xmin + n] * kk[xx, :n][None, :].astype(np.int64)).sum(axis=1) + (1 << (_PRECISION_BITS - 1))
            out[:, xx] = _clip8(acc)
        cur = out
    if dh != sh:
        bounds, kk = pil_bicubic_coeffs(sh, dh)
        out = np.zeros((dh, cur.shape[1]), dtype=np.uint8)
        a = cur.astype(np.int64)
        for yy in range(dh):
            ymin, n = bounds[yy]
            acc = (a[ymin:ymin + n, :] * kk[yy, :n][:, None].astype(np.int64)).sum(axis=0) + (1 << (_PRECISION_BITS - 1))
            out[yy, :] = _clip8(acc)
        cur = out
    return cur
