"""TEST INFRASTRUCTURE -- CPU restatement of the reference's OCR pre-processing.

Follows ``pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py::preprocess_for_book_cover`` (:147-160):
``cv2.imread`` (BGR) -> ``to_grayscale`` (:25-30, cv2.COLOR_BGR2GRAY) -> ``resize(1.5)`` (:131-138, cv2.INTER_CUBIC) ->
``denoise(3)`` (:32-37, cv2.GaussianBlur 3x3 sigma 3) -> ``increase_contrast(1.9)`` (:69-83, PIL ImageEnhance.Contrast) ->
``increase_brightness(1.2)`` (:85-99, PIL ImageEnhance.Brightness) -> ``clahe(2.5)`` (:48-56, cv2.createCLAHE 8x8 tiles) ->
``sharpen(0.3)`` (:101-115, PIL ImageFilter.UnsharpMask(radius 1, percent 30, threshold 3)).

PARITY: PINNED by the reference's own stored vectors -- five (input PNG, pre-processed PNG) pairs of the LEGACY chain
(pipeline_components/img_to_json/ocr_testing/preprocessing/image_preprocessor.py:221-252: same stage functions, parameters sigma 5,
contrast 1.3, CLAHE 2.0, unsharp 20 %, no brightness step), committed under tests/golden/legacy_preprocess/ and replayed by
tests/test_oracle_cpu.py::test_legacy_preprocess_fixtures: bit-exact on the smallest pair, <= 64 of 1.3 M pixels off on the others,
all of them next to a cubic-resize value within 2e-5 of a rounding boundary (Intel IPP's float32 evaluation inside cv2.resize, see
resize_cubic_u8).  That pins BGR2GRAY (OpenCV 4's 15-bit coefficients), the cubic resize (IPP semantics, not OpenCV's fixed-point
path), GaussianBlur's 8.8 fixed-point kernel, CLAHE (clip / redistribute / LUT rounding / float blend) and, once more, the three
PIL stages, which are also pinned against Pillow itself (tests/test_oracle_cpu.py runs the real ImageEnhance / ImageFilter).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import math

import numpy as np


# ------------------------------------------------------------------------------------------------ cv2.cvtColor(BGR2GRAY), 8u
def bgr2gray(bgr: np.ndarray) -> np.ndarray:
    """cv2 color_rgb.simd.hpp RGB2Gray<uchar>, OpenCV 4: 15-bit fixed point, B 3735, G 19235, R 9798, round to nearest.
    (OpenCV 3's 14-bit 1868/9617/4899 does NOT reproduce the reference's stored outputs; the 15-bit one does.)"""
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ cv2.resize INTER_CUBIC, 8u
# opencv-python wheels (x86-64, every platform) are built with Intel IPP, and cv::resize hands 8-bit INTER_CUBIC to
# ippiResizeCubic_8u (B = 0, C = 0.75, i.e. the same A = -0.75 kernel; imgproc/src/resize.cpp::ipp_resize: only NEAREST, AREA and
# 8-bit LINEAR are kept away from IPP) -- NOT to OpenCV's own 11-bit fixed-point HResizeCubic / VResizeCubic, which the round-1
# oracle restated and which is ~10 % of pixels off the reference's stored outputs.  IPP evaluates the cubic in floating point;
# what the reference's five stored (input, output) pairs pin (tests/golden/legacy_preprocess, tests/test_oracle_cpu.py) is:
#   value = sum_j sum_i wy_j wx_i src[clamp(y0 + j), clamp(x0 + i)],  w = cubic convolution weights (A = -0.75) at the exact
#   rational phase t = frac((d + 0.5) * src / dst - 0.5),  result = round-half-to-EVEN(value), borders replicated.
# The restatement below computes exactly that mathematical value (float64, and exact integer arithmetic for the pixels whose
# value is within 1e-9 of a rounding boundary).  IPP's float32 evaluation differs from it only on pixels whose exact value lies
# within ~2e-5 of x.5 (<= 64 of 1.3 M pixels in the reference's fixtures, 0 on the smallest).
_A_NUM, _A_DEN = -3, 4


def _cubic_axis_exact(dst: int, src: int):
    """Per destination index: first tap (floor - 1), the four weights as float64 and as exact integers over K = 4 (2 dst)^3."""
    d = np.arange(dst, dtype=np.int64)
    num = (2 * d + 1) * src - dst                       # f = num / (2 dst)
    den = 2 * dst
    s = num // den                                      # floor
    n = num - s * den                                   # t = n / den in [0, 1)
    K = 4 * den ** 3
    no = n.astype(object)
    D = den

    def inner(x):                                       # 4 D^3 ((A + 2) x^3 - (A + 3) x^2 + 1), x = x / D, A = -3/4
        return 5 * x ** 3 - 9 * D * x ** 2 + 4 * D ** 3

    def outer(x):                                       # 4 D^3 (A x^3 - 5 A x^2 + 8 A x - 4 A)
        return -3 * x ** 3 + 15 * D * x ** 2 - 24 * D * D * x + 12 * D ** 3

    ci = np.stack([outer(no + D), inner(no), inner(D - no), outer(2 * D - no)], axis=-1)     # object ints, rows sum to K
    cf = (ci / K).astype(np.float64)                    # big-int true division: correctly rounded doubles
    return s - 1, cf, ci, K


def _resize_cubic_value(src: np.ndarray, dw: int, dh: int):
    H, W = src.shape
    x0, cx, ix, KX = _cubic_axis_exact(dw, W)
    y0, cy, iy, KY = _cubic_axis_exact(dh, H)
    s = src.astype(np.float64)
    cols = [np.clip(x0 + k, 0, W - 1) for k in range(4)]
    rows = [np.clip(y0 + k, 0, H - 1) for k in range(4)]
    hor = sum(s[:, cols[k]] * cx[:, k][None, :] for k in range(4))
    val = sum(hor[rows[k], :] * cy[:, k][:, None] for k in range(4))
    return val, (rows, cols, iy, ix, KX * KY)


def resize_cubic_near_ties(src: np.ndarray, dw: int, dh: int, eps: float) -> np.ndarray:
    """Mask of destination pixels whose exact bicubic value lies within ``eps`` of a rounding boundary (x.5): the only pixels on
    which a float32 evaluation (Intel IPP inside cv2.resize) can differ from the exactly rounded value."""
    val, _ = _resize_cubic_value(src, dw, dh)
    return np.abs(val - np.floor(val) - 0.5) < eps


def resize_cubic_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC) for one 8-bit channel as the IPP-backed wheels compute it (see
    above): exact bicubic (A = -0.75), replicated borders, round half to even."""
    val, (rows, cols, iy, ix, D) = _resize_cubic_value(src, dw, dh)
    out = np.rint(val)
    fl = np.floor(val)
    ys, xs = np.nonzero(np.abs(val - fl - 0.5) < 1e-9)          # decide these exactly: 2 ex <> (2 n + 1) KX KY
    if ys.size:
        si = src.astype(np.int64)
        for y, x in zip(ys.tolist(), xs.tolist()):
            ex = 0
            for j in range(4):
                row = si[rows[j][y]]
                ex += int(iy[y, j]) * sum(int(ix[x, i]) * int(row[cols[i][x]]) for i in range(4))
            n = int(fl[y, x])
            c = 2 * ex - (2 * n + 1) * D
            out[y, x] = n + 1 if c > 0 else (n if c < 0 else n + (n & 1))
    return np.clip(out, 0, 255).astype(np.uint8)


# OpenCV's OWN 8-bit cubic path (imgproc/src/resize.cpp HResizeCubic / VResizeCubic, 11-bit coefficients), which builds WITHOUT
# IPP (e.g. aarch64 wheels) run.  Kept as a named alternative: it is ~10 % of pixels off the reference's stored outputs.
def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """imgproc resize.cpp interpolateCubic, A = -0.75, evaluated in float32 like the C code."""
    A = np.float32(-0.75)
    x = x.astype(np.float32)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def _cubic_axis(dst: int, src: int):
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    co = np.clip(np.rint(_cubic_coeffs(f).astype(np.float64) * 2048.0).astype(np.int64), -32768, 32767)   # cvRound: half to even
    return s - 1, co


def resize_cubic_fixedpoint_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """Non-IPP cv2.resize INTER_CUBIC, 8-bit: int32 horizontal pass with 2^11-scaled coefficients, vertical pass + (v + 2^21) >> 22."""
    H, W = src.shape
    x0, cx = _cubic_axis(dw, W)
    y0, cy = _cubic_axis(dh, H)
    s = src.astype(np.int64)
    hor = sum(s[:, np.clip(x0 + k, 0, W - 1)] * cx[:, k][None, :] for k in range(4))
    out = sum(hor[np.clip(y0 + k, 0, H - 1), :] * cy[:, k][:, None] for k in range(4))
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def resize_scale_u8(src: np.ndarray, scale: float) -> np.ndarray:
    """ImagePreprocessor.resize (:131-138): new size = int(h * s), int(w * s)."""
    h, w = src.shape
    return resize_cubic_u8(src, int(w * scale), int(h * scale))


# ------------------------------------------------------------------------------------------------ cv2.GaussianBlur 3x3, 8u
def gaussian_kernel3_fixed(sigma: float):
    """getGaussianKernelBitExact + fixed-point conversion (8.8) with error diffusion so that the taps sum to 256."""
    k = [math.exp(-((i - 1) ** 2) / (2.0 * sigma * sigma)) for i in range(3)]
    tot = sum(k)
    k = [v / tot * 256.0 for v in k]
    out, err = [], 0.0
    for v in k:
        r = int(math.floor(v + err + 0.5))
        err += v - r
        out.append(r)
    return out


def gaussian_blur3_u8(src: np.ndarray, sigma: float) -> np.ndarray:
    """cv2.GaussianBlur(src, (3, 3), sigma) for 8-bit input: separable fixed-point smoothing (ufixedpoint16 rows, then
    (sum + 2^15) >> 16), BORDER_REFLECT_101."""
    k = gaussian_kernel3_fixed(sigma)
    p = np.pad(src.astype(np.int64), 1, mode="reflect")
    hor = k[0] * p[:, :-2] + k[1] * p[:, 1:-1] + k[2] * p[:, 2:]
    ver = k[0] * hor[:-2] + k[1] * hor[1:-1] + k[2] * hor[2:]
    return np.clip((ver + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ PIL ImageEnhance (mode L)
def _pil_blend_u8(in1, in2: np.ndarray, alpha: float) -> np.ndarray:
    """libImaging/Blend.c: interpolation truncates, extrapolation (alpha outside [0, 1]) clips then truncates.  The C code
    computes (int)in1 + alpha * ((int)in2 - (int)in1) in float."""
    a = np.float32(alpha)
    i1 = np.asarray(in1, dtype=np.int64)
    t = (i1.astype(np.float32) + a * (in2.astype(np.int64) - i1).astype(np.float32)).astype(np.float32)   # float arithmetic
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int64))).astype(np.uint8)


def pil_contrast_L(img: np.ndarray, factor: float) -> np.ndarray:
    """ImageEnhance.Contrast: blend(constant image of int(mean + 0.5), image, factor)."""
    mean = int(float(img.astype(np.float64).sum()) / img.size + 0.5)
    return _pil_blend_u8(mean, img, factor)


def pil_brightness_L(img: np.ndarray, factor: float) -> np.ndarray:
    """ImageEnhance.Brightness: blend(black image, image, factor)."""
    return _pil_blend_u8(0, img, factor)


# ------------------------------------------------------------------------------------------------ cv2 CLAHE, 8u
def clahe_u8(src: np.ndarray, clip_limit: float = 2.5, tiles=(8, 8)) -> np.ndarray:
    """imgproc clahe.cpp: reflect-101 padding to a multiple of the tile grid, per-tile clipped histogram with redistribution,
    LUT = cvRound(cumsum * 255 / tile_area) (float), bilinear blend of the four neighbouring tile LUTs in float."""
    tx, ty = tiles
    H, W = src.shape
    if H % ty or W % tx:
        ext = np.pad(src, ((0, ty - H % ty), (0, tx - W % tx)), mode="reflect")
    else:
        ext = src
    th, tw = ext.shape[0] // ty, ext.shape[1] // tx
    area = th * tw
    lut_scale = np.float32(255.0) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((ty, tx, 256), dtype=np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(ext[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    k = 0
                    while k < 256 and residual > 0:
                        hist[k] += 1
                        k += step
                        residual -= 1
            cs = np.cumsum(hist).astype(np.float32) * lut_scale
            luts[j, i] = np.clip(np.rint(cs.astype(np.float64)), 0, 255).astype(np.uint8)   # cvRound(float): half to even
    inv_th, inv_tw = np.float32(1.0) / np.float32(th), np.float32(1.0) / np.float32(tw)
    yy = np.arange(H, dtype=np.float32) * inv_th - np.float32(0.5)
    xx = np.arange(W, dtype=np.float32) * inv_tw - np.float32(0.5)
    y1 = np.floor(yy).astype(np.int64)
    x1 = np.floor(xx).astype(np.int64)
    ya = (yy - y1.astype(np.float32)).astype(np.float32)
    xa = (xx - x1.astype(np.float32)).astype(np.float32)
    y2, x2 = np.minimum(y1 + 1, ty - 1), np.minimum(x1 + 1, tx - 1)
    y1, x1 = np.maximum(y1, 0), np.maximum(x1, 0)
    v = src.astype(np.int64)
    l11 = luts[y1[:, None], x1[None, :], v].astype(np.float32)
    l12 = luts[y1[:, None], x2[None, :], v].astype(np.float32)
    l21 = luts[y2[:, None], x1[None, :], v].astype(np.float32)
    l22 = luts[y2[:, None], x2[None, :], v].astype(np.float32)
    xa_, ya_ = xa[None, :], ya[:, None]
    xa1, ya1 = (np.float32(1) - xa_).astype(np.float32), (np.float32(1) - ya_).astype(np.float32)
    res = ((l11 * xa1 + l12 * xa_).astype(np.float32) * ya1 + (l21 * xa1 + l22 * xa_).astype(np.float32) * ya_).astype(np.float32)
    return np.clip(np.rint(res.astype(np.float64)), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ PIL UnsharpMask (mode L)
def _pil_box_radius(radius: float, passes: int = 3) -> float:
    """libImaging/BoxBlur.c::_gaussian_blur_radius (float arithmetic)."""
    sigma2 = np.float32(radius * radius / passes)
    L = np.float32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = np.float32(math.floor((float(L) - 1.0) / 2.0))
    a = np.float32((2 * l + 1) * (l * (l + 1) - 3 * sigma2))
    a = np.float32(a / np.float32(6 * (sigma2 - (l + 1) * (l + 1))))
    return float(np.float32(l + a))


def _pil_box_blur_rows(img: np.ndarray, fradius: float) -> np.ndarray:
    """One horizontal pass of ImagingLineBoxBlur8 on every row: (sum of the 2r+1 window * ww + the two pixels just outside
    * fw + 2^23) >> 24 with edge replication; ww, fw as in ImagingHorizontalBoxBlur."""
    r = int(fradius)
    ww = int(np.float32(1 << 24) / (np.float32(fradius) * np.float32(2) + np.float32(1)))     # float division, truncated to UINT32
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    W = img.shape[1]
    s = img.astype(np.int64)
    idx = np.arange(W)
    acc = np.zeros_like(s)
    for d in range(-r, r + 1):
        acc += s[:, np.clip(idx + d, 0, W - 1)]
    far = s[:, np.clip(idx - r - 1, 0, W - 1)] + s[:, np.clip(idx + r + 1, 0, W - 1)]
    return (((acc * ww + far * fw) & 0xFFFFFFFF) + (1 << 23) >> 24).astype(np.uint8)


def pil_gaussian_blur_L(img: np.ndarray, radius: float) -> np.ndarray:
    """ImageFilter.GaussianBlur / ImagingGaussianBlur: three box-blur passes per axis (rows, then columns)."""
    fr = _pil_box_radius(radius, 3)
    out = img
    for _ in range(3):
        out = _pil_box_blur_rows(out, fr)
    out = np.ascontiguousarray(out.T)
    for _ in range(3):
        out = _pil_box_blur_rows(out, fr)
    return np.ascontiguousarray(out.T)


def pil_unsharp_L(img: np.ndarray, radius: float = 1.0, percent: int = 30, threshold: int = 3) -> np.ndarray:
    """libImaging/UnsharpMask.c, mode L: diff = in - blur; |diff| > threshold -> clip8(in + diff * percent / 100) (C integer
    division, truncating towards zero), else in."""
    blur = pil_gaussian_blur_L(img, radius).astype(np.int64)
    i = img.astype(np.int64)
    diff = i - blur
    prod = diff * percent
    q = np.where(prod >= 0, prod // 100, -((-prod) // 100))
    sharp = np.clip(i + q, 0, 255)
    return np.where(np.abs(diff) > threshold, sharp, i).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ the chain
def preprocess_chain(bgr: np.ndarray, scale=1.5, blur_sigma=3.0, contrast=1.9, brightness=1.2, clahe_clip=2.5, unsharp_percent=30) -> np.ndarray:
    """The ImagePreprocessor stage sequence both versions of preprocess_for_book_cover run (a stage whose parameter is 0 is skipped)."""
    g = bgr2gray(bgr)
    if scale:
        g = resize_scale_u8(g, scale)
    if blur_sigma:
        g = gaussian_blur3_u8(g, blur_sigma)
    if contrast:
        g = pil_contrast_L(g, contrast)
    if brightness:
        g = pil_brightness_L(g, brightness)
    if clahe_clip:
        g = clahe_u8(g, clahe_clip, (8, 8))
    if unsharp_percent:
        g = pil_unsharp_L(g, 1.0, unsharp_percent, 3)
    return g


def preprocess_for_book_cover(bgr: np.ndarray) -> np.ndarray:
    """pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py:147-160 on a decoded BGR page -> the 8-bit gray image the
    reference then saves / OCRs."""
    return preprocess_chain(bgr, 1.5, 3.0, 1.9, 1.2, 2.5, 30)


LEGACY = dict(scale=1.5, blur_sigma=5.0, contrast=1.3, brightness=0.0, clahe_clip=2.0, unsharp_percent=20)


def preprocess_for_book_cover_legacy(bgr: np.ndarray) -> np.ndarray:
    """pipeline_components/img_to_json/ocr_testing/preprocessing/image_preprocessor.py:221-252 (the version that produced the
    reference's stored results/images/*_preprocessed.png): sigma 5, contrast 1.3, no brightness step, CLAHE 2.0, unsharp 20 %."""
    return preprocess_chain(bgr, **LEGACY)


STEPS = ["original", "grayscale", "resize(scale_factor=1.5)", "denoise(strength=3)", "increase_contrast(factor=1.9)",
         "increase_brightness(factor=1.2)", "clahe(clip_limit=2.5)", "sharpen(amount=0.3)"]
