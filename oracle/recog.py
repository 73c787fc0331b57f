"""Oracle (test infrastructure): box crops -> recogniser input -> greedy CTC text.

Restates ``easyocr/utils.py::{get_image_list,four_point_transform,calculate_ratio,
compute_ratio_and_resize,CTCLabelConverter.decode_greedy}`` and
``easyocr/recognition.py::{AlignCollate,NormalizePAD,contrast_grey,adjust_contrast_grey,
recognizer_predict,custom_mean,get_text}`` (easyocr==1.7.2, un-vendored) plus the OpenCV
4.10 ``getPerspectiveTransform`` / ``warpPerspective`` (bilinear, INTER_BITS=5 fixed point)
they call.  Reached from ``reader.readtext(path, paragraph=False, batch_size=1, workers=0)``
(``pipeline_demo/extractor/enhanced_extractor.py:520``): with batch_size=1 upstream
recognises ONE box per call, so every crop is padded to its own ``ceil(ratio)*64``.
PARITY UNPINNED.
"""
from __future__ import annotations

import math

import numpy as np

from . import imgproc

# english_g2 character set (easyocr/config.py): class 0 is the CTC blank.
_SYMBOLS = "0123456789!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~ €"
CHARSET = _SYMBOLS + "ABCDEFGHIJKLMNOPQRSTUVWXYZ" + "abcdefghijklmnopqrstuvwxyz"
CHARACTER = ["[blank]"] + list(CHARSET)
NUM_CLASS = len(CHARACTER)
assert NUM_CLASS == 97


def calculate_ratio(width, height):
    ratio = width / height
    if ratio < 1.0:
        ratio = 1.0 / ratio
    return ratio


def compute_ratio_and_resize(img, width, height, model_height):
    """utils.py::compute_ratio_and_resize (PIL LANCZOS==1 is passed to cv2 => INTER_LINEAR)."""
    ratio = width / height
    if ratio < 1.0:
        ratio = calculate_ratio(width, height)
        img = imgproc.resize_linear_u8(img, (model_height, int(model_height * ratio)))
    else:
        img = imgproc.resize_linear_u8(img, (int(model_height * ratio), model_height))
    return img, ratio


# ------------------------------------------------------------- perspective warp
def solve_linear(a, b):
    """Gaussian elimination with partial pivoting in float64 (cv::solve DECOMP_LU order)."""
    n = len(b)
    A = [[float(v) for v in row] + [float(bv)] for row, bv in zip(a, b)]
    for i in range(n):
        k = i
        for j in range(i + 1, n):
            if abs(A[j][i]) > abs(A[k][i]):
                k = j
        if abs(A[k][i]) < 2.220446049250313e-16 * 100:
            return None
        if k != i:
            A[i], A[k] = A[k], A[i]
        d = -1.0 / A[i][i]
        for j in range(i + 1, n):
            alpha = A[j][i] * d
            for c in range(i + 1, n + 1):
                A[j][c] += alpha * A[i][c]
    x = [0.0] * n
    for i in range(n - 1, -1, -1):
        s = A[i][n]
        for c in range(i + 1, n):
            s -= A[i][c] * x[c]
        x[i] = s / A[i][i]
    return x


def get_perspective_transform(src, dst):
    """cv2.getPerspectiveTransform(src[4,2] f32, dst[4,2] f32) -> 3x3 float64."""
    a = [[0.0] * 8 for _ in range(8)]
    b = [0.0] * 8
    for i in range(4):
        sx, sy = float(src[i][0]), float(src[i][1])
        dx, dy = float(dst[i][0]), float(dst[i][1])
        a[i][0] = a[i + 4][3] = sx
        a[i][1] = a[i + 4][4] = sy
        a[i][2] = a[i + 4][5] = 1.0
        a[i][6] = -sx * dx
        a[i][7] = -sy * dx
        a[i + 4][6] = -sx * dy
        a[i + 4][7] = -sy * dy
        b[i] = dx
        b[i + 4] = dy
    x = solve_linear(a, b)
    if x is None:
        x = [0.0] * 8
    return np.array(x + [1.0], dtype=np.float64).reshape(3, 3)


def invert3x3(m):
    """cv::invert for a 3x3 double matrix (closed form, as Matx/invert's 3x3 branch)."""
    m = np.asarray(m, dtype=np.float64)
    a = m
    d = (a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
         + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))
    out = np.zeros((3, 3), dtype=np.float64)
    if d != 0.0:
        d = 1.0 / d
        out[0, 0] = (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) * d
        out[0, 1] = (a[0, 2] * a[2, 1] - a[0, 1] * a[2, 2]) * d
        out[0, 2] = (a[0, 1] * a[1, 2] - a[0, 2] * a[1, 1]) * d
        out[1, 0] = (a[1, 2] * a[2, 0] - a[1, 0] * a[2, 2]) * d
        out[1, 1] = (a[0, 0] * a[2, 2] - a[0, 2] * a[2, 0]) * d
        out[1, 2] = (a[0, 2] * a[1, 0] - a[0, 0] * a[1, 2]) * d
        out[2, 0] = (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]) * d
        out[2, 1] = (a[0, 1] * a[2, 0] - a[0, 0] * a[2, 1]) * d
        out[2, 2] = (a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]) * d
    return out


_INTER_BITS = 5
_INTER_TAB_SIZE = 1 << _INTER_BITS
_REMAP_COEF_BITS = 15


def warp_perspective_u8(src, M, dsize_wh):
    """cv2.warpPerspective(src, M, (w, h)) : INTER_LINEAR, BORDER_CONSTANT(0), uint8 HW."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    Mi = invert3x3(M)
    sh, sw = src.shape
    xs = np.arange(dw, dtype=np.float64)[None, :]
    ys = np.arange(dh, dtype=np.float64)[:, None]
    X0 = Mi[0, 0] * xs + (Mi[0, 1] * ys + Mi[0, 2])
    Y0 = Mi[1, 0] * xs + (Mi[1, 1] * ys + Mi[1, 2])
    W0 = Mi[2, 0] * xs + (Mi[2, 1] * ys + Mi[2, 2])
    with np.errstate(divide="ignore", invalid="ignore"):
        W = np.where(W0 != 0, _INTER_TAB_SIZE / W0, 0.0)
    fX = np.clip(X0 * W, -2147483648.0, 2147483647.0)
    fY = np.clip(Y0 * W, -2147483648.0, 2147483647.0)
    X = np.rint(fX).astype(np.int64)
    Y = np.rint(fY).astype(np.int64)
    sx = np.clip(X >> _INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> _INTER_BITS, -32768, 32767)
    ax = (X & (_INTER_TAB_SIZE - 1)).astype(np.int64)
    ay = (Y & (_INTER_TAB_SIZE - 1)).astype(np.int64)
    # weights (1-fx)(1-fy) * 2^15 with fx = ax/32: exact integers (32-ax)*(32-ay)*32
    w00 = (32 - ax) * (32 - ay) * 32
    w01 = ax * (32 - ay) * 32
    w10 = (32 - ax) * ay * 32
    w11 = ax * ay * 32
    s = src.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        v = s[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)]
        return np.where(ok, v, 0)

    acc = tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11
    out = (acc + (1 << (_REMAP_COEF_BITS - 1))) >> _REMAP_COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8)


def four_point_transform(image, rect):
    """utils.py::four_point_transform."""
    rect = np.asarray(rect, dtype=np.float32)
    (tl, tr, br, bl) = rect
    widthA = np.sqrt(((br[0] - bl[0]) ** 2) + ((br[1] - bl[1]) ** 2))
    widthB = np.sqrt(((tr[0] - tl[0]) ** 2) + ((tr[1] - tl[1]) ** 2))
    maxWidth = max(int(widthA), int(widthB))
    heightA = np.sqrt(((tr[0] - br[0]) ** 2) + ((tr[1] - br[1]) ** 2))
    heightB = np.sqrt(((tl[0] - bl[0]) ** 2) + ((tl[1] - bl[1]) ** 2))
    maxHeight = max(int(heightA), int(heightB))
    dst = np.array([[0, 0], [maxWidth - 1, 0], [maxWidth - 1, maxHeight - 1], [0, maxHeight - 1]], dtype="float32")
    M = get_perspective_transform(rect, dst)
    return warp_perspective_u8(image, M, (maxWidth, maxHeight))


def get_image_list(horizontal_list, free_list, img, model_height=64, sort_output=True):
    """utils.py::get_image_list -> ([(box, crop uint8 [64-ish, w])], max_width)."""
    image_list = []
    maximum_y, maximum_x = img.shape
    max_ratio_hori, max_ratio_free = 1, 1
    for box in free_list:
        rect = np.array(box, dtype="float32")
        transformed_img = four_point_transform(img, rect)
        if transformed_img.shape[0] == 0 or transformed_img.shape[1] == 0:
            continue
        ratio = calculate_ratio(transformed_img.shape[1], transformed_img.shape[0])
        new_width = int(model_height * ratio)
        if new_width == 0:
            pass
        else:
            crop_img, ratio = compute_ratio_and_resize(transformed_img, transformed_img.shape[1], transformed_img.shape[0], model_height)
            image_list.append((box, crop_img))
            max_ratio_free = max(ratio, max_ratio_free)
    max_ratio_free = math.ceil(max_ratio_free)
    for box in horizontal_list:
        x_min = max(0, box[0])
        x_max = min(box[1], maximum_x)
        y_min = max(0, box[2])
        y_max = min(box[3], maximum_y)
        crop_img = img[y_min:y_max, x_min:x_max]
        width = x_max - x_min
        height = y_max - y_min
        if width <= 0 or height <= 0:
            # upstream would raise ZeroDivisionError / cv2.error here; the product returns an error status
            raise ValueError("empty horizontal box after clamping")
        ratio = calculate_ratio(width, height)
        new_width = int(model_height * ratio)
        if new_width == 0:
            pass
        else:
            crop_img, ratio = compute_ratio_and_resize(crop_img, width, height, model_height)
            image_list.append(([[x_min, y_min], [x_max, y_min], [x_max, y_max], [x_min, y_max]], crop_img))
            max_ratio_hori = max(ratio, max_ratio_hori)
    max_ratio_hori = math.ceil(max_ratio_hori)
    max_ratio = max(max_ratio_hori, max_ratio_free)
    max_width = math.ceil(max_ratio) * model_height
    if sort_output:
        image_list = sorted(image_list, key=lambda item: item[0][0][1])
    return image_list, max_width


# ------------------------------------------------------------ AlignCollate etc.
def contrast_grey(img):
    high = np.percentile(img, 90)
    low = np.percentile(img, 10)
    return (high - low) / np.maximum(10, high + low), high, low


def adjust_contrast_grey(img, target=0.4):
    contrast, high, low = contrast_grey(img)
    if contrast < target:
        img = img.astype(int)
        ratio = 200.0 / np.maximum(10, high - low)
        img = (img - low + 25) * ratio
        img = np.maximum(np.full(img.shape, 0), np.minimum(np.full(img.shape, 255), img)).astype(np.uint8)
    return img


def align_collate_one(crop: np.ndarray, imgH: int, imgW: int, adjust_contrast: float = 0.0) -> np.ndarray:
    """AlignCollate.__call__ for one image -> float32 [1, imgH, imgW] in [-1, 1]."""
    image = crop
    if adjust_contrast > 0:
        image = adjust_contrast_grey(image, target=adjust_contrast)
    h, w = image.shape
    ratio = w / float(h)
    if math.ceil(imgH * ratio) > imgW:
        resized_w = imgW
    else:
        resized_w = math.ceil(imgH * ratio)
    resized = imgproc.pil_resize_bicubic_u8(image, (resized_w, imgH))
    t = resized.astype(np.float32) / np.float32(255.0)          # ToTensor
    t = (t - np.float32(0.5)) / np.float32(0.5)
    out = np.zeros((1, imgH, imgW), dtype=np.float32)
    out[0, :, :resized_w] = t
    if imgW != resized_w:
        out[0, :, resized_w:] = t[:, resized_w - 1:resized_w]
    return out


def custom_mean(x):
    return x.prod() ** (2.0 / np.sqrt(len(x)))


def decode_greedy(text_index: np.ndarray, length) -> list:
    """CTCLabelConverter.decode_greedy (ignore_idx = [0])."""
    texts = []
    index = 0
    chars = np.array(CHARACTER)
    for l in length:
        t = text_index[index:index + l]
        a = np.insert(~((t[1:] == t[:-1])), 0, True)
        b = ~np.isin(t, np.array([0]))
        c = a & b
        texts.append("".join(chars[t[c.nonzero()]]))
        index += l
    return texts


class _BeamEntry:
    """easyocr/utils.py::BeamEntry (no language model: prText stays 1)."""

    def __init__(self):
        self.prTotal = 0
        self.prNonBlank = 0
        self.prBlank = 0
        self.prText = 1
        self.labeling = ()


def ctc_beam_search(mat: np.ndarray, beam_width: int = 5, ignore_idx=(0,)) -> list:
    """easyocr/utils.py::ctcBeamSearch(mat, classes, ignore_idx, lm=None, beamWidth) -> list of class indices of the best labelling.

    Restated with upstream's quirks kept: the candidate set of a step is every class with probability >= 0.5/maxC INCLUDING the blank
    (class 0), so a labelling may carry explicit blanks; beams are ranked by prTotal * prText with Python's stable sort over dict
    insertion order; the final string drops ``ignore_idx`` classes and any symbol equal to its predecessor IN THE LABELLING.  mat is
    float32 and every product/sum stays float32 (python int/float operands are weak scalars in numpy arithmetic).
    """
    mat = np.asarray(mat, dtype=np.float32)
    maxT, maxC = mat.shape
    last = {(): _BeamEntry()}
    last[()].prBlank = 1
    last[()].prTotal = 1

    def ranked(state):
        beams = sorted(state.values(), reverse=True, key=lambda x: x.prTotal * x.prText)
        return [b.labeling for b in beams]

    for t in range(maxT):
        curr = {}
        for labeling in ranked(last)[:beam_width]:
            prNonBlank = 0
            if labeling:
                prNonBlank = last[labeling].prNonBlank * mat[t, labeling[-1]]
            prBlank = last[labeling].prTotal * mat[t, 0]
            e = curr.setdefault(labeling, _BeamEntry())
            e.labeling = labeling
            e.prNonBlank += prNonBlank
            e.prBlank += prBlank
            e.prTotal += prBlank + prNonBlank
            e.prText = last[labeling].prText
            for c in np.where(mat[t, :] >= 0.5 / maxC)[0]:
                c = int(c)
                new = labeling + (c,)
                if labeling and labeling[-1] == c:
                    prNonBlank = mat[t, c] * last[labeling].prBlank
                else:
                    prNonBlank = mat[t, c] * last[labeling].prTotal
                e2 = curr.setdefault(new, _BeamEntry())
                e2.labeling = new
                e2.prNonBlank += prNonBlank
                e2.prTotal += prNonBlank
        last = curr
    best = ranked(last)[0]
    return [l for i, l in enumerate(best) if l not in ignore_idx and not (i > 0 and best[i - 1] == best[i])]


def decode_beamsearch(mat: np.ndarray, beam_width: int = 5) -> list:
    """CTCLabelConverter.decode_beamsearch: one ctcBeamSearch per sequence of mat [b, T, C] (converter ignore_idx = [0])."""
    return ["".join(CHARACTER[i] for i in ctc_beam_search(m, beam_width)) for m in mat]


def softmax_f32(logits: np.ndarray) -> np.ndarray:
    """F.softmax(preds, dim=2) in float32."""
    import torch

    return torch.softmax(torch.from_numpy(np.ascontiguousarray(logits, dtype=np.float32)), dim=-1).numpy()


def predict_from_logits(logits: np.ndarray, ignore_idx=(), decoder="greedy", beam_width=5):
    """recognizer_predict tail for a batch of logits [b, T, C] -> [[text, conf]] (decoder 'greedy' or 'beamsearch'; the confidence is
    the greedy path's custom_mean for both, as upstream computes it)."""
    preds_prob = softmax_f32(logits)
    if len(ignore_idx):
        preds_prob[:, :, list(ignore_idx)] = 0.0
    pred_norm = preds_prob.sum(axis=2)
    preds_prob = preds_prob / np.expand_dims(pred_norm, axis=-1)
    preds_prob = preds_prob.astype(np.float32)
    b, T, _ = preds_prob.shape
    preds_index = preds_prob.argmax(axis=2).reshape(-1)
    preds_str = decode_greedy(preds_index, [T] * b) if decoder == "greedy" else decode_beamsearch(preds_prob, beam_width)
    values = preds_prob.max(axis=2)
    indices = preds_prob.argmax(axis=2)
    result = []
    for pred, v, i in zip(preds_str, values, indices):
        max_probs = v[i != 0]
        if len(max_probs) == 0:
            max_probs = np.array([0])
        result.append([pred, custom_mean(max_probs)])
    return result


def make_rotated_img_list(rotation_info, img_list):
    """utils.py::make_rotated_img_list: the list extended by scipy.ndimage.rotate(crop, angle, reshape=True) per angle; for the eligible
    angles (90, 180, 270) scipy returns exactly np.rot90(crop, angle // 90) (checked against scipy 1.x in this container)."""
    result_img_list = img_list[:]
    for angle in rotation_info:
        if angle not in (90, 180, 270):
            raise ValueError("rotation_info angles must be 90, 180 or 270")
        for img_info in img_list:
            result_img_list.append((img_info[0], np.ascontiguousarray(np.rot90(img_info[1], angle // 90))))
    return result_img_list


def set_result_with_confidence(results):
    """utils.py::set_result_with_confidence: per box the augmentation (row) with the highest confidence; the first maximum wins."""
    final_result = []
    for col_ix in range(len(results[0])):
        best_row = max([(row_ix, results[row_ix][col_ix][2]) for row_ix in range(len(results))], key=lambda x: x[1])[0]
        final_result.append(results[best_row][col_ix])
    return final_result


def get_text(recognizer_fn, imgH, imgW, image_list, contrast_ths=0.1, adjust_contrast=0.5):
    """recognition.py::get_text with batch_size=1; ``recognizer_fn(x [1,1,H,W] f32) -> logits [1,T,C]``."""
    coord = [item[0] for item in image_list]
    img_list = [item[1] for item in image_list]
    result1 = []
    for img in img_list:
        x = align_collate_one(img, imgH, imgW)[None]
        result1 += predict_from_logits(recognizer_fn(x))
    low_confident_idx = [i for i, item in enumerate(result1) if item[1] < contrast_ths]
    result2 = []
    for i in low_confident_idx:
        x = align_collate_one(img_list[i], imgH, imgW, adjust_contrast=adjust_contrast)[None]
        result2 += predict_from_logits(recognizer_fn(x))
    result = []
    for i, (box, pred1) in enumerate(zip(coord, result1)):
        if i in low_confident_idx:
            pred2 = result2[low_confident_idx.index(i)]
            if pred1[1] > pred2[1]:
                result.append((box, pred1[0], pred1[1]))
            else:
                result.append((box, pred2[0], pred2[1]))
        else:
            result.append((box, pred1[0], pred1[1]))
    return result
