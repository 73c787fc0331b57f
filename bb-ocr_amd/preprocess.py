"""The reference's OCR pre-processing on the device.

``pipeline_demo/ocr_testing/preprocessing/image_preprocessor.py::preprocess_for_book_cover(image_path, output_path=None)``
(:147-160) returns ``(image, output_path, steps_applied)``; the extractor calls it two to three times per page
(``enhanced_extractor.py:431,634,775``) on the CPU.  ``preprocess_for_book_cover`` here has the same signature and return shape
and runs the seven stages as HIP kernels (csrc/preproc.hip) through ``bbocr_preprocess_book_cover``; ``..._device`` keeps the
result in HBM so it can go straight into ``Reader.readtext_device``.

Decoding is host work (PIL).  ``cv2.imread`` applies the EXIF orientation and returns BGR; ``_imread_bgr`` does the same with
Pillow (the JPEG decoders of OpenCV and Pillow builds may differ in the last bit of a pixel, which is outside this backend).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

LEGACY_STEPS = ["original", "grayscale", "resize(scale_factor=1.5)", "denoise(strength=5)", "increase_contrast(factor=1.3)",
                "clahe(clip_limit=2.0)", "sharpen(amount=0.2)"]
STEPS = ["original", "grayscale", "resize(scale_factor=1.5)", "denoise(strength=3)", "increase_contrast(factor=1.9)",
         "increase_brightness(factor=1.2)", "clahe(clip_limit=2.5)", "sharpen(amount=0.3)"]


def _imread_bgr(image_path):
    from PIL import Image, ImageOps

    pil = Image.open(image_path)
    pil = ImageOps.exif_transpose(pil)
    rgb = np.asarray(pil.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def preprocess_bgr_device(reader, bgr_dev, legacy=False, **overrides):
    """uint8 torch tensor [H,W,3] (BGR, on the reader's device) -> uint8 torch tensor [int(H*1.5), int(W*1.5)] (gray).
    ``legacy=True``: the parameters of pipeline_components/.../image_preprocessor.py:221-252; ``overrides``: fields of
    ``bbocr_preproc_params`` (a stage whose parameter is 0 is skipped)."""
    import torch

    from . import _lib

    if not isinstance(bgr_dev, torch.Tensor) or bgr_dev.ndim != 3:
        raise ValueError("expected a uint8 [H,W,3] device tensor")
    H, W, ch = bgr_dev.shape
    if (ch != 3 or bgr_dev.dtype != torch.uint8 or not bgr_dev.is_contiguous() or not bgr_dev.is_cuda
            or bgr_dev.device.index != reader.device_index):
        raise ValueError(f"expected a contiguous uint8 [H,W,3] tensor on {reader.device}")
    q = _lib.bbocr_preproc_params()
    reader._lib.bbocr_preproc_defaults(C.byref(q), int(bool(legacy)))
    for k, v in overrides.items():
        if not hasattr(q, k):
            raise ValueError(f"unknown pre-processing parameter {k!r}")
        setattr(q, k, v)
    oh, ow = C.c_int(), C.c_int()
    reader._check(reader._lib.bbocr_preprocess_chain(reader._h, C.c_void_p(bgr_dev.data_ptr()), H, W, C.byref(q), C.c_void_p(None), C.byref(oh),
                                                    C.byref(ow)))
    out = torch.empty((oh.value, ow.value), dtype=torch.uint8, device=bgr_dev.device)
    reader._check(reader._lib.bbocr_preprocess_chain(reader._h, C.c_void_p(bgr_dev.data_ptr()), H, W, C.byref(q), C.c_void_p(out.data_ptr()),
                                                    C.byref(oh), C.byref(ow)))
    return out


def preprocess_for_book_cover(image_path, output_path=None, reader=None, legacy=False):
    """Drop-in for the reference function: ``(gray uint8 array, output_path, steps_applied)``.  ``image_path`` may also be a
    decoded BGR array.  ``reader`` supplies the device context (any ``bb_ocr_amd.Reader``).  ``legacy=True`` runs the older
    ``preprocess_for_book_cover`` of pipeline_components/img_to_json/ocr_testing/preprocessing/image_preprocessor.py:221-252."""
    if reader is None:
        raise ValueError("preprocess_for_book_cover needs a bb_ocr_amd.Reader (device context)")
    if isinstance(image_path, np.ndarray):
        bgr = np.ascontiguousarray(image_path)
    else:
        if not os.path.exists(image_path):
            raise ValueError(f"Could not load image from {image_path}")      # image_preprocessor.py:19-20
        bgr = _imread_bgr(image_path)
    out = preprocess_bgr_device(reader, reader._to_dev(bgr), legacy=legacy).cpu().numpy()
    if output_path:
        from PIL import Image

        os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
        Image.fromarray(out).save(output_path)
    return out, output_path, list(LEGACY_STEPS if legacy else STEPS)
