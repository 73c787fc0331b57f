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

STEPS = ["original", "grayscale", "resize(scale_factor=1.5)", "denoise(strength=3)", "increase_contrast(factor=1.9)",
         "increase_brightness(factor=1.2)", "clahe(clip_limit=2.5)", "sharpen(amount=0.3)"]


def _imread_bgr(image_path):
    from PIL import Image, ImageOps

    pil = Image.open(image_path)
    pil = ImageOps.exif_transpose(pil)
    rgb = np.asarray(pil.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def preprocess_bgr_device(reader, bgr_dev):
    """uint8 torch tensor [H,W,3] (BGR, on the reader's device) -> uint8 torch tensor [int(H*1.5), int(W*1.5)] (gray)."""
    import torch

    H, W, ch = bgr_dev.shape
    if ch != 3 or bgr_dev.dtype != torch.uint8 or not bgr_dev.is_contiguous():
        raise ValueError("expected a contiguous uint8 [H,W,3] tensor")
    oh, ow = C.c_int(), C.c_int()
    reader._check(reader._lib.bbocr_preprocess_book_cover(reader._h, C.c_void_p(bgr_dev.data_ptr()), H, W, C.c_void_p(None), C.byref(oh), C.byref(ow)))
    out = torch.empty((oh.value, ow.value), dtype=torch.uint8, device=bgr_dev.device)
    reader._check(reader._lib.bbocr_preprocess_book_cover(reader._h, C.c_void_p(bgr_dev.data_ptr()), H, W, C.c_void_p(out.data_ptr()), C.byref(oh),
                                                         C.byref(ow)))
    return out


def preprocess_for_book_cover(image_path, output_path=None, reader=None):
    """Drop-in for the reference function: ``(gray uint8 array, output_path, steps_applied)``.  ``image_path`` may also be a
    decoded BGR array.  ``reader`` supplies the device context (any ``bb_ocr_amd.Reader``)."""
    if reader is None:
        raise ValueError("preprocess_for_book_cover needs a bb_ocr_amd.Reader (device context)")
    if isinstance(image_path, np.ndarray):
        bgr = np.ascontiguousarray(image_path)
    else:
        if not os.path.exists(image_path):
            raise ValueError(f"Could not load image from {image_path}")      # image_preprocessor.py:19-20
        bgr = _imread_bgr(image_path)
    out = preprocess_bgr_device(reader, reader._to_dev(bgr)).cpu().numpy()
    if output_path:
        from PIL import Image

        os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
        Image.fromarray(out).save(output_path)
    return out, output_path, list(STEPS)
