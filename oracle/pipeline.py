"""Oracle (test infrastructure): the whole ``readtext`` path on CPU, batch 1 per page and per box.

Restates ``easyocr/easyocr.py::Reader.{detect,recognize,readtext}`` and
``easyocr/detection.py::{test_net,get_textbox}`` for the call the reference makes:
``reader.readtext(path, paragraph=False, batch_size=1, workers=0)``
(``pipeline_demo/extractor/enhanced_extractor.py:520``).  Also the ``cpu_baseline`` leg
of bench.py (kind "port").  PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np
import torch

from . import boxes as obox
from . import imgproc, nets, recog


class OracleReader:
    """CPU mirror of ``easyocr.Reader(['en'])`` built from explicit state-dicts (no downloads)."""

    def __init__(self, craft_state: dict, crnn_state: dict, num_threads: int | None = None):
        if num_threads:
            torch.set_num_threads(num_threads)
        self.detector = nets.load_state_dict_any(nets.CRAFT(), craft_state).eval()
        self.recognizer = nets.load_state_dict_any(nets.CRNN(), crnn_state).eval()
        self.imgH = 64

    # -- detector ----------------------------------------------------------
    @torch.no_grad()
    def heatmap(self, img_rgb: np.ndarray, canvas_size=2560, mag_ratio=1.0):
        x, ratio = imgproc.detector_input(img_rgb, canvas_size, mag_ratio)
        y, _ = self.detector(torch.from_numpy(x)[None])
        y = y[0].numpy()
        return np.ascontiguousarray(y[:, :, 0]), np.ascontiguousarray(y[:, :, 1]), ratio

    def detect(self, img_rgb, min_size=20, text_threshold=0.7, low_text=0.4, link_threshold=0.4, canvas_size=2560,
               mag_ratio=1.0, slope_ths=0.1, ycenter_ths=0.5, height_ths=0.5, width_ths=0.5, add_margin=0.1):
        st, sl, ratio = self.heatmap(img_rgb, canvas_size, mag_ratio)
        h, f, _ = obox.detect_from_heatmap(st, sl, ratio, min_size=min_size, text_threshold=text_threshold,
                                           low_text=low_text, link_threshold=link_threshold, slope_ths=slope_ths,
                                           ycenter_ths=ycenter_ths, height_ths=height_ths, width_ths=width_ths,
                                           add_margin=add_margin)
        return h, f

    # -- recogniser --------------------------------------------------------
    @torch.no_grad()
    def _logits(self, x: np.ndarray) -> np.ndarray:
        return self.recognizer(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).numpy()

    def recognize(self, img_cv_grey, horizontal_list, free_list, contrast_ths=0.1, adjust_contrast=0.5, rotation_info=None):
        """Reader.recognize: per-box branch (batch_size == 1), or -- with rotation_info -- the batched branch it then takes."""
        if rotation_info:
            image_list, max_width = recog.get_image_list(horizontal_list, free_list, img_cv_grey, model_height=self.imgH)
            image_len = len(image_list)
            if image_list:
                image_list = recog.make_rotated_img_list(rotation_info, image_list)
                max_width = max(max_width, self.imgH)
            result = recog.get_text(self._logits, self.imgH, int(max_width), image_list, contrast_ths, adjust_contrast)
            if horizontal_list + free_list:
                result = recog.set_result_with_confidence([result[image_len * i:image_len * (i + 1)] for i in range(len(rotation_info) + 1)])
            return result
        result = []
        for bbox in horizontal_list:
            image_list, max_width = recog.get_image_list([bbox], [], img_cv_grey, model_height=self.imgH)
            result += recog.get_text(self._logits, self.imgH, int(max_width), image_list, contrast_ths, adjust_contrast)
        for bbox in free_list:
            image_list, max_width = recog.get_image_list([], [bbox], img_cv_grey, model_height=self.imgH)
            result += recog.get_text(self._logits, self.imgH, int(max_width), image_list, contrast_ths, adjust_contrast)
        return result

    def readtext(self, image, **kw):
        det_keys = ("min_size", "text_threshold", "low_text", "link_threshold", "canvas_size", "mag_ratio",
                    "slope_ths", "ycenter_ths", "height_ths", "width_ths", "add_margin")
        img, img_cv_grey = imgproc.reformat_input(image)
        h, f = self.detect(img, **{k: kw[k] for k in det_keys if k in kw})
        return self.recognize(img_cv_grey, h, f, kw.get("contrast_ths", 0.1), kw.get("adjust_contrast", 0.5), kw.get("rotation_info"))
