"""Register this backend under the module name ``easyocr``.

``pipeline_demo/extractor/enhanced_extractor.py`` does ``import easyocr`` (:19), annotates with
``easyocr.Reader`` (:98), constructs ``easyocr.Reader(["en"], gpu=use_gpu)`` (:153) and calls
``readtext`` (:520).  After ``bb_ocr_amd.install()`` (or with ``BB_OCR_BACKEND=mi355x`` set when
``bb_ocr_amd`` is imported) those lines run unchanged on the MI355X backend.
"""
from __future__ import annotations

import os
import sys
import types

_PREV = None


def install(force: bool = True):
    """Make ``import easyocr`` resolve to a module exposing this backend's ``Reader``."""
    global _PREV
    from . import reader

    if "easyocr" in sys.modules and not force:
        return sys.modules["easyocr"]
    _PREV = sys.modules.get("easyocr")
    m = types.ModuleType("easyocr")
    m.__doc__ = "bb_ocr_amd stand-in for easyocr (MI355X backend)"
    m.Reader = reader.Reader
    m.__version__ = "1.7.2+bb_ocr_amd"
    m.__bbocr_backend__ = "mi355x"
    sys.modules["easyocr"] = m
    return m


def uninstall():
    global _PREV
    cur = sys.modules.get("easyocr")
    if cur is not None and getattr(cur, "__bbocr_backend__", None) == "mi355x":
        if _PREV is not None:
            sys.modules["easyocr"] = _PREV
        else:
            del sys.modules["easyocr"]
    _PREV = None


if os.environ.get("BB_OCR_BACKEND", "").strip().lower() == "mi355x":
    install()
