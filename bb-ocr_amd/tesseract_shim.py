"""``pytesseract.image_to_string`` stand-in.

The reference's alternative engine (``enhanced_extractor.py:523-526``, ``compare_ocr_engines.py:110``) calls
``pytesseract.image_to_string(PIL.Image) -> str`` and uses the text with its newline line breaks.  This module offers the same
call on top of the MI355X backend: word boxes from ``Reader.readtext`` are put into reading order -- rows by vertical overlap
of their boxes, left to right inside a row -- words of a row joined by one space, rows by ``"\\n"``, text ending in ``"\\n"``
like Tesseract's plain-text renderer.  It is a different recogniser, so the characters are this backend's, not Tesseract's; what
is mirrored is the call shape the scripts depend on.  ``install()`` registers it as ``sys.modules["pytesseract"]``.
"""
from __future__ import annotations

import sys
import threading
import types

import numpy as np

_lock = threading.Lock()
_reader = None


class TesseractNotFoundError(EnvironmentError):
    """Name kept for scripts that catch ``pytesseract.TesseractNotFoundError`` (never raised here)."""


def _get_reader():
    global _reader
    with _lock:
        if _reader is None:
            from .reader import Reader

            _reader = Reader(["en"], gpu=True)
        return _reader


def set_reader(reader):
    """Use an existing ``Reader`` (its weights) instead of constructing one on first use."""
    global _reader
    with _lock:
        _reader = reader


def lines_from_results(results, overlap=0.5):
    """``[(bbox, text, conf)]`` -> list of rows (each a list of result items) in reading order.  A box joins the current row
    when at least ``overlap`` of the smaller height overlaps the row's running vertical extent."""
    items = []
    for bbox, text, conf in results:
        ys = [float(p[1]) for p in bbox]
        xs = [float(p[0]) for p in bbox]
        items.append((min(ys), max(ys), min(xs), (bbox, text, conf)))
    items.sort(key=lambda t: (0.5 * (t[0] + t[1]), t[2]))
    rows = []
    for y0, y1, x0, it in items:
        placed = False
        if rows:
            r = rows[-1]
            inter = min(y1, r["y1"]) - max(y0, r["y0"])
            if inter >= overlap * max(1e-6, min(y1 - y0, r["y1"] - r["y0"])):
                r["items"].append((x0, it))
                r["y0"], r["y1"] = min(r["y0"], y0), max(r["y1"], y1)
                placed = True
        if not placed:
            rows.append({"y0": y0, "y1": y1, "items": [(x0, it)]})
    return [[it for _, it in sorted(r["items"], key=lambda t: t[0])] for r in rows]


def image_to_string(image, lang=None, config="", nice=0, output_type="string", timeout=0, reader=None):
    """PIL image / numpy array / path -> text with ``\\n`` line breaks (the call shape of ``pytesseract.image_to_string``)."""
    rd = reader or _get_reader()
    if hasattr(image, "convert"):
        arr = np.asarray(image.convert("RGB"))
        results = rd.readtext(np.ascontiguousarray(arr))
    else:
        results = rd.readtext(image)
    rows = lines_from_results(results)
    text = "\n".join(" ".join(it[1] for it in row) for row in rows)
    text = text + "\n" if text else ""
    if output_type in ("string", "STRING"):
        return text
    if output_type in ("dict", "DICT"):
        return {"text": text}
    if output_type in ("bytes", "BYTES"):
        return text.encode("utf-8")
    raise ValueError(f"unsupported output_type {output_type!r}")


_PREV = None


def install(reader=None, force=True):
    """Make ``import pytesseract`` resolve to this shim."""
    global _PREV
    if reader is not None:
        set_reader(reader)
    if "pytesseract" in sys.modules and not force:
        return sys.modules["pytesseract"]
    _PREV = sys.modules.get("pytesseract")
    m = types.ModuleType("pytesseract")
    m.__doc__ = "bb_ocr_amd stand-in for pytesseract (MI355X backend)"
    m.image_to_string = image_to_string
    m.TesseractNotFoundError = TesseractNotFoundError
    m.Output = types.SimpleNamespace(STRING="string", DICT="dict", BYTES="bytes")
    m.pytesseract = types.SimpleNamespace(tesseract_cmd="bb_ocr_amd")
    m.get_tesseract_version = lambda: "bb_ocr_amd"
    m.__bbocr_backend__ = "mi355x"
    sys.modules["pytesseract"] = m
    return m


def uninstall():
    global _PREV
    cur = sys.modules.get("pytesseract")
    if cur is not None and getattr(cur, "__bbocr_backend__", None) == "mi355x":
        if _PREV is not None:
            sys.modules["pytesseract"] = _PREV
        else:
            del sys.modules["pytesseract"]
    _PREV = None
