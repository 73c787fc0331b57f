"""bb_ocr_amd -- MI355X-native OCR backend behind BB-OCR's ``easyocr.Reader.readtext`` call site.

The directory is named ``bb-ocr_amd``; import it as ``bb_ocr_amd`` (the repo-root module
``bb_ocr_amd.py`` maps the name).  Scope: SURVEY.md section 8 -- the detector + recogniser hot
path only.  ``csrc/`` holds the HIP kernels and the C ABI (``include/bbocr.h``); this
package is the thin Python host that mirrors the slice of the easyocr interface the
reference uses.
"""
from .reader import CHARACTER, CHARSET, Reader, reformat_input  # noqa: F401
from .install import install, uninstall  # noqa: F401

__all__ = ["Reader", "install", "uninstall", "reformat_input", "CHARSET", "CHARACTER", "freeze_gc"]


def freeze_gc():
    """Call once after start-up (models loaded, first page read) in a long-running host process -- the reference's batch processor and UI
    worker are such processes.  A 64-page ``readtext`` result is ~25 000 small Python objects (boxes, strings); allocating them makes
    CPython's cyclic collector run, and its periodic full (generation-2) pass walks every object ``import torch`` created: measured
    46 ms every ~8 steps on the MI355X host = 8 % of the step.  ``gc.freeze()`` moves everything alive NOW into the permanent generation,
    so later full passes only scan what was allocated since (well under 1 ms).  Nothing is disabled and no work is skipped."""
    import gc

    gc.collect()
    gc.freeze()

