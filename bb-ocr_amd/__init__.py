"""bb_ocr_amd -- MI355X-native OCR backend behind BB-OCR's ``easyocr.Reader.readtext`` call site.

The directory is named ``bb-ocr_amd``; import it as ``bb_ocr_amd`` (the repo-root module
``bb_ocr_amd.py`` maps the name).  Scope: SURVEY.md section 8 -- the detector + recogniser hot
path only.  ``csrc/`` holds the HIP kernels and the C ABI (``include/bbocr.h``); this
package is the thin Python host that mirrors the slice of the easyocr interface the
reference uses.
"""
from .reader import CHARACTER, CHARSET, Reader, reformat_input  # noqa: F401
from .install import install, uninstall  # noqa: F401

__all__ = ["Reader", "install", "uninstall", "reformat_input", "CHARSET", "CHARACTER"]
