"""CPU oracle for the BB-OCR hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch fp32 / numpy / PIL) of the algorithm
behind ``easyocr.Reader.readtext()`` as it is called from the reference at
``pipeline_demo/extractor/enhanced_extractor.py:520``.  The arithmetic of that
call lives in the un-vendored third-party dependency ``easyocr==1.7.2``
(``pipeline_demo/requirements.txt:7``), with ``opencv-python==4.10.0.84``
(``:5``) and ``Pillow==10.4.0`` (``:4``) underneath.  None of those sources nor
the weight files exist under ``/root/reference`` or in this image, so every
function here restates the *published* upstream algorithm and cites the
upstream module it follows.

PARITY: partly pinned.  The reference holds two kinds of known-answer data for this path:
  * five (input image -> pre-processed image) pairs of its legacy ``preprocess_for_book_cover``
    (``pipeline_components/books/dataset/book*.png`` -> ``.../ocr_testing/results/images/book*_preprocessed.png``), committed under
    ``tests/golden`` -- they PIN ``oracle/preprocess.py`` (all seven stages) and the BGR2GRAY formula that ``oracle/imgproc.py``
    shares with it: one pair bit-exact, <= 64 of 1.4 M pixels off on the others, all of them next to a float32 near-tie of Intel IPP's
    cubic resize (``tests/test_oracle_cpu.py::test_legacy_preprocess_fixtures``);
  * 7 (image -> joined text) pairs that need the real ``craft_mlt_25k.pth`` / ``english_g2.pth`` weights, which
    are not available offline: the detector / recogniser networks, the box geometry and the text stay PARITY UNPINNED;
    ``tests/test_golden_replay.py`` replays the 7 pairs when ``BBOCR_WEIGHTS_DIR`` is set.
Everything else is pinned against itself (seeded golden vectors under ``tests/golden``) and, where one exists, against an independent
implementation (Pillow, scipy).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``bb_ocr_amd``) never does.
"""
