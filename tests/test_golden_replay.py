"""Opt-in replay of the reference's own known-answer data for this path: the (image -> joined EasyOCR text) pairs of
pipeline_components/img_to_json/ocr_testing/results/json/ocr_comparison_*.json.  Needs the real
upstream checkpoints (craft_mlt_25k.pth, english_g2.pth) in $BBOCR_WEIGHTS_DIR; without them parity against EasyOCR is
UNPINNED and these tests skip."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = json.load(open(os.path.join(HERE, "golden", "reference_pairs.json")))
WDIR = os.environ.get("BBOCR_WEIGHTS_DIR", "")
needs_weights = pytest.mark.skipif(not (WDIR and os.path.exists(os.path.join(WDIR, "craft_mlt_25k.pth"))), reason="real EasyOCR weights not available offline")


def _image_path(pair):
    if pair["committed_copy"]:
        return os.path.join(HERE, "golden", pair["committed_copy"])
    p = os.path.join("/root/reference", pair["image"])
    return p if os.path.exists(p) else None


def test_pairs_file_is_consistent():
    assert len(PAIRS) == 7
    for p in PAIRS:
        assert len(p["easyocr_text"]) == p["text_length"]
        if p["committed_copy"]:
            assert os.path.exists(os.path.join(HERE, "golden", p["committed_copy"]))


@needs_weights
@pytest.mark.parametrize("pair", PAIRS, ids=[p["source"].split("_")[-1] for p in PAIRS])
def test_oracle_reproduces_reference_text(pair):
    import torch

    from bb_ocr_amd import weights
    from oracle import pipeline

    path = _image_path(pair)
    if path is None:
        pytest.skip("input image not present on this machine")
    cs, rs = weights.load_checkpoint_dir(WDIR)
    ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
    text = " ".join(r[1] for r in ref.readtext(path))
    assert text == pair["easyocr_text"]


@needs_weights
@pytest.mark.gpu
@pytest.mark.parametrize("pair", PAIRS, ids=[p["source"].split("_")[-1] for p in PAIRS])
def test_hip_backend_reproduces_reference_text(pair):
    import bb_ocr_amd

    path = _image_path(pair)
    if path is None:
        pytest.skip("input image not present on this machine")
    reader = bb_ocr_amd.Reader(["en"], model_storage_directory=WDIR)
    text = " ".join(r[1] for r in reader.readtext(path, paragraph=False, batch_size=1, workers=0))
    assert text == pair["easyocr_text"]
