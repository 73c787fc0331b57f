import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def states():
    from bb_ocr_amd import weights

    return weights.designed_craft_state(0), weights.synthetic_crnn_state(0)


@pytest.fixture(scope="session")
def oracle_reader(states):
    import torch

    from oracle import pipeline

    cs, rs = states
    return pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})


@pytest.fixture(scope="session")
def reader(states):
    """The HIP backend.  Fails (not skips) if the extension is missing on a GPU box."""
    import bb_ocr_amd

    r = bb_ocr_amd.Reader(["en"], gpu=True, weights=states)
    yield r
    r.close()
