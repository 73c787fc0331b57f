"""CPU: property tests (hypothesis) of the host-side C++ stages through the C ABI against the oracle -- SURVEY.md section 4 asks for
them for box grouping and CTC decoding.  Random, ragged and degenerate inputs: empty lists, single boxes, duplicates, boxes on one
line / in one column, slanted quads, zero-area quads, coordinates at the int32 scale of a 2560 canvas."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def lib():
    from bb_ocr_amd import _lib

    return _lib.load()


def _quad(draw):
    """An int polygon as detection.get_textbox emits it: axis-aligned word box, slanted quad, or a degenerate sliver."""
    kind = draw(st.sampled_from(["word", "word", "word", "slant", "sliver"]))
    x = draw(st.integers(-5, 2500))
    y = draw(st.integers(-5, 2500))
    w = draw(st.integers(1, 400))
    h = draw(st.integers(1, 80))
    if kind == "word":
        return [x, y, x + w, y, x + w, y + h, x, y + h]
    if kind == "slant":
        d = draw(st.integers(-60, 60))
        return [x, y, x + w, y + d, x + w, y + d + h, x, y + h]
    return [x, y, x + w, y, x + w, y, x, y]


@st.composite
def _polys(draw):
    n = draw(st.integers(0, 40))
    polys = [_quad(draw) for _ in range(n)]
    if polys and draw(st.booleans()):                      # words of one text line: similar y, increasing x (the merge path)
        y0 = draw(st.integers(0, 2000))
        x = draw(st.integers(0, 200))
        for _ in range(draw(st.integers(2, 12))):
            w, h, gap, dy = draw(st.integers(10, 120)), draw(st.integers(18, 40)), draw(st.integers(0, 60)), draw(st.integers(-6, 6))
            polys.append([x, y0 + dy, x + w, y0 + dy, x + w, y0 + dy + h, x, y0 + dy + h])
            x += w + gap
    if polys and draw(st.booleans()):
        polys.append(list(polys[0]))                       # an exact duplicate
    return polys


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(polys=_polys(), width_ths=st.sampled_from([0.5, 1.0, 0.0]), add_margin=st.sampled_from([0.1, 0.0, 0.3]), min_size=st.sampled_from([20, 0, 60]),
       slope_ths=st.sampled_from([0.1, 0.0, 0.5]))
def test_group_text_box_matches_oracle(lib, polys, width_ths, add_margin, min_size, slope_ths):
    """utils.group_text_box + Reader.detect's min_size filter: C++ (boxpost.cpp) == oracle on arbitrary polygon lists, in order."""
    from bb_ocr_amd import _lib
    from oracle import boxes as obox

    p = _lib.bbocr_params()
    lib.bbocr_default_params(C.byref(p))
    p.width_ths, p.add_margin, p.min_size, p.slope_ths = width_ths, add_margin, min_size, slope_ths
    arr = np.array(polys, dtype=np.int32).reshape(-1, 8)
    bl = C.POINTER(_lib.bbocr_boxlist)()
    ptr = arr.ctypes.data_as(C.POINTER(C.c_int)) if len(arr) else None
    assert lib.bbocr_host_group_boxes(ptr, len(arr), C.byref(p), C.byref(bl)) == 0
    b = bl.contents
    hori = [[b.hori[i * 4 + k] for k in range(4)] for i in range(b.hori_off[1])]
    free = [[b.free_q[i * 8 + k] for k in range(8)] for i in range(b.free_off[1])]
    lib.bbocr_free_boxlist(bl)
    oh, of = obox.group_text_box([list(map(int, q)) for q in polys], slope_ths, 0.5, 0.5, width_ths, add_margin)
    oh = [i for i in oh if max(i[1] - i[0], i[3] - i[2]) > min_size]
    of = [i for i in of if max(max(c[0] for c in i) - min(c[0] for c in i), max(c[1] for c in i) - min(c[1] for c in i)) > min_size]
    assert hori == [list(map(int, x)) for x in oh]
    assert np.allclose(np.array(free, dtype=np.float64).reshape(-1, 4, 2), np.array(of, dtype=np.float64).reshape(-1, 4, 2), rtol=0, atol=1e-9)


@st.composite
def _prob_rows(draw):
    """Renormalised class probabilities as ctc_rows_kernel writes them: [n sequences][T][cs], a few dominant classes per step, exact
    ties and all-blank stretches included."""
    n = draw(st.integers(1, 4))
    T = draw(st.integers(1, 24))
    C_, cs = 97, 112
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    mat = np.zeros((n, T, cs), np.float32)
    for i in range(n):
        for t in range(T):
            k = int(rng.integers(1, 5))
            idx = rng.choice(C_, size=k, replace=False)
            w = rng.random(k).astype(np.float32) + np.float32(0.05)
            if rng.random() < 0.2:
                w[:] = w[0]                                  # exact ties between candidates
            if rng.random() < 0.25:
                idx[0] = 0                                   # blank among the candidates
            mat[i, t, idx] = w / w.sum()
    return mat


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(mat=_prob_rows(), beam=st.sampled_from([1, 2, 5, 10]))
def test_ctc_beam_search_matches_oracle(lib, mat, beam):
    """decoder='beamsearch' (easyocr/utils.py::ctcBeamSearch without a language model): ctc_beam.cpp == oracle/recog.py::ctc_beam_search,
    index for index, on arbitrary probability rows."""
    from oracle import recog

    n, T, cs = mat.shape
    flat = np.ascontiguousarray(mat.reshape(-1), dtype=np.float32)
    off = (C.c_int * (n + 1))()
    idx = (C.c_int * (n * T + 1))()
    assert lib.bbocr_host_ctc_beam(flat.ctypes.data_as(C.POINTER(C.c_float)), n, T, 97, cs, beam, off, idx) == 0
    for i in range(n):
        want = recog.ctc_beam_search(mat[i, :, :97], beam_width=beam)
        assert [idx[k] for k in range(off[i], off[i + 1])] == list(want), (i, beam)


def test_host_entry_points_reject_bad_arguments(lib):
    """Status codes, never crashes: null pointers, negative counts, non-positive sizes."""
    from bb_ocr_amd import _lib

    bl = C.POINTER(_lib.bbocr_boxlist)()
    assert lib.bbocr_host_group_boxes(None, 3, None, C.byref(bl)) != 0
    assert lib.bbocr_host_group_boxes(None, -1, None, C.byref(bl)) != 0
    off = (C.c_int * 2)()
    idx = (C.c_int * 4)()
    assert lib.bbocr_host_ctc_beam(None, 1, 2, 97, 112, 5, off, idx) != 0
    one = (C.c_float * 224)()
    assert lib.bbocr_host_ctc_beam(one, 1, 2, 97, 96, 5, off, idx) != 0        # cs < C
    assert lib.bbocr_host_ctc_beam(one, 1, 2, 97, 112, 0, off, idx) != 0       # beam width 0
    out = (C.c_int * 8)()
    assert lib.bbocr_host_component_polys(None, None, 1, 10, 10, 1.0, out) != 0


@st.composite
def _blob_maps(draw):
    """Region / affinity maps made of a few rotated boxes and ellipses, touching the borders, overlapping, one pixel wide, with link
    bridges -- the shapes getDetBoxes_core's min-area-rectangle path has to survive."""
    h, w = draw(st.integers(24, 72)), draw(st.integers(24, 96))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    text = np.zeros((h, w), np.float32)
    link = np.zeros((h, w), np.float32)
    for _ in range(int(rng.integers(1, 7))):
        cx, cy = rng.uniform(-4, w + 4), rng.uniform(-4, h + 4)
        lw, lh = rng.uniform(0.6, 22), rng.uniform(0.6, 7)
        ang = float(rng.choice([0.0, 0.0, rng.uniform(-1.5, 1.5)]))
        ca, sa = np.cos(ang), np.sin(ang)
        u = (xx - cx) * ca + (yy - cy) * sa
        v = -(xx - cx) * sa + (yy - cy) * ca
        d = np.maximum(np.abs(u) / lw, np.abs(v) / lh) if rng.random() < 0.6 else np.sqrt((u / lw) ** 2 + (v / lh) ** 2)
        amp = float(rng.uniform(0.5, 1.4))
        text = np.maximum(text, (amp * np.clip(1.5 - d, 0, 1)).astype(np.float32))
        if rng.random() < 0.4:
            link = np.maximum(link, (0.9 * np.clip(1.4 - np.maximum(np.abs(u - lw) / (0.8 * lw + 0.5), np.abs(v) / lh), 0, 1)).astype(np.float32))
    return text, link


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(maps=_blob_maps(), ratio=st.sampled_from([1.0, 0.7306, 0.5]))
def test_component_polygons_match_oracle(lib, maps, ratio):
    """craft_utils.getDetBoxes_core's per-component geometry + adjustResultCoordinates + get_textbox: boxpost.cpp (dilation of the row
    extremes, convex hull, rotating calipers, boxPoints, diamond fix, int cast) == oracle, polygon for polygon."""
    from oracle import boxes as obox
    from test_abi_host_cpu import _components_from_heat

    text, link = maps
    comps, rows = _components_from_heat(text, link)
    det, _, _ = obox.get_det_boxes_core(text, link)
    want = obox.boxes_to_int_polys(obox.adjust_result_coordinates(det, 1 / ratio, 1 / ratio)) if len(det) else []
    assert len(want) == len(comps)
    if not len(comps):
        return
    out = np.zeros((len(comps), 8), np.int32)
    rc = lib.bbocr_host_component_polys(comps.ctypes.data_as(C.POINTER(C.c_int)), rows.ctypes.data_as(C.POINTER(C.c_int)), len(comps),
                                        text.shape[1], text.shape[0], ratio, out.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0
    assert np.array_equal(out, np.array(want, dtype=np.int32).reshape(-1, 8))
