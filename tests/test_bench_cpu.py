"""CPU: the pieces of bench.py that decide what the JSON line says (no GPU work): the parity counter behind `parity_in_run`, the
config table, the seeded page set."""
import numpy as np


def test_parity_counter_counts_boxes_texts_and_pages():
    import bench

    box = lambda x: [[x, 0], [x + 9, 0], [x + 9, 5], [x, 5]]
    want = [[(box(0), "alpha", 0.9), (box(20), "beta", 0.8)], [(box(5), "gamma", 0.7)], []]
    same = [[(box(0), "alpha", 0.91), (box(20), "beta", 0.5)], [(box(5), "gamma", 0.7)], []]
    p = bench.parity(want, same, "mixed")
    assert p["all_identical"] and p["boxes_identical"] == "3/3" and p["texts_identical"] == "3/3" and p["pages_identical"] == "3/3"
    assert p["first_text_difference"] is None and p["mode"] == "mixed"          # confidences do not enter: boxes and strings do
    off = [[(box(0), "alpha", 0.9), (box(21), "beta", 0.8)], [(box(5), "gamna", 0.7)], []]
    p = bench.parity(want, off, "bf16")
    assert not p["all_identical"] and p["boxes_identical"] == "2/3" and p["texts_identical"] == "2/3" and p["pages_identical"] == "1/3"
    assert p["first_text_difference"] == {"oracle": "gamma", "gpu": "gamna"}
    # a page with a missing box can never count as identical, free (float) boxes compare as numbers
    fl = [[([[0.5, 0.0], [9.5, 0.0], [9.5, 5.0], [0.5, 5.0]], "x", 0.5)]]
    assert bench.parity(fl, [[([[0.5, 0], [9.5, 0], [9.5, 5], [0.5, 5]], "x", 0.4)]], "fp16")["all_identical"]
    assert bench.parity(want[:1], [[(box(0), "alpha", 0.9)]], "fp16")["pages_identical"] == "0/1"


def test_configs_name_the_baseline_workloads():
    import bench

    assert bench.CONFIGS["p1"][:3] == (1280, 960, 64) and "configs[2]" in bench.CONFIGS["p1"][6]
    assert bench.CONFIGS["a4"][:3] == (2480, 3504, 16) and bench.CONFIGS["a4"][5] == "fp16" and "configs[4]" in bench.CONFIGS["a4"][6]
    assert set(bench.DTYPE) == {"bf16", "fp16", "exact", "mixed", "exact_rec"} and bench.PEAK_BF16_TFLOPS == 2500.0
    assert abs(bench.CRAFT_GFLOP_PER_PAGE["p1"] - 874.22) < 1e-9 and abs(bench.CRAFT_GFLOP_PER_PAGE["a4"] - 3322.03) < 1e-9


def test_rendered_pages_are_seeded_and_distinct():
    import bench

    a = bench.render_pages("p1", 320, 192, 3, 0, 3, 3)
    b = bench.render_pages("p1", 320, 192, 3, 0, 3, 3)
    assert len(a) == 3 and all(np.array_equal(x, y) for x, y in zip(a, b))          # same seeds -> same pages (every rank, every run)
    assert not np.array_equal(a[0], a[1]) and a[0].shape == (192, 320, 3) and a[0].dtype == np.uint8
    assert np.array_equal(a[0][..., 0], a[0][..., 1]) and not np.array_equal(a[1][..., 0], a[1][..., 1])      # every other page on tinted stock
    assert np.array_equal(bench.render_pages("p1", 320, 192, 3, 1, 2, 2)[0], a[1])                              # a shard starts at its global index
