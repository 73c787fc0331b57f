"""One-off sweep of the f2 device stages against the CPU restatement on random plane shapes (tile / strip / cell edges of the LDS-tiled
resize, the fused box passes and the cell-wise CLAHE blend).  Test infrastructure: imports oracle/.

  python tools/preprocess_fuzz.py [cases] [seed]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bb_ocr_amd
from bb_ocr_amd import weights
from oracle import preprocess as pp

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))


def stage(k, a, param, dh=None, dw=None):
    src = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dh, dw = dh or a.shape[0], dw or a.shape[1]
    dst = torch.empty((dh, dw), dtype=torch.uint8, device="cuda")
    r._check(r._lib.bbocr_op_preprocess_stage(r._h, k, C.c_void_p(src.data_ptr()), a.shape[0], a.shape[1], C.c_void_p(dst.data_ptr()), dh, dw, float(param)))
    return dst.cpu().numpy()


bad = 0
for i in range(cases):
    H = int(rng.integers(9, 700))
    W = int(rng.integers(3, 260)) * 4 if i % 3 else int(rng.integers(9, 900))
    kind = i % 4
    if kind == 0:
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    elif kind == 1:
        img = rng.normal(128, 50, (H, W)).clip(0, 255).astype(np.uint8)
    elif kind == 2:                                                     # flat steps: exact ties of the cubic resize, clipped CLAHE bins
        img = (np.add.outer(np.arange(H) // 7, np.arange(W) // 5) * 9 % 256).astype(np.uint8)
    else:
        img = np.full((H, W), int(rng.integers(0, 256)), dtype=np.uint8)
        img[H // 3:, W // 2:] = int(rng.integers(0, 256))
    f = float(rng.choice([1.5, 1.5, 0.75, 2.0, 1.1, 3.0]))
    dh, dw = max(int(H * f), 8), max(int(W * f), 8)
    checks = [("resize", lambda: (stage(0, img, 0, dh, dw), pp.resize_cubic_u8(img, dw, dh))),
              ("unsharp", lambda: (stage(5, img, 1.0), pp.pil_unsharp_L(img, 1.0, 30, 3))),
              ("gauss", lambda: (stage(1, img, 3.0), pp.gaussian_blur3_u8(img, 3.0))),
              ("contrast", lambda: (stage(2, img, 1.9), pp.pil_contrast_L(img, 1.9)))]
    if H >= 16 and W >= 16:
        clip = float(rng.choice([2.5, 1.0, 8.0, 40.0]))
        checks.append((f"clahe{clip}", lambda: (stage(4, img, clip), pp.clahe_u8(img, clip, (8, 8)))))
    for name, fn in checks:
        got, want = fn()
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH {name} shape {(H, W)} -> {(dh, dw)} kind {kind}: {int((got != want).sum())} pixels", flush=True)
print(f"{cases} shapes, {bad} mismatching stage results")
sys.exit(1 if bad else 0)
