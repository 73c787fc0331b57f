"""Diagnostic: recogniser network time vs sequence length / count (bbocr_crnn_logits on random crops)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import weights
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
for n, imgW in ((64, 640), (64, 2560), (64, 4928), (256, 2560), (776, 4928)):
    T = imgW // 4 - 1
    x = (torch.randn(n, 64, imgW, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty((n, T, 112), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for it in range(3):
        t0 = time.perf_counter()
        r._check(r._lib.bbocr_crnn_logits(r._h, C.c_void_p(x.data_ptr()), n, imgW, C.c_void_p(out.data_ptr())))
        dt = time.perf_counter() - t0
    print(f"n={n:4d} imgW={imgW:5d} T={T:5d}: {dt*1e3:8.2f} ms")
