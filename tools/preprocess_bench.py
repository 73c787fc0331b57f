"""Timing of the f2 pre-processing chain on one 5712x4284 page (the reference's largest photographs): device time and algorithmic
bytes.  (The oracle is test infrastructure and is not imported here.)

  python tools/preprocess_bench.py [H W]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import weights, preprocess as dev_pp

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4284, 5712)
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
rng = np.random.default_rng(0)
bgr = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
for _ in range(2):
    out = dev_pp.preprocess_bgr_device(r, bgr)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    out = dev_pp.preprocess_bgr_device(r, bgr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
dh, dw = out.shape
plane = dh * dw
# gray: 3 in + 1 out (source size); resize: 1 (src) + 1; blur 1+1; clahe hist 1, apply 1+1; six box passes 2 each; unsharp 2+1
alg = H * W * 4 + H * W + plane + 2 * plane + plane + 2 * plane + 12 * plane + 3 * plane
print(f"{H}x{W} -> {dh}x{dw}: {ms:.2f} ms per page (host-synchronous call, one wait at the end of the chain), "
      f"algorithmic {alg/1e6:.0f} MB -> {alg/ms/1e6:.0f} GB/s")
