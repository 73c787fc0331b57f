"""BASELINE.json configs[4] shape check: dense A4 @300 dpi pages (2480x3504) through the whole path (canvas_size resize to 2560)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
pg, words = synth.page(4242, width=2480, height=3504, lines=80, font_size=26, line_pitch=42, margin=60)
rgb = torch.from_numpy(np.stack([pg] * n)).cuda()
out = r.readtext_device(rgb)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = r.readtext_device(rgb)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} A4 pages: {dt*1e3:.1f} ms ({n/dt:.1f} pages/s); boxes per page {[len(p) for p in out][:4]} (words drawn {len(words)}); copies identical {all(p == out[0] for p in out)}")
print(r.stage_times())
