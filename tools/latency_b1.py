"""Single-page latency of Reader.readtext (the reference's call pattern: one page per call), host array in, result out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bb_ocr_amd
from bb_ocr_amd import synth, weights
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
pages = [synth.page(500 + i)[0] for i in range(4)]
for p in pages:
    r.readtext(p)
ts = []
for i in range(24):
    t0 = time.perf_counter()
    out = r.readtext(pages[i % 4])
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"readtext(1280x960 page, {len(out)} boxes): p50 {ts[len(ts)//2]:.2f} ms, p90 {ts[int(len(ts)*0.9)]:.2f} ms, min {ts[0]:.2f} ms")
print(r.stage_times())
