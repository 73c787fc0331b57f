"""Single-page latency of Reader.readtext (the reference's call pattern: one page per call): host array in / JPEG path in, result out,
with the host-side pieces timed separately."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import bb_ocr_amd, bench
from bb_ocr_amd import synth
from bb_ocr_amd.reader import reformat_input
cs, rs, _ = bench.load_states("trained")
r = bb_ocr_amd.Reader(["en"], weights=(cs, rs))
kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24)
pages = [synth.page(1234 + i, colour=bool(i & 1), **kw)[0] for i in range(4)]
def pct(ts):
    ts = sorted(ts); return f"p50 {ts[len(ts)//2]:.2f} ms, p90 {ts[int(len(ts)*0.9)]:.2f} ms, min {ts[0]:.2f} ms"
for p in pages:
    r.readtext(p)
ts = []
for i in range(32):
    t0 = time.perf_counter(); out = r.readtext(pages[i % 4]); ts.append((time.perf_counter() - t0) * 1e3)
print(f"readtext(ndarray 1280x960, {len(out)} boxes): {pct(ts)}")
print("  stage ms of the last call:", {k: round(v, 2) for k, v in r.stage_times().items()})
import torch
dev = [torch.from_numpy(p[None]).cuda() for p in pages]
ts = []
for i in range(32):
    t0 = time.perf_counter(); out = r.readtext_device(dev[i % 4]); ts.append((time.perf_counter() - t0) * 1e3)
print(f"readtext_device(resident page): {pct(ts)}")
ts = []
for i in range(32):
    t0 = time.perf_counter(); a = r._to_dev(pages[i % 4][None]); ts.append((time.perf_counter() - t0) * 1e3)
print(f"_to_dev(3.7 MB page): {pct(ts)}")
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i, pg in enumerate(pages):
        paths.append(os.path.join(d, f"p{i}.jpg")); Image.fromarray(pg).save(paths[-1], quality=95)
    for p in paths: r.readtext(p)
    ts, td = [], []
    for i in range(32):
        t0 = time.perf_counter(); out = r.readtext(paths[i % 4]); ts.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); reformat_input(paths[i % 4], device_gray=True, parallel_decode=True); td.append((time.perf_counter() - t0) * 1e3)
    print(f"readtext(JPEG path): {pct(ts)};  decode alone (RGB + Y on two threads): {pct(td)}")
