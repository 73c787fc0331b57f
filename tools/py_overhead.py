"""Diagnostic: wall time of one bench step (reader.readtext_device on 64 resident pages) split into Python-side pieces."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
uniq = [synth.page(1234 + i)[0] for i in range(8)]
rgb = torch.from_numpy(np.stack([uniq[i % 8] for i in range(64)])).cuda()
r.readtext_device(rgb, None)
r.set_profiling(int(os.environ.get("PROF", "1")))
if os.environ.get("NOGC") == "1":
    gc.disable()
if os.environ.get("FREEZE") == "1":
    bb_ocr_amd.freeze_gc()
torch.cuda.synchronize()
T0 = time.perf_counter()
for it in range(24):
    t0 = time.perf_counter()
    out = r.readtext_device(rgb, None)
    t1 = time.perf_counter()
    st = r.stage_times()
    t2 = time.perf_counter()
    print(f"step wall {1e3*(t1-t0):7.2f} ms  lib total {st['total']:7.2f}  stage_times() {1e3*(t2-t1):5.2f} ms  gc {gc.get_count()}", flush=True)
torch.cuda.synchronize()
print(f"loop avg {1e3*(time.perf_counter()-T0)/24:.2f} ms/step")
