"""Profiling helper: run the detector alone on N synthetic 1280x960 pages (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sub = int(sys.argv[3]) if len(sys.argv) > 3 else n          # pages per detector pass (default: all n in one pass)
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)), det_sub_batch=sub,
                      precision=os.environ.get("BBOCR_PRECISION", "bf16"))
pg = synth.page(1234)[0]
rgb = torch.from_numpy(np.stack([pg] * n)).cuda()
for _ in range(reps):
    heat, ratio = r.heatmap_device(rgb)
print("ok", heat.shape, r.stage_times()["detector_net"])
