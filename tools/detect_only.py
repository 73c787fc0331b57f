"""Profiling helper: run the detector alone on N synthetic 1280x960 pages (used under rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)), det_sub_batch=n)
pg = synth.page(1234)[0]
rgb = torch.from_numpy(np.stack([pg] * n)).cuda()
for _ in range(reps):
    heat, ratio = r.heatmap_device(rgb)
print("ok", heat.shape, r.stage_times()["detector_net"])
