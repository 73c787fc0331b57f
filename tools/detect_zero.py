"""Diagnostic: the detector on all-zero weights and pages (under rocprofv3, as tools/detect_only.py): layers whose rate rises on zero data are
bound by the board's power limit, the others by their own schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cs, rs = weights.designed_craft_state(0), weights.synthetic_crnn_state(0)
cs = {k: (np.zeros_like(v) if ("weight" in k and v.ndim == 4) or k.endswith(".bias") else v) for k, v in cs.items()}
r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), det_sub_batch=n)
rgb = torch.zeros((n, 960, 1280, 3), dtype=torch.uint8).cuda()
for _ in range(reps):
    heat, ratio = r.heatmap_device(rgb)
print("ok", heat.shape, float(heat.abs().max()), r.stage_times()["detector_net"])
