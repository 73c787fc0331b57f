"""A/B helper: run the detector on the same 8 synthetic pages under two environments and compare the heat-maps bit for bit.

  BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so python tools/ab_heat.py BBOCR_UP4_FUSED=0 BBOCR_UP4_FUSED=1

(the knobs exist in diagnostic builds only: make -C bb-ocr_amd/csrc DIAG=1 OUT=../libbbocr_diag.so)
"""
import os, subprocess, sys, tempfile
import numpy as np

child = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
pages = np.stack([synth.page(100 + i)[0] for i in range(8)])
rgb = torch.from_numpy(pages).cuda()
heat, ratio = r.heatmap_device(rgb)
for _ in range(3):
    heat, ratio = r.heatmap_device(rgb)
print("detector_net ms (8 pages)", r.stage_times()["detector_net"])
np.save(sys.argv[1], heat.cpu().numpy())
'''
outs = []
for i, env in enumerate(sys.argv[1:3]):
    e = dict(os.environ)
    k, v = env.split("=")
    e[k] = v
    f = os.path.join(tempfile.gettempdir(), f"ab_heat_{i}.npy")
    print(env, subprocess.run([sys.executable, "-c", child, f], env=e, capture_output=True, text=True).stdout.strip())
    outs.append(np.load(f))
d = np.abs(outs[0].astype(np.float64) - outs[1].astype(np.float64))
print("max |diff|", d.max(), "identical" if np.array_equal(outs[0], outs[1]) else "DIFFERENT", "heat max", outs[0].max())
