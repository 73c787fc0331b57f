"""A/B helper on a detector where every layer matters: run the NOISE-SENSITIVE CRAFT (tests/conftest.py) on 3 pages of 1280x960 (many
tiles per persistent workgroup) + 4 pages of 640x480 under a list of environments (diagnostic build: BBOCR_* knobs) and compare every
heat-map with the first environment's, bit for bit.  The designed detector's lattice-valued maps (tools/ab_heat.py) can hide a
low-order difference; this one cannot.

  BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so python tools/ab_heat_noise.py [fp16|bf16] X=1 BBOCR_UP4_FUSED=0 BBOCR_UP3_POST=0 ...
"""
import os, subprocess, sys, tempfile
import numpy as np

child = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth
from conftest import noise_sensitive_craft
cs, rs = noise_sensitive_craft()
r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision=sys.argv[2])
big = torch.from_numpy(np.stack([synth.page(50 + i)[0] for i in range(3)])).cuda()
small = torch.from_numpy(np.stack([synth.page(910 + i, width=640, height=480, lines=10, margin=24, colour=bool(i & 1))[0] for i in range(4)])).cuda()
a = r.heatmap_device(big)[0].cpu().numpy(); b = r.heatmap_device(small)[0].cpu().numpy()
a2 = r.heatmap_device(big)[0].cpu().numpy()
print("repeatable" if np.array_equal(a, a2) else "NOT REPEATABLE")
np.savez(sys.argv[1], a=a, b=b)
'''
prec = sys.argv[1]
outs = []
for i, env in enumerate(sys.argv[2:]):
    e = dict(os.environ)
    k, v = env.split("=")
    e[k] = v
    f = os.path.join(tempfile.gettempdir(), f"ab_heat_noise_{i}.npz")
    res = subprocess.run([sys.executable, "-c", child, f, prec], env=e, capture_output=True, text=True)
    z = np.load(f)
    outs.append((z["a"], z["b"]))
    if i == 0:
        print(f"{env:28s} reference ({res.stdout.strip()})")
        continue
    da = int((outs[0][0] != z["a"]).sum()); db = int((outs[0][1] != z["b"]).sum())
    mx = max(float(np.abs(outs[0][0] - z["a"]).max()), float(np.abs(outs[0][1] - z["b"]).max()))
    print(f"{env:28s} {res.stdout.strip():15s} differing values 1280x960x3: {da} of {z['a'].size}, 640x480x4: {db} of {z['b'].size}; max |diff| {mx:.3g}", flush=True)
