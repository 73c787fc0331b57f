"""A/B helper for the RECOGNISER half (diagnostic build): the whole readtext result -- boxes, texts, confidences as doubles -- of 8 bench
pages + 2 faint-ink pages (contrast retry live) under a list of BBOCR_* environments, compared with the first one's exactly.

  BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so python tools/ab_rec_noise.py fp16 X=1 BBOCR_CONV_RESW64=0 BBOCR_BN256_XPROJ=0 ...
"""
import os, pickle, subprocess, sys, tempfile

child = r'''
import sys, os, pickle
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bb_ocr_amd, bench
from bb_ocr_amd import synth
cs, rs, _ = bench.load_states("trained")
r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision=sys.argv[2])
kw = dict(width=1280, height=960, lines=24, line_pitch=38, margin=24)
pages = [synth.page(1234 + i, colour=bool(i & 1), **kw)[0] for i in range(8)] + [synth.page(9000 + i, faint=0.5, **kw)[0] for i in range(2)]
rgb = torch.from_numpy(np.stack(pages)).cuda()
out = r.readtext_device(rgb)
out2 = r.readtext_device(rgb)
print("repeatable" if out == out2 else "NOT REPEATABLE", sum(len(p) for p in out), "boxes", round(r.stage_times()["contrast_retry"], 2), "ms retry")
pickle.dump(out, open(sys.argv[1], "wb"))
'''
prec = sys.argv[1]
ref = None
for i, env in enumerate(sys.argv[2:]):
    e = dict(os.environ)
    k, v = env.split("=")
    e[k] = v
    f = os.path.join(tempfile.gettempdir(), f"ab_rec_{i}.pkl")
    res = subprocess.run([sys.executable, "-c", child, f, prec], env=e, capture_output=True, text=True)
    out = pickle.load(open(f, "rb"))
    if ref is None:
        ref = out
        print(f"{env:28s} reference ({res.stdout.strip()})")
        continue
    n = sum(len(p) for p in ref)
    same = sum(a == b for pa, pb in zip(ref, out) for a, b in zip(pa, pb))
    print(f"{env:28s} {res.stdout.strip():40s} results identical (box, text, confidence as a double): {same} of {n}", flush=True)
