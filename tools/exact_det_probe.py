"""GPU probe: heat-map error / threshold flips of every precision mode against the fp32 CPU oracle on the noise-sensitive detector."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bb_ocr_amd
from bb_ocr_amd import synth
from conftest import noise_sensitive_craft
from oracle import pipeline

cs, rs = noise_sensitive_craft()
ref = pipeline.OracleReader({k: torch.from_numpy(v) for k, v in cs.items()}, {k: torch.from_numpy(v) for k, v in rs.items()})
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pages = [synth.page(910 + i, width=640, height=480, lines=10, margin=24, colour=bool(i & 1))[0] for i in range(n)]
want = [ref.heatmap(p) for p in pages]
out = {}
for prec in ("fp16", "exact"):
    r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision=prec)
    t0 = time.time()
    heat, _ = r.heatmap_device(torch.from_numpy(np.stack(pages)).cuda())
    dt = time.time() - t0
    flips = 0; err = 0.0; near = 0
    for i, (st, sl, _) in enumerate(want):
        h = heat[i].cpu().numpy()
        flips += int(((h[..., 0] > 0.4) != (st > 0.4)).sum() + ((h[..., 1] > 0.4) != (sl > 0.4)).sum() + ((h[..., 0] >= 0.7) != (st >= 0.7)).sum())
        err = max(err, float(np.abs(h[..., 0] - st).max()), float(np.abs(h[..., 1] - sl).max()))
        near += int((np.abs(st - 0.4) < 1e-5).sum() + (np.abs(sl - 0.4) < 1e-5).sum() + (np.abs(st - 0.7) < 1e-5).sum())
    out[prec] = {"flips": flips, "max_abs_err": err, "oracle_pixels_within_1e-5_of_a_threshold": near, "first_call_s": dt}
    r.close()
print(json.dumps(out))
