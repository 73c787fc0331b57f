"""GPU probe: are the detector's heat-maps bit-reproducible -- call to call (REPS calls), after other work has dirtied the work buffers,
page by page vs batched, on two page sizes (a 15 x 20-tile grid: one tile per workgroup of the persistent tail kernels; 1280x960: many
tiles per workgroup) -- in every precision mode?  Noise-sensitive detector: every layer feeds every pixel (a designed detector's
lattice-valued maps hide a flaky low-order bit).  Prints one JSON object; exit code 1 if anything differs."""
import hashlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bb_ocr_amd
from bb_ocr_amd import synth
from conftest import noise_sensitive_craft

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cs, rs = noise_sensitive_craft()
small = np.stack([synth.page(910 + i, width=640, height=480, lines=10, margin=24, colour=bool(i & 1))[0] for i in range(4)])
big = np.stack([synth.page(50 + i)[0] for i in range(3)])
out, bad = {}, False
for prec in ("bf16", "fp16", "exact"):
    r = bb_ocr_amd.Reader(["en"], weights=(cs, rs), precision=prec)
    res = {}
    for name, pages, other in (("640x480x4", small, big), ("1280x960x3", big, small)):
        rgb, rgb_other = torch.from_numpy(pages).cuda(), torch.from_numpy(other).cuda()
        first = r.heatmap_device(rgb)[0].cpu().numpy()
        diffs = [int((first != r.heatmap_device(rgb)[0].cpu().numpy()).sum()) for _ in range(REPS if prec != "exact" else 2)]
        r.readtext_device(rgb_other)                     # dirties every work buffer with another shape
        after = int((first != r.heatmap_device(rgb)[0].cpu().numpy()).sum())
        single = int((first != np.concatenate([r.heatmap_device(rgb[i:i + 1])[0].cpu().numpy() for i in range(len(pages))])).sum())
        res[name] = {"digest": hashlib.sha256(first.tobytes()).hexdigest()[:16], "values": int(first.size), "differing_values_call_to_call_max": max(diffs),
                     "differing_after_other_work": after, "differing_page_by_page": single}
        bad |= bool(max(diffs) or after or single)
    out[prec] = res
    r.close()
print(json.dumps(out))
sys.exit(1 if bad else 0)
