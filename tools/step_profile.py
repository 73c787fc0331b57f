"""Diagnostic: where one 64-page readtext step spends its time outside the GPU stages (C call vs Python result marshalling)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import bb_ocr_amd
from bb_ocr_amd import synth, weights, _lib
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
uniq = [synth.page(1000 + i)[0] for i in range(8)]
rgb = torch.from_numpy(np.stack([uniq[i % 8] for i in range(64)])).cuda()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = r._params({})
    res = C.POINTER(_lib.bbocr_result)()
    B, H, W, _ = rgb.shape
    rc = r._lib.bbocr_readtext_batch(r._h, C.c_void_p(rgb.data_ptr()), C.c_void_p(None), B, H, W, C.byref(p), C.byref(res))
    t1 = time.perf_counter()
    out = r._collect(res, 1)
    t2 = time.perf_counter()
    st = r.stage_times()
    print(f"C call {1e3*(t1-t0):7.2f} ms  (lib total {st['total']:7.2f}, stages sum {sum(v for k, v in st.items() if k != 'total'):7.2f})   _collect {1e3*(t2-t1):6.2f} ms   boxes {sum(len(x) for x in out)}")
