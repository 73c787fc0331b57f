"""Diagnostic: one recogniser pass over n random crops of one width (bbocr_crnn_logits), for a kernel trace of lstm8_kernel (tools/lstm_abl.sh)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bb_ocr_amd
from bb_ocr_amd import weights
n, imgW = int(sys.argv[1]), int(sys.argv[2])
r = bb_ocr_amd.Reader(["en"], weights=(weights.designed_craft_state(0), weights.synthetic_crnn_state(0)))
T = imgW // 4 - 1
x = (torch.randn(n, 64, imgW, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty((n, T, 112), dtype=torch.float32, device="cuda")
for it in range(3):
    r._check(r._lib.bbocr_crnn_logits(r._h, C.c_void_p(x.data_ptr()), n, imgW, C.c_void_p(out.data_ptr())))
torch.cuda.synchronize()
print("ok", n, imgW, T)
