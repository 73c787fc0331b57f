#!/bin/bash
# timing-only ablations of the trunk kernel (diagnostic build; results are garbage, only durations mean something):
# BBOCR_CONV_DBG bits  8 = no epilogue, 16 = no weight DMA in the k-loop, 32 = no patch DMA in the k-loop.   tools/ab_dbg_lt.sh "0 16 32 48 8 56"
export BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so
for v in $1; do
  timeout -k 10 200 bash tools/lt.sh dbg$v BBOCR_CONV_DBG=$v > /dev/null || exit 1
  echo "== BBOCR_CONV_DBG=$v"
  python3 - gpurun_out/lt_dbg$v/layer_table.txt <<'PY'
import re, sys
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+<.*>\s+([0-9.]+) us\s+([0-9.]+) TFLOP/s", line)
    if m and m.group(1) in ("conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2", "conv5_1", "fc6", "up1b", "up3b+4y"): print(m.group(1), m.group(2), end="; ")
print()
PY
  tail -1 gpurun_out/lt_dbg$v/layer_table.txt
done
