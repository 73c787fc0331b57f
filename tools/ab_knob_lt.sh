#!/bin/bash
# A/B of one diagnostic knob on ONE box (diagnostic build): heat-maps compared, then the detector layer table with the knob 0 / 1, alternating twice
# usage: tools/ab_knob_lt.sh BBOCR_S1_POST <tag>
V=$1; T=$2
export BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so
timeout -k 10 300 python tools/ab_heat.py $V=0 $V=1 > gpurun_out/ab_$T.log 2>&1 || exit 1
for r in 1 2; do
timeout -k 10 200 bash tools/lt.sh ${T}_off$r $V=0 > /dev/null || exit 1
timeout -k 10 200 bash tools/lt.sh ${T}_on$r $V=1 > /dev/null || exit 1
done
cat gpurun_out/ab_$T.log
for r in 1 2; do for k in off on; do echo "$k$r $(grep 'per page' gpurun_out/lt_${T}_$k$r/layer_table.txt)"; done; done
for k in off1 on1 off2 on2; do echo "== $k"; grep -E "^conv2_2|^up4" gpurun_out/lt_${T}_$k/layer_table.txt | cut -c1-90; done
