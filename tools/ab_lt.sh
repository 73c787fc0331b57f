#!/bin/bash
# A/B of two library builds on ONE box: heat-maps bit-compared, then the detector layer table of each, alternating twice
# usage: tools/ab_lt.sh <base.so> <new.so> <tag>
B=$1; N=$2; T=$3
timeout -k 10 200 python tools/ab_heat.py BBOCR_LIB_PATH=$B BBOCR_LIB_PATH=$N > gpurun_out/ab_$T.log 2>&1 || exit 1
for r in 1 2; do
timeout -k 10 200 bash tools/lt.sh ${T}_base$r BBOCR_LIB_PATH=$B > /dev/null || exit 1
timeout -k 10 200 bash tools/lt.sh ${T}_new$r BBOCR_LIB_PATH=$N > /dev/null || exit 1
done
tail -1 gpurun_out/ab_$T.log
for r in 1 2; do for k in base new; do echo "$k$r $(grep 'per page' gpurun_out/lt_${T}_$k$r/layer_table.txt)"; done; done
paste <(awk '{print $1, $(NF-6)}' gpurun_out/lt_${T}_base1/layer_table.txt) <(awk '{print $(NF-6)}' gpurun_out/lt_${T}_new1/layer_table.txt) <(awk '{print $(NF-6)}' gpurun_out/lt_${T}_base2/layer_table.txt) <(awk '{print $(NF-6)}' gpurun_out/lt_${T}_new2/layer_table.txt)
