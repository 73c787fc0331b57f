#!/bin/bash
# layer table of the detector on all-zero data: tools/lt_zero.sh <tag>
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/lt_$1; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/detect_zero.py 32 2 > $O/run.log 2>&1
cd $R; python3 tools/layer_table.py $(ls $O/tr/*/*kernel_trace.csv | head -1) 32 > $O/layer_table.txt; rm -rf $O/tr; tail -3 $O/layer_table.txt
