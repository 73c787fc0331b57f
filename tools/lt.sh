#!/bin/bash
# layer table of the detector alone (32 pages): tools/lt.sh <tag> [env assignments...]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/lt_$1; shift; mkdir -p $O; cd /tmp
env "$@" true
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/detect_only.py 32 2 > $O/run.log 2>&1
cd $R; python3 tools/layer_table.py $(ls $O/tr/*/*kernel_trace.csv | head -1) 32 > $O/layer_table.txt; rm -rf $O/tr; tail -8 $O/layer_table.txt
