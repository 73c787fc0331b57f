#!/bin/bash
# one bench step split into detector / recogniser parts: tools/sb.sh <tag> [bench args...]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/sb_$1; shift; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-pages 0 "$@" > $O/run.log 2>&1
cd $R; python3 tools/step_breakdown.py $(ls $O/tr/*/*kernel_trace.csv | head -1) launches > $O/breakdown.txt; rm -rf $O/tr; cat $O/breakdown.txt
