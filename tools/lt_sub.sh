#!/bin/bash
# per-layer detector table as a function of the pass size (pages per pass): does a pass whose low-layer activations fit the 256 MB
# Infinity Cache run those layers faster?   tools/lt_sub.sh "2 4 8 32"
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/lt_sub; mkdir -p $O
for sub in $1; do
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$sub -- python3 $R/tools/detect_only.py 32 2 $sub > $O/run_$sub.log 2>&1
  cd $R; python3 tools/layer_table.py $(ls $O/tr_$sub/*/*kernel_trace.csv | head -1) $sub > $O/layer_table_sub$sub.txt; rm -rf $O/tr_$sub
  echo "== $sub pages per pass"; awk '{print $1, $(NF-6), $(NF-4)}' $O/layer_table_sub$sub.txt 2>/dev/null | tr '\n' ';'; echo; tail -1 $O/layer_table_sub$sub.txt
done
