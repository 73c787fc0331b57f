#!/bin/bash
# per-layer detector table as a function of the pass size (pages per pass): does a pass whose low-layer activations fit the 256 MB
# Infinity Cache run those layers faster?   tools/lt_sub.sh "2 4 8 32"
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/lt_sub; mkdir -p $O
for sub in $1; do
  cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_$sub -- python3 $R/tools/detect_only.py 32 2 $sub > $O/run_$sub.log 2>&1
  cd $R; python3 tools/layer_table.py $(ls $O/tr_$sub/*/*kernel_trace.csv | head -1) $sub > $O/layer_table_sub$sub.txt; rm -rf $O/tr_$sub
  echo "== $sub pages per pass"
  python3 - $O/layer_table_sub$sub.txt <<'PY'
import re, sys
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+<.*>\s+([0-9.]+) us\s+([0-9.]+) TFLOP/s", line)
    if m: print(f"{m.group(1)} {m.group(2)} us {m.group(3)} TF;", end=" ")
print()
PY
  tail -1 $O/layer_table_sub$sub.txt
done
