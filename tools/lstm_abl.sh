#!/bin/bash
# lstm8_kernel time per launch for the in-tree library and variant builds: tools/lstm_abl.sh <n> <imgW> <lib.so>...
export TMPDIR=/tmp; R=$PWD; N=$1; W=$2; shift; shift
for lib in "" "$@"; do
  O=$R/gpurun_out/lstm_abl; rm -rf $O; mkdir -p $O; cd /tmp
  export BBOCR_LIB_PATH=$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/lstm_one.py $N $W > $O.log 2>&1
  cd $R
  python3 - "$lib" $(ls $O/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[2])))
for r in rows:
    if 'lstm8' in r['Name']:
        print('lib=%-24s lstm8 %d calls, avg %.1f us' % ((sys.argv[1] or 'in-tree').split('/')[-1], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
rm -rf $R/gpurun_out/lstm_abl
