#!/bin/bash
# lstm8_kernel time per launch against the x prefetch distance (diagnostic build): tools/lstm_pf.sh <n> <imgW> <pf>...
export TMPDIR=/tmp; R=$PWD; N=$1; W=$2; shift; shift
export BBOCR_LIB_PATH=$R/bb-ocr_amd/libbbocr_diag.so
for pf in "$@"; do
  O=$R/gpurun_out/lstm_pf; rm -rf $O; mkdir -p $O; cd /tmp
  export BBOCR_LSTM_PF=$pf
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/lstm_one.py $N $W > $O.log 2>&1
  cd $R
  python3 - "$pf" $(ls $O/*/*kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'lstm8' in r['Name']:
        print('pf=%s lstm8 %d calls, avg %.1f us' % (sys.argv[1], int(r['Calls']), float(r['AverageNs']) / 1e3))
PY
done
rm -rf $R/gpurun_out/lstm_pf
