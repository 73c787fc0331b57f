#!/bin/bash
# usage: tools/ab_libs.sh <reps> <lib.so>...  -- alternate short bench.py runs on ONE box over the in-tree library and the given builds
REPS=$1; shift
for i in $(seq $REPS); do
for lib in "" "$@"; do
BBOCR_LIB_PATH=$lib python bench.py --steps 6 --warmup 2 --cpu-pages 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stage_ms_per_step_rank0'];print('lib=%-28s'%('${lib:-in-tree}'.split('/')[-1]),round(d['value'],1),round(d['ms_per_step'],2),'det',round(s['detector_net'],2),'rec',round(s['recognizer_net'],2),'total',round(s['total'],2))"
done
done
