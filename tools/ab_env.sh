# The BBOCR_* A/B knobs are read by DIAGNOSTIC builds only: make -C bb-ocr_amd/csrc DIAG=1 OUT=../libbbocr_diag.so, then
#   export BBOCR_LIB_PATH=$PWD/bb-ocr_amd/libbbocr_diag.so
# usage: tools/ab_env.sh VAR "v1 v2 ..." [reps]   -- alternate bench.py runs on ONE box with VAR set to each value
VAR=$1; VALS=$2; REPS=${3:-3}
for i in $(seq $REPS); do
for v in $VALS; do
env $VAR=$v python bench.py --steps 6 --warmup 2 --cpu-pages 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stage_ms_per_step_rank0'];print('$VAR=$v',round(d['value'],1),round(d['ms_per_step'],2),'det',round(s['detector_net'],2),'rec',round(s['recognizer_net'],2),'total',round(s['total'],2))"
done
done
