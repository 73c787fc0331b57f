# usage: tools/ab_lib.sh <other libbbocr.so> [reps]  -- alternate bench.py runs on ONE box between the in-tree library and another build
OTHER=$1; REPS=${2:-3}
for i in $(seq $REPS); do
for lib in "" "$OTHER"; do
BBOCR_LIB_PATH=$lib python bench.py --steps 6 --warmup 2 --cpu-pages 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stage_ms_per_step_rank0'];print('lib=${lib:-in-tree}',round(d['value'],1),round(d['ms_per_step'],2),'det',round(s['detector_net'],2),'rec',round(s['recognizer_net'],2),'retry',round(s['contrast_retry'],2),'total',round(s['total'],2))"
done
done
