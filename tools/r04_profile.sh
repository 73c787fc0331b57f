#!/bin/bash
# Round-4 profiling pass on one MI355X box (run from the repo root through gpurun); summaries are copied into profiles/ afterwards.
# Same command as bench.py's default run (fp16 = the library default, two calls in flight), legs and CPU leg switched off so the kernel
# statistics are those of the timed workload alone; then the one-call-at-a-time run, the exact mode (split-fp16 detector + recogniser),
# the A4 share, the per-layer table of the detector alone and the two HBM counter passes (separate --pmc runs, kernel-trace only).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04prof
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_default -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-pages 0 --legs '' > $O/bench_default.log 2>&1 && echo "bench default (fp16, 2 in flight) profiled" && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_serial -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-pages 0 --legs '' --in-flight 1 > $O/bench_serial.log 2>&1 && echo "bench serial profiled" && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_exact -- python3 $R/bench.py --steps 4 --warmup 2 --cpu-pages 0 --legs '' --precision exact > $O/bench_exact.log 2>&1 && echo "bench exact profiled" && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_a4 -- python3 $R/bench.py --config a4 --steps 4 --warmup 2 --cpu-pages 0 --legs '' > $O/bench_a4.log 2>&1 && echo "bench a4 profiled" && \
rocprofv3 --kernel-trace --output-format csv -d $O/det32 -- python3 $R/tools/detect_only.py 32 2 > $O/det32.log 2>&1 && echo "det32 traced" && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/detect_only.py 32 1 > $O/pmc_f.log 2>&1 && echo "pmc fetch" && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/detect_only.py 32 1 > $O/pmc_w.log 2>&1 && echo "pmc write"
cd $R
python3 tools/layer_table.py $(ls $O/det32/*/*kernel_trace.csv | head -1) 32 > $O/layer_table.txt 2>&1; tail -3 $O/layer_table.txt
python3 tools/pmc_traffic.py $O/pmc_f $O/pmc_w 32 $O/pmc_hbm.json > $O/pmc_traffic.txt 2>&1; tail -2 $O/pmc_traffic.txt
# keep only the summaries (the traces are hundreds of MB)
for d in bench_default bench_serial bench_exact bench_a4; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; grep '^{' $O/$d.log > $O/$d.json; done
rm -rf $O/bench_default $O/bench_serial $O/bench_exact $O/bench_a4 $O/det32 $O/pmc_f $O/pmc_w
ls -la $O
