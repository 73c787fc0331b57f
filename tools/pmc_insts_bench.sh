#!/bin/bash
# executed instruction mix per kernel over one bench step (detector + recogniser): tools/pmc_insts_bench.sh <tag> [bench args]
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmcb_$1; shift; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $O/c -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-pages 0 "$@" > $O/run.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], "c", "*", "*counter_collection.csv")), key=os.path.getsize)
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:r["Kernel_Name"].find("(")] if "(" in r["Kernel_Name"] else r["Kernel_Name"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in disp.values():
    a = agg.setdefault(d["kernel"], collections.Counter())
    a["n"] += 1
    for k, v in d.items():
        if k != "kernel": a[k] += v
print(f"{'kernel':60s} {'calls':>5s} {'waves':>9s} {'valu-mfma/w':>11s} {'mfma/w':>8s} {'salu/w':>8s} {'lds/w':>7s} {'vmem/w':>7s} {'cycles/w':>9s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    w = a["SQ_WAVES"] or 1
    print(f"{k[:60]:60s} {a['n']:5d} {int(w):9d} {(a['SQ_INSTS_VALU']-a['SQ_INSTS_MFMA'])/w:11.0f} {a['SQ_INSTS_MFMA']/w:8.0f} {a['SQ_INSTS_SALU']/w:8.0f} {a['SQ_INSTS_LDS']/w:7.0f} {(a['SQ_INSTS_VMEM_RD']+a['SQ_INSTS_VMEM_WR'])/w:7.0f} {a['SQ_WAVE_CYCLES']/w:9.0f}")
PY
rm -rf $O/c
