#!/bin/bash
# executed instruction mix per detector launch (8 pages): tools/pmc_insts.sh <tag>
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmci_$1; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/c -- python3 $R/tools/detect_only.py 8 1 > $O/run.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], "c", "*", "*counter_collection.csv")), key=os.path.getsize)
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"][:r["Kernel_Name"].find("(")], "grid": int(r["Grid_Size"]) // int(r["Workgroup_Size"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
rows = [d for k, d in sorted(disp.items()) if "conv" in d["kernel"]][-26:]
print(f"{'kernel':62s} {'WGs':>7s} {'valu/wave':>10s} {'salu':>8s} {'lds':>7s} {'mfma':>7s} {'vmem':>7s}")
for d in rows:
    w = d.get("SQ_WAVES", 1) or 1
    print(f"{d['kernel'][:62]:62s} {d['grid']:7d} {d.get('SQ_INSTS_VALU',0)/w:10.0f} {d.get('SQ_INSTS_SALU',0)/w:8.0f} {d.get('SQ_INSTS_LDS',0)/w:7.0f} {d.get('SQ_INSTS_MFMA',0)/w:7.0f} {(d.get('SQ_INSTS_VMEM_RD',0)+d.get('SQ_INSTS_VMEM_WR',0))/w:7.0f}")
PY
rm -rf $O/c
