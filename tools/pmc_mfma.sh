#!/bin/bash
# rocprof-reported MFMA utilisation per detector launch (north_star: "rocprof-reported ... MFMA utilisation against the chip's peak"):
#   tools/pmc_mfma.sh <pages> <out.txt>
# One --pmc pass (own run, --kernel-trace only beside it): SQ_INSTS_MFMA (wave-level MFMA instructions, chip total), SQ_VALU_MFMA_BUSY_CYCLES,
# SQ_BUSY_CYCLES, SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY (quad-cycles), SQ_LDS_BANK_CONFLICT, GRBM_GUI_ACTIVE (shader cycles).
# GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (/ 8 / duration = 1.9-2.0 GHz on the trunk, the clock round 2 read from s_memtime ratios).
#   clock       = GRBM_GUI_ACTIVE / 8 / kernel duration
#   utilisation = 16 cycles x SQ_INSTS_MFMA / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)    (v_mfma_f32_16x16x32_* occupies its SIMD's matrix pipe for 16 cycles;
#                                                                                     SQ_VALU_MFMA_BUSY_CYCLES = 16 x SQ_INSTS_MFMA exactly, both are collected)
#   of peak     = utilisation x clock / 2.4 GHz  (= the TFLOP/s of the layer over 2.5 PFLOP/s, up to zero-padded MFMA rows)
export TMPDIR=/tmp; R=$PWD; N=${1:-32}; OUT=${2:-$R/gpurun_out/pmc_mfma.txt}; O=$R/gpurun_out/pmc_mfma_tmp; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d $O/c -- python3 $R/tools/detect_only.py $N 2 > $O/run.log 2>&1
cd $R
python3 - $O $N > $OUT <<'PY'
import csv, glob, os, sys, collections
o, npages = sys.argv[1], int(sys.argv[2])
f = max(glob.glob(os.path.join(o, "c", "*", "*counter_collection.csv")), key=os.path.getsize)
kt = max(glob.glob(os.path.join(o, "c", "*", "*kernel_trace.csv")), key=os.path.getsize)
dur = {int(r["Dispatch_Id"]): int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
conv = [(k, d) for k, d in sorted(disp.items()) if any(s in d["kernel"] for s in ("conv3x3_dma", "conv1x1_dma", "conv_mfma", "conv3x3_resw", "conv3x3_up4"))]
start = max(i for i, (k, d) in enumerate(conv) if "conv3x3_dma" in d["kernel"] and ", true" in d["kernel"])
names = ["conv1_2(+1_1)", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "fc6", "fc7", "up1a", "up1b", "up2y", "up2s",
         "up2b", "up3y", "up3s", "up3b+4y", "up4s+b", "cls0", "cls2", "cls4(+tail)"]
print(f"# {npages} pages, last detector pass; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA util = 16 x SQ_INSTS_MFMA / (1024 SIMDs x GRBM_GUI_ACTIVE / 8);")
print(f"# of 2.5 PF = util x clock / 2.4 GHz; parked / stall / issue = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY shares of SQ_WAVE_CYCLES")
print(f"{'layer':14s} {'us':>9s} {'clock GHz':>9s} {'MFMA util':>9s} {'of 2.5 PF':>9s} {'parked %':>8s} {'stall %':>8s} {'issue %':>8s} {'LDS confl':>10s}")
tm = tg = 0.0
for i, (k, d) in enumerate(conv[start:start + len(names)]):
    gui = (d.get("GRBM_GUI_ACTIVE", 0) or 1) / 8.0
    assert abs(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) - 16.0 * d.get("SQ_INSTS_MFMA", 0)) <= 1e-6 * max(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 1), 1), "MFMA_BUSY != 16 x INSTS_MFMA"
    us = dur.get(k, 0) / 1e3
    util = 16.0 * d.get("SQ_INSTS_MFMA", 0) / (1024.0 * gui)
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    tm += 16.0 * d.get("SQ_INSTS_MFMA", 0); tg += 1024.0 * gui
    print(f"{names[i]:14s} {us:9.1f} {gui / max(us, 1e-9) / 1e3:9.2f} {util:9.3f} {util * (gui / max(us, 1e-9) / 1e3) / 2.4:9.3f} "
          f"{100 * d.get('SQ_WAIT_ANY', 0) / wc:8.1f} {100 * d.get('SQ_WAIT_INST_ANY', 0) / wc:8.1f} {100 * d.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.1f} {int(d.get('SQ_LDS_BANK_CONFLICT', 0)):10d}")
print(f"whole pass: MFMA pipe utilisation {tm / tg:.3f} at the delivered clock (cycle-weighted over the {len(names)} conv launches)")
PY
rm -rf $O
cat $OUT
