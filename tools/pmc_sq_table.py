"""Per-layer SQ counters of the last detector pass from `rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv
-- python3 tools/detect_only.py 8 2`:   python tools/pmc_sq_table.py <outdir> > profiles/r01_pmc_sq.csv"""
import csv, glob, os, sys, collections
f = max(glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")), key=os.path.getsize)
names = ["conv1_2(+conv1_1)", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "fc6", "fc7", "up1a",
         "up1b", "up2y", "up2s", "up2b", "up3y", "up3s", "up3b", "up4y", "up4s", "up4b", "cls0", "cls2", "cls4(+tail)"]
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = int(r["Dispatch_Id"])
    d = disp.setdefault(k, {"kernel": r["Kernel_Name"], "grid": r["Grid_Size"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
conv = [(k, d) for k, d in sorted(disp.items()) if any(s in d["kernel"] for s in ("conv3x3_dma", "conv1x1_dma", "conv_mfma"))]
start = max(i for i, (k, d) in enumerate(conv) if "conv3x3_dma" in d["kernel"] and ", true" in d["kernel"])
w = csv.writer(sys.stdout)
w.writerow(["layer", "dispatch", "kernel", "grid"] + cols + ["wait_any_pct", "wait_inst_pct", "active_pct"])
for i, (k, d) in enumerate(conv[start:start + len(names)]):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    w.writerow([names[i], k, d["kernel"], d["grid"]] + [int(d.get(c, 0)) for c in cols] +
               [round(100 * d.get("SQ_WAIT_ANY", 0) / wc, 1), round(100 * d.get("SQ_WAIT_INST_ANY", 0) / wc, 1), round(100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 1)])
