"""HBM traffic of one detector pass from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out_f -- python3 tools/detect_only.py 32 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out_w -- python3 tools/detect_only.py 32 1
  python tools/pmc_traffic.py out_f out_w 32 profiles/r01_pmc_hbm.json

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of wide (16 B per lane) coalesced reads -- every read of these kernels is one -- so reads = 2 x FETCH_SIZE.
"""
import csv, glob, json, os, sys

def per_layer(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getsize)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv3x3_dma", "conv1x1_dma", "conv_mfma", "conv3x3_resw", "conv3x3_up4"))]
    start = max(i for i, r in enumerate(conv) if "conv3x3_dma" in r["Kernel_Name"] and ", true" in r["Kernel_Name"])   # fused conv1_2 = first launch of a pass
    return [(r["Kernel_Name"][:r["Kernel_Name"].find("(")], float(r["Counter_Value"])) for r in conv[start:]]

fd, wd, npages, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
F, W = per_layer(fd, "FETCH_SIZE"), per_layer(wd, "WRITE_SIZE")
assert len(F) == len(W), (len(F), len(W))
names = ["conv1_2(+conv1_1)", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "fc6", "fc7", "up1a",
         "up1b", "up2y", "up2s", "up2b", "up3y", "up3s", "up3b", "up4y", "up4s", "up4b", "cls0", "cls2", "cls4(+tail)"]
if any("conv3x3_up4" in kn for kn, _ in F):      # upconv4 fused: one launch for the 1x1 over s1 and the 3x3
    names = names[:22] + ["up4s+up4b (fused)"] + names[24:]
if any("conv3x3_dma" in kn and len(kn[kn.find("<") + 1:kn.find(">")].split(", ")) > 9 and kn[kn.find("<") + 1:kn.find(">")].split(", ")[9] == "1" for kn, _ in F):      # upconv4's y-half 1x1 applied in upconv3.3x3's epilogue
    i = names.index("up3b")
    names = names[:i] + ["up3b+up4y (fused)"] + names[i + 2:]
layers = []
for i, ((kn, f), (_, w)) in enumerate(zip(F, W)):
    rd, wr = 2.0 * f * 1024 / npages, w * 1024 / npages
    layers.append({"layer": names[i] if i < len(names) else kn, "read_MB_per_page": round(rd / 1e6, 2), "write_MB_per_page": round(wr / 1e6, 2)})
    print(f"{layers[-1]['layer']:18s} read {rd/1e6:8.2f} MB/page  write {wr/1e6:8.2f} MB/page")
rd = sum(l["read_MB_per_page"] for l in layers); wr = sum(l["write_MB_per_page"] for l in layers)
print(f"total read {rd:.1f} MB/page, write {wr:.1f} MB/page, {len(layers)} launches per pass")
json.dump({"pages": npages, "launches_per_pass": len(layers), "read_MB_per_page": rd, "write_MB_per_page": wr,
           "correction": "reads = 2 x FETCH_SIZE KiB (gfx950, 16 B/lane streams), writes = WRITE_SIZE KiB", "layers": layers}, open(out, "w"), indent=1)
