"""Per-layer table of the detector from a rocprofv3 --kernel-trace CSV (tools/detect_only.py run): duration and TFLOP/s."""
import csv, sys
MACS = [('conv1_2',45.298 + 2.123),('conv2_1',22.649),('conv2_2',45.298),('conv3_1',22.649),('conv3_2',45.298),('conv3_3',45.298),('conv4_1',22.649),
        ('conv4_2',45.298),('conv4_3',45.298),('conv5_1',11.325),('conv5_2',11.325),('fc6',22.649),('fc7',5.033),('up1a',3.775),('up1b',5.662),
        ('up2y',0.315),('up2s',2.517),('up2b',5.662),('up3y',0.315),('up3s',2.517),('up3b',5.662),('up4y',0.315),('up4s',2.517),('up4b',5.662),('cls0',2.831),('cls2',2.831),('cls4',1.416)]
path, npages = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a detector pass starts at conv1_1 (BBOCR_FUSE1=0) or at the fused conv1_2 launch (the only <..., true> 3x3 instantiation)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('conv1_1_kernel') or ('conv3x3_dma' in r['Kernel_Name'] and ', true' in r['Kernel_Name'])]
start = idx[-1]
k = 0; tot = 0.0; other = 0.0
for r in rows[start:]:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'conv_mfma' in n or 'conv3x3_dma' in n or 'conv1x1_dma' in n or 'conv3x3_resw' in n or 'conv3x3_up4' in n:
        name, m = MACS[k]; k += 1
        if 'conv3x3_up4' in n:          # upconv4's 1x1 over s1 and its 3x3 in one launch
            name, m = 'up4s+b', m + MACS[k][1]; k += 1
        targs = n[n.find('<') + 1:n.find('>')].split(', ')
        if name == 'up3b' and 'conv3x3_dma' in n and len(targs) > 9 and targs[9] == '1':      # upconv4's 1x1 over u3b applied in upconv3.3x3's epilogue (EPI = 1)
            name, m = 'up3b+4y', m + MACS[k][1]; k += 1
        print(f"{name:8s} {n[n.find('<'):n.find('>')+1]:24s} {d:9.1f} us {m*npages*2/(d*1e-6)/1e3:8.1f} TFLOP/s  grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
        tot += d
    else:
        print(f"         {n[:40]:40s} {d:9.1f} us"); other += d
    if k >= len(MACS):
        break
print(f"conv_mfma total {tot:.1f} us, other kernels {other:.1f} us, per page {(tot+other)/npages:.1f} us")
