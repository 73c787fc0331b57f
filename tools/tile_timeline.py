"""Diagnostic: per-workgroup timeline of the LDS-DMA 3x3 conv kernel (BBOCR_CONV_STAMPS).

  BBOCR_CONV_STAMPS=gpurun_out/stamps python tools/detect_only.py 8 1 && python tools/tile_timeline.py gpurun_out/stamps

Each file holds, per workgroup, s_memtime at {start, prologue landed, main loop done, stores drained}.
"""
import glob, os, re, sys
import numpy as np

d = sys.argv[1]
def pers(f, m):
    seq, cin, cout, H, W, grid, bn, ring = map(int, m.groups())
    t = np.fromfile(f, dtype=np.uint64).reshape(grid, -1, 4)
    hw = t[:, 0, 3]
    xcc, cu, sh, se, slot = (hw >> 32) & 15, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, hw & 15
    key = (xcc * 8 + se) * 32 + sh * 16 + cu
    t = t.astype(np.int64)
    valid = t[:, :, 0] != 0
    nk = cin // 32 * 9
    main = (t[:, :, 1] - t[:, :, 0])[valid]
    epi = (t[:, :, 2] - t[:, :, 1])[valid]
    per = np.diff(t[:, :, 0], axis=1)[valid[:, 1:]]
    q = lambda x: "%6d/%6d/%6d" % tuple(np.percentile(x, [10, 50, 90]))
    print(f"{seq:03d} PERS {cin:4d}->{cout:4d} {H}x{W} bn{bn} grid {grid}: main {q(main)} ({np.median(main)/nk:6.1f}/k-step) epilogue {q(epi)} period {q(per)}"
          f"  CUs {len(set(key.tolist()))} slots {sorted(set(slot.tolist()))}")
    # phase between the two workgroups of a CU: offset of their main-loop starts (tile 3) relative to the period
    ph, deltas = [], set()
    P = float(np.median(per))
    for k in set(key.tolist()):
        w = np.nonzero(key == k)[0]
        if len(w) == 2 and valid[w[0], 3] and valid[w[1], 3]:
            dt = abs(int(t[w[0], 3, 0]) - int(t[w[1], 3, 0]))
            ph.append((dt % P) / P)
            deltas.add(int(abs(int(w[0]) - int(w[1]))))
    if ph:
        print(f"     co-resident phase offset (per mille of a period) {q(np.array(ph) * 1000)}; pairs {len(ph)}; block-id deltas {sorted(deltas)[:8]}")


for f in sorted(glob.glob(os.path.join(d, "stamps_*.bin"))):
    if f.endswith("_pers.bin"):
        pers(f, re.search(r"stamps_(\d+)_(\d+)x(\d+)_(\d+)x(\d+)_g(\d+)_bn(\d+)_r(\d+)", f))
        continue
    m = re.search(r"stamps_(\d+)_(\d+)x(\d+)_(\d+)x(\d+)_g(\d+)_bn(\d+)_r(\d+)", f)
    seq, cin, cout, H, W, grid, bn, ring = map(int, m.groups())
    t = np.fromfile(f, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    t = t[(t != 0).all(axis=1)]
    pro, main, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    span = t[:, 3].max() - t[:, 0].min()
    tot = (t[:, 3] - t[:, 0]).sum()
    nk = cin // 32 * 9
    q = lambda x: "%6d/%6d/%6d" % tuple(np.percentile(x, [10, 50, 90]))
    print(f"{seq:03d} {cin:4d}->{cout:4d} {H}x{W} bn{bn} r{ring} grid {grid:6d}: prologue {q(pro)}  main {q(main)} ({np.median(main)/nk:6.1f}/k-step)  "
          f"epilogue+drain {q(epi)}  resident WGs {tot/span:6.1f}  main share {main.sum()/tot:.3f}")
