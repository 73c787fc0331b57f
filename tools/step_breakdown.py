"""Diagnostic: one bench step from a rocprofv3 --kernel-trace CSV, split at the CCL kernels into detector / post-detector
parts: per kernel name calls + busy time, and the idle time between kernels.

  python tools/step_breakdown.py <kernel_trace.csv>
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# steps are delimited by gray_kernel (first kernel of readtext_batch) -- take the last complete step
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('gray_kernel')]
a = starts[-1]
step = rows[a:]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
ccl = max(i for i, r in enumerate(step) if r['Kernel_Name'].startswith('ccl_rowext')) + 1   # end of the last box extraction
def part(rs, label):
    busy = collections.OrderedDict(); cnt = collections.Counter()
    gaps = 0; prev_end = None; biggaps = []
    for r in rs:
        n = r['Kernel_Name']; n = n[:n.find('(')] if '(' in n else n
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        busy[n] = busy.get(n, 0) + (e - s); cnt[n] += 1
        if prev_end is not None and s > prev_end:
            gaps += s - prev_end
            if s - prev_end > 60000: biggaps.append(((s - prev_end) / 1e3, n))
        prev_end = max(prev_end or 0, e)
    span = int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])
    print(f"== {label}: span {span/1e6:.2f} ms, idle between kernels {gaps/1e6:.2f} ms, {len(rs)} launches")
    for n, b in sorted(busy.items(), key=lambda kv: -kv[1]):
        print(f"   {n[:60]:60s} {cnt[n]:5d} calls {b/1e6:8.3f} ms")
    print("   gaps > 100 us before:", ", ".join(f"{g:.0f}us->{n[:24]}" for g, n in biggaps[:12]))
part(step[:ccl], "detector")
part(step[ccl:], "recogniser (after the last box extraction)")
print(f"step span {(t1 - t0)/1e6:.2f} ms")

if len(sys.argv) > 2 and sys.argv[2] == "launches":        # every conv / GEMM / LSTM launch of the recogniser part, in order
    for r in step[ccl:]:
        n = r['Kernel_Name']
        if any(k in n for k in ('conv', 'lstm')):
            print(f"   {n[:n.find('(')][:58]:58s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} us")
