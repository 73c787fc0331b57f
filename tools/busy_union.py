"""GPU busy fraction of a rocprofv3 --kernel-trace CSV: union of all kernel intervals over the span of the last `steps` bench steps
(steps delimited by gray_kernel launches), and how much of the span has two or more kernels resident.

  python tools/busy_union.py <kernel_trace.csv> [steps]
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r['Start_Timestamp']))
g = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('gray_kernel')]
a = g[-steps - 1] if len(g) > steps else 0
b = g[-1]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows[a:b]]
t0, t1 = iv[0][0], max(e for _, e in iv)
ev = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
busy = multi = 0
depth = 0
prev = t0
for t, d in ev:
    if depth >= 1: busy += t - prev
    if depth >= 2: multi += t - prev
    depth += d
    prev = t
span = t1 - t0
print(f"{b - a} launches over {span/1e6:.2f} ms ({steps} steps: {span/1e6/steps:.2f} ms per step): busy {busy/span:.4f}, idle {(span-busy)/1e6:.2f} ms "
      f"({(span-busy)/span:.4f}), two or more kernels resident {multi/span:.4f}")
