"""Per-kernel mean durations of one rocprofv3 kernel-trace csv directory (tools/slowmode_probe.sh): which kernel is stretched in a slow process."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 3:]     # skip warm-up
acc = defaultdict(list)
for s, e, n in rows:
    acc[n.split("(")[0][-48:]].append((e - s) / 1e3)
out = []
for n, v in acc.items():
    if sum(v) > 2000:
        v.sort()
        out.append(f"{n}: n={len(v)} min {v[0]:.0f} med {v[len(v) // 2]:.0f} max {v[-1]:.0f} us")
print(" | ".join(sorted(out)))
