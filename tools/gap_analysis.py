"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace run (second half of the trace): total busy / span, the largest gaps by
(kernel, next kernel) pair, the largest kernels.   python tools/gap_analysis.py <rocprof output dir>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]   # second half: timed rollouts
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels", len(rows), "span ms", span/1e6, "busy ms", busy/1e6, "idle share", 1 - busy/span)
from collections import defaultdict
gap_after = defaultdict(list)
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gap_after[(a["Kernel_Name"][:50], b["Kernel_Name"][:50])].append(g)
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%9.1f us total  n=%5d  mean %6.2f us   %s -> %s" % (sum(v)/1e3, len(v), sum(v)/len(v)/1e3, k[0], k[1]))
dur = defaultdict(list)
for r in rows: dur[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%9.1f us total  n=%5d  mean %6.2f us   %s" % (sum(v)/1e3, len(v), sum(v)/len(v)/1e3, k))
