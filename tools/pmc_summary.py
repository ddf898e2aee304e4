#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv) + mean duration from kernel_trace.csv."""
import csv, re, sys, collections
d = sys.argv[1]; pre = sys.argv[2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n)
    return n[:46]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{d}/{pre}_counter_collection.csv")):
    cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in cnt for c in cnt[k]})
print(f"{'kernel':46s} {'n':>5s} {'avg_us':>8s} " + " ".join(f"{n[:18]:>18s}" for n in names))
for k in sorted(cnt, key=lambda k: -sum(dur[k])):
    if not k.startswith("k_"): continue
    print(f"{k:46s} {len(dur[k]):5d} {sum(dur[k])/len(dur[k])/1e3:8.1f} " + " ".join(f"{sum(cnt[k][n])/max(1,len(cnt[k][n])):18.0f}" for n in names))
