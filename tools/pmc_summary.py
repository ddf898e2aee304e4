#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (<pre>_counter_collection.csv, optional) and mean duration (<pre>_kernel_trace.csv).
usage: pmc_summary.py DIR PREFIX [--tail FRAC] [--json OUT]
  --tail FRAC   only the last FRAC of every kernel's dispatches (bench.py --warmup W --steps K: FRAC = K / (W + K) is the timed window)
  --json OUT    also dump {kernel: {"n":, "avg_us":, counter: mean, ...}}"""
import collections, csv, json, os, re, sys

d, pre = sys.argv[1], sys.argv[2]
frac, out_json = 1.0, None
a = sys.argv[3:]
while a:
    if a[0] == "--tail": frac = float(a[1])
    elif a[0] == "--json": out_json = a[1]
    a = a[2:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n)
    return n[:46]


def tail(v):
    return v[len(v) - max(1, int(round(len(v) * frac))):]


dur = collections.defaultdict(list)
rows = sorted(csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
cc = f"{d}/{pre}_counter_collection.csv"
if os.path.exists(cc):
    crow = list(csv.DictReader(open(cc)))
    key = "Dispatch_Id" if crow and "Dispatch_Id" in crow[0] else None
    if key: crow.sort(key=lambda r: int(r[key]))
    for r in crow:
        cnt[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in cnt for c in cnt[k]})
doc = {}
print(f"{'kernel':46s} {'n':>5s} {'avg_us':>8s} {'total_ms':>9s} " + " ".join(f"{n[:18]:>18s}" for n in names))
for k in sorted(dur, key=lambda k: -sum(tail(dur[k]))):
    if not k.startswith("k_"): continue
    t = tail(dur[k])
    rec = {"n": len(t), "avg_us": sum(t) / len(t) / 1e3}
    for n in names:
        v = tail(cnt[k][n]) if cnt[k][n] else []
        rec[n] = sum(v) / max(1, len(v))
    doc[k] = rec
    print(f"{k:46s} {len(t):5d} {rec['avg_us']:8.1f} {sum(t) / 1e6:9.3f} " + " ".join(f"{rec[n]:18.0f}" for n in names))
if out_json:
    json.dump(doc, open(out_json, "w"), indent=1)
