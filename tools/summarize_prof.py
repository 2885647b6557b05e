#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace stats + PMC passes) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    n = name.split("(")[0]
    for pre in ("void moka::", "moka::"):
        n = n.replace(pre, "")
    return n[:70]


# ---- kernel trace: durations ----
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("== kernel trace (ns) ==")
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:70s} {len(v):6d} {sum(v)/len(v)/1e3:10.1f} {min(v)/1e3:10.1f} {max(v)/1e3:10.1f} {sum(v)/1e6:10.2f}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --stats ==")
    print(open(f).read())

# ---- PMC passes ----
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"== PMC {os.path.basename(d)} (per-dispatch average) ==")
        for k, cs in acc.items():
            if not ("k_stage" in k or "k_fe" in k):
                continue
            for c, v in cs.items():
                print(f"{k:70s} {c:32s} n={len(v):4d} avg={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
