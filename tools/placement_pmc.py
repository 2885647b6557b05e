#!/usr/bin/env python3
"""Per state of a tools/placement_roulette.py run under rocprofv3 --pmc: mean duration and counters of its mode-2 stage launches (22 per state and
pass: 11 steps x 2 launches, states one after the other, two passes).   tools/placement_pmc.py <rocprof output dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not cc:
    sys.exit("no counter_collection.csv under " + d)
rows = list(csv.DictReader(open(cc[0])))
# one row per (dispatch, counter)
disp = defaultdict(dict)
meta = {}
for r in rows:
    k = int(r["Dispatch_Id"])
    disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
    meta[k] = (r["Kernel_Name"], int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0))
m2 = [k for k in sorted(disp) if "k_stage_rec2c<6, 10, 2" in meta[k][0]]
names = sorted({c for k in m2 for c in disp[k]})
print("   state pass    avg_us  " + "  ".join(f"{n:>28s}" for n in names))
per = 22
for i in range(0, len(m2) - per + 1, per):
    grp = m2[i:i + per][6:]                      # skip the warm-up steps' launches
    st, ps = (i // per) % 6, (i // per) // 6
    dur = sum(meta[k][2] - meta[k][1] for k in grp) / len(grp) / 1e3
    print(f"   {st:5d} {ps:4d} {dur:9.1f}  " + "  ".join(f"{sum(disp[k].get(n, 0.0) for k in grp) / len(grp):28.4e}" for n in names))
