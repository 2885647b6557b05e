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


# ---- the bench line of the traced run (the same process the kernel trace comes from) ----
import json
import re

bench = None
bj = os.path.join(out, "bench_trace.json")
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith("{"):
            bench = json.loads(line)
    if bench:
        print("== bench.py line of the traced run ==")
        print(json.dumps({k: bench[k] for k in ("value", "ms_per_step", "step_ms", "steps", "warmup", "config", "roofline", "calibration",
                                                "tendency_kernel", "forward_euler_compat") if k in bench}))
        if isinstance(bench.get("config5"), dict) and "ms_per_step" in bench["config5"]:
            print("== its config-5 leg ==")
            print(json.dumps(bench["config5"]))
if os.path.exists(os.path.join(out, "command.txt")):
    print(open(os.path.join(out, "command.txt")).read())

# ---- kernel trace: durations ----
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    dur = defaultdict(list)
    seq = []
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        seq.append((int(r["Start_Timestamp"]), short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    # reproducibility check against the bench line: the LAST steps*4 RK-stage launches (modes 1,2,2,3) before the tendency
    # launches are the timed region + the per-stage pass; take the timed region = launches [warmup*4, (warmup+steps)*4)
    legs = []
    if bench:
        seq.sort()
        f32_main = "f32" in bench["config"]["workload"]
        legs.append(("headline workload " + bench["config"]["workload"], bench, r"k_stage_rec2c_f32<\d+, \d+, [123]" if f32_main else r"k_stage_rec2c<\d+, \d+, [123]"))
        c5 = bench.get("config5")
        if isinstance(c5, dict) and "ms_per_step" in c5:
            legs.append(("config-5 leg", {"warmup": c5["warmup"], "steps": c5["steps"], "ms_per_step": c5["ms_per_step"],
                                          "ms_per_step_wall_mean": c5.get("ms_per_step_wall_mean", c5["ms_per_step"]),
                                          "placement": c5.get("placement", {}),
                                          "roofline": {"algorithmic_bytes_per_launch": c5["roofline"]["achieved"] * 1e9 * c5["ms_per_step"] * 1e-3 / 4,
                                                       "frac": c5["roofline"]["frac"]}}, r"k_stage_rec2c_f32<\d+, \d+, [123]"))
    for leg_name, bl, pat in legs:
        stage = [(n, d) for _, n, d in seq if re.match(pat, n)]
        w, k = bl["warmup"], bl["steps"]
        # in front of the warm-up: the placement search of the set-up (moka_state_optimize_placement: one untimed + five timed dt = 0
        # steps for the baseline and per trial: 24 launches each)
        pl = bl.get("placement", {})
        skip = pl.get("stage_launches", 24 * (1 + pl.get("tries", 0)) if pl.get("ms_before") else 0)
        timed = stage[skip + 4 * w:skip + 4 * (w + k)]
        print(f"== {leg_name} ==")
        if len(timed) == 4 * k:
            per_mode = defaultdict(list)
            for n, d in timed:
                per_mode[n].append(d)
            tot = sum(d for _, d in timed) / k / 1e6
            print("== timed region of the traced run: stage kernels per RK4 step ==")
            for n, v in per_mode.items():
                print(f"{n:70s} launches/step={len(v) / k:.0f} avg_us={sum(v) / len(v) / 1e3:10.1f}")
            per_step = sorted(sum(d for _, d in timed[4 * i:4 * i + 4]) / 1e6 for i in range(k))
            med = (per_step[(k - 1) // 2] + per_step[k // 2]) / 2
            wall = bl.get("ms_per_step_wall_mean", bl["ms_per_step"])
            # 0.3 %: the trace's kernel intervals and the HIP events around the region come from two clocks' worth of rounding, and the
            # profiler's begin / end stamps of back-to-back kernels overlap by a few microseconds
            ok = tot <= wall * 1.003 and med <= bl["ms_per_step"] * 1.003
            print(f"stage kernels per step: mean {tot:.3f} ms, median {med:.3f} ms; bench line of the same run: wall mean {wall:.3f} ms, "
                  f"ms_per_step (median of per-step event times) {bl['ms_per_step']:.3f} ms "
                  f"-> {'CONSISTENT' if ok else 'INCONSISTENT (kernel time exceeds the timed region)'}")
            b_contract = bl["roofline"]["algorithmic_bytes_per_launch"] * 4
            print(f"roofline frac recomputed from this trace (contract bytes / kernel time / 8 TB/s) = {b_contract / (tot * 1e-3) / 8e12:.4f}; "
                  f"bench.py printed {bl['roofline']['frac']:.4f}")
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
