#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from a tools/profile.sh summary (FETCH_SIZE / WRITE_SIZE passes + the exact 32-byte counters).

Usage: tools/make_traffic.py gpurun_out/prof_<tag>/summary.txt > profiles/pmc_traffic.json

The profiled command is the default bench.py run, which times BASELINE config 4 (k_stage_rec2c) and config 5
(k_stage_rec2c_f32) in one process, so one summary yields both records.
Counter unit of FETCH_SIZE / WRITE_SIZE is KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half
the bytes of 16-byte-per-lane coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is.  Cross-check without a
correction rule: TCC_EA0_RDREQ_DRAM_32B / TCC_EA0_WRREQ_WRITE_DRAM_32B count 32-byte units (a 128-byte request counts 4).
"""
import json
import os
import re
import subprocess
import sys

src = sys.argv[1]
cfgs = {}
bj = os.path.join(os.path.dirname(src), "bench_trace.json")
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith("{"):
            b = json.loads(line)
            c = b["config"]
            cfgs[c["workload"]] = {k: c.get(k) for k in ("patch_cells", "ordering", "kernel_variant")}
            c5 = b.get("config5")
            if isinstance(c5, dict) and "workload" in c5:
                cfgs[c5["workload"]] = {"patch_cells": c5.get("patch_cells"), "ordering": c.get("ordering"), "kernel_variant": c.get("kernel_variant")}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None

vals = {}     # (family, mode, counter) -> per-dispatch average
for line in open(src):
    m = re.match(r"(k_stage_rec2c(?:_f32)?)<\d+, \d+, (\d+)[^>]*>\s+(\S+)\s+n=\s*\d+\s+avg=([\d.e+]+)", line)
    if m:                                      # template arguments: <ME, ME2, MODE[, threads[, waves per SIMD]]>
        vals[(m.group(1), int(m.group(2)), m.group(3))] = float(m.group(4))

modes = [1, 2, 2, 3]  # the four stage launches of one RK4 step
out = {
    "_source": f"{src}: rocprofv3 --pmc passes of `python3 bench.py --steps 3 --warmup 1 --no-cpu --tend-iters 3` (tools/profile.sh; one "
               "counter group per pass).  FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE "
               "reports half the bytes of 16-byte-per-lane coalesced reads, so the read side is doubled; WRITE_SIZE is exact.  Per launch = mean "
               "over the four stage launches of one RK4 step (modes 1,2,2,3).  *_exact = the 32-byte request counters x 32 B (no correction "
               "rule involved).  These bytes cross the L2 <-> fabric interface; Infinity-Cache hits are included (profiles/r03_traffic_attribution.txt).",
}
for fam, workload in (("k_stage_rec2c", "config4_1M_x60"), ("k_stage_rec2c_f32", "config5_3.7M_x80_f32")):
    if (fam, 1, "FETCH_SIZE") not in vals:
        continue
    fetch = [vals[(fam, m, "FETCH_SIZE")] for m in modes]
    write = [vals[(fam, m, "WRITE_SIZE")] for m in modes]
    rec = {
        "config": cfgs.get(workload, {}),
        "commit": commit,
        "kernel": fam,
        "stage_fetch_KiB_raw": fetch,
        "stage_write_KiB": write,
        "stage_bytes_per_launch": int(sum(2 * f + w for f, w in zip(fetch, write)) / 4 * 1024),
        "stage_bytes_per_launch_uncorrected": int(sum(f + w for f, w in zip(fetch, write)) / 4 * 1024),
        "tendency_fetch_KiB_raw": vals.get((fam, 0, "FETCH_SIZE")),
        "tendency_write_KiB": vals.get((fam, 0, "WRITE_SIZE")),
    }
    if rec["tendency_fetch_KiB_raw"] is not None:
        rec["tendency_bytes_per_launch"] = int((2 * rec["tendency_fetch_KiB_raw"] + rec["tendency_write_KiB"]) * 1024)
    rd, wr = "TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"
    if (fam, 1, rd) in vals and (fam, 1, wr) in vals:
        rec["stage_read_bytes_exact"] = [int(vals[(fam, m, rd)] * 32) for m in modes]
        rec["stage_write_bytes_exact"] = [int(vals[(fam, m, wr)] * 32) for m in modes]
        rec["stage_bytes_per_launch_exact"] = int(sum(vals[(fam, m, rd)] + vals[(fam, m, wr)] for m in modes) * 32 / 4)
        if (fam, 0, rd) in vals:
            rec["tendency_bytes_per_launch_exact"] = int((vals[(fam, 0, rd)] + vals[(fam, 0, wr)]) * 32)
    out[workload] = rec
print(json.dumps(out, indent=2))
