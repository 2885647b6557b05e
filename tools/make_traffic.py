#!/usr/bin/env python3
"""Build profiles/pmc_traffic.json from a tools/profile.sh summary (FETCH_SIZE / WRITE_SIZE passes).

Usage: tools/make_traffic.py gpurun_out/prof_<tag>/summary.txt <workload> > profiles/pmc_traffic.json
Counter unit is KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the
bytes of 16-byte-per-lane coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is.
"""
import json
import re
import sys

import os
import subprocess

src, workload = sys.argv[1], sys.argv[2]
# optional: the bench line of the profiled run (gives the configuration the counters belong to)
cfg = {}
bj = os.path.join(os.path.dirname(src), "bench_trace.json")
if os.path.exists(bj):
    for line in open(bj):
        if line.startswith("{"):
            c = json.loads(line)["config"]
            cfg = {k: c.get(k) for k in ("patch_cells", "ordering", "kernel_variant")}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None
vals = {}
for line in open(src):
    m = re.match(r"(k_stage_\w+)<\d+, \d+, (\d+)[^>]*>\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+avg=([\d.e+]+)", line)
    if m:                                      # template arguments: <ME, ME2, MODE[, threads]>
        vals[(int(m.group(2)), m.group(3))] = float(m.group(4))
        kernel = m.group(1)
modes = [1, 2, 2, 3]  # the four stage launches of one RK4 step
fetch = [vals[(m, "FETCH_SIZE")] for m in modes]
write = [vals[(m, "WRITE_SIZE")] for m in modes]
out = {
    "_source": f"{src}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu --tend-iters 3` (tools/profile.sh). Counter unit is KiB. "
               "gfx950 correction per MI355X_MICROARCH.md section HBM: FETCH_SIZE reports half the bytes of "
               "16-byte-per-lane coalesced reads, so the read side is doubled; WRITE_SIZE is exact. "
               "Per launch = mean over the four stage launches of one RK4 step (modes 1,2,2,3).",
    "_kernel": kernel,
    workload: {
        "config": cfg,
        "commit": commit,
        "stage_fetch_KiB_raw": fetch,
        "stage_write_KiB": write,
        "stage_bytes_per_launch": int(sum(2 * f + w for f, w in zip(fetch, write)) / 4 * 1024),
        "stage_bytes_per_launch_uncorrected": int(sum(f + w for f, w in zip(fetch, write)) / 4 * 1024),
        "tendency_fetch_KiB_raw": vals[(0, "FETCH_SIZE")],
        "tendency_write_KiB": vals[(0, "WRITE_SIZE")],
        "tendency_bytes_per_launch": int((2 * vals[(0, "FETCH_SIZE")] + vals[(0, "WRITE_SIZE")]) * 1024),
    },
}
print(json.dumps(out, indent=2))
