#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pitch; export TMPDIR=/tmp
for r in 1 2; do for w in config4_1M_x60 exp_1M_x64; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 20 --warmup 5 > gpurun_out/pitch/${w}_$r.json 2> gpurun_out/pitch/${w}_$r.err || { echo "$w failed"; tail -3 gpurun_out/pitch/${w}_$r.err; continue; }
  python3 - gpurun_out/pitch/${w}_$r.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
print(f"{sys.argv[2]}: {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) + f"  tendency {d.get('tendency_kernel', {}).get('avg_launch_ms', float('nan')):.3f} ms  FE {d.get('forward_euler_compat', {}).get('ms_per_step', float('nan')):.3f}")
PY
done; done
