#!/bin/bash
# Interleaved A/B of one moka_set_tuning switch on one GPU box:  gpurun -- 'bash tools/ab_tuning.sh <tag> <key> <valA> <valB> [rounds] [bench args...]'
# Prints ms_per_step / stages / tendency / Forward Euler of every run (A = key set to valA, B = valB).
TAG=$1; KEY=$2; VA=$3; VB=$4; ROUNDS=${5:-2}; shift 5 || true
OUT=gpurun_out/abt_$TAG; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for side in A B; do
    if [ $side = A ]; then V=$VA; else V=$VB; fi
    python3 bench.py --no-cpu --steps 20 --warmup 5 --tuning $KEY=$V "$@" > $OUT/${side}_$r.json 2> $OUT/${side}_$r.err || { echo "$side $r failed"; tail -3 $OUT/${side}_$r.err; continue; }
    python3 - $OUT/${side}_$r.json $side $r $KEY $V <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
print(f"{sys.argv[2]}{sys.argv[3]} (key {sys.argv[4]} = {sys.argv[5]}): {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) +
      f"  tendency {d.get('tendency_kernel', {}).get('avg_launch_ms', float('nan')):.3f} ms  FE lean {d.get('forward_euler_compat', {}).get('ms_per_step', float('nan')):.3f}"
      f" all-arrays {d.get('forward_euler_compat', {}).get('ms_per_step_all_arrays_stored', float('nan')):.3f}  gather probe {d.get('calibration', {}).get('gather_GBs_before', float('nan')):.0f} GB/s", flush=True)
PY
  done
done
