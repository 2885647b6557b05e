#!/bin/bash
# Interleaved comparison of several builds of the library on one GPU box (through gpurun):
#   tools/ab_libs.sh <tag> <rounds> "<name1> <name2> ..." [bench args...]     name "-" = the product library
# Prints ms_per_step / stages / tendency / Forward Euler of every run.
TAG=$1; ROUNDS=$2; NAMES=$3; shift 3
OUT=gpurun_out/abl_$TAG; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for n in $NAMES; do
    if [ "$n" = "-" ]; then unset MOKA_HIP_LIB; else export MOKA_HIP_LIB=$PWD/mpas-ocean.jl_amd/libmoka_hip_$n.so; fi
    python3 bench.py --no-cpu "$@" > $OUT/${n}_$r.json 2> $OUT/${n}_$r.err || { echo "$n $r failed"; tail -3 $OUT/${n}_$r.err; continue; }
    python3 - $OUT/${n}_$r.json $n $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
fe = d.get("forward_euler_compat", {})
print(f"{sys.argv[2]:>10s} r{sys.argv[3]}: {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) +
      f"  tendency {d.get('tendency_kernel', {}).get('avg_launch_ms', float('nan')):.3f} ms  FE lean {fe.get('ms_per_step', float('nan')):.3f}"
      f" all {fe.get('ms_per_step_all_arrays_stored', float('nan')):.3f}", flush=True)
PY
  done
done
