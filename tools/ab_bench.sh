#!/bin/bash
# Interleaved A/B of the product library against an experiment build on one GPU box (through gpurun):
#   make -C mpas-ocean.jl_amd exp EXP="-DMOKA_EXP_..."      (here, before gpurun: the .so files travel with the snapshot)
#   gpurun -- 'bash tools/ab_bench.sh <tag> [rounds] [bench args...]'
# Prints ms_per_step / tendency ms of every run; A = libmoka_hip.so, B = libmoka_hip_exp.so.
TAG=${1:-ab}; ROUNDS=${2:-3}; shift 2 || true
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
EXPLIB=$PWD/mpas-ocean.jl_amd/libmoka_hip_exp.so
for r in $(seq 1 $ROUNDS); do
  for side in A B; do
    if [ $side = B ]; then export MOKA_HIP_LIB=$EXPLIB; else unset MOKA_HIP_LIB; fi
    python3 bench.py --no-cpu --steps 20 --warmup 5 "$@" > $OUT/${side}_$r.json 2> $OUT/${side}_$r.err || { echo "$side $r failed"; tail -3 $OUT/${side}_$r.err; continue; }
    python3 - $OUT/${side}_$r.json $side $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
print(f"{sys.argv[2]}{sys.argv[3]}: {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) +
      f"  tendency {d.get('tendency_kernel', {}).get('avg_launch_ms', float('nan')):.3f} ms  FE {d.get('forward_euler_compat', {}).get('ms_per_step', float('nan')):.3f}")
PY
  done
done
