#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f32exp; export TMPDIR=/tmp
EXPLIB=$PWD/mpas-ocean.jl_amd/libmoka_hip_exp.so
run() { # tag lib P
  if [ "$2" = B ]; then export MOKA_HIP_LIB=$EXPLIB; else unset MOKA_HIP_LIB; fi
  timeout -k 10 400 python3 bench.py --workload config5_3.7M_x80_f32 --no-cpu --steps 10 --warmup 3 --patch-cells $3 > gpurun_out/f32exp/$1.json 2> gpurun_out/f32exp/$1.err || { echo "$1 failed"; tail -3 gpurun_out/f32exp/$1.err; return; }
  python3 - gpurun_out/f32exp/$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
print(f"{sys.argv[2]}: {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) + f"  tendency {d.get('tendency_kernel', {}).get('avg_launch_ms', float('nan')):.3f} ms  P={d['config'].get('patch_cells')}")
PY
}
run A24 A 0
run B24 B 0
run B20 B 20
run B16 B 16
run A20 A 20
run A24b A 0
