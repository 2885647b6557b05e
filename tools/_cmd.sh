#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in BASE NOCELLS NOROWS NOEDGES NOF BASE; do
  if [ $v = BASE ]; then unset MOKA_HIP_LIB; else export MOKA_HIP_LIB=$PWD/mpas-ocean.jl_amd/libmoka_abl_$v.so; fi
  rm -rf gpurun_out/abl_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$v -- python3 tools/nonlinear_timing.py 320 60 > gpurun_out/abl_$v.log 2>&1 || { tail -3 gpurun_out/abl_$v.log; continue; }
  f=$(ls -t gpurun_out/abl_$v/*/*kernel_stats.csv | head -1)
  python3 - "$f" $v <<'PY'
import csv,sys
r={x["Name"][:40]:float(x["AverageNs"])/1e3 for x in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], " ".join(f"{k.split('moka::')[-1][:22]}={v:.0f}" for k,v in r.items() if "nl" in k or "rec2c<6, 10, 2" in k))
PY
done
