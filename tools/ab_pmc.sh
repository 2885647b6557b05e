#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the stage kernels for the product library (A) and the experiment build (B), one pass each.
#   gpurun -- 'bash tools/ab_pmc.sh <tag>'
TAG=${1:-abpmc}; OUT=gpurun_out/abpmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
EXPLIB=$PWD/mpas-ocean.jl_amd/libmoka_hip_exp.so
for side in A B; do
  if [ $side = B ]; then export MOKA_HIP_LIB=$EXPLIB; else unset MOKA_HIP_LIB; fi
  for PMC in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/${side}/pmc_$PMC -- python3 bench.py --steps 3 --warmup 1 --no-cpu --tend-iters 3 > /dev/null 2> $OUT/${side}_$PMC.err
  done
  echo "== $side =="; python3 tools/summarize_prof.py $OUT/$side | grep -E "k_stage_rec2c<6, 10, [0123]" | awk '{print $1,$2,$3,$4,$5,$7}'
done
