#!/bin/bash
# tools/pmc_one.sh <tag> "<counters>" [bench args] : one PMC pass, prints k_stage* rows
TAG=$1; PMC=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_x -- python3 bench.py --steps 2 --warmup 1 --no-cpu --tend-iters 3 "$@" > /dev/null 2> $OUT/err.txt
python3 tools/summarize_prof.py $OUT | grep -E "k_stage|k_fe" 
