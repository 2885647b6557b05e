#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py,
# summaries into gpurun_out/prof_<tag>/.  Usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu --tend-iters 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$NAME -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_$NAME.err || echo "pmc pass $PMC failed" >> $OUT/errors.txt
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
