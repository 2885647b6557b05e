#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py, summaries into
# gpurun_out/prof_<tag>/.  Usage: tools/profile.sh <tag> [bench args...]
# The trace is taken with the driver's own bench command line (--steps 20 --warmup 5), program directly after `--`, so that
# the kernel durations of profiles/ are comparable with the driver's BENCH record: tools/summarize_prof.py checks that
# the stage kernels of one step sum to no more than the ms_per_step bench.py printed in the same (traced) run -- for the headline
# workload (config 4) and for the config-5 leg of the same run -- and prints the run's own calibration beside them.
# PMC passes use fewer steps (counters are per dispatch and do not depend on the step count).  One --pmc group per pass
# (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), never combined with any other trace domain than --kernel-trace.
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
TRACE_ARGS="--steps 20 --warmup 5 --no-cpu $*"
PMC_ARGS="--steps 3 --warmup 1 --no-cpu --tend-iters 3 $*"
echo "bench args (trace): $TRACE_ARGS" > $OUT/command.txt
echo "bench args (pmc):   $PMC_ARGS" >> $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $TRACE_ARGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
echo "trace done"
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_128B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE"; do
  NAME=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$NAME -- python3 bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_$NAME.err || echo "pmc pass $PMC failed" >> $OUT/errors.txt
  echo "pmc pass $NAME done"
done
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
