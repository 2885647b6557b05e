#!/bin/bash
# gpurun -- 'bash tools/placement_pmc.sh <tag>': tools/placement_roulette.py (six states alive together, fast and slow placements in one process) under
# rocprofv3 --pmc, one small counter group per pass (more than two TCC counters per pass exceed what the hardware collects at once); tools/placement_pmc.py
# then lists, per state, the mean duration and counters of its mode-2 stage launches.
TAG=${1:-pp}; OUT=gpurun_out/placement_pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  d=$OUT/pmc_$(echo $grp | tr ' ' '_' | cut -c1-40)
  echo "== pass: $grp"
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 tools/placement_roulette.py 6 > $d.out 2> $d.err || { echo "pass failed"; tail -2 $d.err | cut -c1-200; continue; }
  grep "^state" $d.out | head -6
  python3 tools/placement_pmc.py $d
done
