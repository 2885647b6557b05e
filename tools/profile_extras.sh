#!/bin/bash
# GPU box helper: rocprofv3 --kernel-trace --stats of the paths bench.py's default run does not cover
# (config 5 fp32 storage, reverse mode, optional nonlinear terms); per-kernel averages into gpurun_out/prof_extras/.
set -u
OUT=gpurun_out/prof_extras
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 bench.py --workload config5_3.7M_x80_f32 --steps 3 --warmup 1 --no-cpu --tend-iters 3 > $OUT/c5.json 2> $OUT/c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adj -- python3 tools/adjoint_timing.py 320 60 2 > $OUT/adj.txt 2> $OUT/adj.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nl -- python3 tools/nonlinear_timing.py 320 60 > $OUT/nl.txt 2> $OUT/nl.err
for t in c5 adj nl; do
  echo "== $t: rocprofv3 --kernel-trace --stats =="
  f=$(ls $OUT/$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -14 "$f"
done > $OUT/summary.txt
cat $OUT/summary.txt
