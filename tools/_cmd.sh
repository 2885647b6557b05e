#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_r02g gpurun_out/prof_extras
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail gpurun_out/bench_default.err; exit 1; }
bash tools/profile.sh r02g > gpurun_out/profile_r02g.log 2>&1 || { tail -20 gpurun_out/profile_r02g.log; exit 1; }
grep -n "CONSISTENT\|roofline frac" gpurun_out/prof_r02g/summary.txt
bash tools/profile_extras.sh > gpurun_out/extras.log 2>&1
grep -h "ms/step\|ms per" gpurun_out/prof_extras/*.txt
timeout -k 10 400 python3 bench.py --workload config5_3.7M_x80_f32 --no-cpu --steps 10 --warmup 3 > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
for w in 8 2; do timeout -k 10 300 python3 tools/rank_timing.py $w 0 2>&1 | grep "world" | sed 's/owned cells.*ms per step://'; done
