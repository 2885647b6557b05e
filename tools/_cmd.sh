#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
timeout -k 10 1000 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/gpu_tests_$r.log 2>&1 || { tail -40 gpurun_out/gpu_tests_$r.log; exit 1; }
tail -1 gpurun_out/gpu_tests_$r.log
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
