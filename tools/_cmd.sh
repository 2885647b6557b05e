#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
for o in 0 1 0 1 0 1; do
  timeout -k 10 300 python3 tools/rank_timing.py 8 0 $o 2>&1 | grep "world" | sed 's/owned cells.*ms per step://' || exit 1
done
