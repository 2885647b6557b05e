#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 700 python3 tools/fuzz_parity.py 420 11 > gpurun_out/fuzz_parity.log 2>&1 || { tail -20 gpurun_out/fuzz_parity.log; exit 1; }
tail -2 gpurun_out/fuzz_parity.log
timeout -k 10 400 python3 tools/fuzz_cluster.py 200 5 > gpurun_out/fuzz_cluster.log 2>&1 || { tail -20 gpurun_out/fuzz_cluster.log; exit 1; }
tail -2 gpurun_out/fuzz_cluster.log
