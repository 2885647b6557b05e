#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "nonlinear or del2" -x > gpurun_out/nl_tests.log 2>&1 || { tail -40 gpurun_out/nl_tests.log; exit 1; }
tail -3 gpurun_out/nl_tests.log
timeout -k 10 300 python3 tools/nonlinear_timing.py 320 60 2>&1 | grep "ms per"
