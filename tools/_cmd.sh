#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "fp32" -x > gpurun_out/f32_tests.log 2>&1 || { tail -40 gpurun_out/f32_tests.log; exit 1; }
tail -3 gpurun_out/f32_tests.log
