#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_r02f
bash tools/profile.sh r02f > gpurun_out/profile_r02f.log 2>&1 || { tail -20 gpurun_out/profile_r02f.log; exit 1; }
tail -5 gpurun_out/profile_r02f.log
