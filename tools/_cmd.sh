#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "nonlinear or del2" -x > gpurun_out/nl_tests.log 2>&1 || { tail -40 gpurun_out/nl_tests.log; exit 1; }
tail -2 gpurun_out/nl_tests.log
rm -rf gpurun_out/nlprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/nlprof -- python3 tools/nonlinear_timing.py 320 60 > gpurun_out/nlprof.log 2>&1
grep "ms per" gpurun_out/nlprof.log
f=$(ls -t gpurun_out/nlprof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(r["Name"][:70].ljust(70), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
PY
