#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/ -q -m gpu -k "adjoint or tape or reverse or gradient" -x > gpurun_out/adj_tests.log 2>&1 || { tail -40 gpurun_out/adj_tests.log; exit 1; }
tail -3 gpurun_out/adj_tests.log
rm -rf gpurun_out/adjprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/adjprof -- python3 tools/adjoint_timing.py 320 60 2 > gpurun_out/adjprof.log 2>&1
grep "ms/step" gpurun_out/adjprof.log
f=$(ls -t gpurun_out/adjprof/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r["Name"][:90].ljust(90), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
PY
