timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fp32 or config5 or stream_ordered" > gpurun_out/r02_f32_tests.log 2>&1; tail -3 gpurun_out/r02_f32_tests.log | cut -c1-300
timeout -k 10 300 python3 bench.py --workload config5_3.7M_x80_f32 --no-cpu > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5.err; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_c5.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], [round(p["ms"],3) for p in d["roofline"]["per_stage"]], d["tendency_kernel"], d["roofline"]["frac"])
PY
timeout -k 10 200 python tools/fuzz_cluster.py 60 3 2>&1 | tail -2
timeout -k 10 200 python tools/fuzz_parity.py 60 5 2>&1 | tail -2
