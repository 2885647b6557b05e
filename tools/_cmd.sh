export MOKA_HIP_LIB=$PWD/mpas-ocean.jl_amd/libmoka_hip_exp.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rk4_bitwise" > gpurun_out/r02_ptile_tests.log 2>&1; tail -3 gpurun_out/r02_ptile_tests.log | cut -c1-300
for v in 14 0 14; do for P in 14; do
timeout -k 10 200 python3 bench.py --no-cpu --steps 20 --warmup 5 --variant $v --patch-cells $P > gpurun_out/ptile_v${v}_P${P}.json 2> gpurun_out/ptile_v${v}_P${P}.err && python3 - gpurun_out/ptile_v${v}_P${P}.json $v $P <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
ps = d["roofline"].get("per_stage", [])
print(f"v{sys.argv[2]} P{sys.argv[3]}: {d['ms_per_step']:.3f} ms/step  stages " + " ".join(f"{p['ms']:.3f}" for p in ps) + f"  tendency {d['tendency_kernel']['avg_launch_ms']:.3f}  FE {d['forward_euler_compat']['ms_per_step']:.3f}")
PY
done; done
