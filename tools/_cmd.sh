timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "adjoint" > gpurun_out/r02_adj_tests2.log 2>&1; tail -3 gpurun_out/r02_adj_tests2.log | cut -c1-300
timeout -k 10 300 python tools/adjoint_timing.py 320 60 2 2>&1 | tail -2
