timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_partition_gloo.py -q -m gpu -k "adjoint or stream_ordered or forward_euler_on or ipc or two_ranks" > gpurun_out/r02_adj_tests.log 2>&1; tail -4 gpurun_out/r02_adj_tests.log | cut -c1-300
timeout -k 10 300 python tools/adjoint_timing.py 320 60 2 2>&1 | tail -3
timeout -k 10 300 python tools/rank_timing.py 8 0 2>&1 | tail -1
