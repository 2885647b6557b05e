#!/usr/bin/env python3
"""GPU box helper: time taped RK4 / FE steps and the reverse sweep on a BASELINE-size sphere.
   python tools/adjoint_timing.py [m=320] [K=60] [steps=3]"""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L               # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

m, K, n = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 320), (2, 60), (3, 3)))
mesh = mg.icosahedral_mesh(m)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True)
for method, name in ((mk.RungeKutta4, "RK4"), (mk.ForwardEuler, "FE")):
    tape = mk.AdjointTape(Prog, n)
    for rep in range(2):                   # first pass warms up (lazy allocations)
        b.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            tape.step(dts, 0, method=method if method is mk.RungeKutta4 else None)
        b.synchronize(); t1 = time.perf_counter()
        L.check(L.lib().moka_adjoint_seed_sum_sq_ssh(tape._h), b._h)
        L.check(L.lib().moka_adjoint_sweep(tape._h), b._h)
        b.synchronize(); t2 = time.perf_counter()
    print(f"{name}: {mesh.nCells} cells x {K}: taped forward {1e3 * (t1 - t0) / n:.2f} ms/step, "
          f"seed + reverse sweep {1e3 * (t2 - t1) / n:.2f} ms/step (n = {n})")
    tape.close()
