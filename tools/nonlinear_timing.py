#!/usr/bin/env python3
"""GPU box helper: ms per RK4 step with the optional nonlinear terms (generic kernels) vs the reference's linear terms."""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

m, K = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 320), (2, 60)))
mesh = mg.icosahedral_mesh(m)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True)
for nl in (False, True):
    mk.set_nonlinear(Prog, nl)
    mk.run_steps(Prog, mk.RungeKutta4, dts, 3)
    b.synchronize(); t0 = time.perf_counter()
    mk.run_steps(Prog, mk.RungeKutta4, dts, 10)
    b.synchronize(); t1 = time.perf_counter()
    print(f"{'nonlinear' if nl else 'linear   '}: {mesh.nCells} cells x {K}: {1e3 * (t1 - t0) / 10:.2f} ms per RK4 step")
