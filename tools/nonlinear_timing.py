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
pc = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # patch_cells (0 = the plan's default)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True, patch_cells=pc)
print('patch_cells', Setup.mesh.info().get('patch_cells'), flush=True)
from moka_hip import lib as L              # noqa: E402
shapes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 0, 2, 3]
for nl, shape in [(False, 0)] + [(True, sh) for sh in shapes]:
    L.check(L.lib().moka_set_tuning(5, shape))      # launch shape of the nonlinear stage kernel (include/moka_hip.h)
    mk.set_nonlinear(Prog, nl)
    mk.run_steps(Prog, mk.RungeKutta4, dts, 3)
    b.synchronize(); t0 = time.perf_counter()
    mk.run_steps(Prog, mk.RungeKutta4, dts, 10)
    b.synchronize(); t1 = time.perf_counter()
    print(f"{'nonlinear shape ' + str(shape) if nl else 'linear           '}: {mesh.nCells} cells x {K}: {1e3 * (t1 - t0) / 10:.2f} ms per RK4 step", flush=True)
    if nl and shape in (0, 2, 3):          # the 13-stream form (moka_set_tuning key 7; k_stage_nl5 only), interleaved with the line above
        L.check(L.lib().moka_set_tuning(7, 1))
        mk.run_steps(Prog, mk.RungeKutta4, dts, 3)
        b.synchronize(); t0 = time.perf_counter()
        mk.run_steps(Prog, mk.RungeKutta4, dts, 12)
        b.synchronize(); t1 = time.perf_counter()
        L.check(L.lib().moka_set_tuning(7, 0))
        print(f"nonlinear shape {shape}, 13 streams: {1e3 * (t1 - t0) / 12:.2f} ms per RK4 step", flush=True)
