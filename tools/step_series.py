#!/usr/bin/env python3
"""GPU box helper: per-step times of config 4's RK4 step (one HIP event per step) behind host pauses of 0 / 0.05 / 0.3 / 1 s --
how many steps the device needs to be back in its steady state (power-limited clocks) after the host left it idle.
    python3 tools/step_series.py"""
import datetime as dt, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd")); sys.path.insert(0, ROOT)
import moka_hip as mk
from moka_hip import meshgen as mg
mesh = mg.icosahedral_mesh(320)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, 60)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True, placement_tries=12)
mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
def series(n, pause):
    b.synchronize(); time.sleep(pause)
    b.marks_reset(); b.mark()
    for _ in range(n):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4); b.mark()
    return b.marks_read()
for pause in (0.0, 0.05, 0.3, 1.0, 0.0):
    ms = series(60, pause)
    print(f"pause {pause:4.2f} s: first 5 {' '.join(f'{x:.3f}' for x in ms[:5])} | steps 6-25 median {sorted(ms[5:25])[10]:.3f} | steps 26-60 median {sorted(ms[25:])[17]:.3f}", flush=True)
