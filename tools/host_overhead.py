import sys, time, os
sys.path.insert(0, "mpas-ocean.jl_amd")
import numpy as np, torch
import moka_hip as mk
from moka_hip import meshgen as mg, parallel as mp
mesh = mg.icosahedral_mesh(32); K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
model = mp.DistributedModel(mesh, ssh, u, h, rest, dts, b, 0, 1, transport="gloo")
for _ in range(20): model.step_rk4()
b.synchronize(); t0 = time.perf_counter()
N = 300
for _ in range(N): model.step_rk4()
b.synchronize(); t1 = time.perf_counter()
print("dist path world=1: %.1f us per step (10242 cells x 60)" % ((t1 - t0) / N * 1e6))
import datetime as dt
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True)
for _ in range(20): mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
b.synchronize(); t0 = time.perf_counter()
for _ in range(N): mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
b.synchronize(); t1 = time.perf_counter()
print("plain moka_step_rk4: %.1f us per step" % ((t1 - t0) / N * 1e6))
