#!/usr/bin/env python3
"""GPU box helper: what the boundary costs when the caller's host arrays cross PCIe -- upload of the prognostic state,
one RK4 step, download (config 4).  The state normally stays resident; this is the PCIe-inclusive figure DESIGN.md quotes."""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

mesh = mg.icosahedral_mesh(320); K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True)
nbytes = (u.size + h.size + ssh.size) * 8
for rep in range(3):
    b.synchronize(); t0 = time.perf_counter()
    Prog.normalVelocity[-1].set(u); Prog.layerThickness[-1].set(h); Prog.ssh[-1].set(ssh)
    b.synchronize(); t1 = time.perf_counter()
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    b.synchronize(); t2 = time.perf_counter()
    uu, hh, ss = Prog.normalVelocity[-1].get(), Prog.layerThickness[-1].get(), Prog.ssh[-1].get()
    t3 = time.perf_counter()
    print(f"upload {1e3 * (t1 - t0):.1f} ms ({nbytes / (t1 - t0) / 1e9:.1f} GB/s incl. renumbering), step {1e3 * (t2 - t1):.2f} ms, "
          f"download {1e3 * (t3 - t2):.1f} ms ({nbytes / (t3 - t2) / 1e9:.1f} GB/s); round trip per step: "
          f"{mesh.nCells * K / (t3 - t0) / 1e6:.0f} M cell-updates/s")
