#!/usr/bin/env python3
"""A/B of the lean Forward-Euler launch forms, interleaved on one box: the general instances (stage-kernel modes 5 / 6, outputs tested
at run time; moka_set_tuning key 9 = 0) against the lean instances (modes 10 / 11; key 9 = 1), fp32 storage also as 512-thread
workgroups (key 1 mask bits 10 / 11).     python3 tools/fe_lean_ab.py [workload=config5_3.7M_x80_f32] [rounds=3]"""
import datetime as dt
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
sys.path.insert(0, ROOT)
import bench                                  # noqa: E402  (workload table)
import moka_hip as mk                         # noqa: E402
from moka_hip import lib as L                 # noqa: E402
from moka_hip import meshgen as mg            # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config5_3.7M_x80_f32"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m, K, sbytes, stretch = (tuple(bench.WORKLOADS[wl]) + (8, 1.0))[:4]
mesh = mg.icosahedral_mesh(m, stretch=stretch)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True, state_bytes=sbytes)
lib = L.lib()
forms = [("general instances (key 9 = 0)", 0, 1, 1), ("lean instances (key 9 = 1)", 1, 1, 1), ("lean instances, vertex pass as its own launch (key 3 = 0)", 1, 1, 0)]
if sbytes == 4:
    forms.append(("lean instances, 512 threads", 1, 1 | (1 << 10) | (1 << 11), 1))
for flags, what in ((3, "reference_compat (stale thickness, accumulating vorticity)"), (0, "flags 0")):
    print(f"{wl}: lean Forward-Euler step, {what}", flush=True)
    for r in range(rounds):
        line = []
        for name, key9, mask, key3 in forms:
            L.check(lib.moka_set_tuning(9, key9)); L.check(lib.moka_set_tuning(1, mask)); L.check(lib.moka_set_tuning(3, key3))
            for _ in range(3):
                mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
            b.synchronize(); b.marks_reset(); b.mark()
            for _ in range(12):
                mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
                b.mark()
            ms = sorted(b.marks_read())
            line.append(f"{name}: {ms[len(ms) // 2]:.3f} ms")
        print("   " + "   ".join(line), flush=True)
L.check(lib.moka_set_tuning(9, 1)); L.check(lib.moka_set_tuning(1, 1)); L.check(lib.moka_set_tuning(3, 1))
