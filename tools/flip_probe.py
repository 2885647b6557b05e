#!/usr/bin/env python3
"""GPU box helper: ONE state, many short batches of RK4 steps with pauses of different length in between: does the per-step time flip between
batches (the 6.8 / 7.5 ms states of profiles/r03_variants.txt) without anything being re-allocated?   python tools/flip_probe.py"""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
lib = L.lib()
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True)
st = Prog._state


def batch(n):
    b.marks_reset(); b.mark()
    for _ in range(n):
        L.check(lib.moka_step_rk4(st._h, dts), b._h)
        b.mark()
    ms = b.marks_read()
    return ms


for pause in (0.0, 0.005, 0.05, 0.5, 2.0, 0.0):
    out = []
    for i in range(10):
        if pause:
            b.synchronize(); time.sleep(pause)
        ms = batch(12)
        out.append(sorted(ms)[len(ms) // 2])
    print(f"pause {pause:5.3f} s, medians of 10 batches of 12 steps: " + " ".join(f"{x:.2f}" for x in out), flush=True)
ms = batch(400)
print("400 steps back to back, every 20th: " + " ".join(f"{x:.2f}" for x in ms[::20]), flush=True)
