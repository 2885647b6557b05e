#!/usr/bin/env python3
"""GPU box helper: where do the prognostic arrays of a state sit (device addresses) before and after moka_state_optimize_placement,
and which re-rolls did it keep?  Looks for what distinguishes a kept allocation from a dropped one.
    python3 tools/placement_addresses.py [rounds=3]"""
import ctypes as C
import datetime as dt
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import bench                               # noqa: E402
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m, K = bench.WORKLOADS["config4_1M_x60"][:2]
mesh = mg.icosahedral_mesh(m)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
backend = mk.MokaHIP(0)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
names = {(f, lv): f"{('prev', 'cur', 'rk1', 'rk2')[lv]}.{('ssh', 'normalVelocity', 'layerThickness')[f]}" for f in range(3) for lv in range(4)}


def addresses(st):
    out = {}
    for (f, lv), n in names.items():
        a = C.c_uint64()
        L.check(L.lib().moka_state_array_address(st._h, f, lv, C.byref(a)), backend._h)
        out[n] = a.value
    return out


for r in range(rounds):
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, backend, multilayer=True, placement_tries=1)
    L.check(L.lib().moka_step_rk4(Prog._state._h, 0.0), backend._h)          # allocates the provisional states
    a0 = addresses(Prog._state)
    rep = Prog._state.optimize_placement(24)
    a1 = addresses(Prog._state)
    print(f"== state {r}: {rep['ms_before']:.3f} -> {rep['ms_after']:.3f} ms; kept: {[(t['field'], round(t['ms_old'] - t['ms_new'], 3)) for t in rep['trials'] if t['kept']]}")
    base = min(a0.values())
    for n in sorted(a0, key=lambda k: a0[k]):
        moved = "" if a0[n] == a1[n] else f"  -> {a1[n]:#014x} (offset {a1[n] - base:+d}, mod 4 GiB {a1[n] % (1 << 32):#x}, mod 64 MiB {a1[n] % (1 << 26):#x})"
        print(f"   {n:22s} {a0[n]:#014x}  offset {a0[n] - base:12d}  mod 4 GiB {a0[n] % (1 << 32):#011x}  mod 64 MiB {a0[n] % (1 << 26):#09x}{moved}")
    Prog._state.close()
    Setup.mesh.close()
