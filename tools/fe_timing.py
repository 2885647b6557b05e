#!/usr/bin/env python3
"""The reference's live integrator (reference_compat Forward Euler) on config 4, piece by piece (HIP events):
whole step with the stale layerThicknessEdge gathered (stage-kernel mode 4) and formed from the previous level (mode 6),
and the step without the stale flag (mode 5).  Run under rocprofv3 --kernel-trace --stats for the per-kernel split.

    python3 tools/fe_timing.py [workload=config4_1M_x60]
"""
import datetime as dt
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
sys.path.insert(0, ROOT)
import bench                                  # noqa: E402  (workload table)
import moka_hip as mk                         # noqa: E402
from moka_hip import lib as L                 # noqa: E402
from moka_hip import meshgen as mg            # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config4_1M_x60"
m, K, sbytes, stretch = (tuple(bench.WORKLOADS[wl]) + (8, 1.0))[:4]
mesh = mg.icosahedral_mesh(m, stretch=stretch)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
b = mk.MokaHIP(0)
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, b, multilayer=True, state_bytes=sbytes)
out = {"workload": wl, "nCells": mesh.nCells, "K": K}
n = 20
# (name, key 2: stale thickness from the previous level, flags, key 3: vertex pass inside the launches, key 4: lean steps)
for name, key2, flags, key3, key4 in (("all_arrays_stored_mode4_gathered_hEdge", 0, 3, 1, 0),
                                      ("all_arrays_stored_mode4_vertex_pass_as_its_own_launch", 0, 3, 0, 0),
                                      ("all_arrays_stored_mode6_from_previous_level", 1, 3, 1, 0),
                                      ("all_arrays_stored_mode5_fresh", 1, 2, 1, 0),
                                      ("lean_mode6_reference_compat", 1, 3, 1, 1),
                                      ("lean_mode6_vertex_pass_as_its_own_launch", 1, 3, 0, 1),
                                      ("lean_mode5_fresh_accumulating_vorticity", 1, 2, 1, 1),
                                      ("lean_mode5_flags0", 1, 0, 1, 1)):
    L.check(L.lib().moka_set_tuning(2, key2))
    L.check(L.lib().moka_set_tuning(3, key3))
    L.check(L.lib().moka_set_tuning(4, key4))
    for _ in range(3):
        mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
    b.synchronize()
    b.marks_reset(); b.mark()
    for _ in range(n):
        mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        b.mark()
    ms = sorted(b.marks_read())
    out[name] = {"median_ms": ms[len(ms) // 2], "min_ms": ms[0], "max_ms": ms[-1], "path": L.lib().moka_last_fe_path(Prog._state._h),
                 "arrays_pending": L.lib().moka_fe_lazy_pending(Prog._state._h)}
L.check(L.lib().moka_set_tuning(2, 1))
L.check(L.lib().moka_set_tuning(3, 1))
L.check(L.lib().moka_set_tuning(4, 1))
print(json.dumps(out))
