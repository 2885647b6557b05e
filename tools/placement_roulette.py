#!/usr/bin/env python3
"""GPU box helper: ONE process, several states of config 4 alive at the same time (so each sits in different memory): ms per RK4 step of each.
If some are fast and some slow, the speed belongs to the memory a state was given, and choosing among a few placements at set-up cures the
slow sessions (profiles/r03_variants.txt).   python tools/placement_roulette.py [nstates=5]"""
import datetime as dt
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
lib = L.lib()
hm = mk.HorzMesh(mesh)
vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
M = mk.Mesh(hm, vm, backend=b)
states = []


def time_state(st):
    for _ in range(3):
        L.check(lib.moka_step_rk4(st._h, dts), b._h)
    b.stage_timing(True)
    b.marks_reset(); b.mark()
    for _ in range(8):
        L.check(lib.moka_step_rk4(st._h, dts), b._h)
        b.mark()
    ms = sorted(b.marks_read())
    st4, _ = b.stage_timing_read()
    b.stage_timing(False)
    return ms[len(ms) // 2], st4


for i in range(n):
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    states.append(Prog)
    med, st4 = time_state(Prog._state)
    print(f"state {i}: {med:.3f} ms/step  stages " + " ".join(f"{x:.3f}" for x in st4), flush=True)
print("again, all still alive:")
for i, Prog in enumerate(states):
    med, st4 = time_state(Prog._state)
    print(f"state {i}: {med:.3f} ms/step  stages " + " ".join(f"{x:.3f}" for x in st4), flush=True)
