#!/usr/bin/env python3
"""GPU box helper: repeat one reverse-mode Forward-Euler case many times and report any run that differs from the oracle
(looks for timing-dependent behaviour).   python tools/stress_adjoint.py [reps=300] [m=13] [K=34] [P=12] [flags=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np                         # noqa: E402
import moka_hip as mk                      # noqa: E402
import oracle as orc                       # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

reps, m, K, P, flags = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 300), (2, 13), (3, 34), (4, 12), (5, 0)))
mesh = mg.icosahedral_mesh(m)
r = np.random.default_rng(5)
rest = np.full((mesh.nCells, K), 1000.0 / K) + r.uniform(0, 0.1, (mesh.nCells, K))
h = rest + r.uniform(-1, 1, (mesh.nCells, K)); u = r.uniform(-1, 1, (mesh.nEdges, K)); ssh = h.sum(1) - rest.sum(1)
b = mk.MokaHIP(0)
hm = mk.HorzMesh(mesh); vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
M = mk.Mesh(hm, vm, backend=b, patch_cells=P)
om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
st = orc.OracleState(om, ssh, u, h); adj = orc.OracleAdjoint(st)
for _ in range(2):
    adj.step_fe(20.0, flags)
gS, gU, gH, gE = adj.gradient_sum_sq_ssh()
bad = 0
for i in range(reps):
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    tape = mk.AdjointTape(Prog, 2)
    for _ in range(2):
        tape.step(np.array([20.0]), flags)
    fwd_ok = np.array_equal(Prog.ssh[-1].get(), st.ssh[1]) and np.array_equal(Prog.normalVelocity[-1].get(), st.u[1])
    g = tape.gradient()
    d = {k: int((g[k] != e).sum()) for k, e in (("ssh", gS), ("normalVelocity", gU), ("layerThickness", gH))}
    if any(d.values()) or not fwd_ok:
        bad += 1
        w = np.argwhere(g["normalVelocity"] != gU)
        print(f"rep {i}: forward ok {fwd_ok}, differing entries {d}, first u mismatch at {w[:3].tolist()}", flush=True)
    tape.close(); Prog._state.close()
print(f"stress_adjoint: {bad} of {reps} repetitions differed (cells {mesh.nCells}, K {K}, P {P}, flags {flags})")
