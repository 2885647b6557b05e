#!/usr/bin/env python3
"""GPU box helper: randomized parity sweep of the tuned kernels against the oracle (bit for bit).
Random meshes (icosahedral with edge flips, planar periodic), even K in 34..64 (Float64) or K % 4 == 0 (fp32 storage),
patch sizes, level masks, Forward-Euler flags; RK4 + FE (both storage types) + tendencies, reverse mode of FE and RK4 runs,
the three forms of the nonlinear kernels.   python tools/fuzz_parity.py [seconds=150] [seed=0]"""
import datetime as dt
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np                         # noqa: E402
import moka_hip as mk                      # noqa: E402
import oracle as orc                       # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

budget, seed = (float(sys.argv[1]) if len(sys.argv) > 1 else 150.0), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = np.random.default_rng(seed)
b = mk.MokaHIP(0)
t0, n = time.time(), 0
stats = {"f32": 0, "f32_fe": 0, "fe_tuned": 0, "fe_generic": 0, "masked": 0, "cells_max": 0, "adjoint": 0, "adjoint_rk4": 0, "nonlinear": 0,
         "rk4_13_streams": 0, "placement": 0, "rows": 0}
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    if kind == 0:
        mesh = mg.icosahedral_mesh(int(rng.integers(3, 14)), flips=int(rng.integers(0, 10)), seed=int(rng.integers(0, 1000)))
        dtv = 20.0
    elif kind == 1:
        mesh = mg.icosahedral_mesh(int(rng.integers(3, 18)))
        dtv = 20.0
    else:
        mesh = mg.planar_hex_mesh(2 * int(rng.integers(1, 9)), 2 * int(rng.integers(1, 9)), 1000.0, f0=1e-4)
        dtv = 2.0
    f32 = bool(rng.integers(0, 4) == 0)
    K = int(4 * rng.integers(1, 33)) if f32 else int(2 * rng.integers(17, 33))
    P = int(rng.choice([0, 0, 5, 8, 12, 16, 20, 24]))
    r = np.random.default_rng(int(rng.integers(0, 1 << 30)))
    rest = np.full((mesh.nCells, K), 1000.0 / K) + r.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + r.uniform(-1, 1, (mesh.nCells, K))
    u = r.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    mlt = np.full(mesh.nEdges, K, dtype=np.int32)
    if rng.integers(0, 2):
        sel = r.random(mesh.nEdges) < 0.3
        mlt[sel] = r.integers(0, K + 1, int(sel.sum()))
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
    vm.maxLevelEdge.Top[:] = mlt
    try:
        M = mk.Mesh(hm, vm, backend=b, patch_cells=P, state_bytes=4 if f32 else 8)
        Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    except mk.MokaError as exc:            # e.g. fp32 shapes the library refuses
        continue
    tag = f"case {n}: kind {kind} cells {mesh.nCells} K {K} P {P} f32 {f32} masks {int((mlt < K).sum())}"
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
    st = orc.OracleState(om, ssh, u, h, mixed=f32)
    Diag, Tend = mk.DiagnosticVars(None, M, Prog._state), mk.TendencyVars(None, M, Prog._state)
    for _ in range(2):
        L.check(L.lib().moka_step_rk4(Prog._state._h, dtv), b._h)
        st.step_rk4(dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]), tag + " rk4 u"
    assert np.array_equal(Prog.layerThickness[-1].get(), st.h[1]), tag + " rk4 h"
    assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1]), tag + " rk4 ssh"
    if f32:     # the Forward-Euler step of an fp32-storage state: nothing carried over right after RK4, then random flags
        flags = int(rng.choice([0, 1, 2, 3]))
        for fl in (0, flags, flags):
            L.check(L.lib().moka_step_fe(Prog._state._h, dtv, fl), b._h)
            st.step_fe(dtv, fl)
        for name, got, exp in (("u", Prog.normalVelocity[-1].get(), st.u[1]), ("h", Prog.layerThickness[-1].get(), st.h[1]),
                               ("ssh", Prog.ssh[-1].get(), st.ssh[1]), ("hEdge", Diag.layerThicknessEdge.get(), st.hEdge),
                               ("F", Diag.thicknessFlux.get(), st.F), ("div", Diag.velocityDivCell.get(), st.div),
                               ("vort", Diag.relativeVorticity.get(), st.vort), ("tendU", Tend.tendNormalVelocity.get(), st.tendU),
                               ("tendH", Tend.tendLayerThickness.get(), st.tendH)):
            assert np.array_equal(got, exp), f"{tag} f32 fe flags {flags} {name}"
        stats["f32_fe"] += 1
    if not f32:
        flags = int(rng.choice([0, 1, 2, 3]))
        for _ in range(2):
            L.check(L.lib().moka_step_fe(Prog._state._h, dtv, flags), b._h)
            st.step_fe(dtv, flags)
        for name, got, exp in (("u", Prog.normalVelocity[-1].get(), st.u[1]), ("h", Prog.layerThickness[-1].get(), st.h[1]),
                               ("ssh", Prog.ssh[-1].get(), st.ssh[1]), ("hEdge", Diag.layerThicknessEdge.get(), st.hEdge),
                               ("F", Diag.thicknessFlux.get(), st.F), ("div", Diag.velocityDivCell.get(), st.div),
                               ("vort", Diag.relativeVorticity.get(), st.vort), ("tendU", Tend.tendNormalVelocity.get(), st.tendU),
                               ("tendH", Tend.tendLayerThickness.get(), st.tendH)):
            assert np.array_equal(got, exp), f"{tag} fe flags {flags} path {L.lib().moka_last_fe_path(Prog._state._h)} {name}"
        stats["fe_tuned" if L.lib().moka_last_fe_path(Prog._state._h) >= 1 else "fe_generic"] += 1
        if n % 4 == 1 and not (mlt < K).any():                      # reverse mode of two more Forward-Euler steps
            Prog2 = mk.PrognosticVars(st.ssh[1], st.u[1], st.h[1], 2, M)
            tape = mk.AdjointTape(Prog2, 2)
            st2 = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
            adj = orc.OracleAdjoint(st2)
            fl = int(rng.choice([0, 1, 2, 3]))
            for _ in range(2):
                tape.step(np.array([dtv]), fl)
                adj.step_fe(dtv, fl)
            g = tape.gradient()
            gS, gU, gH, gE = adj.gradient_sum_sq_ssh()
            assert np.array_equal(g["ssh"], gS) and np.array_equal(g["normalVelocity"], gU) and \
                np.array_equal(g["layerThickness"], gH), tag + f" adjoint flags {fl}"
            tape.close(); Prog2._state.close()
            stats["adjoint"] += 1
        if n % 4 == 2 and not (mlt < K).any():                      # reverse mode of two RK4 steps (chunk kernels, fused cell pass)
            Prog2 = mk.PrognosticVars(st.ssh[1], st.u[1], st.h[1], 2, M)
            tape = mk.AdjointTape(Prog2, 2)
            st2 = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
            adj = orc.OracleAdjointRK4(st2)
            for _ in range(2):
                tape.step(dtv, method=mk.RungeKutta4)
                adj.step_rk4(dtv)
            g = tape.gradient()
            gU, gH = adj.gradient_sum_sq_ssh()
            assert np.array_equal(g["normalVelocity"], gU) and np.array_equal(g["layerThickness"], gH), tag + " adjoint rk4"
            tape.close(); Prog2._state.close()
            stats["adjoint_rk4"] += 1
        if n % 4 == 3 and mesh.kiteAreasOnVertex is not None:      # nonlinear terms + Del2, one RK4 step, any of the three kernel forms
            visc = float(rng.choice([0.0, 0.01 * float(mesh.dcEdge.min()) ** 2 / dtv]))
            form = int(rng.choice([0, 0, 0, 4, 3]))
            b.set_kernel_variant(form)
            # launch shape of the patch form (moka_set_tuning key 5) and, at random, few resident vertex rows (key 6: the path of the
            # patches that list more vertices than the LDS holds)
            shape, cap = int(rng.choice([0, 0, 1, 2, 3])), int(rng.choice([0, 0, 24, 48]))
            L.check(L.lib().moka_set_tuning(5, shape)); L.check(L.lib().moka_set_tuning(6, cap))
            Prog3 = mk.PrognosticVars(st.ssh[1], st.u[1], st.h[1], 2, M)
            mk.set_nonlinear(Prog3, True, visc_del2=visc)
            nl = orc.OracleNonlinear(om, visc_del2=visc)
            st3 = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
            # at random with moka_set_tuning key 7: the 13-stream form where the state qualifies (moka_state_rk4_streams tells: the
            # default patch kernel on hexagon-dominated meshes, even 34 <= K <= 64)
            s13 = bool(rng.integers(0, 2))
            L.check(L.lib().moka_set_tuning(7, int(s13)))
            streams = L.lib().moka_state_rk4_streams(Prog3._state._h)
            assert streams == 16 or s13, tag
            L.check(L.lib().moka_step_rk4(Prog3._state._h, dtv), b._h)
            b.set_kernel_variant(0)
            L.check(L.lib().moka_set_tuning(5, 0)); L.check(L.lib().moka_set_tuning(6, 0)); L.check(L.lib().moka_set_tuning(7, 0))
            (nl.step_rk4_s13 if streams == 13 else nl.step_rk4)(st3, dtv)
            assert np.array_equal(Prog3.normalVelocity[-1].get(), st3.u[1]) and np.array_equal(Prog3.layerThickness[-1].get(), st3.h[1]), \
                tag + f" nonlinear visc {visc} form {form} shape {shape} cap {cap} streams {streams}"
            stats["nonlinear_13"] = stats.get("nonlinear_13", 0) + int(streams == 13)
            Prog3._state.close()
            stats["nonlinear"] += 1
    # round 4: the per-array placement search leaves every array as it is (mid-run, whatever is lazily pending), sampled rows
    # equal the whole-field download, and -- Float64 -- the opt-in 13-stream RK4 form equals its oracle twin
    if n % 3 == 0:
        before = {k: f.get() for k, f in (("u", Prog.normalVelocity[-1]), ("h", Prog.layerThickness[-1]), ("s", Prog.ssh[-1]),
                                         ("u0", Prog.normalVelocity[0]), ("h0", Prog.layerThickness[0]), ("s0", Prog.ssh[0]))}
        rep = Prog._state.optimize_placement(int(rng.integers(1, 6)))
        assert rep["ms_after"] <= rep["ms_before"], tag + " placement"
        for k, f in (("u", Prog.normalVelocity[-1]), ("h", Prog.layerThickness[-1]), ("s", Prog.ssh[-1]),
                     ("u0", Prog.normalVelocity[0]), ("h0", Prog.layerThickness[0]), ("s0", Prog.ssh[0])):
            assert np.array_equal(f.get(), before[k]), tag + f" placement changed {k}"
        ids = r.integers(0, mesh.nEdges, 50)
        assert np.array_equal(Prog.normalVelocity[-1].rows(ids), before["u"][ids]), tag + " rows"
        stats["placement"] += 1; stats["rows"] += 1
    if not f32 and n % 3 == 1:
        L.check(L.lib().moka_set_tuning(7, 1))
        try:
            for _ in range(2):
                L.check(L.lib().moka_step_rk4(Prog._state._h, dtv), b._h)
                st.step_rk4_s13(dtv)
        finally:
            L.check(L.lib().moka_set_tuning(7, 0))
        assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1]) and \
            np.array_equal(Prog.ssh[-1].get(), st.ssh[1]) and np.array_equal(Prog.normalVelocity[0].get(), st.u[0]), tag + " 13-stream rk4"
        stats["rk4_13_streams"] += 1
    stats["f32"] += int(f32); stats["masked"] += int((mlt < K).any()); stats["cells_max"] = max(stats["cells_max"], mesh.nCells)
    Prog._state.close(); M.close()
    n += 1
    if n % 10 == 0:
        print(f"{n} cases, {time.time() - t0:.0f}s", flush=True)
print(f"fuzz_parity: {n} random cases bit-identical to the oracle (seed {seed}); {stats}")
