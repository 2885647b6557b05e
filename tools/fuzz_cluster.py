#!/usr/bin/env python3
"""GPU box helper: randomized sweep of the distributed steps (all ranks of a partition in one process: parallel.LocalCluster,
direct transport -- push kernels into the neighbours' fields + flag words -- or buffered transport with stream-ordered device
copies, chosen at random; RK4 steps, and on fp64 states also the reference's Forward-Euler step with random compat flags)
against the single-domain oracle, bit for bit; every third fp64 case also tapes two steps and reverses them across the ranks.
   python tools/fuzz_cluster.py [seconds=120] [seed=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np                         # noqa: E402
import oracle as orc                       # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402
from moka_hip import parallel as par       # noqa: E402

budget, seed = (float(sys.argv[1]) if len(sys.argv) > 1 else 120.0), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = np.random.default_rng(seed)
t0, n, skipped = time.time(), 0, 0
n_rev = {"rk4": 0, "fe": 0}
meshes = {}
while time.time() - t0 < budget:
    m = int(rng.integers(8, 28))
    mesh = meshes.setdefault(m, mg.icosahedral_mesh(m))
    world = int(rng.integers(2, 9))
    f32 = bool(rng.integers(0, 3) == 0)
    K = int(4 * rng.integers(9, 21)) if f32 else int(rng.choice([1, 3, 8, 34, 40, 60, 64, 70]))
    P = int(rng.choice([0, 0, 8, 12, 16]))
    r = np.random.default_rng(int(rng.integers(0, 1 << 30)))
    rest = np.full((mesh.nCells, K), 1000.0 / K) + r.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + r.uniform(-1, 1, (mesh.nCells, K))
    u = r.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    tag = f"case {n}: m {m} world {world} K {K} P {P} f32 {f32}"
    direct = bool(rng.integers(0, 4) != 0)
    fe_flags = int(rng.choice([3, 2, 1, 0])) if K > 1 else int(rng.choice([7, 0]))
    tag += f" direct {direct} fe_flags {fe_flags}"
    try:
        cl = par.LocalCluster(mesh, ssh, u, h, rest, 20.0, world, patch_cells=P, state_bytes=4 if f32 else 8, direct=direct)
    except Exception as exc:               # noqa: BLE001  shapes the library refuses (e.g. fp32 with a huge straddling patch)
        skipped += 1
        continue
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h, mixed=f32)
    cl.exchange_state()
    for _ in range(3):
        cl.step_rk4()
        ref.step_rk4(20.0)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, ref.u[1]), tag + " u"
    assert np.array_equal(gh, ref.h[1]), tag + " h"
    assert np.array_equal(gs, ref.ssh[1]), tag + " ssh"
    if not f32:
        for _ in range(2):
            cl.step_fe(fe_flags)
            ref.step_fe(20.0, fe_flags)
        gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1]), tag + " FE"
        d = cl.gather_diagnostics(mesh, K)
        for name, exp in (("hEdge", ref.hEdge), ("F", ref.F), ("div", ref.div), ("vort", ref.vort), ("tendU", ref.tendU), ("tendH", ref.tendH)):
            assert np.array_equal(d[name], exp), tag + " FE " + name
    if not f32 and (K % 2 == 0 or K == 1) and n % 3 == 0:
        # reverse mode across the ranks from the state reached: two taped steps of a random integrator, the assembled gradient
        # of sum(ssh^2) against the single-domain oracle adjoint
        st2 = orc.OracleState(om, ref.ssh[1], ref.u[1], ref.h[1])
        st2.hEdge[...] = ref.hEdge; st2.vort[...] = ref.vort
        cl.tape(2)
        if rng.integers(0, 2):
            adj = orc.OracleAdjointRK4(st2)
            for _ in range(2):
                cl.step_rk4_taped()
                adj.step_rk4(20.0)
            gU, gH = adj.gradient_sum_sq_ssh()
            gu, gh = cl.adjoint_gradient(2, mesh.nCells, mesh.nEdges, K, overlap=bool(rng.integers(0, 2)))   # stages by cell class or whole
            assert np.array_equal(gu, gU) and np.array_equal(gh, gH), tag + " reverse RK4"
            n_rev["rk4"] += 1
        else:
            adj = orc.OracleAdjoint(st2)
            for _ in range(2):
                cl.step_fe_taped(fe_flags)
                adj.step_fe(20.0, fe_flags)
            gS, gU, gH, gE = adj.gradient_sum_sq_ssh()
            g = cl.adjoint_gradient_fe(2, mesh, K)
            assert np.array_equal(g["ssh"], gS) and np.array_equal(g["normalVelocity"], np.asarray(gU).reshape(mesh.nEdges, K)) and \
                np.array_equal(g["layerThickness"], np.asarray(gH).reshape(mesh.nCells, K)) and \
                np.array_equal(g["layerThicknessEdge"], np.asarray(gE).reshape(mesh.nEdges, K)), tag + " reverse FE"
            n_rev["fe"] += 1
    cl.close()
    if not f32 and n % 5 == 4:
        # the optional nonlinear terms on the same partition (two-ring halo, whole-mesh stages), random Del2 viscosity
        visc = float(rng.choice([0.0, 0.01 * float(mesh.dcEdge.min()) ** 2 / 20.0]))
        cn = par.LocalCluster(mesh, ssh, u, h, rest, 20.0, world, patch_cells=P, direct=bool(rng.integers(0, 2)) and K % 2 == 0 and 34 <= K <= 64,
                              nonlinear=True, visc_del2=visc)
        onl, stn = (orc.OracleNonlinear(om, visc_del2=visc) if visc else orc.OracleNonlinear(om)), orc.OracleState(om, ssh, u, h)
        cn.exchange_state()
        from moka_hip import lib as L_
        parts = all(L_.lib().moka_rk4_dist_parts_available(m._halo) for m in cn.models)
        for i in range(2):
            # stage kernel over boundary / interior patches with the interior's preparation pass under the exchange, or whole-mesh stages
            (cn.step_rk4 if parts and (cn.direct or i or rng.integers(0, 2)) else cn.step_rk4_whole)()
            onl.step_rk4(stn, 20.0)
        gs, gu, gh = cn.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, stn.u[1]) and np.array_equal(gh, stn.h[1]) and np.array_equal(gs, stn.ssh[1]), tag + f" nonlinear visc {visc}"
        cn.close()
        n_rev["nonlinear"] = n_rev.get("nonlinear", 0) + 1
    n += 1
    if n % 10 == 0:
        print(f"{n} cases, {time.time() - t0:.0f}s", flush=True)
print(f"fuzz_cluster: {n} random partitions bit-identical to the single-domain oracle (seed {seed}, {skipped} refused; "
      f"reverse mode / nonlinear terms across the ranks: {n_rev})")
