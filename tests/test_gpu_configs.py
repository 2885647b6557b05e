"""GPU tests of the BASELINE.json configurations at their STATED sizes ("configs untested"), through the
C ABI, against the CPU oracle:

* configs 2 / 3: the 40 962-cell icosahedral sphere (m = 64), K = 1 and K = 60 -- RK4 and the reference's live
  reference_compat Forward-Euler step, every field bit for bit;
* config 4: 1 024 002 cells x 60 layers -- two whole RK4 steps bit for bit against the oracle (OpenMP on the box's cores);
* config 5: the Schmidt-stretched sphere with fp32 storage -- at reduced size (m = 32 / 64, K = 80, stretch 4.47) bit for
  bit against the storage-emulating oracle, and at full size (3 696 642 cells x 80 layers) through size-independent
  properties (finite, flux form conserves mass per level, identical layers stay identical bit for bit);
* the reference's own six known-answer constants tied to the fused TENDENCY kernels (K8 / K9), not only to the
  stand-alone operators: tendU = -g grad(ssh) has the gradient norms, tendH = -div(u hEdge) the divergence norms
  (test/ocn/test_Operators.jl:52-53, 72-73).
"""
import ctypes as C
import datetime as dt
import json
import os

import numpy as np
import pytest

import oracle as orc
import moka_hip as mk
from analytic import PlanarSetup, areas, error_measures
from moka_hip import lib as L
from moka_hip import meshgen as mg

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "operator_norms.json")))
CONFIG = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=10)},
          "time_integration": {"config_dt": dt.timedelta(seconds=400), "config_number_of_time_levels": 2},
          "output": {"output_interval": dt.timedelta(hours=1)}}
STRETCH = 4.47          # bench.py's config 5: Schmidt factor of the 3-60 km mesh


@pytest.fixture(scope="module")
def backend():
    b = mk.MokaHIP(0)          # raises MokaError if the HIP extension or the GPU is missing: no fallback
    yield b
    b.close()


_MESHES = {}


def sphere(m, stretch=1.0):
    if (m, stretch) not in _MESHES:
        _MESHES[(m, stretch)] = mg.icosahedral_mesh(m, stretch=stretch)
    return _MESHES[(m, stretch)]


def host_threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(16, n))


def prog_fields(Prog):
    return {"u1": Prog.normalVelocity[-1].get(), "h1": Prog.layerThickness[-1].get(), "ssh1": Prog.ssh[-1].get(),
            "u0": Prog.normalVelocity[0].get(), "h0": Prog.layerThickness[0].get(), "ssh0": Prog.ssh[0].get()}


def oracle_prog(st):
    return {"u1": st.u[1], "h1": st.h[1], "ssh1": st.ssh[1], "u0": st.u[0], "h0": st.h[0], "ssh0": st.ssh[0]}


def all_fields(Prog, Diag, Tend):
    d = prog_fields(Prog)
    d.update({"hEdge": Diag.layerThicknessEdge.get(), "F": Diag.thicknessFlux.get(), "div": Diag.velocityDivCell.get(),
              "vort": Diag.relativeVorticity.get(), "tendU": Tend.tendNormalVelocity.get(),
              "tendH": Tend.tendLayerThickness.get()})
    return d


def oracle_all(st):
    d = oracle_prog(st)
    d.update({"hEdge": st.hEdge, "F": st.F, "div": st.div, "vort": st.vort, "tendU": st.tendU, "tendH": st.tendH})
    return d


# ------------------------------------------------------------------------------------------------
# configs 2 and 3 at their stated size: m = 64 (40 962 cells), K = 1 and K = 60
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [1, 60])
def test_config2_config3_m64_rk4_and_forward_euler_bitwise(backend, K):
    mesh = sphere(64)
    assert mesh.nCells == 40962
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
    dts = Setup.timeManager.timeStep.total_seconds()         # the clock holds dt at microsecond resolution
    orc.set_threads(host_threads())
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    for _ in range(3):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(dts)
    got, exp = all_fields(Prog, Diag, Tend), oracle_all(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), ("rk4", k)
    # hipGraph replay of the launch-bound small configuration continues bit-identically
    mk.run_steps(Prog, mk.RungeKutta4, dts, 8)
    for _ in range(8):
        st.step_rk4(dts)
    got, exp = prog_fields(Prog), oracle_prog(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), ("rk4 replay", k)
    # the reference's live integrator (time_integration.jl:150-193) with its quirks; level-1-only kernels when K = 1
    flags = mk.REFERENCE_COMPAT if K == 1 else (mk.REFERENCE_COMPAT & ~4)
    for _ in range(3):
        mk.ocn_timestep(np.array([dts]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(dts, flags)
    got, exp = all_fields(Prog, Diag, Tend), oracle_all(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), ("fe", k)
    orc.set_threads(1)
    Prog._state.close(); Setup.mesh.close()


# ------------------------------------------------------------------------------------------------
# config 4 at full size: two RK4 steps of the 1 024 002-cell x 60-layer sphere, bit for bit
# ------------------------------------------------------------------------------------------------
def test_config4_full_size_two_rk4_steps_bitwise(backend):
    mesh = sphere(320)
    K = 60
    assert mesh.nCells == 1024002
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
    dts = Setup.timeManager.timeStep.total_seconds()
    orc.set_threads(host_threads())
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    for _ in range(2):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(dts)
    orc.set_threads(1)
    got, exp = prog_fields(Prog), oracle_prog(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    # the lazily produced stage-4 tendencies as well
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU)
    assert np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
    Prog._state.close(); Setup.mesh.close()


def test_config4_full_size_eight_way_partition_direct(backend):
    """BASELINE config 4 as the 8-GPU run partitions it: 1 024 002 cells x 60 layers, RCB into 8 parts, one
    context per rank on this one GPU, the DIRECT transport (push kernels storing into the neighbours' fields, flag words) --
    two RK4 steps, every rank's owned rows bit for bit against the single-domain oracle."""
    from moka_hip import parallel as par
    mesh = sphere(320)
    K = 60
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    orc.set_threads(host_threads())
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dts, 8, direct=True)
    assert cl.direct
    # the shares are what bench.py --gpus 8 gives its ranks: balanced, each with a boundary and an interior launch
    owned = [m.lm.n_owned_cells for m in cl.models]
    assert sum(owned) == mesh.nCells and max(owned) - min(owned) <= 8
    assert all(0 < m.p_boundary < m.p_owned for m in cl.models)
    cl.exchange_state()
    for _ in range(2):
        cl.step_rk4()
        st.step_rk4(dts)
    orc.set_threads(1)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, st.u[1]), "normalVelocity"
    assert np.array_equal(gh, st.h[1]), "layerThickness"
    assert np.array_equal(gs, st.ssh[1]), "ssh"
    cl.close()
    _MESHES.pop((320, 1.0), None)        # ~1 GB of host arrays


# ------------------------------------------------------------------------------------------------
# config 5 at reduced size WITH the stretch: fp32 storage, K = 80, bit for bit against the storage-emulating oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,P", [(32, 0), (64, 0), (32, 12)])
def test_config5_stretched_fp32_reduced_size_bitwise(backend, m, P):
    mesh = sphere(m, STRETCH)
    K = 80
    # the Schmidt transformation really produces a variable-resolution mesh (BASELINE: 3-60 km at m = 608)
    assert mesh.dcEdge.max() / mesh.dcEdge.min() > 10.0
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True,
                                                       patch_cells=P, state_bytes=4)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
    dts = Setup.timeManager.timeStep.total_seconds()
    orc.set_threads(host_threads())
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    tu, th, ossh = om.tendencies_clean(u, h, mixed=True)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu)
    assert np.array_equal(Tend.tendLayerThickness.get(), th)
    assert np.array_equal(Prog.ssh[-1].get(), ossh)
    st = orc.OracleState(om, ssh, u, h, mixed=True)
    for _ in range(2):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(dts)
    got, exp = prog_fields(Prog), oracle_prog(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
    # the reference's live integrator on the same state: a Forward-Euler step that carries nothing over from the RK4 steps,
    # then two with the reference's stale layerThicknessEdge and accumulating vorticity (all levels)
    for flags in (0, 3, 3):
        mk.ocn_timestep(np.array([dts]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(dts, flags)
    orc.set_threads(1)
    got, exp = prog_fields(Prog), oracle_prog(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(Diag.layerThicknessEdge.get(), st.hEdge) and np.array_equal(Diag.thicknessFlux.get(), st.F)
    assert np.array_equal(Diag.velocityDivCell.get(), st.div) and np.array_equal(Diag.relativeVorticity.get(), st.vort)
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
    Prog._state.close(); Setup.mesh.close()


# ------------------------------------------------------------------------------------------------
# config 5 at FULL size (3 696 642 cells x 80 layers, fp32 storage) through size-independent properties
# ------------------------------------------------------------------------------------------------
def test_config5_full_size_properties(backend):
    mesh = sphere(608, STRETCH)
    K = 80
    assert mesh.nCells == 3696642
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=4)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    th = Tend.tendLayerThickness.get()
    assert np.all(np.isfinite(th))
    # (a) flux form: sum_c areaCell * tendH[k,c] = 0 for every level, to round-off of the summands -- which are stored
    #     fp32 on such a state (accumulated in fp64): 2^-24 relative each
    tot = (mesh.areaCell[:, None] * th).sum(0)
    scale = (mesh.areaCell[:, None] * np.abs(th)).sum(0)
    assert np.all(np.abs(tot) <= 6e-8 * scale)
    # (b) the stored state widens exactly
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    assert np.array_equal(Prog.layerThickness[-1].get(), f32(h))
    del th, tot, scale
    # (b') BIT-LEVEL check at this size: the row-sampled oracle (oracle.SubMesh, cross-checked against the full
    #     oracle in tests/test_oracle_operators.py) evaluates ~10 000 sampled cells and as many edges from the rows their
    #     stencils gather -- the same C loop nests, the same rounding points -- for the tendency launch and for each of the
    #     four stage launches of an RK4 step (kernel modes 0, 1, 2, 2, 3), every stage from the rows the GPU itself produced.
    #     The sample: the first and the last patch of the library's order, the twelve pentagons and their edges, the finest
    #     and the coarsest cells, edge rows beyond byte offset 2^31 of the 3.55 GB normalVelocity field, and random ones.
    rng = np.random.default_rng(608)
    lib, ctx, hS = L.lib(), backend._h, Prog._state._h
    cperm = np.empty(mesh.nCells, dtype=np.int32); eperm = np.empty(mesh.nEdges, dtype=np.int32)
    L.check(lib.moka_mesh_permutation(Setup.mesh._h, L.CELL, L.i32(cperm)), ctx)
    L.check(lib.moka_mesh_permutation(Setup.mesh._h, L.EDGE, L.i32(eperm)), ctx)
    pent = np.flatnonzero(mesh.nEdgesOnCell == 5)
    by_area = np.argsort(mesh.areaCell)
    far = np.arange(int(2**31 // (K * 4)) + 1, mesh.nEdges)          # library rows whose byte offset needs the 32nd bit
    assert far.size > 10**6
    s_cells = np.unique(np.concatenate([cperm[:48], cperm[-48:], pent, by_area[:300], by_area[-300:], rng.integers(0, mesh.nCells, 9000)]))
    pe = mesh.edgesOnCell[pent].reshape(-1)
    fe_ = mesh.edgesOnCell[by_area[:100]].reshape(-1)
    s_edges = np.unique(np.concatenate([eperm[:150], eperm[-150:], pe[pe > 0] - 1, eperm[rng.choice(far, 3000, replace=False)],
                                        fe_[fe_ > 0] - 1, rng.integers(0, mesh.nEdges, 6000)]))
    sm = orc.SubMesh(mesh, s_cells, s_edges, K, rest.sum(1), max_level_edge_top=K)
    sc, se, ccl, ecl = sm.sampled_cells_global, sm.sampled_edges_global, sm.cells, sm.edges
    U, H, S = Prog.normalVelocity[-1], Prog.layerThickness[-1], Prog.ssh[-1]
    # tendency launch (mode 0): accumulated in fp64, stored fp32
    tu, th_s, _ = sm.tendencies(U.rows(ecl), H.rows(ccl), mixed=True)
    assert np.array_equal(Tend.tendNormalVelocity.rows(se), f32(tu)), "tendNormalVelocity rows at 3.7 M x 80"
    assert np.array_equal(Tend.tendLayerThickness.rows(sc), f32(th_s)), "tendLayerThickness rows at 3.7 M x 80"
    # the four stage launches, one by one (the piecewise form of moka_step_rk4: a halo object without neighbours)
    z64 = np.zeros(1, dtype=np.int64)
    hh_ = C.c_void_p()
    info = Setup.mesh.info()
    L.check(lib.moka_halo_create(hS, 0, None, L.i64(z64), None, L.i64(z64), None, L.i64(z64), None, L.i64(z64), 0, info["nPatches"],
                                 C.byref(hh_)), ctx)
    a_s, b_s = (dts / 2., dts / 2., dts, 0.0), (dts / 6., dts / 3., dts / 3., dts / 6.)
    L.check(lib.moka_rk4_dist_begin(hh_, dts), ctx)
    cur_u, cur_h = U.rows(se), H.rows(sc)                         # Curr at the sampled entities (level 1 until the step ends)
    new_u, new_h = cur_u.copy(), cur_h.copy()                     # the New accumulator starts from Curr (stage 1 aliases them)
    pu_cl, ph_cl = U.rows(ecl), H.rows(ccl)                       # provisional state of stage 1 = Curr, on the closure
    for s_ in range(1, 5):
        L.check(lib.moka_rk4_dist_stage(hh_, s_, 2), ctx)
        pu2, ph2, ssh2, nu2, nh2 = sm.rk_stage(pu_cl, ph_cl, cur_u, cur_h, new_u, new_h, a_s[s_ - 1], b_s[s_ - 1], mixed=True)
        # New accumulator = the previous level's buffers (level 0) while the step is open
        assert np.array_equal(Prog.normalVelocity[0].rows(se), nu2), f"New normalVelocity after stage {s_}"
        assert np.array_equal(Prog.layerThickness[0].rows(sc), nh2), f"New layerThickness after stage {s_}"
        new_u, new_h = nu2, nh2
        if s_ < 4:
            lvl = 3 if s_ == 2 else 2                             # stage 1, 3 -> provisional state R1 (level 2), stage 2 -> R2 (level 3)
            assert np.array_equal(U.rows(se, level=lvl), pu2), f"provisional normalVelocity after stage {s_}"
            assert np.array_equal(H.rows(sc, level=lvl), ph2), f"provisional layerThickness after stage {s_}"
            assert np.array_equal(S.rows(sc, level=lvl), ssh2), f"provisional ssh after stage {s_}"
            pu_cl, ph_cl = U.rows(ecl, level=lvl), H.rows(ccl, level=lvl)
        else:                                                     # stage 4 leaves ssh of the new state
            exp_ssh = f32(np.array([orc.ksum(col) for col in nh2]) - rest.sum(1)[sc])
            assert np.array_equal(Prog.ssh[0].rows(sc), exp_ssh), "ssh of the new level"
    L.check(lib.moka_rk4_dist_end(hh_), ctx)
    lib.moka_halo_destroy(hh_)
    assert np.array_equal(U.rows(se), new_u) and np.array_equal(H.rows(sc), new_h)     # the levels have swapped: New is current
    del sm, pu_cl, ph_cl
    tu = Tend.tendNormalVelocity.get()
    assert np.all(np.isfinite(tu))
    del tu
    # (c) K identical layers (h_k = h/K, u_k = u_1) stay identical bit for bit through RK4 steps (N3 invariant),
    #     and the state stays finite
    u1 = np.repeat(u[:, :1], K, axis=1)
    hK = np.repeat(h.sum(1, keepdims=True) / K, K, axis=1)
    for f, a in ((Prog.normalVelocity, u1), (Prog.layerThickness, hK)):
        f[0].set(a); f[-1].set(a)
    del u1, hK
    for _ in range(3):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    uK = Prog.normalVelocity[-1].get()
    assert np.all(np.isfinite(uK))
    assert np.abs(uK - uK[:, :1]).max() == 0.0
    del uK
    hh = Prog.layerThickness[-1].get()
    assert np.abs(hh - hh[:, :1]).max() == 0.0
    sshK = Prog.ssh[-1].get()
    assert np.all(np.isfinite(sshK))
    # ssh is the fp32-rounded column sum minus restingThicknessSum (N3): recompute it from the downloaded thickness
    exp = np.array([orc.ksum(col) for col in hh[:2000]]) - rest.sum(1)[:2000]
    assert np.array_equal(sshK[:2000], f32(exp))
    Prog._state.close(); Setup.mesh.close()
    _MESHES.pop((608, STRETCH), None)


# ------------------------------------------------------------------------------------------------
# the reference's known-answer constants through the fused tendency kernels (K8 / K9)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [1, 10, 60])
def test_reference_constants_through_the_tendency_kernels(backend, K):
    mesh = mg.planar_hex_mesh(48, 48, 1.0)                   # f = 0: the Coriolis term vanishes identically
    ts = PlanarSetup(mesh, K)
    g = 9.80616                                              # pressure_gradient.jl:63
    zeros_c, zeros_e = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K))
    rest = zeros_c.copy()
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)

    # K9: ssh = h_analytic (carried by level 1, restingThickness = 0), u = 0  =>  tendU = -g grad(ssh) on every level
    h = zeros_c.copy()
    h[:, 0] = ts.h()[:, 0]
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, h[:, 0].copy(), zeros_e, h, rest, CONFIG, backend, multilayer=True)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    tu = Tend.tendNormalVelocity.get()
    otu, oth, ossh = om.tendencies_clean(zeros_e, h)
    assert np.array_equal(tu, otu) and np.array_equal(Prog.ssh[-1].get(), ossh)
    for grad in (-tu / g, -otu / g):                         # GPU and oracle
        linf, l2 = error_measures(grad, ts.grad_h_edge(), areas(mesh)["edge"])
        assert abs(linf - GOLD["grad"]["L_inf"]) < GOLD["atol"] and abs(l2 - GOLD["grad"]["L_two"]) < GOLD["atol"]
    Prog._state.close(); Setup.mesh.close()

    # K5 + K7 + K8: u = F_edge, h = 1  =>  hEdge = 1, flux = F, tendH = -div(F)
    ones = np.ones((mesh.nCells, K))
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ones.sum(1), ts.F_edge(), ones, rest, CONFIG, backend, multilayer=True)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    th = Tend.tendLayerThickness.get()
    otu, oth, ossh = om.tendencies_clean(ts.F_edge(), ones)
    assert np.array_equal(th, oth)
    for div in (-th, -oth):
        linf, l2 = error_measures(div, ts.div_F(), areas(mesh)["cell"])
        assert abs(linf - GOLD["div"]["L_inf"]) < GOLD["atol"] and abs(l2 - GOLD["div"]["L_two"]) < GOLD["atol"]
    Prog._state.close(); Setup.mesh.close()
