"""Optional nonlinear (vector-invariant TRiSK) tendencies -- SURVEY.md section 8(f) rank 4 / note N4.  NOT in the
reference (parity unpinned): pinned by properties of the scheme itself."""
import numpy as np
import pytest

import oracle as orc
from moka_hip import meshgen as mg

G = 9.80616


def tc2_state(mesh, u0=38.61, h0=2998.0):
    """Williamson et al. (1992) test case 2: steady solid-body rotation u = u0 cos(lat), gh = gh0 - (a Omega u0 + u0^2/2) sin^2(lat)."""
    a, om = mg.RADIUS_EARTH, mg.OMEGA_EARTH
    R = np.hypot(np.hypot(mesh.xCell, mesh.yCell), mesh.zCell)
    latC = np.arcsin(mesh.zCell / R)
    h = h0 - (a * om * u0 + 0.5 * u0 * u0) * np.sin(latC) ** 2 / G
    RE = np.hypot(np.hypot(mesh.xEdge, mesh.yEdge), mesh.zEdge)
    latE = np.arcsin(mesh.zEdge / RE)
    u = u0 * np.cos(latE) * np.cos(mesh.angleEdge)            # zonal wind projected on the edge normal
    return u.reshape(-1, 1), h.reshape(-1, 1)


def rel_norms(mesh, tu, th, u, h):
    # characteristic sizes: Coriolis acceleration f*u and thickness change rate h*u/a
    su = np.sqrt(np.mean(tu ** 2)) / np.sqrt(np.mean((mesh.fEdge.reshape(-1, 1) * 38.61) ** 2) + 1e-30)
    sh = np.sqrt(np.mean(th ** 2)) / (np.mean(h) * 38.61 / mg.RADIUS_EARTH)
    return su, sh


def test_steady_solid_body_rotation_converges():
    """TC2 is an exact steady state of the nonlinear equations: the discrete tendencies are pure truncation error and
    shrink with the grid spacing (first order in RMS on this un-optimised geodesic grid: the error sits at the twelve
    pentagons, as for TRiSK in general), while the reference's LINEAR form is left with the O(u0 / 2 a Omega) = 4 %
    imbalance of the dropped advection terms whatever the resolution."""
    errs, lin = [], []
    for m in (8, 16, 32):
        mesh = mg.icosahedral_mesh(m)
        u, h = tc2_state(mesh)
        rest = np.full((mesh.nCells, 1), 2998.0)
        om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1), max_level_edge_top=1)
        tu, th, ssh, diag = orc.OracleNonlinear(om).tendencies(u, h)
        errs.append(rel_norms(mesh, tu, th, u, h))
        tul, thl, _ = om.tendencies_clean(u, h)
        lin.append(rel_norms(mesh, tul, thl, u, h))
        assert np.array_equal(th, thl)                         # the thickness equation is the same in both forms
    for a, b in zip(errs, errs[1:]):
        assert b[0] < a[0] / 1.8 and b[1] < a[1] / 2.5, errs
    assert errs[-1][0] < 4e-3 and errs[-1][1] < 5e-4, errs
    assert lin[-1][0] > 0.9 * lin[-2][0] > 0.015                 # the linear form does not converge to balance
    assert lin[-1][0] > 5 * errs[-1][0]


def test_rest_state_and_pure_height_gradient():
    mesh = mg.icosahedral_mesh(6)
    K = 2
    rng = np.random.default_rng(1)
    rest = np.full((mesh.nCells, K), 500.0)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    nl = orc.OracleNonlinear(om)
    tu, th, ssh, _ = nl.tendencies(np.zeros((mesh.nEdges, K)), rest)
    assert not tu.any() and not th.any() and not ssh.any()
    h = rest + rng.uniform(-1, 1, rest.shape)
    tu, th, ssh, d = nl.tendencies(np.zeros((mesh.nEdges, K)), h)
    tul, thl, sshl = om.tendencies_clean(np.zeros((mesh.nEdges, K)), h)
    assert np.array_equal(tu, tul) and np.array_equal(th, thl) and np.array_equal(ssh, sshl)   # u = 0: only -g grad(ssh) is left
    assert not d["ke"].any()


def test_kite_areas_partition_the_triangles():
    for mesh in (mg.icosahedral_mesh(6), mg.planar_hex_mesh(8, 6, 1000.0), mg.icosahedral_mesh(6, flips=4, seed=3)):
        assert mesh.kiteAreasOnVertex.shape == (mesh.nVertices, 3)
        assert mesh.maxEdges != 6 or (mesh.kiteAreasOnVertex > 0).all()      # flipped (non-Delaunay) edges give signed kites
        rel = np.abs(mesh.kiteAreasOnVertex.sum(1) / mesh.areaTriangle - 1)
        assert rel.max() < (1e-9 if mesh.maxEdges == 6 else 0.5)
        # the kites of a cell add up to its area
        acc = np.zeros(mesh.nCells)
        np.add.at(acc, mesh.cellsOnVertex.reshape(-1) - 1, mesh.kiteAreasOnVertex.reshape(-1))
        assert np.allclose(acc, mesh.areaCell, rtol=1e-9)


def test_rk4_nonlinear_holds_the_steady_state_better_than_linear():
    mesh = mg.icosahedral_mesh(12)
    u, h = tc2_state(mesh)
    rest = np.full((mesh.nCells, 1), 2998.0)
    ssh = h[:, 0] - rest[:, 0]
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1), max_level_edge_top=1)
    nl = orc.OracleNonlinear(om)
    dt = 0.3 * float(mesh.dcEdge.min()) / np.sqrt(G * 3000.0)
    a, b = orc.OracleState(om, ssh, u, h), orc.OracleState(om, ssh, u, h)
    for _ in range(40):
        nl.step_rk4(a, dt)
        b.step_rk4(dt)
    drift_nl = np.abs(a.h[1] - h).max()
    drift_lin = np.abs(b.h[1] - h).max()
    assert np.isfinite(drift_nl) and drift_nl < 0.25 * drift_lin, (drift_nl, drift_lin)


# ---- Del2 momentum mixing (the reference's uncalled sketch, horizontal_momentum_mixing.jl:53-80) --------------------
def _del2_setup(mesh, K, seed=5):
    rng = np.random.default_rng(seed)
    rest = np.full((mesh.nCells, K), 400.0)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    h = rest + rng.uniform(-1, 1, rest.shape)
    return om, u, h


@pytest.mark.parametrize("visc", [0.0, 1.0])
def test_rk4_nonlinear_13_stream_twin_against_the_running_sum(visc):
    """oracle_step_rk4_nonlinear_s13 (twin of moka_set_tuning key 7 on a nonlinear state) is the same Runge-Kutta step as
    oracle_step_rk4_nonlinear_del2 up to round-off: <= 1e-12 relative after one step, <= 1e-10 after 40, and not bit-identical
    (so the GPU test of the form does compare against the twin, not against the running sum by accident)."""
    mesh = mg.icosahedral_mesh(12)
    K = 6
    om, u, h = _del2_setup(mesh, K, seed=11)
    rest = np.full((mesh.nCells, K), 400.0)
    ssh = h.sum(1) - rest.sum(1)
    dt = 0.2 * float(mesh.dcEdge.min()) / np.sqrt(G * 2400.0)
    v = visc * 0.01 * float(mesh.dcEdge.min()) ** 2 / dt
    nl = orc.OracleNonlinear(om, visc_del2=v)
    a, b = orc.OracleState(om, ssh, u, h), orc.OracleState(om, ssh, u, h)
    rel = lambda x, y: np.max(np.abs(x - y)) / np.max(np.abs(y))
    nl.step_rk4(a, dt); nl.step_rk4_s13(b, dt)
    assert rel(b.u[1], a.u[1]) <= 1e-12 and rel(b.h[1], a.h[1]) <= 1e-12 and rel(b.ssh[1], a.ssh[1]) <= 1e-10
    assert np.array_equal(a.u[0], b.u[0]) and np.array_equal(a.h[0], b.h[0])          # the level the step started from
    for _ in range(39):
        nl.step_rk4(a, dt); nl.step_rk4_s13(b, dt)
    assert rel(b.u[1], a.u[1]) <= 1e-10 and rel(b.h[1], a.h[1]) <= 1e-10
    assert not np.array_equal(a.u[1], b.u[1])


@pytest.mark.parametrize("mesh", [mg.icosahedral_mesh(6), mg.planar_hex_mesh(8, 6, 1000.0)], ids=["ico6", "planar"])
def test_del2_term_is_built_from_the_reference_operators(mesh):
    """The extra tendency is exactly viscDel2 * (GradientOnEdge(DivergenceOnCell(u)) - d(CurlOnVertex(u))/dv): each
    piece compared with the oracle's restatement of the reference's own operator kernels (K1-K4)."""
    K, visc = 3, 2.5e3
    om, u, h = _del2_setup(mesh, K)
    base = orc.OracleNonlinear(om).tendencies(u, h)
    tu, th, ssh, d = orc.OracleNonlinear(om, visc_del2=visc).tendencies(u, h)
    assert np.array_equal(th, base[1]) and np.array_equal(ssh, base[2])
    assert np.array_equal(d["velocityDivCell"], om.divergence_on_cell(u))
    assert np.array_equal(d["relativeVorticity"], om.curl_on_vertex(u))
    v1, v2 = mesh.verticesOnEdge[:, 0] - 1, mesh.verticesOnEdge[:, 1] - 1
    c1, c2 = mesh.cellsOnEdge[:, 0] - 1, mesh.cellsOnEdge[:, 1] - 1
    div, zeta = d["velocityDivCell"], d["relativeVorticity"]
    term = ((div[c2] - div[c1]) * (1.0 / mesh.dcEdge)[:, None] - (zeta[v2] - zeta[v1]) * (1.0 / mesh.dvEdge)[:, None]) * visc
    assert np.array_equal(tu, base[0] + term)
    # viscDel2 = 0 is the plain nonlinear form, bit for bit
    assert np.array_equal(orc.OracleNonlinear(om, visc_del2=0.0).tendencies(u, h)[0], base[0])


@pytest.mark.parametrize("mesh", [mg.icosahedral_mesh(8), mg.planar_hex_mesh(10, 8, 1000.0)], ids=["ico8", "planar"])
def test_del2_dissipates_kinetic_energy(mesh):
    """Discrete integration by parts on a closed C-grid: sum_e dc dv u D(u) = -(sum_c A div^2 + sum_v A_tri zeta^2),
    so the mixing term can only remove kinetic energy."""
    om, u, h = _del2_setup(mesh, 1, seed=9)
    visc = 1.0
    base = orc.OracleNonlinear(om).tendencies(u, h)[0]
    tu, _, _, d = orc.OracleNonlinear(om, visc_del2=visc).tendencies(u, h)
    D = tu - base
    lhs = float(np.sum(mesh.dcEdge * mesh.dvEdge * u[:, 0] * D[:, 0]))
    rhs = -float(np.sum(mesh.areaCell * d["velocityDivCell"][:, 0] ** 2) + np.sum(mesh.areaTriangle * d["relativeVorticity"][:, 0] ** 2))
    assert rhs < 0 and abs(lhs - rhs) < 1e-9 * abs(rhs), (lhs, rhs)


def test_del2_damps_a_run():
    mesh = mg.icosahedral_mesh(8)
    om, u, h = _del2_setup(mesh, 1, seed=2)
    u *= 0.1
    ssh = h[:, 0] - 400.0
    dt = 0.3 * float(mesh.dcEdge.min()) / np.sqrt(G * 400.0)
    visc = 0.02 * float(mesh.dcEdge.min()) ** 2 / dt                      # well inside the diffusive stability limit
    ke = {}
    for nu in (0.0, visc):
        nl = orc.OracleNonlinear(om, visc_del2=nu)
        st = orc.OracleState(om, ssh, u, h)
        for _ in range(20):
            nl.step_rk4(st, dt)
        ke[nu] = float(np.sum(mesh.dcEdge * mesh.dvEdge * st.u[1][:, 0] ** 2))
        assert np.isfinite(ke[nu])
    assert ke[visc] < 0.8 * ke[0.0], ke
