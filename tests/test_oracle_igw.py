"""Step-loop checks of the CPU oracle on the reference's inertia-gravity-wave case
(src/inertialGravityWave.jl; driver sequence src/driver/mpas_ocean.jl:20-53)."""
import json
import math
import os

import numpy as np
import pytest

import oracle as orc
from moka_hip import meshgen as mg

EXP = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "igw_expectations.json")))


def run(res, scheme, nsteps=None):
    mesh = mg.igw_mesh(res)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    dt = mg.igw_dt(mesh)
    nsteps = int(EXP["run_hours"] * 3600 / dt) if nsteps is None else nsteps
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1))
    st = orc.OracleState(om, ssh, u, h)
    for _ in range(nsteps):
        if scheme == "fe_compat":
            st.step_fe(dt, orc.FE_REFERENCE_COMPAT)
        elif scheme == "fe_clean":
            st.step_fe(dt, 0)
        else:
            st.step_rk4(dt)
    es, eu = mg.igw_exact(mesh, nsteps * dt)
    rms = lambda a: float(np.sqrt(np.mean(a * a)))
    return mesh, st, dt, rms(st.ssh[1] - es), rms(st.u[1][:, 0] - eu)


def test_igw_sizing_and_dt():
    """polaris sizing + dt rule of init.jl:118: 200 km -> 50x50, dt = 400 s; 100 km -> 100 s."""
    m = mg.igw_mesh(200.0)
    assert (m.nCells, m.nEdges, m.nVertices) == (2500, 7500, 5000)
    assert mg.igw_dt(m) == 400 and mg.igw_dt(mg.igw_mesh(100.0)) == 100


@pytest.mark.parametrize("scheme", ["fe_compat", "fe_clean", "rk4"])
def test_igw_200km(scheme):
    _, _, dt, e_ssh, e_u = run(200.0, scheme)
    exp = EXP["cases"]["200km"][scheme]
    assert math.isclose(e_ssh, exp["ssh"], rel_tol=EXP["rtol"])
    assert math.isclose(e_u, exp["u"], rel_tol=EXP["rtol"])


def test_igw_rk4_second_order_in_space():
    _, _, _, e200, u200 = run(200.0, "rk4")
    _, _, _, e100, u100 = run(100.0, "rk4")
    exp = EXP["cases"]["100km"]["rk4"]
    assert math.isclose(e100, exp["ssh"], rel_tol=EXP["rtol"])
    assert 3.0 < e200 / e100 < 5.0 and 3.0 < u200 / u100 < 5.0


def test_fe_compat_first_step_has_zero_thickness_tendency():
    """Quirk 0.6(i): thicknessFlux uses the stale (zero-initialised) layerThicknessEdge on step 1
    (DiagnosticVars.jl:90-93,113-116), so tendLayerThickness == 0 and h does not move."""
    mesh, st, dt, _, _ = run(200.0, "fe_compat", nsteps=1)
    assert np.all(st.tendH == 0.0)
    assert np.array_equal(st.h[1], st.h[0])
    assert not np.array_equal(st.u[1], st.u[0])
    # hEdge has been refreshed at the end of diagnostic_compute!, F is still zero
    assert np.all(st.F == 0.0) and np.all(st.hEdge > 0)


def test_fe_compat_vorticity_accumulates():
    mesh, st1, dt, _, _ = run(200.0, "fe_compat", nsteps=1)
    _, st2, _, _, _ = run(200.0, "fe_compat", nsteps=2)
    om = st1.om
    z2 = om.curl_on_vertex(st1.u[1])            # curl of the state entering step 2
    assert np.allclose(st2.vort, st1.vort + z2, rtol=1e-12, atol=1e-18)


def test_time_levels_after_step():
    """advanceTimeLevels!: level 1 (index 0) holds the pre-step state (time_integration.jl:10-40)."""
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1))
    st = orc.OracleState(om, ssh, u, h)
    st.step_rk4(400.0)
    assert np.array_equal(st.u[0][:, 0], u[:, 0]) and np.array_equal(st.h[0], h)
    assert np.allclose(st.ssh[1], st.h[1][:, 0] - rest[:, 0], rtol=0, atol=0)


def test_mass_conservation_rk4():
    mesh, st, dt, _, _ = run(200.0, "rk4", nsteps=20)
    mass0 = (mesh.areaCell * st.om.arrays["restingThicknessSum"]).sum() + 0.0
    ssh0, _ = mg.igw_exact(mesh, 0.0)
    m0 = (mesh.areaCell * (1000.0 + ssh0)).sum()
    m1 = (mesh.areaCell * st.h[1][:, 0]).sum()
    assert abs(m1 - m0) / m0 < 1e-13


def test_layer_split_invariance_N3():
    """SURVEY N3 invariant: K identical layers with h_k = h/K, u_k = u reproduce the
    single-layer ssh to round-off (clean RK4, maxLevelEdgeTop = K)."""
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    K = 4
    om1 = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1), max_level_edge_top=1)
    omK = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    s1 = orc.OracleState(om1, ssh, u, h)
    sK = orc.OracleState(omK, ssh, np.repeat(u, K, axis=1), np.repeat(h / K, K, axis=1))
    for _ in range(10):
        s1.step_rk4(400.0)
        sK.step_rk4(400.0)
    assert np.abs(sK.ssh[1] - s1.ssh[1]).max() < 1e-9
    assert np.abs(sK.u[1] - s1.u[1]).max() < 1e-12


def test_mixed_precision_oracle_tracks_fp64():
    """fp32-storage emulation (config 5; not a reference feature): stored values are fp32-representable, the
    IGW solution stays within fp32 round-off accumulation of the fp64 run and on the same error curve."""
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.reshape(mesh.nCells, -1).sum(1), max_level_edge_top=1)
    a = orc.OracleState(om, ssh, u, h)
    b = orc.OracleState(om, ssh, u, h, mixed=True)
    for _ in range(30):
        a.step_rk4(400.0)
        b.step_rk4(400.0)
    for arr in (b.u[1], b.h[1], b.ssh[1]):
        assert np.array_equal(arr, arr.astype(np.float32).astype(np.float64))
    # h ~ 1000 m stored in fp32: 6e-5 m resolution, so ssh differs by a few 1e-4 m after 30 steps
    assert np.abs(a.h[1] - b.h[1]).max() < 5e-3
    assert np.abs(a.u[1] - b.u[1]).max() < 5e-5
    with pytest.raises(ValueError):
        b.step_fe(400.0)


def test_rk4_13_stream_form_against_the_reference_form():
    """The opt-in 13-stream RK4 form (oracle_step_rk4_s13, twin of moka_set_tuning key 7): New is formed in stage 4 from the
    provisional states instead of being accumulated through the stages (time_integration.jl:134-135).  Same Runge-Kutta step,
    other round-off -- bounded here on BASELINE config 3 (40 962 cells x 60 levels): <= 1e-12 relative after one step,
    <= 1e-10 after 100 steps (BASELINE.md's fp64 tolerance), and on the IGW case it sits on the same error curve."""
    mesh = mg.icosahedral_mesh(64)
    K = 60
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    a, b = orc.OracleState(om, ssh, u, h), orc.OracleState(om, ssh, u, h)
    rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())      # noqa: E731
    threads0 = orc.lib().oracle_get_threads()
    orc.set_threads(min(8, os.cpu_count() or 1))                         # 200 oracle steps of 2.5 M cell-layers: ~25 s on 8 threads
    for n in range(1, 101):
        a.step_rk4(dts)
        b.step_rk4_s13(dts)
        if n == 1:
            assert not np.array_equal(a.u[1], b.u[1])                    # (it IS another rounding: the test would be vacuous otherwise)
            assert rel(b.u[1], a.u[1]) <= 1e-12 and rel(b.h[1], a.h[1]) <= 1e-12 and np.abs(b.ssh[1] - a.ssh[1]).max() <= 1e-12 * 4000.0
    assert rel(b.u[1], a.u[1]) <= 1e-10 and rel(b.h[1], a.h[1]) <= 1e-10 and np.abs(b.ssh[1] - a.ssh[1]).max() <= 1e-10 * 4000.0
    orc.set_threads(threads0)
    # the stage-4 tendencies and the end-of-step diagnostics are the same functions of (nearly) the same state
    assert rel(b.tendU, a.tendU) <= 1e-8 and rel(b.F, a.F) <= 1e-10
    # IGW: same error against the analytic solution
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    dt = mg.igw_dt(mesh)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1))
    st = orc.OracleState(om, ssh, u, h)
    nsteps = int(EXP["run_hours"] * 3600 / dt)
    for _ in range(nsteps):
        st.step_rk4_s13(dt)
    es, eu = mg.igw_exact(mesh, nsteps * dt)
    rms = lambda x: float(np.sqrt(np.mean(x * x)))                      # noqa: E731
    exp = EXP["cases"]["200km"]["rk4"]
    assert math.isclose(rms(st.ssh[1] - es), exp["ssh"], rel_tol=EXP["rtol"]) and math.isclose(rms(st.u[1][:, 0] - eu), exp["u"], rel_tol=EXP["rtol"])
