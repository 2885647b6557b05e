"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the
same seeded inputs.  fp64 throughout; the bar is BIT-EXACT equality (np.array_equal), which is
stricter than the 1e-12 relative tolerance BASELINE.md states -- the kernels evaluate every
expression in the reference's operand order with FMA contraction off."""
import ctypes as C
import datetime as dt
import json
import math
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
IGW = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "igw_expectations.json")))


@pytest.fixture(scope="module")
def backend():
    b = mk.MokaHIP(0)          # raises MokaError if the HIP extension or the GPU is missing
    yield b
    b.close()


_MESH_CACHE = {}
# kernel variants this build of the library carries: the product has 11 (default) / 4 (column) / 3 (generic); the round-1
VARIANTS = [v for v in (11, 3, 4) if L.lib().moka_kernel_variant_available(v)]


def get_mesh(name):
    if name not in _MESH_CACHE:
        _MESH_CACHE[name] = {"planar48": lambda: mg.planar_hex_mesh(48, 48, 1.0),
                             "planar": lambda: mg.planar_hex_mesh(20, 18, 1000.0, f0=1e-4),
                             "igw200": lambda: mg.igw_mesh(200.0),
                             "ico16": lambda: mg.icosahedral_mesh(16),
                             "ico32": lambda: mg.icosahedral_mesh(32),
                             "ico12f": lambda: mg.icosahedral_mesh(12, flips=8, seed=4)}[name]()
    return _MESH_CACHE[name]


def device_mesh(backend, mesh, K, rest=None, multilayer=True, ordering=L.ORDER_DEFAULT, P=0):
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=multilayer)
    return mk.Mesh(hm, vm, backend=backend, ordering=ordering, patch_cells=P)


# ------------------------------------------------------------------------------------------------
# operators: reference known-answer vectors through the C ABI + bitwise vs the oracle
# ------------------------------------------------------------------------------------------------
def test_operator_known_answers_and_bitwise(backend):
    mesh = get_mesh("planar48")
    K = GOLD["mesh"]["nVertLevels"]
    M = device_mesh(backend, mesh, K, multilayer=False)
    om = orc.OracleMesh(mesh, K)
    ts = PlanarSetup(mesh, K)
    grad = np.zeros((mesh.nEdges, K))
    mk.GradientOnEdge(grad, ts.h(), M)
    linf, l2 = error_measures(grad, ts.grad_h_edge(), areas(mesh)["edge"])
    assert abs(linf - GOLD["grad"]["L_inf"]) < GOLD["atol"] and abs(l2 - GOLD["grad"]["L_two"]) < GOLD["atol"]
    assert np.array_equal(grad, om.gradient_on_edge(ts.h()))

    div, temp = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K))
    mk.DivergenceOnCell(div, ts.F_edge(), temp, M)
    linf, l2 = error_measures(div, ts.div_F(), areas(mesh)["cell"])
    assert abs(linf - GOLD["div"]["L_inf"]) < GOLD["atol"] and abs(l2 - GOLD["div"]["L_two"]) < GOLD["atol"]
    otemp = np.zeros_like(temp)
    assert np.array_equal(div, om.divergence_on_cell(ts.F_edge(), temp=otemp))
    assert np.array_equal(temp, otemp)

    curl = np.zeros((mesh.nVertices, K))
    mk.CurlOnVertex(curl, ts.F_edge(), M)
    linf, l2 = error_measures(curl, ts.curl_F(), areas(mesh)["vertex"])
    assert abs(linf - GOLD["curl"]["L_inf"]) < GOLD["atol"] and abs(l2 - GOLD["curl"]["L_two"]) < GOLD["atol"]
    assert np.array_equal(curl, om.curl_on_vertex(ts.F_edge()))
    # CurlOnVertex! accumulates (Operators.jl:135,142)
    mk.CurlOnVertex(curl, ts.F_edge(), M)
    assert np.array_equal(curl, om.curl_on_vertex(ts.F_edge(), curl=om.curl_on_vertex(ts.F_edge())))

    # interpolateCell2Edge! touches level 1 only (Operators.jl:207-208)
    e = np.full((mesh.nEdges, K), 7.0)
    mk.interpolateCell2Edge(e, ts.h(), M)
    oe = np.full((mesh.nEdges, K), 7.0)
    om.interpolate_cell2edge(ts.h(), nlev=1, out=oe)
    assert np.array_equal(e, oe)
    M.close()


def _lin(shape, idx1):
    """Julia linear index (1-based) into a (nVertLevels, n) array == flat index into our (n, K) C-order array."""
    return np.unravel_index(idx1 - 1, shape)


@pytest.mark.parametrize("K", [1, 10])
def test_operator_reverse_and_forward_mode_against_central_differences(backend, K):
    """test/enzyme/test_Enzyme_Operators.jl restated through the C ABI: d grad[kEnd] / d h[kBegin] (:42-131) and
    d div[kEnd] / d F[kBegin] (:137-225) in reverse AND forward mode on the 48 x 48 planar mesh with the reference's analytic
    fields, against central differences of the HIP operators themselves with the reference's step (eps = 1e-8 relative) and
    tolerance (atol = 1e-6).  The reference's own index pairs are (1, 1) and (2, 1) at nVertLevels = 1."""
    mesh = get_mesh("planar48")
    M = device_mesh(backend, mesh, K, multilayer=False)
    ts = PlanarSetup(mesh, K)
    eps = 1e-8

    def fd(op, x, kin, kout):
        xp, xm = x.copy(), x.copy()
        xp[kin] += abs(xp[kin]) * eps
        xm[kin] -= abs(xm[kin]) * eps
        return (op(xp)[kout] - op(xm)[kout]) / (xp[kin] - xm[kin])

    def grad_op(h):
        g = np.zeros((mesh.nEdges, K)); mk.GradientOnEdge(g, h, M); return g

    def div_op(v):
        d, t = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K)); mk.DivergenceOnCell(d, v, t, M); return d

    scalar = ts.h()
    for k_begin, k_end in ((1, 1), (K + 1, 3 * K + 1), (2 * K, 4 * K)):
        kin, kout = _lin((mesh.nCells, K), k_begin), _lin((mesh.nEdges, K), k_end)
        d_grad, d_h = np.zeros((mesh.nEdges, K)), np.zeros((mesh.nCells, K))
        d_grad[kout] = 1.0
        mk.GradientOnEdge_vjp(d_grad, d_h, M)                      # autodiff(Reverse, ...)   :61-67
        rev = d_h[kin]
        assert not d_grad.any()
        d_grad, d_h = np.zeros((mesh.nEdges, K)), np.zeros((mesh.nCells, K))
        d_h[kin] = 1.0
        mk.GradientOnEdge_jvp(d_grad, d_h, M)                      # autodiff(Forward, ...)   :82-99
        fwd = d_grad[kout]
        ref = fd(grad_op, scalar, kin, kout)
        assert abs(rev - ref) < 1e-6 and abs(fwd - ref) < 1e-6, ("grad", k_begin, k_end, rev, fwd, ref)
        if k_begin == 1:
            assert rev != 0.0
    vec = ts.F_edge()
    for k_begin, k_end in ((2, 1), (1, 1), (2 * K + 1, 1), (5 * K, K)):
        kin, kout = _lin((mesh.nEdges, K), k_begin), _lin((mesh.nCells, K), k_end)
        d_div, d_vec, d_temp = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K)), np.zeros((mesh.nEdges, K))
        d_div[kout] = 1.0
        mk.DivergenceOnCell_vjp(d_div, d_vec, d_temp, M)           # :160-167
        rev = d_vec[kin]
        assert not d_div.any() and not d_temp.any()
        d_div, d_vec, d_temp = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K)), np.zeros((mesh.nEdges, K))
        d_vec[kin] = 1.0
        mk.DivergenceOnCell_jvp(d_div, d_vec, d_temp, M)           # :182-194
        fwd = d_div[kout]
        ref = fd(div_op, vec, kin, kout)
        assert abs(rev - ref) < 1e-6 and abs(fwd - ref) < 1e-6, ("div", k_begin, k_end, rev, fwd, ref)
        if (k_begin, k_end, K) == (2, 1, 1):
            assert rev != 0.0
    M.close()


@pytest.mark.parametrize("meshname,K,ordering", [("planar48", 10, 0), ("ico16", 5, 0), ("ico12f", 60, 0), ("ico16", 3, 2), ("planar", 1, 1)])
def test_operator_transposes_bitwise_against_the_oracle(backend, meshname, K, ordering):
    """moka_*_vjp against the oracle's transposes, bit for bit, on random cotangents -- spheres with pentagons and flipped
    edges, every ordering of the layout pass (the summation order is tied to the CALLER's numbering) -- with non-zero shadows
    to accumulate into; and the forward mode against the oracle's operators applied to the tangents."""
    mesh = get_mesh(meshname)
    M = device_mesh(backend, mesh, K, multilayer=False, ordering=ordering)
    om = orc.OracleMesh(mesh, K)
    rng = np.random.default_rng(K + 100)
    R = lambda n: rng.standard_normal((n, K))
    # gradient
    d_grad, d_h = R(mesh.nEdges), R(mesh.nCells)
    o_grad, o_h = d_grad.copy(), d_h.copy()
    mk.GradientOnEdge_vjp(d_grad, d_h, M); om.gradient_on_edge_vjp(o_h, o_grad)
    assert np.array_equal(d_h, o_h) and not d_grad.any() and not o_grad.any()
    t_h, t_grad = R(mesh.nCells), R(mesh.nEdges)
    mk.GradientOnEdge_jvp(t_grad, t_h, M)
    assert np.array_equal(t_grad, om.gradient_on_edge(t_h))
    # divergence, with and without a temp shadow
    for with_temp in (True, False):
        d_div, d_vec, d_temp = R(mesh.nCells), R(mesh.nEdges), R(mesh.nEdges)
        o_div, o_vec, o_temp = d_div.copy(), d_vec.copy(), (d_temp.copy() if with_temp else np.zeros((mesh.nEdges, K)))
        mk.DivergenceOnCell_vjp(d_div, d_vec, d_temp if with_temp else None, M)
        om.divergence_on_cell_vjp(o_vec, o_temp, o_div)
        assert np.array_equal(d_vec, o_vec) and not d_div.any(), with_temp
        if with_temp:
            assert not d_temp.any()
    t_vec, t_div, t_temp = R(mesh.nEdges), R(mesh.nCells), R(mesh.nEdges)
    mk.DivergenceOnCell_jvp(t_div, t_vec, t_temp, M)
    o_temp = np.zeros((mesh.nEdges, K))
    assert np.array_equal(t_div, om.divergence_on_cell(t_vec, temp=o_temp)) and np.array_equal(t_temp, o_temp)
    # curl: the primal accumulates, so the output's shadow stays and the forward mode accumulates
    d_curl, d_vec = R(mesh.nVertices), R(mesh.nEdges)
    o_vec, keep = d_vec.copy(), d_curl.copy()
    mk.CurlOnVertex_vjp(d_curl, d_vec, M); om.curl_on_vertex_vjp(o_vec, keep)
    assert np.array_equal(d_vec, o_vec) and np.array_equal(d_curl, keep)
    t_vec, t_curl = R(mesh.nEdges), R(mesh.nVertices)
    exp = om.curl_on_vertex(t_vec, curl=t_curl.copy())
    mk.CurlOnVertex_jvp(t_curl, t_vec, M)
    assert np.array_equal(t_curl, exp)
    with pytest.raises(mk.MokaError):
        mk.GradientOnEdge_vjp(np.zeros((mesh.nEdges, K + 1)), np.zeros((mesh.nCells, K)), M)
    M.close()


def test_operator_argument_errors(backend):
    mesh = get_mesh("planar")
    M = device_mesh(backend, mesh, 2)
    with pytest.raises(mk.MokaError):
        mk.GradientOnEdge(np.zeros((mesh.nEdges, 3)), np.zeros((mesh.nCells, 2)), M)
    with pytest.raises(mk.MokaError, match="nlev"):
        mk.interpolateCell2Edge(np.zeros((mesh.nEdges, 2)), np.zeros((mesh.nCells, 2)), M, nlev=5)
    M.close()


# ------------------------------------------------------------------------------------------------
# fused tendency kernel vs oracle, across lane-group widths, orderings, patch sizes
# ------------------------------------------------------------------------------------------------
def random_state(mesh, K, seed):
    rng = np.random.default_rng(seed)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    return ssh, u, h, rest


@pytest.mark.parametrize("meshname,K,ordering,P", [
    ("planar", 1, L.ORDER_RCB, 0), ("planar", 1, L.ORDER_NONE, 16), ("ico16", 1, L.ORDER_RCB, 0),
    ("ico16", 2, L.ORDER_RCM, 8), ("ico16", 3, L.ORDER_RCB, 0), ("ico16", 5, L.ORDER_RCB, 5),
    ("ico16", 10, L.ORDER_RCB, 0), ("ico16", 17, L.ORDER_RCB, 0), ("ico16", 33, L.ORDER_RCM, 0),
    ("ico16", 60, L.ORDER_RCB, 0), ("ico16", 64, L.ORDER_NONE, 0), ("ico16", 80, L.ORDER_RCB, 0),
    ("ico32", 60, L.ORDER_RCB, 32), ("ico16", 130, L.ORDER_RCB, 64), ("ico32", 60, L.ORDER_RCB, 0),
    ("ico16", 8, L.ORDER_RCB, 0), ("ico16", 34, L.ORDER_RCB, 0), ("ico16", 100, L.ORDER_RCB, 0), ("planar", 60, L.ORDER_RCB, 0),
    ("ico16", 60, L.ORDER_RCM, 0), ("ico16", 60, L.ORDER_NONE, 24),
    ("ico12f", 60, L.ORDER_RCB, 0), ("ico12f", 1, L.ORDER_RCB, 0), ("ico12f", 10, L.ORDER_RCM, 0), ("ico12f", 80, L.ORDER_RCB, 0), ("ico32", 60, L.ORDER_RCB, 12), ("ico16", 62, L.ORDER_RCB, 12), ("ico32", 60, L.ORDER_RCB, 8), ("ico16", 34, L.ORDER_RCB, 7), ("planar", 60, L.ORDER_RCB, 8), ("planar", 60, L.ORDER_RCB, 12), ("ico16", 8, L.ORDER_RCB, 12),
])
def test_fused_tendency_bitwise(backend, meshname, K, ordering, P):
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 11 + K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True,
                                                       ordering=ordering, patch_cells=P)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    tu, th, ossh = om.tendencies_clean(u, h)
    info = Setup.mesh.info()
    for variant in VARIANTS:   # 11 default, 4 plain column, 3 generic index 
        backend.set_kernel_variant(variant)
        Tend.tendNormalVelocity.set(np.full_like(tu, np.nan)); Tend.tendLayerThickness.set(np.full_like(th, np.nan))
        Prog.ssh[-1].set(ssh)
        mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
        assert np.array_equal(Tend.tendNormalVelocity.get(), tu), variant
        assert np.array_equal(Tend.tendLayerThickness.get(), th), variant
        assert np.array_equal(Prog.ssh[-1].get(), ossh), variant
    backend.set_kernel_variant(0)
    if K % 2 == 0 and 8 <= K <= 64 and P == 12:
        assert 0 < info["ldsBytesPerBlock"] <= 80 * 1024, "LDS-tiled kernel (variant 2) should fit two workgroups per CU"
        assert info["maxPatchRows"] <= 136 and info["maxPatchCells"] <= 16, "tiled kernel (variant 9) must be exercised at P = 12"
    Prog._state.close(); Setup.mesh.close()


def test_max_level_edge_top_masks_levels(backend):
    """k <= maxLevelEdgeTop[e] loops (pressure_gradient.jl:61, coriolis.jl:69, horizontal_advection.jl:63)."""
    mesh = get_mesh("ico16")
    K = 6
    ssh, u, h, rest = random_state(mesh, K, 5)
    mlt = np.random.default_rng(2).integers(0, K + 1, mesh.nEdges).astype(np.int32)
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest)
    vm.maxLevelEdge.Top[:] = mlt
    M = mk.Mesh(hm, vm, backend=backend)
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    Diag, Tend = mk.DiagnosticVars(None, M, Prog._state), mk.TendencyVars(None, M, Prog._state)
    mk.computeTendency(M, Diag, Prog, Tend)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
    tu, th, _ = om.tendencies_clean(u, h)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu) and np.array_equal(Tend.tendLayerThickness.get(), th)
    Prog._state.close(); M.close()


# ------------------------------------------------------------------------------------------------
# Forward Euler: the live reference step, quirks included
# ------------------------------------------------------------------------------------------------
CONFIG = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=10)},
          "time_integration": {"config_dt": dt.timedelta(seconds=400), "config_number_of_time_levels": 2},
          "output": {"output_interval": dt.timedelta(hours=1)}}


def all_fields(Prog, Diag, Tend):
    return {"ssh0": Prog.ssh[0].get(), "ssh1": Prog.ssh[-1].get(), "u0": Prog.normalVelocity[0].get(),
            "u1": Prog.normalVelocity[-1].get(), "h0": Prog.layerThickness[0].get(), "h1": Prog.layerThickness[-1].get(),
            "hEdge": Diag.layerThicknessEdge.get(), "F": Diag.thicknessFlux.get(), "div": Diag.velocityDivCell.get(),
            "vort": Diag.relativeVorticity.get(), "tendU": Tend.tendNormalVelocity.get(),
            "tendH": Tend.tendLayerThickness.get()}


def oracle_fields(st):
    return {"ssh0": st.ssh[0], "ssh1": st.ssh[1], "u0": st.u[0], "u1": st.u[1], "h0": st.h[0], "h1": st.h[1],
            "hEdge": st.hEdge, "F": st.F, "div": st.div, "vort": st.vort, "tendU": st.tendU, "tendH": st.tendH}


@pytest.mark.parametrize("flags", [7, 0, 1, 2, 3])
def test_forward_euler_igw_bitwise(backend, flags):
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1))
    st = orc.OracleState(om, ssh, u, h)
    for step in range(12):
        mk.ocn_timestep(np.array([400.0]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(400.0, flags)
        if step in (0, 1, 11):
            got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
            for k in exp:
                assert np.array_equal(got[k], exp[k]), (k, step, flags)
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("K,flags,multilayer", [(3, 7, False), (3, 3, True), (4, 0, True), (70, 3, True)])
def test_forward_euler_multilayer_bitwise(backend, K, flags, multilayer):
    """K > 1: flags 7 = strict reference (only level 1 evolves); otherwise N3 semantics."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, K, 3)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=multilayer)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K if multilayer else 1)
    st = orc.OracleState(om, ssh, u, h)
    for step in range(4):
        mk.ocn_timestep(np.array([30.0]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(30.0, flags)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("meshname,K,flags,P", [("ico16", 60, 3, 0), ("ico16", 60, 0, 0), ("ico16", 60, 1, 0), ("ico16", 60, 2, 0),
                                                ("ico12f", 60, 3, 0), ("planar", 60, 3, 0), ("ico16", 34, 3, 0), ("ico16", 64, 0, 12),
                                                ("ico32", 60, 3, 24), ("ico12f", 40, 0, 7)])
def test_forward_euler_tuned_path_bitwise(backend, meshname, K, flags, P):
    """Even 34 <= K <= 64, all levels stepped: moka_step_fe runs in the tuned stage kernel (modes 4 / 5) plus the vertex
    pass.  Every field of the three structs against the oracle after several steps, stale and fresh flux thickness,
    accumulating and plain vorticity, pentagons / flipped edges (masked slots), then graph replay of a longer run."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 40 + K + flags)
    dtv = 2.0 if meshname == "planar" else 20.0
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, patch_cells=P)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    for step in range(4):
        mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(dtv, flags)
        # the tuned kernel, not the generic one; with the stale-thickness flag every step after the first forms that thickness
        # from the previous level's layerThickness (path 2, k_stage_rec2c mode 6) instead of gathering the stored array
        assert L.lib().moka_last_fe_path(Prog._state._h) == (2 if (flags & 1) and step > 0 else 1)
        if step in (0, 3):
            got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
            for k in exp:
                assert np.array_equal(got[k], exp[k]), (k, step)
    mk.run_steps(Prog, mk.ForwardEuler, dtv, 7, flags)
    for _ in range(7):
        st.step_fe(dtv, flags)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), (k, "replay")
    # an RK4 step in between: its lazily produced diagnostics are what the next stale-thickness FE step reads
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    st.step_rk4(dtv)
    mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
    st.step_fe(dtv, flags)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("sbytes,K", [(8, 60), (4, 80)])
def test_forward_euler_stale_thickness_from_the_previous_level(backend, sbytes, K):
    """Mode 6 of the stage kernels (round 3): with MOKA_FE_STALE_HEDGE the flux thickness of step n is the layerThicknessEdge
    step n - 1 stored = the interpolation of the layerThickness that is the PREVIOUS time level at step n.  The kernel forms it
    from those rows (a third level set keeps them readable while the step writes the new level) -- but only while the stored
    array really is that interpolation.  Whatever writes either array in between (an upload of layerThicknessEdge or of the
    previous level's layerThickness, an RK4 step, advanceTimeLevels!, diagnostic_compute!) must send the next step back to the
    stored array (path 1); results equal the oracle bit for bit throughout, previous time level included."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, K, 77)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=sbytes)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h, mixed=sbytes == 4)
    lib, hS = L.lib(), Prog._state._h
    rng = np.random.default_rng(3)

    def step(expect_path, flags=3):
        L.check(lib.moka_step_fe(hS, 20.0, flags), backend._h)
        st.step_fe(20.0, flags)
        assert lib.moka_last_fe_path(hS) == expect_path, (lib.moka_last_fe_path(hS), expect_path)

    def check(tag):
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), (k, tag)

    step(1); step(2); step(2)
    check("three steps")                                   # includes the previous time level (three rotating sets)
    # the caller overwrites Diag.layerThicknessEdge: it is no longer the interpolation of anything
    he = (1000.0 / K + rng.uniform(-1, 1, (mesh.nEdges, K)))
    if sbytes == 4:
        he = he.astype(np.float32).astype(np.float64)
    Diag.layerThicknessEdge.set(he); st.hEdge[:] = he
    step(1); step(2)
    check("after an upload of layerThicknessEdge")
    # the caller overwrites the previous level's layerThickness: the stored array is still what the reference would use
    h0 = Prog.layerThickness[0].get() + 0.25
    Prog.layerThickness[0].set(h0); st.h[0][:] = Prog.layerThickness[0].get()
    step(1); step(2)
    check("after an upload of the previous level")
    # a step without the flag in between (path 1, mode 5) keeps the relation: the next stale step may use the previous level
    step(1, flags=2); step(2, flags=3)
    check("fresh, then stale")
    if sbytes == 8:
        mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=20.0))
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4); st.step_rk4(20.0)
        step(1); step(2)
        check("after an RK4 step")
        # advanceTimeLevels! on its own (time_integration.jl:10-40): previous <- current, under the stored layerThicknessEdge
        mk.advanceTimeLevels(Prog)
        for a in (st.ssh, st.u, st.h):
            a[0][...] = a[1]
        step(1); step(2)
        check("after advanceTimeLevels!")
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("sbytes,K,flags", [(8, 60, 3), (8, 60, 0), (4, 80, 3), (8, 34, 1), (4, 40, 2)])
def test_forward_euler_lean_steps_produce_every_array_on_demand(backend, sbytes, K, flags):
    """A Forward-Euler step of all levels is LEAN by default: it stores the new time level and relativeVorticity; the step's
    tendNormalVelocity, tendLayerThickness, thicknessFlux, velocityDivCell and layerThicknessEdge are produced on the first read,
    from the level the step started from (moka_fe_lazy_pending).  Whatever reads them -- a download, sumArray, an upload that
    overwrites one of their inputs, the piecewise reference calls, an RK4 step in between -- sees the bits a step that stores
    everything (moka_set_tuning(4, 0)) and the oracle produce."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, K, 90 + K + flags)
    lib = L.lib()
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    dtv = 20.0

    def fresh():
        Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=sbytes)
        return Setup, Diag, Tend, Prog, orc.OracleState(om, ssh, u, h, mixed=sbytes == 4)

    def step(Prog, st, n=1, fl=flags):
        for _ in range(n):
            L.check(lib.moka_step_fe(Prog._state._h, dtv, fl), backend._h)
            st.step_fe(dtv, fl)

    def check(Prog, Diag, Tend, st, tag):
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), (k, tag)

    # (a) every step stores everything: the reference point
    L.check(lib.moka_set_tuning(4, 0))
    try:
        Setup, Diag, Tend, Prog, st = fresh()
        step(Prog, st, 4)
        assert lib.moka_fe_lazy_pending(Prog._state._h) == 0
        check(Prog, Diag, Tend, st, "eager")
        Prog._state.close(); Setup.mesh.close()
    finally:
        L.check(lib.moka_set_tuning(4, 1))
    # (b) lean steps
    Setup, Diag, Tend, Prog, st = fresh()
    hS = Prog._state._h
    step(Prog, st, 1)
    assert lib.moka_fe_lazy_pending(hS) == (0 if flags & 1 else 1)     # the first stale-thickness step has a stored array to read
    step(Prog, st, 3)
    assert lib.moka_fe_lazy_pending(hS) == 1
    # one read produces all of them
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU)
    assert lib.moka_fe_lazy_pending(hS) == 0
    check(Prog, Diag, Tend, st, "after four lean steps")
    step(Prog, st, 2)
    assert lib.moka_fe_lazy_pending(hS) == 1
    # sumArray of a pending array (run_loop.jl:47-51 on Diag.thicknessFlux)
    out = C.c_double()
    L.check(lib.moka_sum_sq(hS, L.F_THICKNESS_FLUX, 1, C.byref(out)), backend._h)
    F = np.ascontiguousarray(st.F)
    assert out.value == orc.lib().oracle_sum_sq(orc._p(F), F.size) and lib.moka_fe_lazy_pending(hS) == 0
    step(Prog, st, 1)
    # the caller overwrites an INPUT of the pending arrays (the previous level): they are produced from what the step saw
    u0 = Prog.normalVelocity[0].get() * 0.5
    Prog.normalVelocity[0].set(u0); st.u[0][:] = Prog.normalVelocity[0].get()
    assert lib.moka_fe_lazy_pending(hS) == 0
    check(Prog, Diag, Tend, st, "upload of the previous level while arrays were pending")
    step(Prog, st, 2)
    check(Prog, Diag, Tend, st, "two more lean steps")
    Prog._state.close(); Setup.mesh.close()
    # (c) an RK4 step supersedes pending arrays; the Forward-Euler step after it starts from the RK4 diagnostics
    if sbytes == 8:
        Setup, Diag, Tend, Prog, st = fresh()
        step(Prog, st, 3)
        mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4); st.step_rk4(dtv)
        assert lib.moka_fe_lazy_pending(Prog._state._h) == 0
        step(Prog, st, 3)
        check(Prog, Diag, Tend, st, "RK4 in between")
        Prog._state.close(); Setup.mesh.close()
    # (d) lean launches run their own kernel instances (stage-kernel modes 10 / 11: the optional outputs compiled out); with
    # moka_set_tuning(9, 0) they go through the general Forward-Euler instances (outputs tested at run time): the same bits
    L.check(lib.moka_set_tuning(9, 0))
    try:
        Setup, Diag, Tend, Prog, st = fresh()
        step(Prog, st, 4)
        assert lib.moka_fe_lazy_pending(Prog._state._h) == 1
        check(Prog, Diag, Tend, st, "lean steps through the general instances")
        Prog._state.close(); Setup.mesh.close()
    finally:
        L.check(lib.moka_set_tuning(9, 1))


def test_forward_euler_tuned_path_level_masks(backend):
    """maxLevelEdgeTop < K on random edges: the masked branches of modes 4 / 5 (divergence ignores the mask, the
    thickness tendency and the velocity tendency honour it)."""
    mesh = get_mesh("ico16")
    K = 60
    ssh, u, h, rest = random_state(mesh, K, 9)
    mlt = np.random.default_rng(4).integers(0, K + 1, mesh.nEdges).astype(np.int32)
    mlt[np.random.default_rng(5).random(mesh.nEdges) < 0.5] = K
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
    vm.maxLevelEdge.Top[:] = mlt
    M = mk.Mesh(hm, vm, backend=backend)
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    Diag, Tend = mk.DiagnosticVars(None, M, Prog._state), mk.TendencyVars(None, M, Prog._state)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
    for flags in (3, 0):
        st = orc.OracleState(om, Prog.ssh[-1].get(), Prog.normalVelocity[-1].get(), Prog.layerThickness[-1].get())
        st.hEdge[:] = Diag.layerThicknessEdge.get(); st.vort[:] = Diag.relativeVorticity.get()
        for _ in range(3):
            L.check(L.lib().moka_step_fe(Prog._state._h, 20.0, flags), backend._h)
            assert L.lib().moka_last_fe_path(Prog._state._h) in (1, 2)
            st.step_fe(20.0, flags)
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            if k not in ("ssh0", "u0", "h0"):
                assert np.array_equal(got[k], exp[k]), (k, flags)
    Prog._state.close(); M.close()


@pytest.mark.parametrize("nx,ny,K", [(4, 4, 60), (4, 6, 1), (6, 4, 34), (2, 4, 60), (4, 2, 8)])
def test_tiny_periodic_meshes_bitwise(backend, nx, ny, K):
    """Smallest doubly periodic hexagon meshes: a cell meets the same neighbour across both boundaries, an edge's
    edgesOnEdge list repeats entries, and the whole mesh is one or two patches -- RK4, Forward Euler and the
    tendencies still equal the oracle bit for bit."""
    mesh = mg.planar_hex_mesh(nx, ny, 1000.0, f0=1e-4)
    ssh, u, h, rest = random_state(mesh, K, 100 + nx * ny + K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    tu, th, _ = om.tendencies_clean(u, h)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu) and np.array_equal(Tend.tendLayerThickness.get(), th)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=2.0))
    for _ in range(3):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(2.0)
    for _ in range(3):
        mk.ocn_timestep(np.array([2.0]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=3)
        st.step_fe(2.0, 3)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    Prog._state.close(); Setup.mesh.close()


def test_reference_call_sequence_piecewise(backend):
    """The separately exported reference entry points, called in the order of time_integration.jl:163-177,
    give the same Diag/Tend arrays as the oracle's restatement of that order."""
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=rest.sum(1))
    st = orc.OracleState(om, ssh, u, h)
    M = Setup.mesh
    for _ in range(2):
        mk.advanceTimeLevels(Prog)
        mk.diagnostic_compute(M, Diag, Prog)
        mk.computeNormalVelocityTendency(Tend, Prog, Diag, M, None)
        mk.computeLayerThicknessTendency(Tend, Prog, Diag, M, None)
        orc.lib().oracle_diagnostic_compute(om.ref, orc._p(st.hEdge), orc._p(st.F), orc._p(st.div), orc._p(st.vort),
                                            orc._p(st.u[1]), orc._p(st.h[1]), 1)
        orc.lib().oracle_normal_velocity_tendency(om.ref, orc._p(st.tendU), orc._p(st.ssh[1]), orc._p(st.u[1]), 1)
        orc.lib().oracle_layer_thickness_tendency(om.ref, orc._p(st.tendH), orc._p(st.F), 1)
        got = all_fields(Prog, Diag, Tend)
        for k in ("hEdge", "F", "div", "vort", "tendU", "tendH"):
            assert np.array_equal(got[k], oracle_fields(st)[k]), k
    Prog._state.close(); M.close()


# ------------------------------------------------------------------------------------------------
# RK4 stage loop
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("meshname,K,nsteps,variant", [("igw200", 1, 10, 0), ("ico16", 1, 5, 0), ("ico32", 60, 2, 0), ("ico12f", 60, 3, 0), ("ico12f", 3, 3, 0),
                                                         ("ico32", 60, 3, 11), ("ico16", 60, 3, 4), ("ico16", 80, 2, 4), ("ico16", 60, 3, 3), ("ico16", 33, 2, 0),
                                                         ("ico16", 100, 2, 0)])
def test_rk4_bitwise(backend, meshname, K, nsteps, variant):
    backend.set_kernel_variant(variant)
    mesh = get_mesh(meshname)
    if meshname == "igw200":
        ssh, u, h, rest = mg.igw_initial_state(mesh)
        dtv = 400.0
    else:
        ssh, u, h, rest = random_state(mesh, K, 9)
        dtv = 20.0
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True,
                                                       patch_cells=12 if variant == 11 else 0)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
    for _ in range(nsteps):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(dtv)
    # includes Diag (diagnostic_compute! of the new state, :147) and Tend (stage-4 tendencies), produced lazily
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    # an RK4 step followed by a reference-compat Forward-Euler step sees those diagnostics (stale hEdge)
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    st.step_rk4(dtv)
    mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler)
    st.step_fe(dtv, 7 if K == 1 else 3)
    if K == 1:
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), k
    backend.set_kernel_variant(0)
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("meshname,K,P", [("ico32", 60, 0), ("ico16", 64, 0), ("ico12f", 34, 12), ("ico32", 60, 5)])
def test_two_patches_per_workgroup_bitwise(backend, meshname, K, P):
    """moka_set_tuning(8, mask): large launches of the Float64 stage kernel walk TWO consecutive patches per 512-thread workgroup
    (one staging phase, one row cache over both patches' own edges).  Same entities, same arithmetic: the tendency launch, RK4 steps
    in the reference's and in the 13-stream form, with every mode paired (bit 16 of the mask drops the size threshold for the test),
    odd patch counts and partial level masks included, equal the oracle bit for bit -- and equal the unpaired launches."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 31)
    mlt = np.full(mesh.nEdges, K, dtype=np.int32)
    if meshname == "ico12f":
        r = np.random.default_rng(8)
        sel = r.random(mesh.nEdges) < 0.3
        mlt[sel] = r.integers(0, K + 1, int(sel.sum()))
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
    vm.maxLevelEdge.Top[:] = mlt
    M = mk.Mesh(hm, vm, backend=backend, patch_cells=P)
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    Diag, Tend = mk.DiagnosticVars(None, M, Prog._state), mk.TendencyVars(None, M, Prog._state)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
    st = orc.OracleState(om, ssh, u, h)
    lib, hS = L.lib(), Prog._state._h
    every = (1 << 16) | 0b1110001111                       # modes 0-3 and 7-9, no size threshold
    try:
        L.check(lib.moka_set_tuning(8, every))
        mk.computeTendency(M, Diag, Prog, Tend)
        tu, th, _ = om.tendencies_clean(u, h)
        assert np.array_equal(Tend.tendNormalVelocity.get(), tu) and np.array_equal(Tend.tendLayerThickness.get(), th)
        for _ in range(2):
            L.check(lib.moka_step_rk4(hS, 20.0), backend._h)
            st.step_rk4(20.0)
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), k
        L.check(lib.moka_set_tuning(7, 1))
        for _ in range(2):
            L.check(lib.moka_step_rk4(hS, 20.0), backend._h)
            st.step_rk4_s13(20.0)
        L.check(lib.moka_set_tuning(7, 0))
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), k
    finally:
        L.check(lib.moka_set_tuning(7, 0))
        L.check(lib.moka_set_tuning(8, 0b10000011))           # the default mask: modes 0, 1, 7
    Prog._state.close(); M.close()


@pytest.mark.parametrize("meshname,K,P,nsteps", [("ico16", 60, 0, 3), ("ico32", 34, 0, 2), ("ico12f", 64, 12, 2), ("ico16", 60, 0, 11)])
def test_rk4_13_stream_form_bitwise_against_its_twin(backend, meshname, K, P, nsteps):
    """moka_set_tuning(7, 1): RK4 steps with 13 instead of 16 state streams (stage kernel modes 7 / 8 / 9; New formed by stage 4
    from the own rows of Curr and the provisional states).  NOT the reference's round-off: opt-in, compared bit for bit with its
    own oracle twin (oracle_step_rk4_s13, whose distance to the reference form tests/test_oracle_igw.py bounds) -- every field of
    Prog, both levels, the lazily produced stage-4 tendencies and diagnostics; step by step and through moka_run's graph replay
    (three buffer sets rotate: period 3).  Default off: with the key cleared the same state steps in the reference's form."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 21)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, patch_cells=P)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    lib = L.lib()
    L.check(lib.moka_set_tuning(7, 1))
    try:
        if nsteps > 8:
            mk.run_steps(Prog, mk.RungeKutta4, 20.0, nsteps)            # eager first step + replayed periods + remainder
        else:
            mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=20.0))
            for _ in range(nsteps):
                mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        for _ in range(nsteps):
            st.step_rk4_s13(20.0)
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), k
    finally:
        L.check(lib.moka_set_tuning(7, 0))
    # key cleared: the reference's form again, from the state the 13-stream steps left
    mk.run_steps(Prog, mk.RungeKutta4, 20.0, 2)
    st.step_rk4(20.0); st.step_rk4(20.0)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    Prog._state.close(); Setup.mesh.close()


# ------------------------------------------------------------------------------------------------
# fp32-storage state (BASELINE config 5: "fp32 state with fp64 tendency accumulation").  Not a reference
# feature; the oracle emulates the storage (oracle_step_rk4_mixed) and the bar stays BIT-EXACT.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("meshname,K,P,nsteps", [("ico16", 80, 0, 3), ("ico16", 60, 12, 2), ("ico32", 80, 12, 2), ("ico16", 4, 0, 3),
                                                  ("ico16", 128, 8, 2), ("planar", 8, 0, 4), ("ico12f", 80, 0, 2),
                                                  ("ico16", 64, 32, 2), ("ico16", 36, 5, 2)])
def test_fp32_state_tendency_and_rk4_bitwise(backend, meshname, K, P, nsteps):
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 21 + K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True,
                                                       patch_cells=P, state_bytes=4)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    # upload rounds to fp32, download widens exactly
    assert np.array_equal(Prog.normalVelocity[-1].get(), f32(u))
    assert np.array_equal(Prog.layerThickness[-1].get(), f32(h))
    tu, th, ossh = om.tendencies_clean(u, h, mixed=True)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu)
    assert np.array_equal(Tend.tendLayerThickness.get(), th)
    assert np.array_equal(Prog.ssh[-1].get(), ossh)
    st = orc.OracleState(om, ssh, u, h, mixed=True)
    dtv = 2.0 if meshname == "planar" else 20.0          # the 1 km planar mesh is unstable at 20 s
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
    for _ in range(nsteps):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        st.step_rk4(dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1])
    assert np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
    assert np.array_equal(Prog.normalVelocity[0].get(), st.u[0]) and np.array_equal(Prog.layerThickness[0].get(), st.h[0])
    # stage-4 tendencies, produced lazily from the fp32 provisional state
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
    # sum(ssh^2) in the reference's serial order over the widened values
    tot = L.C.c_double()
    L.check(L.lib().moka_sum_sq(Prog._state._h, L.F_SSH, 1, L.C.byref(tot)), backend._h)
    assert tot.value == st.sum_sq_ssh()
    # graph replay (moka_run) continues bit-identically
    mk.run_steps(Prog, mk.RungeKutta4, dtv, 7)
    for _ in range(7):
        st.step_rk4(dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    # and it stays within fp32 round-off of the fp64 path
    s64 = orc.OracleState(om, f32(ssh), f32(u), f32(h))
    for _ in range(nsteps + 7):
        s64.step_rk4(dtv)
    assert np.abs(st.h[1] - s64.h[1]).max() <= 64 * np.finfo(np.float32).eps * np.abs(s64.h[1]).max()
    # what the fp32 form does not carry fails loudly: level-1-only steps (the reference's default flags at K > 1), and
    # DiagnosticVars after an RK4 step (they come out of Forward-Euler steps only), hence steps that would carry them over
    with pytest.raises(mk.MokaError):
        mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler)
    with pytest.raises(mk.MokaError):
        Diag.layerThicknessEdge.get()
    with pytest.raises(mk.MokaError):
        mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=3)
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("meshname,K,P,flags,nsteps", [("ico16", 80, 0, 3, 4), ("ico16", 60, 12, 0, 3), ("ico32", 80, 0, 1, 3), ("ico16", 4, 0, 2, 4),
                                                        ("ico12f", 80, 0, 3, 3), ("planar", 8, 0, 3, 5), ("ico16", 128, 8, 3, 2)])
def test_fp32_state_forward_euler_bitwise(backend, meshname, K, P, flags, nsteps):
    """The reference's live step (time_integration.jl:150-193: advanceTimeLevels!, diagnostic_compute!, both tendencies,
    updates) on an fp32-storage state, all levels: every array of Prog, Diag and Tend against the storage-emulating oracle
    (oracle_step_fe_mixed), bit for bit -- stale layerThicknessEdge (1) and accumulating vorticity (2) included; then an RK4
    step, then a Forward-Euler step without carried diagnostics."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 31 + K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True,
                                                       patch_cells=P, state_bytes=4)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h, mixed=True)
    dtv = 2.0 if meshname == "planar" else 20.0
    for i in range(nsteps):
        mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=flags)
        st.step_fe(dtv, flags)
        got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
        for k in exp:
            assert np.array_equal(got[k], exp[k]), (k, i)
    mk.run_steps(Prog, mk.ForwardEuler, dtv, 5, flags=flags)           # graph replay
    for _ in range(5):
        st.step_fe(dtv, flags)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))      # integrators mix: RK4, then a step that carries nothing over
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    st.step_rk4(dtv)
    mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=0)
    st.step_fe(dtv, 0)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    Prog._state.close(); Setup.mesh.close()


def test_fp32_state_unsupported_shapes(backend):
    mesh = get_mesh("ico16")
    for K in (1, 6, 132):
        ssh, u, h, rest = random_state(mesh, K, 3)
        with pytest.raises(mk.MokaError):
            mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=4)


def test_driver_replay_igw_forward_euler_and_rk4(backend):
    """Replays src/driver/mpas_ocean.jl:20-53: init -> ocn_init_alarms (dt override, init.jl:118) ->
    timestep[1] = dt -> ocn_run_loop until the simulation alarm rings -> download; RMS error vs the
    analytic solution lands on the oracle's numbers (tests/golden/igw_expectations.json)."""
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    exp = IGW["cases"]["200km"]
    for method, key in ((mk.ForwardEuler, "fe_compat"), (mk.RungeKutta4, "rk4")):
        Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend)
        clock, simAlarm, outAlarm = mk.ocn_init_alarms(Setup)
        assert clock.timeStep == dt.timedelta(seconds=400)
        timestep = np.zeros(1)
        timestep[0] = clock.timeStep.total_seconds()
        sumCPU, sumGPU = np.zeros(1), np.zeros(1)
        total = mk.ocn_run_loop(sumCPU, sumGPU, timestep, Prog, Diag, Tend, Setup, method, clock, simAlarm, outAlarm)
        assert clock.currTime == dt.datetime(1, 1, 1, 10)
        es, eu = mg.igw_exact(mesh, 36000.0)
        got_ssh, got_u = Prog.ssh[-1].get(), Prog.normalVelocity[-1].get()[:, 0]
        rms = lambda a: float(np.sqrt(np.mean(a * a)))
        assert math.isclose(rms(got_ssh - es), exp[key]["ssh"], rel_tol=IGW["rtol"])
        assert math.isclose(rms(got_u - eu), exp[key]["u"], rel_tol=IGW["rtol"])
        # sumArray (run_loop.jl:47-51): strictly serial order, bitwise
        assert total == orc.lib().oracle_sum_sq(orc._p(np.ascontiguousarray(got_ssh)), got_ssh.size)
        Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("method,K,nsteps", [("rk4", 1, 9), ("rk4", 60, 8), ("fe", 1, 11), ("fe", 4, 7), ("rk4", 1, 3)])
def test_moka_run_graph_replay_bitwise(backend, method, K, nsteps):
    """moka_run replays long runs from a hipGraph of two captured steps: same bits as stepping one by one."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, K, 21)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    flags = 3
    mk.run_steps(Prog, mk.RungeKutta4 if method == "rk4" else mk.ForwardEuler, 15.0, nsteps, flags)
    for _ in range(nsteps):
        st.step_rk4(15.0) if method == "rk4" else st.step_fe(15.0, flags)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    Prog._state.close(); Setup.mesh.close()


def test_upload_download_roundtrip_and_errors(backend):
    mesh = get_mesh("ico16")
    K = 7
    ssh, u, h, rest = random_state(mesh, K, 1)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    assert np.array_equal(Prog.normalVelocity[0].get(), u) and np.array_equal(Prog.layerThickness[-1].get(), h)
    assert np.array_equal(Prog.ssh[0].get(), ssh)
    assert np.all(Diag.relativeVorticity.get() == 0) and np.all(Tend.tendLayerThickness.get() == 0)
    with pytest.raises(mk.MokaError, match="nTimeLevels"):
        mk.PrognosticVars(ssh, u, h, 3, Setup.mesh)
    out = np.zeros(3)
    rc = L.lib().moka_state_download(Prog._state._h, 99, 1, L.f64(out))
    assert rc == L.ERR_ARG and b"unknown field" in L.lib().moka_last_error(backend._h)
    Prog._state.close(); Setup.mesh.close()


# ------------------------------------------------------------------------------------------------
# full-size (BASELINE config 4: 1 024 002 cells x 60 layers) through size-independent properties
# ------------------------------------------------------------------------------------------------
def test_full_size_properties(backend):
    mesh = mg.icosahedral_mesh(320)
    K = 60
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    th, tu = Tend.tendLayerThickness.get(), Tend.tendNormalVelocity.get()
    assert np.all(np.isfinite(th)) and np.all(np.isfinite(tu))
    # (a) flux form: sum_c areaCell * tendH[k,c] = 0 for every level, to round-off of the summands
    tot = (mesh.areaCell[:, None] * th).sum(0)
    scale = (mesh.areaCell[:, None] * np.abs(th)).sum(0)
    assert np.all(np.abs(tot) <= 1e-11 * scale)
    # (b) bitwise agreement with the oracle on a random sample of whole columns is impossible without the
    #     full oracle run; instead compare a patch-independent subset: first 3000 cells / edges (caller numbering)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    orc.set_threads(min(16, os.cpu_count() or 1))
    otu, oth, ossh = om.tendencies_clean(u, h)
    orc.set_threads(1)
    assert np.array_equal(th, oth) and np.array_equal(tu, otu) and np.array_equal(Prog.ssh[-1].get(), ossh)
    # (c) K identical layers (h_k = h/K, u_k = u) => same ssh as the single-layer model (N3 invariant), RK4
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dts))
    u1 = u[:, :1].copy()
    hK = np.repeat(h.sum(1, keepdims=True) / K, K, axis=1)
    for f, a in ((Prog.normalVelocity, np.repeat(u1, K, axis=1)), (Prog.layerThickness, hK)):
        f[0].set(a); f[-1].set(a)
    for _ in range(3):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    sshK = Prog.ssh[-1].get()
    uK = Prog.normalVelocity[-1].get()
    assert np.abs(uK - uK[:, :1]).max() == 0.0          # layers stay identical, bit for bit
    Prog._state.close(); Setup.mesh.close()
    S1, D1, T1, P1 = mk.ocn_init_from_arrays(mesh, ssh, u1, h.sum(1, keepdims=True), rest.sum(1, keepdims=True),
                                             CONFIG, backend, multilayer=True)
    mk.changeTimeStep(S1.timeManager, dt.timedelta(seconds=dts))
    for _ in range(3):
        mk.ocn_timestep(P1, D1, T1, S1, mk.RungeKutta4)
    assert np.abs(P1.ssh[-1].get() - sshK).max() < 1e-8
    assert np.abs(P1.normalVelocity[-1].get()[:, 0] - uK[:, 0]).max() < 1e-12
    P1._state.close(); S1.mesh.close()


# ------------------------------------------------------------------------------------------------
# the driver end to end: YAML config + MPAS files -> ocn_init -> alarm-driven loop
# -> write_netcdf, against the oracle replaying the same sequence (src/driver/mpas_ocean.jl:20-53)
# ------------------------------------------------------------------------------------------------
def test_driver_from_yaml_and_mpas_files(tmp_path):
    from moka_hip import driver, mpasio
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    mesh_fp, out_fp, cfg_fp = tmp_path / "igw_mesh.nc", tmp_path / "output.nc", tmp_path / "config.yml"
    mpasio.write_mesh(mesh_fp, mesh, restingThickness=np.asarray(rest).reshape(mesh.nCells, 1),
                      state=(ssh, u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1)))
    cfg_fp.write_text(f"""omega:
  time_management:
    config_do_restart: false
    config_restart_timestamp_name: Restart_timestamp
    config_start_time: 0001-01-01_00:00:00
    config_stop_time: none
    config_run_duration: 0000-00-00_02:00:00
  time_integration:
    config_dt: 0000-00-00_00:05:00
    config_number_of_time_levels: 2
  streams:
    mesh:
      filename_template: {mesh_fp}
    input:
      filename_template: {mesh_fp}
    output:
      filename_template: {out_fp}
      reference_time: 0001-01-01_00:00:00
      output_interval: 0000-00-00_01:00:00
""")
    assert driver.ocn_run(str(cfg_fp)) == str(out_fp)
    dts = float(mg.igw_dt(mesh))                    # ocn_init_alarms overrides config_dt (init.jl:118): 400 s at 200 km
    assert dts == 400.0
    nsteps = int(2 * 3600 / dts)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=np.asarray(rest).reshape(mesh.nCells, -1).sum(1))
    st = orc.OracleState(om, ssh, u, h)
    for _ in range(nsteps):
        st.step_fe(dts)
    ds = mpasio.open_dataset(out_fp)
    try:
        assert ds.attr("dt") == dts and ds.var("time")[0] == nsteps * dts
        # the reference's file holds the state one step behind (adapt_structure rebuilds Prog from its FIRST time level:
        # OutPut.jl:124, PrognosticVars.jl:108-113): index 0 = the level the last step started from
        assert np.array_equal(ds.var("ssh"), st.ssh[0]) and not np.array_equal(st.ssh[0], st.ssh[1])
        assert np.array_equal(ds.var("layerThickness")[0], st.h[0][:, 0])
        assert np.array_equal(ds.var("normalVelocity")[0], st.u[0][:, 0])
    finally:
        ds.close()


def _write_igw_case(tmp_path, hours=2):
    from moka_hip import mpasio
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    mesh_fp, out_fp, cfg_fp = tmp_path / "igw_mesh.nc", tmp_path / "output.nc", tmp_path / "config.yml"
    mpasio.write_mesh(mesh_fp, mesh, restingThickness=np.asarray(rest).reshape(mesh.nCells, 1),
                      state=(ssh, u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1)))
    cfg_fp.write_text(f"""omega:
  time_management:
    config_do_restart: false
    config_start_time: 0001-01-01_00:00:00
    config_stop_time: none
    config_run_duration: 0000-00-00_0{hours}:00:00
  time_integration:
    config_dt: 0000-00-00_00:05:00
    config_number_of_time_levels: 2
  streams:
    mesh:
      filename_template: {mesh_fp}
    input:
      filename_template: {mesh_fp}
    output:
      filename_template: {out_fp}
      reference_time: 0001-01-01_00:00:00
      output_interval: 0000-00-00_01:00:00
""")
    return mesh, (ssh, u, h, rest), cfg_fp, out_fp


def test_driver_in_the_reference_constructor_order(tmp_path):
    """src/driver/mpas_ocean.jl:20-53 line by line through moka_hip/shim.py, the executable transliteration of the Julia
    shim (julia/MokaHIP.jl): the reference's OWN constructors run first -- every array adapted to the backend or made by
    KA.zeros / KA.ones before any device state exists (init.jl:3-30, PrognosticVars.jl:59-106, DiagnosticVars.jl:75-99,
    TendencyVars.jl:51-67), `timestep` a 1-element array on the backend written with a scalar store -- the first
    ocn_timestep binds everything to one library state, write_netcdf gets host arrays back through
    Adapt.adapt_structure(KA.CPU(), x).  Output file bit-identical to the oracle replaying the same sequence."""
    from moka_hip import mpasio, shim
    mesh, (ssh, u, h, rest), cfg_fp, out_fp = _write_igw_case(tmp_path)
    out, arch, clock, (Setup, Diag, Tend, Prog) = shim.ocn_run(str(cfg_fp))
    assert out == str(out_fp) and arch == "GPU"
    dts = float(mg.igw_dt(mesh))                    # ocn_init_alarms overrides config_dt (init.jl:118): 400 s at 200 km
    nsteps = int(2 * 3600 / dts)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=np.asarray(rest).reshape(mesh.nCells, -1).sum(1))
    st = orc.OracleState(om, ssh, u, h)
    for _ in range(nsteps):
        st.step_fe(dts)
    ds = mpasio.open_dataset(out_fp)
    try:
        assert ds.attr("dt") == dts and ds.var("time")[0] == nsteps * dts
        # the reference's file holds the state one step behind (adapt_structure rebuilds Prog from its FIRST time level:
        # OutPut.jl:124, PrognosticVars.jl:108-113): index 0 = the level the last step started from
        assert np.array_equal(ds.var("ssh"), st.ssh[0]) and not np.array_equal(st.ssh[0], st.ssh[1])
        assert np.array_equal(ds.var("layerThickness")[0], st.h[0][:, 0])
        assert np.array_equal(ds.var("normalVelocity")[0], st.u[0][:, 0])
    finally:
        ds.close()
    # what the lazily bound arrays show afterwards: every field of the three structs, both time levels
    f = lambda a: np.asarray(a)
    assert np.array_equal(f(Prog.ssh[0]), st.ssh[0]) and np.array_equal(f(Prog.normalVelocity[0]), st.u[0])
    for got, exp in ((Diag.layerThicknessEdge, st.hEdge), (Diag.thicknessFlux, st.F), (Diag.velocityDivCell, st.div),
                     (Diag.relativeVorticity, st.vort), (Tend.tendNormalVelocity, st.tendU), (Tend.tendLayerThickness, st.tendH)):
        assert np.array_equal(f(got), exp)
    # scalar reads of a bound array cost ONE download per device change (version stamp), scalar writes reach the device
    # before the next step
    s = Prog.ssh[-1].state
    assert Prog.ssh[-1].host_version != s.version            # stale: the steps since binding changed the device
    x = [Prog.ssh[-1][i] for i in range(5)]                  # first scalar read downloads, the rest do not
    assert Prog.ssh[-1].host_version == s.version and x == list(st.ssh[1][:5])
    Prog.layerThickness[-1][3] = Prog.layerThickness[-1][3] + 0.25
    st.h[1][3] += 0.25
    timestep = shim.zeros(Prog.ssh[-1].backend, np.float64, (1,))
    timestep[0] = dts
    shim.ocn_timestep(timestep, Prog, Diag, Tend, Setup, shim.ForwardEuler, backend=Prog.ssh[-1].backend)
    st.step_fe(dts)
    assert np.array_equal(f(Prog.layerThickness[-1]), st.h[1]) and np.array_equal(f(Prog.ssh[-1]), st.ssh[1])
    # the (sumCPU, sumGPU, ...) form of ocn_run_loop (run_loop.jl:26-45) on one more hour
    from moka_hip.timemanager import OneTimeAlarm, attachAlarm
    sim2 = OneTimeAlarm("simulation_end2", clock.currTime + dt.timedelta(hours=1))
    attachAlarm(clock, sim2)
    sumCPU, sumGPU = np.zeros(1), shim.zeros(Prog.ssh[-1].backend, np.float64, (1,))
    got = shim.ocn_run_loop(sumCPU, sumGPU, timestep, Prog, Diag, Tend, Setup, shim.ForwardEuler, clock, sim2,
                            clock.alarms["outputAlarm"], backend=Prog.ssh[-1].backend)
    for _ in range(int(3600 / dts)):
        st.step_fe(dts)
    assert got == st.sum_sq_ssh() == sumCPU[0]
    s.close(); s.mesh.close()


# ------------------------------------------------------------------------------------------------
# reverse mode of the Forward-Euler loop: gradients of sum(ssh^2) from the HIP tape +
# transposed kernels, bit for bit against the oracle's adjoint, whose own pin is finite differences
# (tests/test_oracle_adjoint.py, the reference's test/enzyme/test_Enzyme_end2end.jl check)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("meshname,K,flags,nsteps", [("igw200", 1, 7, 12), ("igw200", 1, 0, 12), ("ico16", 1, 7, 6), ("ico16", 3, 3, 5),
                                                       ("ico16", 60, 0, 3), ("ico12f", 5, 1, 4), ("ico16", 70, 3, 2),
                                                       ("ico16", 60, 3, 3), ("ico12f", 40, 3, 3), ("ico16", 34, 1, 3)])
def test_fe_adjoint_bitwise(backend, meshname, K, flags, nsteps):
    mesh = get_mesh(meshname)
    if meshname == "igw200":
        ssh, u, h, rest = mg.igw_initial_state(mesh)
        u, h, rest = u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1), np.asarray(rest).reshape(mesh.nCells, 1)
        dtv = 400.0
    else:
        ssh, u, h, rest = random_state(mesh, K, 31 + K)
        dtv = 20.0
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    tape = mk.AdjointTape(Prog, nsteps)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjoint(st)
    for _ in range(nsteps):
        tape.step(np.array([dtv]), flags)
        adj.step_fe(dtv, flags)
    assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1])           # the taped step is the ordinary step
    with pytest.raises(mk.MokaError):
        tape.step(dtv, flags)                                      # tape is full
    g = tape.gradient()
    gS, gU, gH, gE = adj.gradient_sum_sq_ssh()
    assert np.array_equal(g["ssh"], gS)
    assert np.array_equal(g["normalVelocity"], gU)
    assert np.array_equal(g["layerThickness"], gH)
    assert np.array_equal(g["layerThicknessEdge"], gE)
    assert np.abs(gU).max() > 0 and np.abs(gH).max() > 0
    # the tape is consumed: it can record again, and the state keeps stepping
    tape.step(dtv, flags)
    adj2 = orc.OracleAdjoint(st)
    adj2.step_fe(dtv, flags)
    g2, o2 = tape.gradient(), adj2.gradient_sum_sq_ssh()
    assert np.array_equal(g2["normalVelocity"], o2[1]) and np.array_equal(g2["layerThickness"], o2[2])
    tape.close(); Prog._state.close(); Setup.mesh.close()


def test_fe_adjoint_central_difference_through_the_c_abi(backend):
    """The reference's own check (test_Enzyme_end2end.jl:128-176), on the GPU path: perturb one entry of the initial
    layerThickness / normalVelocity, rerun, compare (J+ - J-)/dist with the adjoint entry (atol 1e-4 / 1e-2)."""
    mesh = get_mesh("igw200")
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    u, h, rest = u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1), np.asarray(rest).reshape(mesh.nCells, 1)
    dtv, nsteps, k = 400.0, 15, 4

    def run(uu, hh, want_grad=False):
        Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, uu, hh, rest, CONFIG, backend, multilayer=True)
        tape = mk.AdjointTape(Prog, nsteps)
        for _ in range(nsteps):
            tape.step(dtv)
        tot = L.C.c_double()
        L.check(L.lib().moka_sum_sq(Prog._state._h, L.F_SSH, 1, L.C.byref(tot)), backend._h)
        g = tape.gradient() if want_grad else None
        tape.close(); Prog._state.close(); Setup.mesh.close()
        return tot.value, g

    _, g = run(u, h, True)
    for arr, name, atol in ((h, "layerThickness", 1e-4), (u, "normalVelocity", 1e-2)):
        eps = abs(arr[k, 0]) * 1e-6
        p, m_ = arr.copy(), arr.copy()
        p[k, 0] += eps
        m_[k, 0] -= eps
        Jp, _ = run(*( (u, p) if arr is h else (p, h) ))
        Jm, _ = run(*( (u, m_) if arr is h else (m_, h) ))
        fd = (Jp - Jm) / (p[k, 0] - m_[k, 0])
        assert abs(fd - g[name][k, 0]) <= atol, (name, fd, g[name][k, 0])


def test_closing_a_state_takes_its_tapes_along(backend):
    """An explicit close() of the state while a tape on it is alive: the tape is destroyed first (its destructor dereferences
    the state), not later by the garbage collector."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, 8, 3)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    tape = mk.AdjointTape(Prog, 2)
    tape.step(20.0, method=mk.RungeKutta4)
    Prog._state.close()
    assert not tape._h
    tape.close()                     # idempotent
    Setup.mesh.close()
    import gc
    gc.collect()


def test_adjoint_refuses_what_it_does_not_cover(backend):
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, 4, 2)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    tape = mk.AdjointTape(Prog, 2)
    with pytest.raises(mk.MokaError):
        tape.step(10.0, 7)                          # level-1-only stepping with K > 1
    tape.close(); Prog._state.close(); Setup.mesh.close()
    S2, D2, T2, P2 = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=4)
    with pytest.raises(mk.MokaError):
        mk.AdjointTape(P2, 2)                       # fp32-storage state
    P2._state.close(); S2.mesh.close()


@pytest.mark.parametrize("meshname,K,nsteps", [("igw200", 1, 6), ("ico16", 3, 4), ("ico16", 60, 2), ("ico12f", 5, 3), ("ico16", 70, 2),
                                                 ("ico12f", 34, 2), ("ico16", 64, 2)])
def test_rk4_adjoint_bitwise(backend, meshname, K, nsteps):
    mesh = get_mesh(meshname)
    if meshname == "igw200":
        ssh, u, h, rest = mg.igw_initial_state(mesh)
        u, h, rest = u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1), np.asarray(rest).reshape(mesh.nCells, 1)
        dtv = 400.0
    else:
        ssh, u, h, rest = random_state(mesh, K, 41 + K)
        dtv = 20.0
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    tape = mk.AdjointTape(Prog, nsteps)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjointRK4(st)
    for _ in range(nsteps):
        tape.step(dtv, method=mk.RungeKutta4)
        adj.step_rk4(dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
    with pytest.raises(mk.MokaError):
        tape.step(dtv)                                             # one integrator per tape (and it is full)
    g = tape.gradient()
    gU, gH = adj.gradient_sum_sq_ssh()
    assert np.array_equal(g["normalVelocity"], gU)
    assert np.array_equal(g["layerThickness"], gH)
    assert not g["ssh"].any() and not g["layerThicknessEdge"].any()
    assert np.abs(gU).max() > 0 and np.abs(gH).max() > 0
    # an emptied tape may switch integrator
    tape.step(dtv, 0)
    a2 = orc.OracleAdjoint(st)
    a2.step_fe(dtv, 0)
    assert np.array_equal(tape.gradient()["layerThickness"], a2.gradient_sum_sq_ssh()[2])
    tape.close()
    # The stage-4 tendencies of a taped RK4 step are produced lazily from the TAPE's copy of the last provisional state
    # (stage 3 writes it into the tape, not into the RK buffer): read them (a) while the tape lives, (b) after it is gone.
    for close_first in (False, True):
        t2 = mk.AdjointTape(Prog, 2)
        t2.step(dtv, method=mk.RungeKutta4)
        t2.step(dtv, method=mk.RungeKutta4)
        st.step_rk4(dtv); st.step_rk4(dtv)
        if close_first:
            t2.close()
        assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
        assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
        if not close_first:
            t2.close()
    Prog._state.close(); Setup.mesh.close()


# ------------------------------------------------------------------------------------------------
# optional nonlinear terms: NOT in the reference, parity unpinned; the HIP
# kernels reproduce the oracle's restatement of the scheme bit for bit and keep Williamson test case 2 steady
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("meshname,K,nsteps", [("ico16", 1, 4), ("ico16", 3, 3), ("ico16", 60, 2), ("planar", 4, 3), ("ico12f", 5, 2),
                                               ("ico16", 70, 2), ("ico12f", 40, 2), ("planar", 64, 2), ("ico16", 34, 2)])
def test_nonlinear_tendency_and_rk4_bitwise(backend, meshname, K, nsteps):
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 51 + K)
    dtv = 2.0 if meshname == "planar" else 20.0
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    nl = orc.OracleNonlinear(om)
    # default is the reference's linear form
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    assert np.array_equal(Tend.tendNormalVelocity.get(), om.tendencies_clean(u, h)[0])
    mk.set_nonlinear(Prog, True)
    tu, th, ossh, _ = nl.tendencies(u, h)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu)
    assert np.array_equal(Tend.tendLayerThickness.get(), th)
    assert np.array_equal(Prog.ssh[-1].get(), ossh)
    st = orc.OracleState(om, ssh, u, h)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
    for _ in range(nsteps):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        nl.step_rk4(st, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1])
    assert np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU)      # stage-4 tendencies, lazily, nonlinear too
    mk.run_steps(Prog, mk.RungeKutta4, dtv, 6)                          # graph replay
    for _ in range(6):
        nl.step_rk4(st, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    with pytest.raises(mk.MokaError):
        mk.ocn_timestep(np.array([dtv]), Prog, Diag, Tend, Setup, mk.ForwardEuler)
    with pytest.raises(mk.MokaError):
        mk.AdjointTape(Prog, 2)
    mk.set_nonlinear(Prog, False)                                       # and back to the reference's terms
    lin = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    lin.step_rk4(dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), lin.u[1])
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("variant,shape", [(0, 0), (0, 1), (0, 2), (0, 3), (0, 10), (4, 0), (3, 0)])
@pytest.mark.parametrize("meshname,K,visc", [("ico16", 60, 0.0), ("planar", 64, 1.0), ("ico32", 34, 0.0)])
def test_nonlinear_kernel_forms_with_partial_edge_masks(backend, meshname, K, visc, variant, shape):
    """The forms of the nonlinear kernels (variant 0: patch kernels -- launch shape 0 / 2 / 3: potential vorticity of the patch's
    vertices in LDS, 1: q_e of its edge rows in LDS (moka_set_tuning key 5); variant 4: patch kernels gathering the vertex potential
    vorticity; 3: generic lane-group kernels) against the oracle, with maxLevelEdgeTop < K on a third of the edges (the masks of
    horizontal_advection.jl:63 and the edge loop) and, on one mesh, Del2 mixing on top.  Shape 10 = shape 0 with only 40 vertex rows
    resident, so that every patch takes the path of the few patches of a large mesh that list more vertices than the LDS holds."""
    L.check(L.lib().moka_set_tuning(5, shape % 10))
    L.check(L.lib().moka_set_tuning(6, 40 if shape == 10 else 0))
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 91 + K)
    dtv = 2.0 if meshname == "planar" else 20.0
    rng = np.random.default_rng(7)
    mlt = np.where(rng.random(mesh.nEdges) < 0.33, rng.integers(0, K + 1, mesh.nEdges), K).astype(np.int32)
    hm = mk.HorzMesh(mesh)
    vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest)
    vm.maxLevelEdge.Top[:] = mlt
    backend.set_kernel_variant(variant)
    try:
        M = mk.Mesh(hm, vm, backend=backend)
        Prog = mk.PrognosticVars(ssh, u, h, 2, M)
        Diag, Tend = mk.DiagnosticVars(None, M, Prog._state), mk.TendencyVars(None, M, Prog._state)
        om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
        v = visc * 0.01 * float(mesh.dcEdge.min()) ** 2 / dtv
        nl = orc.OracleNonlinear(om, visc_del2=v) if v else orc.OracleNonlinear(om)
        mk.set_nonlinear(Prog, True, visc_del2=v)
        tu, th, ossh, _ = nl.tendencies(u, h)
        mk.computeTendency(M, Diag, Prog, Tend)
        assert np.array_equal(Tend.tendNormalVelocity.get(), tu)
        assert np.array_equal(Tend.tendLayerThickness.get(), th)
        assert np.array_equal(Prog.ssh[-1].get(), ossh)
        st = orc.OracleState(om, ssh, u, h)
        mk.run_steps(Prog, mk.RungeKutta4, dtv, 3)
        for _ in range(3):
            nl.step_rk4(st, dtv)
        assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1])
        assert np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
        assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
        Prog._state.close(); M.close()
    finally:
        backend.set_kernel_variant(0)
        L.check(L.lib().moka_set_tuning(5, 0))
        L.check(L.lib().moka_set_tuning(6, 0))


@pytest.mark.parametrize("meshname,K,visc,served", [("ico16", 60, 0.0, True), ("planar", 64, 1.0, True), ("ico32", 34, 0.0, True),
                                                    ("ico12f", 40, 1.0, False), ("ico16", 70, 0.0, False), ("ico16", 3, 0.0, False)])
def test_nonlinear_rk4_13_stream_form_bitwise_against_its_twin(backend, meshname, K, visc, served):
    """moka_set_tuning(7, 1) on a state with the nonlinear terms: where the stage launch is k_stage_nl5 (even 34 <= K <= 64) the RK4
    step runs in the 13-stream form (StageArgs.rkMode 9 in the last stage; not on ico12f, whose heptagons take the generic kernels) and equals its twin oracle_step_rk4_nonlinear_s13 bit for
    bit -- both levels, the lazily produced stage-4 tendencies, graph replay; elsewhere the key changes nothing (the reference's
    running sum).  Key cleared: the running sum again."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 61 + K)
    dtv = 2.0 if meshname == "planar" else 20.0
    v = visc * 0.01 * float(mesh.dcEdge.min()) ** 2 / dtv
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    nl = orc.OracleNonlinear(om, visc_del2=v) if v else orc.OracleNonlinear(om)
    mk.set_nonlinear(Prog, True, visc_del2=v)
    st = orc.OracleState(om, ssh, u, h)
    step = (lambda: nl.step_rk4_s13(st, dtv)) if served else (lambda: nl.step_rk4(st, dtv))
    lib = L.lib()
    assert lib.moka_state_rk4_streams(Prog._state._h) == 16 and lib.moka_state_rk4_streams(None) == 0
    L.check(lib.moka_set_tuning(7, 1))
    try:
        assert lib.moka_state_rk4_streams(Prog._state._h) == (13 if served else 16)
        mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
        for _ in range(2):
            mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
            step()
        for lev in (0, 1):
            assert np.array_equal(Prog.normalVelocity[lev].get(), st.u[lev]), lev
            assert np.array_equal(Prog.layerThickness[lev].get(), st.h[lev]), lev
            assert np.array_equal(Prog.ssh[lev].get(), st.ssh[lev]), lev
        assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
        mk.run_steps(Prog, mk.RungeKutta4, dtv, 11)                     # eager first step + a replayed period + remainder
        for _ in range(11):
            step()
        assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
        assert np.array_equal(Prog.ssh[-1].get(), st.ssh[1]) and np.array_equal(Prog.normalVelocity[0].get(), st.u[0])
        if served:                                                      # and it is NOT the running sum's round-off
            ref = orc.OracleState(om, ssh, u, h)
            for _ in range(13):
                nl.step_rk4(ref, dtv)
            assert not np.array_equal(ref.u[1], st.u[1])
            assert np.max(np.abs(ref.u[1] - st.u[1])) <= 1e-10 * np.max(np.abs(ref.u[1]))
    finally:
        L.check(lib.moka_set_tuning(7, 0))
    mk.run_steps(Prog, mk.RungeKutta4, dtv, 2)
    nl.step_rk4(st, dtv); nl.step_rk4(st, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("meshname,K,nsteps", [("ico16", 1, 3), ("ico16", 60, 2), ("planar", 4, 3), ("ico12f", 5, 2), ("ico16", 70, 2),
                                               ("ico12f", 40, 2), ("planar", 34, 2)])
def test_del2_mixing_bitwise(backend, meshname, K, nsteps):
    """Del2 momentum mixing (the reference's uncalled sketch, horizontal_momentum_mixing.jl:53-80) on top of the nonlinear
    terms: tendencies, RK4 steps and graph replay against the oracle's restatement, bit for bit."""
    mesh = get_mesh(meshname)
    ssh, u, h, rest = random_state(mesh, K, 77 + K)
    dtv = 2.0 if meshname == "planar" else 20.0
    visc = 0.01 * float(mesh.dcEdge.min()) ** 2 / dtv
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    with pytest.raises(mk.MokaError, match="moka_set_nonlinear"):
        L.check(L.lib().moka_set_viscosity_del2(Prog._state._h, visc), Prog._state.mesh.backend._h)
    mk.set_nonlinear(Prog, True, visc_del2=visc)
    nl = orc.OracleNonlinear(om, visc_del2=visc)
    tu, th, ossh, _ = nl.tendencies(u, h)
    mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
    assert np.array_equal(Tend.tendNormalVelocity.get(), tu)
    assert not np.array_equal(tu, orc.OracleNonlinear(om).tendencies(u, h)[0])
    assert np.array_equal(Tend.tendLayerThickness.get(), th) and np.array_equal(Prog.ssh[-1].get(), ossh)
    st = orc.OracleState(om, ssh, u, h)
    mk.changeTimeStep(Setup.timeManager, dt.timedelta(seconds=dtv))
    for _ in range(nsteps):
        mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
        nl.step_rk4(st, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.layerThickness[-1].get(), st.h[1])
    assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU)
    mk.run_steps(Prog, mk.RungeKutta4, dtv, 5)                          # graph replay
    for _ in range(5):
        nl.step_rk4(st, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st.u[1]) and np.array_equal(Prog.ssh[-1].get(), st.ssh[1])
    mk.set_nonlinear(Prog, True, visc_del2=0.0)                         # mixing off again: the plain nonlinear form
    plain = orc.OracleNonlinear(om)
    st2 = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
    mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)
    plain.step_rk4(st2, dtv)
    assert np.array_equal(Prog.normalVelocity[-1].get(), st2.u[1])
    Prog._state.close(); Setup.mesh.close()


def test_nonlinear_needs_the_extra_mesh_arrays(backend):
    import dataclasses
    mesh = dataclasses.replace(get_mesh("ico16"), kiteAreasOnVertex=None)
    ssh, u, h, rest = random_state(mesh, 2, 3)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    with pytest.raises(mk.MokaError, match="kiteAreasOnVertex"):
        mk.set_nonlinear(Prog, True)
    Prog._state.close(); Setup.mesh.close()


def test_nonlinear_keeps_williamson_tc2_steady_on_the_gpu(backend):
    """Property, not parity: solid-body rotation balanced by its height field is a steady state of the nonlinear
    equations; after 60 RK4 steps the thickness has drifted several times less than with the reference's linear terms."""
    from test_oracle_nonlinear import tc2_state
    mesh = get_mesh("ico16")
    u, h = tc2_state(mesh)
    rest = np.full((mesh.nCells, 1), 2998.0)
    ssh = h[:, 0] - rest[:, 0]
    dtv = 0.3 * float(mesh.dcEdge.min()) / math.sqrt(9.80616 * 3000.0)
    drift = {}
    for nonlinear in (True, False):
        Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
        mk.set_nonlinear(Prog, nonlinear)
        mk.run_steps(Prog, mk.RungeKutta4, dtv, 60)
        drift[nonlinear] = np.abs(Prog.layerThickness[-1].get() - h).max()
        Prog._state.close(); Setup.mesh.close()
    assert np.isfinite(drift[True]) and drift[True] < 0.25 * drift[False], drift


def test_round4_entry_points_arguments_and_edges(backend):
    """moka_state_download_rows / moka_state_array_address / moka_state_optimize_placement(max_tries <= 0) /
    moka_state_placement_launches: argument checks (nothing aborts: a negative status and a message) and the edge cases."""
    mesh = get_mesh("ico16")
    K = 60
    ssh, u, h, rest = random_state(mesh, K, 3)
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True)
    lib, hS, ctx = L.lib(), Prog._state._h, backend._h
    U = Prog.normalVelocity[-1]
    ids = np.array([0, mesh.nEdges - 1, 5, 5], dtype=np.int32)                 # any order, repeats allowed
    assert np.array_equal(U.rows(ids), u[ids]) and np.array_equal(Prog.ssh[-1].rows([3, 1]), ssh[[3, 1]])
    assert U.rows(np.zeros(0, dtype=np.int32)).shape == (0, K)
    out = np.empty((1, K))
    bad = np.array([mesh.nEdges], dtype=np.int32)
    assert lib.moka_state_download_rows(hS, L.F_NORMAL_VELOCITY, 1, 1, L.i32(bad), L.f64(out)) == L.ERR_ARG
    assert b"out of range" in lib.moka_last_error(ctx)
    assert lib.moka_state_download_rows(hS, L.F_NORMAL_VELOCITY, 2, 1, L.i32(ids), L.f64(out)) == L.ERR_ARG     # no RK4 step yet
    assert lib.moka_state_download_rows(hS, L.F_NORMAL_VELOCITY, 4, 1, L.i32(ids), L.f64(out)) == L.ERR_ARG
    assert lib.moka_state_download_rows(hS, L.F_THICKNESS_FLUX, 3, 1, L.i32(ids), L.f64(out)) == L.ERR_ARG
    assert lib.moka_state_download_rows(None, L.F_SSH, 1, 1, L.i32(ids), L.f64(out)) == L.ERR_ARG
    a = C.c_uint64(1)
    L.check(lib.moka_state_array_address(hS, L.F_NORMAL_VELOCITY, 2, C.byref(a)), ctx)
    assert a.value == 0                                                           # the provisional states do not exist yet
    L.check(lib.moka_state_array_address(hS, L.F_LAYER_THICKNESS, 1, C.byref(a)), ctx)
    assert a.value != 0 and a.value % 16 == 0
    assert lib.moka_state_array_address(hS, L.F_THICKNESS_FLUX, 1, C.byref(a)) == L.ERR_ARG
    rep = Prog._state.optimize_placement(0)                                       # measures only
    assert rep["tries"] == 0 and rep["ms_before"] == rep["ms_after"] > 0.0 and rep["stage_launches"] >= 48 and rep["stage_launches"] % 24 == 0
    L.check(lib.moka_state_array_address(hS, L.F_NORMAL_VELOCITY, 2, C.byref(a)), ctx)
    assert a.value != 0                                                           # (the measurement allocated them)
    assert np.array_equal(U.get(), u)
    mk.run_steps(Prog, mk.RungeKutta4, 20.0, 1)
    assert U.rows(ids, level=2).shape == (4, K) and np.all(np.isfinite(U.rows(ids, level=3)))
    Prog._state.close(); Setup.mesh.close()


@pytest.mark.parametrize("sbytes,K", [(8, 60), (4, 80)])
def test_optimize_placement_through_the_c_abi(backend, sbytes, K):
    """moka_state_optimize_placement (include/moka_hip.h): arrays of the state are re-allocated one at a time where that makes the
    RK4 stage launches faster.  Whatever it does, the state is the state: every array of Prog (BOTH time levels, an ssh that is not
    the column sum of layerThickness included), Diag and Tend reads back bit for bit as before the call, in the middle of a
    Forward-Euler run with lazily pending arrays too; the reference's stale-thickness steps (which carry layerThicknessEdge
    and the previous level across steps) and RK4 steps that follow equal the oracle's; ms_after <= ms_before; and once a tape
    or a halo holds the arrays' addresses the call refuses (round 3's search lived in Python, over whole candidate
    states, and left the traces of its trial steps in the state it kept)."""
    mesh = get_mesh("ico16")
    ssh, u, h, rest = random_state(mesh, K, 11)
    ssh = ssh + 0.125                                   # inconsistent with layerThickness on purpose: the caller's array must survive
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, CONFIG, backend, multilayer=True, state_bytes=sbytes)
    rng = np.random.default_rng(5)
    r32 = (lambda x: x.astype(np.float32).astype(np.float64)) if sbytes == 4 else (lambda x: x)
    ssh0, u0, h0 = r32(ssh - 0.25), r32(u + 0.5 * rng.uniform(-1, 1, u.shape)), r32(h + 0.125)      # a previous level of its own
    Prog.ssh[0].set(ssh0); Prog.normalVelocity[0].set(u0); Prog.layerThickness[0].set(h0)
    he = r32(1000.0 / K + rng.uniform(-1, 1, (mesh.nEdges, K)))
    Diag.layerThicknessEdge.set(he)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h, mixed=sbytes == 4)
    st.ssh[0][:] = ssh0; st.u[0][:] = u0; st.h[0][:] = h0; st.hEdge[:] = he
    lib, hS = L.lib(), Prog._state._h

    def prog_fields():
        return {"ssh0": Prog.ssh[0].get(), "ssh1": Prog.ssh[-1].get(), "u0": Prog.normalVelocity[0].get(), "u1": Prog.normalVelocity[-1].get(),
                "h0": Prog.layerThickness[0].get(), "h1": Prog.layerThickness[-1].get()}

    # (an fp32-storage state has DiagnosticVars after Forward-Euler steps only: prognostic fields here, everything below)
    snapshot = (lambda: all_fields(Prog, Diag, Tend)) if sbytes == 8 else prog_fields
    before = snapshot()
    rep = Prog._state.optimize_placement(6)
    assert rep["tries"] <= 6 and rep["ms_after"] <= rep["ms_before"] and rep["ms_before"] > 0.0
    assert all(t["field"] in mk.api._State.FIELD_NAMES and (t["ms_new"] < t["ms_old"]) >= t["kept"] for t in rep["trials"])
    after = snapshot()
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    # the reference's live step, carrying layerThicknessEdge and the previous level across steps: three steps, then once more
    # in the middle of the run (the arrays of the last lean step are pending when the call comes)
    for _ in range(3):
        L.check(lib.moka_step_fe(hS, 20.0, 3), backend._h)
        st.step_fe(20.0, 3)
    assert lib.moka_fe_lazy_pending(hS) == 1
    rep2 = Prog._state.optimize_placement(4)
    assert rep2["ms_after"] <= rep2["ms_before"]
    for _ in range(2):
        L.check(lib.moka_step_fe(hS, 20.0, 3), backend._h)
        st.step_fe(20.0, 3)
    got, exp = all_fields(Prog, Diag, Tend), oracle_fields(st)
    for k in exp:
        assert np.array_equal(got[k], exp[k]), k
    # RK4 steps from here, and a third call between them (the stage-4 tendencies of the step before are pending then)
    mk.run_steps(Prog, mk.RungeKutta4, 20.0, 1); st.step_rk4(20.0)
    Prog._state.optimize_placement(3)
    mk.run_steps(Prog, mk.RungeKutta4, 20.0, 2); st.step_rk4(20.0); st.step_rk4(20.0)
    got, exp = prog_fields(), oracle_fields(st)
    for k in got:
        assert np.array_equal(got[k], exp[k]), k
    if sbytes == 8:
        assert np.array_equal(Tend.tendNormalVelocity.get(), st.tendU) and np.array_equal(Tend.tendLayerThickness.get(), st.tendH)
        tape = mk.AdjointTape(Prog, 2)                   # holds the arrays' addresses: the state stays where it is from now on
        with pytest.raises(mk.MokaError) as ei:
            Prog._state.optimize_placement(2)
        assert ei.value.code == L.ERR_UNSUPPORTED
        tape.close()
        Prog._state.optimize_placement(1)               # ... and may move again once the tape is gone
    Prog._state.close(); Setup.mesh.close()


