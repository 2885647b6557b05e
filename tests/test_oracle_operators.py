"""Pins the CPU oracle to the reference's own known-answer vectors (test/ocn/test_Operators.jl)."""
import json
import os

import numpy as np
import pytest

import oracle as orc
from analytic import PlanarSetup, areas, error_measures
from moka_hip import meshgen

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "operator_norms.json")))


@pytest.fixture(scope="module")
def setup48():
    g = GOLD["mesh"]
    mesh = meshgen.planar_hex_mesh(g["nx"], g["ny"], g["dc"])
    K = g["nVertLevels"]
    return mesh, K, orc.OracleMesh(mesh, K), PlanarSetup(mesh, K)


def test_gradient_known_answer(setup48):
    mesh, K, om, ts = setup48
    linf, l2 = error_measures(om.gradient_on_edge(ts.h()), ts.grad_h_edge(), areas(mesh)["edge"])
    assert abs(linf - GOLD["grad"]["L_inf"]) < GOLD["atol"]
    assert abs(l2 - GOLD["grad"]["L_two"]) < GOLD["atol"]


def test_divergence_known_answer(setup48):
    mesh, K, om, ts = setup48
    linf, l2 = error_measures(om.divergence_on_cell(ts.F_edge()), ts.div_F(), areas(mesh)["cell"])
    assert abs(linf - GOLD["div"]["L_inf"]) < GOLD["atol"]
    assert abs(l2 - GOLD["div"]["L_two"]) < GOLD["atol"]


def test_curl_known_answer(setup48):
    mesh, K, om, ts = setup48
    linf, l2 = error_measures(om.curl_on_vertex(ts.F_edge()), ts.curl_F(), areas(mesh)["vertex"])
    assert abs(linf - GOLD["curl"]["L_inf"]) < GOLD["atol"]
    assert abs(l2 - GOLD["curl"]["L_two"]) < GOLD["atol"]


def test_curl_accumulates(setup48):
    """CurlOnVertex adds into the caller's array (Operators.jl:135 zeroing is commented out)."""
    mesh, K, om, ts = setup48
    once = om.curl_on_vertex(ts.F_edge())
    twice = om.curl_on_vertex(ts.F_edge(), curl=once.copy())
    assert np.allclose(twice, 2 * once, rtol=1e-12, atol=1e-14)


def test_all_levels_identical(setup48):
    mesh, K, om, ts = setup48
    g = om.gradient_on_edge(ts.h())
    assert np.all(g == g[:, :1])


def test_interp_level1_only(setup48):
    """interpolateCell2Edge writes k = 1 only (Operators.jl:207-208)."""
    mesh, K, om, ts = setup48
    out = np.full((mesh.nEdges, K), 7.0)
    om.interpolate_cell2edge(ts.h(), nlev=1, out=out)
    assert np.all(out[:, 1:] == 7.0) and not np.any(out[:, 0] == 7.0)


def test_ksum_is_plain_value_for_one_level():
    assert orc.ksum(np.array([3.25])) == 3.25
    x = np.random.default_rng(1).standard_normal(60)
    assert abs(orc.ksum(x) - x.sum()) < 1e-13


def test_tangential_reconstruction_uniform_flow():
    """weightsOnEdge reconstruct k x n component exactly for uniform flow (SURVEY App. B check);
    pins the sign of the Coriolis term of horizontal_advection_and_coriolis.jl:69-73."""
    mesh = meshgen.planar_hex_mesh(8, 8, 1000.0, f0=1e-4)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=np.zeros(mesh.nCells))
    U = np.array([0.3, -0.7])
    nx, ny = np.cos(mesh.angleEdge), np.sin(mesh.angleEdge)
    un = (U[0] * nx + U[1] * ny)[:, None]
    h = np.full((mesh.nCells, 1), 10.0)
    tu, th, ssh = om.tendencies_clean(un, h)
    ut = -U[0] * ny + U[1] * nx
    assert np.abs(tu[:, 0] - 1e-4 * ut).max() < 1e-18 + 1e-15
    assert np.abs(th).max() < 1e-15


def test_ksum_order_against_a_plain_serial_sum():
    """N3 (a build decision): ssh[c] = sum_k h[k,c] - restingThicknessSum[c] with the sum taken in the order a wavefront
    reduction produces (64 strided partials, then an XOR butterfly), so that the CPU oracle and the GPU agree bit for bit.
    A Julia restatement would more naturally write the plain serial sum over k.  This test states how far the two are
    apart: they are identical for K <= 2, and within K/2 units in the last place of the result for layer thicknesses of
    one sign (the case of the model: h > 0) -- far below the 1e-12 relative tolerance BASELINE.md asks for."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for K in (1, 2, 3, 7, 10, 33, 60, 64, 65, 80, 100, 128):
        for _ in range(200):
            col = rng.uniform(10.0, 70.0, K)                # layer thicknesses, metres
            serial = 0.0
            for x in col:
                serial += x                                  # the plain left-to-right sum
            got = orc.ksum(col)
            if K <= 2:
                assert got == serial
            ulp = np.spacing(abs(serial))
            worst = max(worst, abs(got - serial) / ulp)
            assert abs(got - serial) <= 0.5 * K * ulp
            assert abs(got - serial) <= 1e-13 * abs(serial)
    assert worst >= 1.0                                      # they do differ in the last place: the order is a decision


# ------------------------------------------------------------------------------------------------
# Reverse and forward mode of the stand-alone operators: the reference's test/enzyme/test_Enzyme_Operators.jl restated for
# the oracle twins.  Same mesh (48 x 48 planar periodic), same analytic fields, the same entries (input kBegin, output
# kEnd, linear indices into (nVertLevels, n) arrays), the same check: AD against central differences with a relative
# eps = 1e-8 and atol = 1e-6 (:102-127, :196-221).
# ------------------------------------------------------------------------------------------------
def _lin(a, idx1):
    """Julia linear index (1-based) into a (nVertLevels, n) array == flat index into our (n, K) C-order array."""
    return np.unravel_index(idx1 - 1, a.shape)


def _fd(fun, x, k_in, k_out, eps=1e-8):
    xp, xm = x.copy(), x.copy()
    xp[k_in] += abs(xp[k_in]) * eps
    xm[k_in] -= abs(xm[k_in]) * eps
    return (fun(xp)[k_out] - fun(xm)[k_out]) / (xp[k_in] - xm[k_in])


@pytest.mark.parametrize("K", [1, 10])
def test_gradient_reverse_and_forward_mode_against_central_differences(K):
    g = GOLD["mesh"]
    mesh = meshgen.planar_hex_mesh(g["nx"], g["ny"], g["dc"])
    om, ts = orc.OracleMesh(mesh, K), PlanarSetup(mesh, K)
    scalar = ts.h()
    for k_begin, k_end in ((1, 1), (K + 1, 3 * K + 1), (2 * K, 4 * K)):       # (1, 1) is the reference's pair (:58-59)
        kin, kout = _lin(scalar, k_begin), _lin(np.zeros((mesh.nEdges, K)), k_end)
        # reverse: d_grad[kEnd] = 1 -> d_scalar[kBegin]                                  (:58-68)
        d_grad, d_scalar = np.zeros((mesh.nEdges, K)), np.zeros((mesh.nCells, K))
        d_grad[kout] = 1.0
        om.gradient_on_edge_vjp(d_scalar, d_grad)
        rev = d_scalar[kin]
        assert not d_grad.any()                                                          # overwritten output: shadow zeroed
        # forward: d_scalar[kBegin] = 1 -> d_grad[kEnd] (the operator is linear: its tangent map is itself)   (:80-99)
        t = np.zeros((mesh.nCells, K)); t[kin] = 1.0
        fwd = om.gradient_on_edge(t)[kout]
        fd = _fd(om.gradient_on_edge, scalar, kin, kout)
        assert abs(rev - fd) < 1e-6 and abs(fwd - fd) < 1e-6, (k_begin, k_end, rev, fwd, fd)
        if k_begin == 1:
            assert rev != 0.0                                                            # cell 1 is a cell of edge 1 on this mesh


@pytest.mark.parametrize("K", [1, 10])
def test_divergence_reverse_and_forward_mode_against_central_differences(K):
    g = GOLD["mesh"]
    mesh = meshgen.planar_hex_mesh(g["nx"], g["ny"], g["dc"])
    om, ts = orc.OracleMesh(mesh, K), PlanarSetup(mesh, K)
    vec = ts.F_edge()
    for k_begin, k_end in ((2, 1), (1, 1), (2 * K + 1, 1), (5 * K, K)):                  # (2, 1) is the reference's pair (:155-156)
        kin, kout = _lin(vec, k_begin), _lin(np.zeros((mesh.nCells, K)), k_end)
        d_div, d_vec, d_temp = np.zeros((mesh.nCells, K)), np.zeros((mesh.nEdges, K)), np.zeros((mesh.nEdges, K))
        d_div[kout] = 1.0
        om.divergence_on_cell_vjp(d_vec, d_temp, d_div)
        rev = d_vec[kin]
        assert not d_div.any() and not d_temp.any()
        t = np.zeros((mesh.nEdges, K)); t[kin] = 1.0
        fwd = om.divergence_on_cell(t)[kout]
        fd = _fd(om.divergence_on_cell, vec, kin, kout)
        assert abs(rev - fd) < 1e-6 and abs(fwd - fd) < 1e-6, (k_begin, k_end, rev, fwd, fd)
        if (k_begin, k_end) == (2, 1) and K == 1:
            assert rev != 0.0                                                            # edge 2 is an edge of cell 1 on this mesh


def test_operator_transposes_satisfy_the_adjoint_identity():
    """<J x, y> == <x, J^T y> for grad, div and curl on a sphere with pentagons (random x, y, K = 5): pins the three
    transposes independently of finite differences; accumulation conventions included (shadows of inputs add up)."""
    mesh = meshgen.icosahedral_mesh(6)
    K = 5
    om = orc.OracleMesh(mesh, K)
    rng = np.random.default_rng(8)
    xc, ye = rng.standard_normal((mesh.nCells, K)), rng.standard_normal((mesh.nEdges, K))
    xe, yc, yv = rng.standard_normal((mesh.nEdges, K)), rng.standard_normal((mesh.nCells, K)), rng.standard_normal((mesh.nVertices, K))
    rel = lambda a, b: abs(a - b) / max(abs(a), abs(b))
    # gradient (accumulation: the shadow of the input adds up)
    d_s, d_g = np.zeros((mesh.nCells, K)), ye.copy()
    om.gradient_on_edge_vjp(d_s, d_g)
    assert rel(np.vdot(om.gradient_on_edge(xc), ye), np.vdot(xc, d_s)) < 1e-12
    twice, d_g = d_s.copy(), ye.copy()
    om.gradient_on_edge_vjp(twice, d_g)
    assert np.allclose(twice, 2 * d_s, rtol=1e-9, atol=1e-18)      # a second call adds the same sum onto the first
    # divergence (through temp)
    d_v, d_t, d_d = np.zeros((mesh.nEdges, K)), np.zeros((mesh.nEdges, K)), yc.copy()
    om.divergence_on_cell_vjp(d_v, d_t, d_d)
    assert rel(np.vdot(om.divergence_on_cell(xe), yc), np.vdot(xe, d_v)) < 1e-12
    # a non-zero temp shadow is carried through P1^T as well: d_vec += d_temp * dvEdge
    d_v2, d_t2, d_d2 = np.zeros((mesh.nEdges, K)), ye.copy(), np.zeros((mesh.nCells, K))
    om.divergence_on_cell_vjp(d_v2, d_t2, d_d2)
    assert np.array_equal(d_v2, ye * mesh.dvEdge[:, None]) and not d_t2.any()
    # curl (accumulating primal: the output's shadow stays)
    d_v, d_c = np.zeros((mesh.nEdges, K)), yv.copy()
    om.curl_on_vertex_vjp(d_v, d_c)
    assert np.array_equal(d_c, yv)
    assert rel(np.vdot(om.curl_on_vertex(xe), yv), np.vdot(xe, d_v)) < 1e-12


@pytest.mark.parametrize("mixed", [False, True])
def test_row_sampled_oracle_equals_the_full_oracle(mixed):
    """oracle.SubMesh (the row-sampled oracle the full-size config-5 check uses): tendencies and one fused RK stage at sampled
    cells / edges, evaluated on the sub-mesh of their stencils, equal the full oracle's rows bit for bit -- stretched sphere
    (every cell shape, the twelve pentagons included), partial maxLevelEdgeTop, fp64 and fp32-storage forms."""
    mesh = meshgen.icosahedral_mesh(20, stretch=4.47)
    K = 8
    rng = np.random.default_rng(41)
    rest = np.full((mesh.nCells, K), 4000.0 / K)
    u = 0.1 * rng.uniform(-1, 1, (mesh.nEdges, K))
    h = rest + 0.3 * rng.uniform(-1, 1, (mesh.nCells, K))
    mlt = rng.integers(1, K + 1, mesh.nEdges).astype(np.int32)
    mlt[rng.random(mesh.nEdges) < 0.6] = K
    rnd = (lambda x: x.astype(np.float32).astype(np.float64)) if mixed else (lambda x: x)
    u, h = rnd(u), rnd(h)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=mlt)
    pent = np.flatnonzero(mesh.nEdgesOnCell == 5)
    cells = np.unique(np.concatenate([pent, rng.integers(0, mesh.nCells, 60), [0, mesh.nCells - 1]]))
    edges = np.unique(np.concatenate([mesh.edgesOnCell[pent].reshape(-1)[mesh.edgesOnCell[pent].reshape(-1) > 0] - 1,
                                      rng.integers(0, mesh.nEdges, 100), [0, mesh.nEdges - 1]]))
    sm = orc.SubMesh(mesh, cells, edges, K, rest.sum(1), max_level_edge_top=mlt)
    assert sm.cells.size < mesh.nCells // 2 and sm.edges.size < mesh.nEdges // 2          # a sub-mesh, not the mesh
    # full oracle: unrounded tendencies of the whole mesh (the python wrapper rounds the mixed ones: call the C function)
    tu, th = np.zeros_like(u), np.zeros_like(h)
    ssh = np.zeros(mesh.nCells)
    s1, s2 = np.zeros_like(u), np.zeros_like(u)
    fn = orc.lib().oracle_tendencies_mixed if mixed else orc.lib().oracle_tendencies_clean
    fn(om.ref, orc._p(tu), orc._p(th), orc._p(u), orc._p(h), orc._p(ssh), orc._p(s1), orc._p(s2))
    stu, sth, sssh = sm.tendencies(u[sm.edges], h[sm.cells], mixed)
    assert np.array_equal(stu, tu[edges]) and np.array_equal(sth, th[cells]) and np.array_equal(sssh, ssh[cells])
    # one fused stage from those
    a, b = 7.5, 2.5
    cu, ch = rnd(u + 0.01), rnd(h - 0.02)
    nu, nh = rnd(u - 0.03), rnd(h + 0.04)
    pu2, ph2, ssh2, nu2, nh2 = sm.rk_stage(u[sm.edges], h[sm.cells], cu[edges], ch[cells], nu[edges], nh[cells], a, b, mixed)
    assert np.array_equal(pu2, rnd(cu + a * tu)[edges]) and np.array_equal(ph2, rnd(ch + a * th)[cells])
    assert np.array_equal(nu2, rnd(nu + b * tu)[edges]) and np.array_equal(nh2, rnd(nh + b * th)[cells])
    full_ph2 = rnd(ch + a * th)
    assert np.array_equal(ssh2, rnd(om.update_ssh(full_ph2))[cells])
