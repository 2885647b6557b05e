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
