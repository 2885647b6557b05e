"""Reverse mode of the Forward-Euler loop, CPU side.  Pinned the way the reference
pins its Enzyme adjoint (test/enzyme/test_Enzyme_end2end.jl): d sum(ssh^2) / d initial state against central
differences of the forward model -- here for many entries and through a dot-product identity, not one cell."""
import numpy as np
import pytest

import oracle as orc
from moka_hip import meshgen as mg


def forward_J(om, ssh, u, h, dt, nsteps, flags, hE0=None):
    st = orc.OracleState(om, ssh, u, h)
    if hE0 is not None:
        st.hEdge[...] = hE0
    for _ in range(nsteps):
        st.step_fe(dt, flags)
    return st.sum_sq_ssh()


def gradient(om, ssh, u, h, dt, nsteps, flags, hE0=None):
    st = orc.OracleState(om, ssh, u, h)
    if hE0 is not None:
        st.hEdge[...] = hE0
    adj = orc.OracleAdjoint(st)
    for _ in range(nsteps):
        adj.step_fe(dt, flags)
    return adj.gradient_sum_sq_ssh()


@pytest.mark.parametrize("flags", [orc.FE_REFERENCE_COMPAT, 0])
def test_igw_gradient_vs_central_differences_like_the_reference(flags):
    """The reference's check (cell 5, relative eps 1e-8, atol 1e-4 / 1e-2) on the 200 km IGW case, both FE variants."""
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=np.asarray(rest).reshape(mesh.nCells, -1).sum(1))
    dt, nsteps = 400.0, 20
    gS, gU, gH, gE = gradient(om, ssh, u, h, dt, nsteps, flags)
    u2, h2 = u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1)
    for k in (4, 17, 1203):
        for arr, g, atol in ((h2, gH, 1e-4), (u2, gU, 1e-2)):
            eps = abs(arr[k, 0]) * 1e-6 + 1e-9
            p, m_ = arr.copy(), arr.copy()
            p[k, 0] += eps
            m_[k, 0] -= eps
            args_p = (ssh, u2, p) if arr is h2 else (ssh, p, h2)
            args_m = (ssh, u2, m_) if arr is h2 else (ssh, m_, h2)
            fd = (forward_J(om, *args_p, dt, nsteps, flags) - forward_J(om, *args_m, dt, nsteps, flags)) / (2 * eps)
            assert abs(fd - g[k, 0]) <= atol + 1e-5 * abs(fd), (k, fd, g[k, 0])


@pytest.mark.parametrize("meshname,K,flags", [("ico", 3, 0), ("ico", 3, 3), ("planar", 2, 1), ("ico5", 1, 7), ("ico", 1, 7)])
def test_directional_derivative_identity(meshname, K, flags):
    """<grad J, d> == dJ/d(eps) along random directions d of the whole initial state (incl. ssh_0 and hEdge_0)."""
    mesh = {"ico": lambda: mg.icosahedral_mesh(5), "planar": lambda: mg.planar_hex_mesh(10, 8, 50e3, f0=1e-4),
            "ico5": lambda: mg.icosahedral_mesh(6, flips=5, seed=2)}[meshname]()
    rng = np.random.default_rng(5 + K)
    rest = np.full((mesh.nCells, K), 1000.0 / K)
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1) + rng.uniform(-0.1, 0.1, mesh.nCells)      # ssh_0 is an independent state variable
    hE0 = rng.uniform(0.5, 1.5, (mesh.nEdges, K)) * (1000.0 / K)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    dt = 0.2 * float(mesh.dcEdge.min()) / np.sqrt(9.80616 * 1000.0)
    nsteps = 7
    gS, gU, gH, gE = gradient(om, ssh, u, h, dt, nsteps, flags, hE0)
    for trial in range(3):
        dS, dU, dH, dE = (rng.standard_normal(a.shape) for a in (ssh, u, h, hE0))
        lhs = (gS * dS).sum() + (gU * dU).sum() + (gH * dH).sum() + (gE * dE).sum()
        eps = 1e-5
        Jp = forward_J(om, ssh + eps * dS, u + eps * dU, h + eps * dH, dt, nsteps, flags, hE0 + eps * dE)
        Jm = forward_J(om, ssh - eps * dS, u - eps * dU, h - eps * dH, dt, nsteps, flags, hE0 - eps * dE)
        fd = (Jp - Jm) / (2 * eps)
        assert abs(fd - lhs) <= 2e-6 * max(abs(fd), abs(lhs), 1.0), (trial, fd, lhs)
    if not flags & orc.FE_STALE_HEDGE:
        assert not gE.any()                                   # a refreshed hEdge makes the carried one irrelevant


def test_transposed_coriolis_stencil_is_the_transpose():
    mesh = mg.icosahedral_mesh(4, flips=3, seed=1)
    teoe, tw = orc.transpose_coriolis(mesh)
    nE = mesh.nEdges
    A = np.zeros((nE, nE))
    for e in range(nE):
        for i in range(mesh.nEdgesOnEdge[e]):
            t = mesh.edgesOnEdge[e, i]
            if t > 0:
                A[e, t - 1] += mesh.weightsOnEdge[e, i]
    B = np.zeros((nE, nE))
    for e in range(nE):
        for j in range(teoe.shape[1]):
            if teoe[e, j] > 0:
                B[e, teoe[e, j] - 1] += tw[e, j]
    assert np.array_equal(B, A.T)
    srcs = [list(teoe[e][teoe[e] > 0]) for e in range(nE)]
    assert all(s == sorted(s) for s in srcs)


@pytest.mark.parametrize("meshname,K", [("ico", 3), ("planar", 1), ("ico5", 2)])
def test_rk4_directional_derivative_identity(meshname, K):
    mesh = {"ico": lambda: mg.icosahedral_mesh(5), "planar": lambda: mg.planar_hex_mesh(10, 8, 50e3, f0=1e-4),
            "ico5": lambda: mg.icosahedral_mesh(6, flips=5, seed=2)}[meshname]()
    rng = np.random.default_rng(15 + K)
    rest = np.full((mesh.nCells, K), 1000.0 / K)
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    dt = 0.5 * float(mesh.dcEdge.min()) / np.sqrt(9.80616 * 1000.0)
    nsteps = 5

    def J(uu, hh):
        st = orc.OracleState(om, ssh, uu, hh)
        for _ in range(nsteps):
            st.step_rk4(dt)
        return st.sum_sq_ssh()

    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjointRK4(st)
    for _ in range(nsteps):
        adj.step_rk4(dt)
    ref = orc.OracleState(om, ssh, u, h)
    for _ in range(nsteps):
        ref.step_rk4(dt)
    assert np.array_equal(st.u[1], ref.u[1]) and np.array_equal(st.h[1], ref.h[1])     # taping does not change the run
    gU, gH = adj.gradient_sum_sq_ssh()
    for trial in range(3):
        dU, dH = rng.standard_normal(u.shape), rng.standard_normal(h.shape)
        lhs = (gU * dU).sum() + (gH * dH).sum()
        eps = 1e-5
        fd = (J(u + eps * dU, h + eps * dH) - J(u - eps * dU, h - eps * dH)) / (2 * eps)
        assert abs(fd - lhs) <= 2e-6 * max(abs(fd), abs(lhs), 1.0), (trial, fd, lhs)


def test_rk4_igw_gradient_vs_central_differences():
    mesh = mg.igw_mesh(200.0)
    ssh, u, h, rest = mg.igw_initial_state(mesh)
    u, h = u.reshape(mesh.nEdges, 1), h.reshape(mesh.nCells, 1)
    om = orc.OracleMesh(mesh, 1, resting_thickness_sum=np.asarray(rest).reshape(mesh.nCells, -1).sum(1))
    dt, nsteps = 400.0, 10
    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjointRK4(st)
    for _ in range(nsteps):
        adj.step_rk4(dt)
    gU, gH = adj.gradient_sum_sq_ssh()

    def J(uu, hh):
        s2 = orc.OracleState(om, ssh, uu, hh)
        for _ in range(nsteps):
            s2.step_rk4(dt)
        return s2.sum_sq_ssh()
    for k in (4, 999):
        for arr, g, atol in ((h, gH, 1e-4), (u, gU, 1e-2)):
            eps = abs(arr[k, 0]) * 1e-6 + 1e-9
            p, m_ = arr.copy(), arr.copy()
            p[k, 0] += eps
            m_[k, 0] -= eps
            fd = (J(u, p) - J(u, m_)) / (2 * eps) if arr is h else (J(p, h) - J(m_, h)) / (2 * eps)
            assert abs(fd - g[k, 0]) <= atol + 1e-5 * abs(fd), (k, fd, g[k, 0])
