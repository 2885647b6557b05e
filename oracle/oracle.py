"""ctypes front-end of the CPU oracle (oracle/moka_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (mpas-ocean.jl_amd/moka_hip) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmoka_oracle.so")

FE_STALE_HEDGE, FE_ACCUM_VORT, FE_LEVEL1_ONLY = 1, 2, 4
FE_REFERENCE_COMPAT = 7

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class _OracleMesh(C.Structure):
    _fields_ = [
        ("nCells", C.c_int32), ("nEdges", C.c_int32), ("nVertices", C.c_int32),
        ("maxEdges", C.c_int32), ("maxEdges2", C.c_int32), ("vertexDegree", C.c_int32),
        ("nVertLevels", C.c_int32), ("edgeSignOnVertexLD", C.c_int32),
        ("nEdgesOnCell", _i32p), ("edgesOnCell", _i32p), ("edgeSignOnCell", _i32p),
        ("areaCell", _f64p),
        ("cellsOnEdge", _i32p), ("nEdgesOnEdge", _i32p), ("edgesOnEdge", _i32p),
        ("weightsOnEdge", _f64p), ("dvEdge", _f64p), ("dcEdge", _f64p), ("fEdge", _f64p),
        ("edgesOnVertex", _i32p), ("edgeSignOnVertex", _i32p), ("areaTriangle", _f64p),
        ("maxLevelEdgeTop", _i32p), ("restingThicknessSum", _f64p),
    ]


class _OracleState(C.Structure):
    _fields_ = [
        ("ssh", _f64p * 2), ("u", _f64p * 2), ("h", _f64p * 2),
        ("hEdge", _f64p), ("F", _f64p), ("div", _f64p), ("vort", _f64p),
        ("tendU", _f64p), ("tendH", _f64p),
    ]


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, "moka_oracle.c"), os.path.join(HERE, "moka_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", HERE, "--no-print-directory"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        mp, sp = C.POINTER(_OracleMesh), C.POINTER(_OracleState)
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_get_threads.restype = C.c_int
        L.oracle_divergence_on_cell.argtypes = [mp, _f64p, _f64p, _f64p]
        L.oracle_gradient_on_edge.argtypes = [mp, _f64p, _f64p]
        L.oracle_curl_on_vertex.argtypes = [mp, _f64p, _f64p]
        L.oracle_gradient_on_edge_vjp.argtypes = [mp, _f64p, _f64p]
        L.oracle_divergence_on_cell_vjp.argtypes = [mp, _f64p, _f64p, _f64p]
        L.oracle_curl_on_vertex_vjp.argtypes = [mp, _f64p, _f64p]
        L.oracle_interpolate_cell2edge.argtypes = [mp, _f64p, _f64p, C.c_int]
        L.oracle_thickness_flux.argtypes = [mp, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_normal_velocity_tendency.argtypes = [mp, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_layer_thickness_tendency.argtypes = [mp, _f64p, _f64p, C.c_int]
        L.oracle_update_ssh.argtypes = [mp, _f64p, _f64p, C.c_int]
        L.oracle_ksum.argtypes = [_f64p, C.c_int]
        L.oracle_ksum.restype = C.c_double
        L.oracle_tendencies_clean.argtypes = [mp] + [_f64p] * 7
        L.oracle_step_fe.argtypes = [mp, sp, C.c_double, C.c_int]
        L.oracle_step_rk4.argtypes = [mp, sp, C.c_double, _f64p]
        L.oracle_step_rk4_mixed.argtypes = [mp, sp, C.c_double, _f64p]
        L.oracle_step_rk4_s13.argtypes = [mp, sp, C.c_double, _f64p]
        L.oracle_step_fe_mixed.argtypes = [mp, sp, C.c_double, C.c_int]
        L.oracle_tendencies_mixed.argtypes = [mp] + [_f64p] * 7
        L.oracle_round_f32.argtypes = [_f64p, C.c_int64]
        L.oracle_sum_sq.argtypes = [_f64p, C.c_int64]
        L.oracle_step_fe_adjoint.argtypes = [mp, _i32p, _f64p, C.c_int, C.c_double, C.c_int] + [_f64p] * 12
        L.oracle_tendency_transpose.argtypes = [mp, _i32p, _f64p, C.c_int] + [_f64p] * 8
        L.oracle_tendencies_nonlinear.argtypes = [mp, _i32p, _i32p, _f64p, _f64p] + [_f64p] * 10
        L.oracle_step_rk4_nonlinear.argtypes = [mp, _i32p, _i32p, _f64p, _f64p, sp, C.c_double, _f64p, _f64p]
        L.oracle_tendencies_nonlinear_del2.argtypes = [mp, _i32p, _i32p, _f64p, _f64p] + [_f64p] * 10 + [C.c_double, _f64p, _f64p]
        L.oracle_step_rk4_nonlinear_del2.argtypes = [mp, _i32p, _i32p, _f64p, _f64p, sp, C.c_double, _f64p, _f64p, C.c_double]
        L.oracle_step_rk4_nonlinear_s13.argtypes = [mp, _i32p, _i32p, _f64p, _f64p, sp, C.c_double, _f64p, _f64p, C.c_double]
        L.oracle_sum_sq.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_f64p if a.dtype == np.float64 else _i32p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class OracleMesh:
    """Reference-convention mesh handed to the oracle.  `max_level_edge_top=None` gives the
    reference's all-ones array (VertMesh.jl:32); pass K for the N3 multi-layer semantics."""

    def __init__(self, mesh, K: int, resting_thickness_sum=None, max_level_edge_top=None):
        self.mesh, self.K = mesh, int(K)
        a = self.arrays = {}
        for n in ("nEdgesOnCell", "edgesOnCell", "edgeSignOnCell", "cellsOnEdge", "nEdgesOnEdge",
                  "edgesOnEdge", "edgesOnVertex", "edgeSignOnVertex"):
            a[n] = _c(getattr(mesh, n), np.int32)
        for n in ("areaCell", "weightsOnEdge", "dvEdge", "dcEdge", "fEdge", "areaTriangle"):
            a[n] = _c(getattr(mesh, n), np.float64)
        if max_level_edge_top is None:
            mlt = np.ones(mesh.nEdges, dtype=np.int32)
        elif np.isscalar(max_level_edge_top):
            mlt = np.full(mesh.nEdges, int(max_level_edge_top), dtype=np.int32)
        else:
            mlt = _c(max_level_edge_top, np.int32)
        a["maxLevelEdgeTop"] = mlt
        if resting_thickness_sum is None:
            resting_thickness_sum = np.ones(mesh.nCells)  # test ctor, VertMesh.jl:99-100
        a["restingThicknessSum"] = _c(np.asarray(resting_thickness_sum).reshape(-1), np.float64)
        s = self.c = _OracleMesh()
        s.nCells, s.nEdges, s.nVertices = mesh.nCells, mesh.nEdges, mesh.nVertices
        s.maxEdges, s.maxEdges2, s.vertexDegree = mesh.maxEdges, mesh.maxEdges2, mesh.vertexDegree
        s.nVertLevels = self.K
        s.edgeSignOnVertexLD = a["edgeSignOnVertex"].shape[1]
        for n, arr in a.items():
            setattr(s, n, _p(arr))

    @property
    def ref(self):
        return C.byref(self.c)

    # ---- operators --------------------------------------------------------------------
    def divergence_on_cell(self, vecEdge, temp=None):
        vecEdge = _c(vecEdge, np.float64)
        div = np.zeros((self.mesh.nCells, self.K))
        temp = np.zeros_like(vecEdge) if temp is None else temp
        lib().oracle_divergence_on_cell(self.ref, _p(div), _p(vecEdge), _p(temp))
        return div

    def gradient_on_edge(self, scalarCell):
        scalarCell = _c(scalarCell, np.float64)
        g = np.zeros((self.mesh.nEdges, self.K))
        lib().oracle_gradient_on_edge(self.ref, _p(g), _p(scalarCell))
        return g

    def curl_on_vertex(self, vecEdge, curl=None):
        vecEdge = _c(vecEdge, np.float64)
        curl = np.zeros((self.mesh.nVertices, self.K)) if curl is None else curl
        lib().oracle_curl_on_vertex(self.ref, _p(curl), _p(vecEdge))
        return curl

    # reverse mode of the operators (Enzyme's in-place conventions: see moka_oracle.c); arrays are modified in place
    def gradient_on_edge_vjp(self, dScalar, dGrad):
        lib().oracle_gradient_on_edge_vjp(self.ref, _p(dScalar), _p(dGrad))

    def divergence_on_cell_vjp(self, dVec, dTemp, dDiv):
        lib().oracle_divergence_on_cell_vjp(self.ref, _p(dVec), _p(dTemp), _p(dDiv))

    def curl_on_vertex_vjp(self, dVec, dCurl):
        lib().oracle_curl_on_vertex_vjp(self.ref, _p(dVec), _p(_c(dCurl, np.float64)))

    def interpolate_cell2edge(self, cellValue, nlev=None, out=None):
        cellValue = _c(cellValue, np.float64)
        out = np.zeros((self.mesh.nEdges, self.K)) if out is None else out
        lib().oracle_interpolate_cell2edge(self.ref, _p(out), _p(cellValue), self.K if nlev is None else nlev)
        return out

    def update_ssh(self, h, nlev=None):
        h = _c(h, np.float64)
        ssh = np.zeros(self.mesh.nCells)
        lib().oracle_update_ssh(self.ref, _p(ssh), _p(h), self.K if nlev is None else nlev)
        return ssh

    def tendencies_clean(self, u, h, mixed=False):
        """mixed=True: fp32-stored state (u, h are rounded to fp32 first), fp64 arithmetic."""
        dt = np.float32 if mixed else np.float64
        u, h = _c(_c(u, dt), np.float64), _c(_c(h, dt), np.float64)
        tu, th = np.zeros_like(u), np.zeros_like(h)
        ssh = np.zeros(self.mesh.nCells)
        s1, s2 = np.zeros_like(u), np.zeros_like(u)
        fn = lib().oracle_tendencies_mixed if mixed else lib().oracle_tendencies_clean
        fn(self.ref, _p(tu), _p(th), _p(u), _p(h), _p(ssh), _p(s1), _p(s2))
        if mixed:          # accumulated in fp64, STORED like the state: fp32
            tu, th = tu.astype(np.float32).astype(np.float64), th.astype(np.float32).astype(np.float64)
        return tu, th, ssh


class OracleState:
    """Two time levels of PrognosticVars + DiagnosticVars + TendencyVars, reference layout."""

    def __init__(self, om: OracleMesh, ssh, u, h, mixed=False):
        """mixed=True: fp32-stored state (inputs are rounded to fp32), fp64 arithmetic."""
        m, K = om.mesh, om.K
        self.om, self.mixed = om, bool(mixed)
        sd = np.float32 if mixed else np.float64
        f = lambda a, shape: np.array(np.asarray(a, dtype=sd).astype(np.float64).reshape(shape), order="C", copy=True)
        # nTimeLevels = 2 deep copies (PrognosticVars.jl:44-55)
        self.ssh = [f(ssh, (m.nCells,)) for _ in range(2)]
        self.u = [f(u, (m.nEdges, K)) for _ in range(2)]
        self.h = [f(h, (m.nCells, K)) for _ in range(2)]
        self.hEdge = np.zeros((m.nEdges, K))
        self.F = np.zeros((m.nEdges, K))
        self.div = np.zeros((m.nCells, K))
        self.vort = np.zeros((m.nVertices, K))
        self.tendU = np.zeros((m.nEdges, K))
        self.tendH = np.zeros((m.nCells, K))
        self._work = None
        s = self.c = _OracleState()
        for n in ("ssh", "u", "h"):
            arr = getattr(self, n)
            setattr(s, n, (_f64p * 2)(_p(arr[0]), _p(arr[1])))
        for n in ("hEdge", "F", "div", "vort", "tendU", "tendH"):
            setattr(s, n, _p(getattr(self, n)))

    def step_fe(self, dt, flags=FE_REFERENCE_COMPAT):
        if self.mixed:
            if flags & FE_LEVEL1_ONLY:
                raise ValueError("fp32-state oracle: Forward Euler steps all levels")
            lib().oracle_step_fe_mixed(self.om.ref, C.byref(self.c), float(dt), int(flags))
            return
        lib().oracle_step_fe(self.om.ref, C.byref(self.c), float(dt), int(flags))

    def step_rk4_s13(self, dt):
        """The 13-stream form of the RK4 step (oracle_step_rk4_s13: twin of the library's opt-in form, not the reference's
        round-off); Float64 states."""
        assert not self.mixed
        if self._work is None:
            m, K = self.om.mesh, self.om.K
            self._work = np.zeros(2 * K * (m.nEdges + m.nCells) + m.nCells)
        lib().oracle_step_rk4_s13(self.om.ref, C.byref(self.c), float(dt), _p(self._work))

    def step_rk4(self, dt):
        if self._work is None:
            m, K = self.om.mesh, self.om.K
            self._work = np.zeros(2 * K * (m.nEdges + m.nCells) + m.nCells)
        fn = lib().oracle_step_rk4_mixed if self.mixed else lib().oracle_step_rk4
        fn(self.om.ref, C.byref(self.c), float(dt), _p(self._work))
        if self.mixed:     # the stage-4 tendencies left in Tend: fp64 inside the step, stored fp32
            self.tendU[...] = self.tendU.astype(np.float32)
            self.tendH[...] = self.tendH.astype(np.float32)

    def sum_sq_ssh(self):
        return lib().oracle_sum_sq(_p(self.ssh[1]), self.ssh[1].size)


def set_threads(n: int):
    lib().oracle_set_threads(int(n))


def ksum(col):
    col = _c(col, np.float64)
    return lib().oracle_ksum(_p(col), col.size)


# ------------------------------------------------------------------------------------------------
# reverse mode of the Forward-Euler loop (reference: Enzyme over ocn_run_loop,
# test/enzyme/test_Enzyme_end2end.jl).  Test infrastructure like the rest of this file.
# ------------------------------------------------------------------------------------------------
def transpose_coriolis(mesh):
    """(teoe, tw), both (nEdges, W): for edge e the (source edge s [1-based], weightsOnEdge[i, s]) pairs with
    edgesOnEdge[i, s] == e, sorted by (s, i); 0 / 0.0 pad.  The transpose of the Coriolis stencil."""
    eoe = np.asarray(mesh.edgesOnEdge)
    neoe = np.asarray(mesh.nEdgesOnEdge)
    w = np.asarray(mesh.weightsOnEdge)
    nE, M = eoe.shape
    src = np.repeat(np.arange(1, nE + 1), M)
    slot = np.tile(np.arange(M), nE)
    tgt = eoe.reshape(-1)
    ok = (slot < np.repeat(neoe, M)) & (tgt > 0)
    src, slot, tgt, ww = src[ok], slot[ok], tgt[ok], w.reshape(-1)[ok]
    order = np.lexsort((slot, src, tgt))
    src, tgt, ww = src[order], tgt[order], ww[order]
    cnt = np.bincount(tgt - 1, minlength=nE)
    W = int(cnt.max()) if cnt.size else 0
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    pos = np.arange(tgt.size) - start[tgt - 1]
    teoe = np.zeros((nE, max(W, 1)), dtype=np.int32)
    tw = np.zeros((nE, max(W, 1)))
    teoe[tgt - 1, pos] = src
    tw[tgt - 1, pos] = ww
    return teoe, tw


class OracleAdjoint:
    """Tape + reverse sweep for Forward-Euler runs of an OracleState (flags without LEVEL1_ONLY unless K == 1)."""

    def __init__(self, st: OracleState):
        self.st, self.om = st, st.om
        self.teoe, self.tw = transpose_coriolis(self.om.mesh)
        self.tape = []

    def step_fe(self, dt, flags=FE_REFERENCE_COMPAT):
        st, om = self.st, self.om
        if (flags & FE_LEVEL1_ONLY) and om.K != 1:
            raise ValueError("adjoint: level-1-only stepping is supported for K = 1 only")
        u = st.u[1].copy()
        if flags & FE_STALE_HEDGE:
            hE = st.hEdge.copy()                               # DiagnosticVars.layerThicknessEdge of the previous step
        else:
            hE = np.zeros_like(st.hEdge)
            lib().oracle_interpolate_cell2edge(om.ref, _p(hE), _p(st.h[1]), om.K)
        self.tape.append((u, hE, float(dt), int(flags)))
        st.step_fe(dt, flags)

    def gradient_sum_sq_ssh(self):
        """d sum(ssh_N^2) / d (ssh_0, u_0, h_0, hEdge_0): the reverse sweep over the tape."""
        st, om = self.st, self.om
        m, K = om.mesh, om.K
        lamS = 2.0 * st.ssh[1]
        lamU, lamH, lamE = np.zeros((m.nEdges, K)), np.zeros((m.nCells, K)), np.zeros((m.nEdges, K))
        Enew, csum = np.zeros((m.nEdges, K)), np.zeros(m.nEdges)
        for u, hE, dt, flags in reversed(self.tape):
            oU, oH, oS, oE = np.zeros_like(lamU), np.zeros_like(lamH), np.zeros_like(lamS), np.zeros_like(lamE)
            lib().oracle_step_fe_adjoint(om.ref, _p(self.teoe), _p(self.tw), self.teoe.shape[1], dt, flags, _p(u), _p(hE),
                                         _p(lamU), _p(lamH), _p(lamS), _p(lamE), _p(oU), _p(oH), _p(oS), _p(oE),
                                         _p(Enew), _p(csum))
            lamU, lamH, lamS, lamE = oU, oH, oS, oE
        return lamS, lamU, lamH, lamE


class OracleAdjointRK4:
    """Tape + reverse sweep for RK4 runs (oracle_step_rk4, time_integration.jl:61-148).  The state of the RK4 map is
    (u, h): ssh is recomputed from h inside every tendency evaluation.  Per step the tape holds the four provisional
    states P1..P4 the tendencies were evaluated at; the reverse step is
        kb4 = b4*X;          Pb4 = T'(P4)^T kb4;   kb3 = b3*X + a3*Pb4;  Pb3 = T'(P3)^T kb3;
        kb2 = b2*X + a2*Pb3; Pb2 = T'(P2)^T kb2;   kb1 = b1*X + a1*Pb2;  Pb1 = T'(P1)^T kb1;
        X   = (((X + Pb4) + Pb3) + Pb2) + Pb1."""

    def __init__(self, st: OracleState):
        self.st, self.om = st, st.om
        self.teoe, self.tw = transpose_coriolis(self.om.mesh)
        self.tape = []

    def step_rk4(self, dt):
        st, om = self.st, self.om
        K = om.K
        a = (dt / 2., dt / 2., dt)
        u0, h0 = st.u[1].copy(), st.h[1].copy()
        P = [(u0, h0)]
        pu, ph = u0, h0
        for s in range(3):                                     # provisional states, exactly as the forward step forms them
            tu, th, _ = om.tendencies_clean(pu, ph)
            pu, ph = u0 + a[s] * tu, h0 + a[s] * th
            P.append((pu, ph))
        self.tape.append((P, float(dt)))
        st.step_rk4(dt)

    def _tt(self, u, h, kU, kH):
        m, K = self.om.mesh, self.om.K
        oU, oH = np.zeros((m.nEdges, K)), np.zeros((m.nCells, K))
        En, cs = np.zeros((m.nEdges, K)), np.zeros(m.nEdges)
        lib().oracle_tendency_transpose(self.om.ref, _p(self.teoe), _p(self.tw), self.teoe.shape[1], _p(_c(u, np.float64)),
                                        _p(_c(h, np.float64)), _p(_c(kU, np.float64)), _p(_c(kH, np.float64)), _p(oU), _p(oH),
                                        _p(En), _p(cs))
        return oU, oH

    def gradient_sum_sq_ssh(self):
        """d sum(ssh_N^2) / d (u_0, h_0)."""
        st, om = self.st, self.om
        m, K = om.mesh, om.K
        XU = np.zeros((m.nEdges, K))
        XH = np.repeat((2.0 * st.ssh[1])[:, None], K, axis=1)          # ssh_N = ksum_k h_N - rsum
        for P, dt in reversed(self.tape):
            a = (dt / 2., dt / 2., dt)
            b = (dt / 6., dt / 3., dt / 3., dt / 6.)
            kU, kH = b[3] * XU, b[3] * XH
            PU, PH = self._tt(P[3][0], P[3][1], kU, kH)
            accU, accH = XU + PU, XH + PH
            for s in (2, 1, 0):
                kU, kH = b[s] * XU + a[s] * PU, b[s] * XH + a[s] * PH
                PU, PH = self._tt(P[s][0], P[s][1], kU, kH)
                accU, accH = accU + PU, accH + PH
            XU, XH = accU, accH
        return XU, XH


class OracleNonlinear:
    """The optional nonlinear (potential-vorticity + kinetic-energy) tendencies -- an extension that the reference does
    not have: parity unpinned, pinned by its own properties only.  `visc_del2` != 0 adds the Del2
    momentum mixing of the reference's uncalled sketch (horizontal_momentum_mixing.jl:53-80)."""

    def __init__(self, om: OracleMesh, visc_del2: float = 0.0):
        m = om.mesh
        self.visc_del2 = float(visc_del2)
        if m.kiteAreasOnVertex is None:
            raise ValueError("the nonlinear terms need kiteAreasOnVertex")
        self.om = om
        self.voe = _c(m.verticesOnEdge, np.int32)
        self.cov = _c(m.cellsOnVertex, np.int32)
        self.kite = _c(m.kiteAreasOnVertex, np.float64)
        self.fv = _c(m.fVertex, np.float64)

    def tendencies(self, u, h):
        m, K = self.om.mesh, self.om.K
        u, h = _c(u, np.float64).reshape(m.nEdges, K), _c(h, np.float64).reshape(m.nCells, K)
        tu, th, ssh = np.zeros_like(u), np.zeros_like(h), np.zeros(m.nCells)
        hE, F, qe = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
        qv, ke = np.zeros((m.nVertices, K)), np.zeros_like(h)
        zv, divc = np.zeros_like(qv), np.zeros_like(h)
        lib().oracle_tendencies_nonlinear_del2(self.om.ref, _p(self.voe), _p(self.cov), _p(self.kite), _p(self.fv), _p(tu),
                                               _p(th), _p(u), _p(h), _p(ssh), _p(hE), _p(F), _p(qv), _p(qe), _p(ke),
                                               self.visc_del2, _p(zv), _p(divc))
        return tu, th, ssh, {"pv_vertex": qv, "pv_edge": qe, "ke": ke, "relativeVorticity": zv, "velocityDivCell": divc}

    def step_rk4(self, st: OracleState, dt, s13=False):
        m, K = self.om.mesh, self.om.K
        if st._work is None:
            st._work = np.zeros(2 * K * (m.nEdges + m.nCells) + m.nCells)
        if getattr(st, "_nl_scratch", None) is None:
            st._nl_scratch = np.zeros(4 * K * m.nEdges + 2 * K * m.nVertices + 2 * K * m.nCells)
        fn = lib().oracle_step_rk4_nonlinear_s13 if s13 else lib().oracle_step_rk4_nonlinear_del2
        fn(self.om.ref, _p(self.voe), _p(self.cov), _p(self.kite), _p(self.fv),
           C.byref(st.c), float(dt), _p(st._work), _p(st._nl_scratch), self.visc_del2)

    def step_rk4_s13(self, st: OracleState, dt):
        """The 13-stream form (oracle_step_rk4_nonlinear_s13: twin of the library's opt-in form, its own round-off)."""
        self.step_rk4(st, dt, s13=True)


# ---------------------------------------------------------------------------------------------------------------------
# Row-sampled oracle (test infrastructure): the tendencies / one RK stage at a few thousand SAMPLED cells and edges of a mesh too
# large to run the whole oracle on, from the rows their stencils gather -- the same C loop nests (same slot order, same rounding
# points) run on a SUB-MESH: the sampled entities plus everything their stencils name, renumbered, connectivity slot for slot.
#   tendU[e] reads u of edgesOnEdge[:, e] and ssh (= column sum of h) of cellsOnEdge[:, e]      (K9 / K10)
#   tendH[c] reads u of edgesOnCell[:, c] and h of both cells of each of those edges             (K5 / K7 / K8)
# Entities of the closure that are not sampled keep whatever of their stencil lies inside it (anything else points at entity
# 1): their results are garbage and are never looked at.
# ---------------------------------------------------------------------------------------------------------------------
class SubMesh:
    """mesh: reference-convention arrays (1-based connectivity, 0 = none).  cells / edges: 0-based ids of the sampled entities."""

    def __init__(self, mesh, cells, edges, K, resting_thickness_sum, max_level_edge_top=None):
        import types
        cells = np.unique(np.asarray(cells, dtype=np.int64))
        edges = np.unique(np.asarray(edges, dtype=np.int64))
        eoc = np.asarray(mesh.edgesOnCell)[cells].reshape(-1)
        eoe = np.asarray(mesh.edgesOnEdge)[edges].reshape(-1)
        E = np.unique(np.concatenate([edges, eoc[eoc > 0] - 1, eoe[eoe > 0] - 1]))          # closure, 0-based, sorted
        ecell = eoc[eoc > 0] - 1
        coe_s = np.asarray(mesh.cellsOnEdge)[edges].reshape(-1)
        coe_c = np.asarray(mesh.cellsOnEdge)[ecell].reshape(-1)
        Cc = np.unique(np.concatenate([cells, coe_s[coe_s > 0] - 1, coe_c[coe_c > 0] - 1]))
        self.cells, self.edges = Cc, E                                   # closure ids (global, 0-based): rows to fetch
        self.sample_cells = np.searchsorted(Cc, cells)                   # positions of the sampled entities inside the closure
        self.sample_edges = np.searchsorted(E, edges)
        self.sampled_cells_global, self.sampled_edges_global = cells, edges

        def remap(conn, ids):              # 1-based global -> 1-based local; outside the closure -> 1 (results never read)
            conn = np.asarray(conn, dtype=np.int64)
            pos = np.searchsorted(ids, conn - 1)
            pos = np.minimum(pos, ids.size - 1)
            inside = (conn > 0) & (ids[pos] == conn - 1)
            return np.where(conn > 0, np.where(inside, pos + 1, 1), 0).astype(np.int32)

        m = types.SimpleNamespace()
        m.nCells, m.nEdges, m.nVertices = int(Cc.size), int(E.size), 1
        m.maxEdges, m.maxEdges2, m.vertexDegree = mesh.maxEdges, mesh.maxEdges2, mesh.vertexDegree
        m.nEdgesOnCell = np.asarray(mesh.nEdgesOnCell)[Cc]
        m.edgesOnCell = remap(np.asarray(mesh.edgesOnCell)[Cc], E)
        m.edgeSignOnCell = np.asarray(mesh.edgeSignOnCell)[Cc]
        m.areaCell = np.asarray(mesh.areaCell)[Cc]
        m.cellsOnEdge = remap(np.asarray(mesh.cellsOnEdge)[E], Cc)
        m.nEdgesOnEdge = np.asarray(mesh.nEdgesOnEdge)[E]
        m.edgesOnEdge = remap(np.asarray(mesh.edgesOnEdge)[E], E)
        # an edgesOnEdge slot that left the closure must not be skipped as "0 = none" would be, nor change the sampled sums: it
        # only occurs on non-sampled edges (remap sends it to edge 1)
        m.weightsOnEdge = np.asarray(mesh.weightsOnEdge)[E]
        m.dvEdge, m.dcEdge, m.fEdge = (np.asarray(getattr(mesh, n))[E] for n in ("dvEdge", "dcEdge", "fEdge"))
        m.edgesOnVertex = np.ones((1, mesh.vertexDegree), dtype=np.int32)
        m.edgeSignOnVertex = np.ones((1, np.asarray(mesh.edgeSignOnVertex).shape[1]), dtype=np.int32)
        m.areaTriangle = np.ones(1)
        mlt = None if max_level_edge_top is None else (max_level_edge_top if np.isscalar(max_level_edge_top) else np.asarray(max_level_edge_top)[E])
        self.K = int(K)
        self.rsum = np.asarray(resting_thickness_sum, dtype=np.float64).reshape(-1)[Cc]
        self.om = OracleMesh(m, K, resting_thickness_sum=self.rsum, max_level_edge_top=mlt)

    def tendencies(self, u_rows, h_rows, mixed=False):
        """u_rows (len(self.edges), K), h_rows (len(self.cells), K): the closure's rows of the provisional state.  Returns the
        UNROUNDED fp64 tendencies at the sampled edges / cells (what a fused stage update consumes) and the stored ssh of the
        sampled cells' columns."""
        dt = np.float32 if mixed else np.float64
        u, h = _c(_c(u_rows, dt), np.float64), _c(_c(h_rows, dt), np.float64)
        tu, th = np.zeros_like(u), np.zeros_like(h)
        ssh = np.zeros(self.cells.size)
        s1, s2 = np.zeros_like(u), np.zeros_like(u)
        fn = lib().oracle_tendencies_mixed if mixed else lib().oracle_tendencies_clean
        fn(self.om.ref, _p(tu), _p(th), _p(u), _p(h), _p(ssh), _p(s1), _p(s2))
        return tu[self.sample_edges], th[self.sample_cells], ssh[self.sample_cells]

    def rk_stage(self, pu_rows, ph_rows, cur_u, cur_h, new_u, new_h, a, b, mixed=False):
        """One fused RK4 stage (time_integration.jl:124-125,134-135) at the sampled entities: tendency of the provisional state
        (closure rows pu_rows / ph_rows), Provis' = Curr + a * t, New' = New + b * t (cur_* / new_*: rows of the SAMPLED
        entities), each rounded to storage when `mixed`; ssh' = column sum of the stored Provis' thickness - restingThicknessSum.
        Returns (pu', ph', ssh', nu', nh')."""
        tu, th, _ = self.tendencies(pu_rows, ph_rows, mixed)
        rnd = (lambda x: x.astype(np.float32).astype(np.float64)) if mixed else (lambda x: x)
        pu2, ph2 = rnd(cur_u + a * tu), rnd(cur_h + a * th)
        nu2, nh2 = rnd(new_u + b * tu), rnd(new_h + b * th)
        ssh2 = rnd(np.array([ksum(col) for col in ph2]) - self.rsum[self.sample_cells])
        return pu2, ph2, ssh2, nu2, nh2
