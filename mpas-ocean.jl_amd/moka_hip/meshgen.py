"""Synthetic MPAS/TRiSK mesh generators (stand-ins for the NetCDF mesh files the reference reads).

The reference only *reads* meshes (`src/infra/MPASMesh/HorzMesh.jl:166-290,334-355`); its test
meshes are downloaded (`test/ocn/test_Operators.jl:12`) and unreachable here.  These generators
produce the same fields, dtypes and conventions as the reader would (1-based Int32 connectivity,
0 = "no neighbour", slot index fastest), so everything downstream sees reference-shaped data.

Array convention: a Julia `(slots, n)` column-major matrix is held as a C-order numpy array of
shape `(n, slots)` -- identical bytes.  State arrays `(nVertLevels, n)` are numpy `(n, nVertLevels)`.

Generators
  planar_hex_mesh(nx, ny, dc)        doubly periodic hexagons, MPAS `planar_hex` coordinates
  icosahedral_mesh(m, radius)        geodesic Voronoi sphere, nCells = 10 m^2 + 2
Initial states
  igw_exact / igw_initial_state      `src/inertialGravityWave.jl:1-64`
  sphere_synthetic_state             SURVEY.md section 8(d) synthetic inputs
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

I32 = np.int32
F64 = np.float64

OMEGA_EARTH = 7.292e-5
RADIUS_EARTH = 6371220.0
GRAVITY = 9.80616  # literal in src/ocn/Tendencies/normalVelocity/pressure_gradient.jl:63


@dataclass
class MeshData:
    """Field names follow HorzMesh.jl:64-162 with ASCII spellings (xᶜ -> xCell, fᵉ -> fEdge ...)."""

    nCells: int
    nEdges: int
    nVertices: int
    maxEdges: int
    maxEdges2: int
    vertexDegree: int
    # cells
    xCell: np.ndarray
    yCell: np.ndarray
    zCell: np.ndarray
    fCell: np.ndarray
    areaCell: np.ndarray
    nEdgesOnCell: np.ndarray
    edgesOnCell: np.ndarray
    verticesOnCell: np.ndarray
    cellsOnCell: np.ndarray
    edgeSignOnCell: np.ndarray
    # edges
    xEdge: np.ndarray
    yEdge: np.ndarray
    zEdge: np.ndarray
    fEdge: np.ndarray
    dvEdge: np.ndarray
    dcEdge: np.ndarray
    angleEdge: np.ndarray
    nEdgesOnEdge: np.ndarray
    cellsOnEdge: np.ndarray
    verticesOnEdge: np.ndarray
    edgesOnEdge: np.ndarray
    weightsOnEdge: np.ndarray
    # vertices
    xVertex: np.ndarray
    yVertex: np.ndarray
    zVertex: np.ndarray
    fVertex: np.ndarray
    areaTriangle: np.ndarray
    edgesOnVertex: np.ndarray
    cellsOnVertex: np.ndarray
    edgeSignOnVertex: np.ndarray
    # misc
    on_sphere: bool = False
    sphere_radius: float = 0.0
    is_periodic: bool = True
    meta: dict = field(default_factory=dict)
    # (nVertices, vertexDegree) area of the kite shared by vertex v and cellsOnVertex[v, j]: MPAS mesh files carry it;
    # only the optional nonlinear (potential-vorticity) terms read it -- the reference never does
    kiteAreasOnVertex: np.ndarray | None = None


def sign_index_fields(cellsOnEdge, verticesOnEdge, nEdgesOnCell, edgesOnCell, edgesOnVertex,
                      maxEdges, vertexDegree):
    """edgeSignOnCell / edgeSignOnVertex exactly as `signIndexField!` (HorzMesh.jl:292-332):
    -1 when the cell (vertex) is the edge's first cell (vertex), +1 otherwise; unused slots 0.
    edgeSignOnVertex is allocated (maxEdges, nVertices) with only vertexDegree rows set (:234,318)."""
    nC = edgesOnCell.shape[0]
    nV = edgesOnVertex.shape[0]
    esc = np.zeros((nC, maxEdges), dtype=I32)
    cid = np.arange(1, nC + 1, dtype=I32)
    for i in range(maxEdges):
        act = i < nEdgesOnCell
        e = edgesOnCell[:, i].astype(np.int64) - 1
        e = np.where(act, e, 0)
        first = cellsOnEdge[e, 0] == cid
        esc[:, i] = np.where(act, np.where(first, -1, 1), 0)
    esv = np.zeros((nV, maxEdges), dtype=I32)
    vid = np.arange(1, nV + 1, dtype=I32)
    for j in range(vertexDegree):
        e = edgesOnVertex[:, j].astype(np.int64) - 1
        first = verticesOnEdge[e, 0] == vid
        esv[:, j] = np.where(first, -1, 1)
    return esc, esv


# --------------------------------------------------------------------------------------------
# planar periodic hexagons (SURVEY.md Appendix B recipe; coordinates as MPAS `planar_hex`)
# --------------------------------------------------------------------------------------------

def planar_hex_mesh(nx: int, ny: int, dc: float, f0: float = 0.0) -> MeshData:
    if ny % 2:
        raise ValueError("ny must be even for a doubly periodic hex mesh")
    nC, nE, nV = nx * ny, 3 * nx * ny, 2 * nx * ny
    r, c = np.divmod(np.arange(nC), nx)
    even = (r % 2) == 0

    def cid(rr, cc):
        return (rr % ny) * nx + (cc % nx)

    E = cid(r, c + 1)
    NE = np.where(even, cid(r + 1, c), cid(r + 1, c + 1))
    NW = np.where(even, cid(r + 1, c - 1), cid(r + 1, c))
    W = cid(r, c - 1)
    SW = np.where(even, cid(r - 1, c - 1), cid(r - 1, c))
    SE = np.where(even, cid(r - 1, c), cid(r - 1, c + 1))
    nbr = np.stack([E, NE, NW, W, SW, SE], axis=1)  # CCW from 0 degrees

    xC = np.where(even, dc * (c + 0.5), dc * (c + 1.0)).astype(F64)
    yC = (dc * (r + 1.0) * math.sqrt(3.0) / 2.0).astype(F64)

    # edges: cell owns d = 0,1,2 (towards E, NE, NW)
    eid = lambda cell, d: 3 * cell + d
    cellsOnEdge = np.zeros((nE, 2), dtype=I32)
    angleEdge = np.zeros(nE)
    xE = np.zeros(nE)
    yE = np.zeros(nE)
    for d in range(3):
        e = eid(np.arange(nC), d)
        cellsOnEdge[e, 0] = np.arange(nC) + 1
        cellsOnEdge[e, 1] = nbr[:, d] + 1
        ang = d * math.pi / 3.0
        angleEdge[e] = ang
        xE[e] = xC + 0.5 * dc * math.cos(ang)
        yE[e] = yC + 0.5 * dc * math.sin(ang)
    dcEdge = np.full(nE, dc, dtype=F64)
    dvEdge = np.full(nE, dc / math.sqrt(3.0), dtype=F64)

    edgesOnCell0 = np.stack([eid(np.arange(nC), 0), eid(np.arange(nC), 1), eid(np.arange(nC), 2),
                             eid(W, 0), eid(SW, 1), eid(SE, 2)], axis=1)
    nEdgesOnCell = np.full(nC, 6, dtype=I32)

    # vertices: cell owns top (t=0) and upper-right (t=1)
    vid = lambda cell, t: 2 * cell + t
    xV = np.zeros(nV)
    yV = np.zeros(nV)
    xV[vid(np.arange(nC), 0)] = xC
    yV[vid(np.arange(nC), 0)] = yC + dc / math.sqrt(3.0)
    xV[vid(np.arange(nC), 1)] = xC + dc / 2.0
    yV[vid(np.arange(nC), 1)] = yC + dc / (2.0 * math.sqrt(3.0))
    A = np.arange(nC)
    edgesOnVertex0 = np.zeros((nV, 3), dtype=np.int64)
    cellsOnVertex0 = np.zeros((nV, 3), dtype=np.int64)
    edgesOnVertex0[vid(A, 0)] = np.stack([eid(A, 1), eid(A, 2), eid(NW, 0)], axis=1)
    cellsOnVertex0[vid(A, 0)] = np.stack([A, NE, NW], axis=1)
    edgesOnVertex0[vid(A, 1)] = np.stack([eid(A, 0), eid(A, 1), eid(E, 2)], axis=1)
    cellsOnVertex0[vid(A, 1)] = np.stack([A, E, NE], axis=1)

    # verticesOnEdge ordered so that v1 -> v2 = k x n  (Appendix B)
    # edge d of cell A has endpoints: d0: (SE.top , A.ur) ; d1: (A.ur, A.top) ; d2: (A.top, W.ur)
    verticesOnEdge0 = np.zeros((nE, 2), dtype=np.int64)
    verticesOnEdge0[eid(A, 0)] = np.stack([vid(SE, 0), vid(A, 1)], axis=1)
    verticesOnEdge0[eid(A, 1)] = np.stack([vid(A, 1), vid(A, 0)], axis=1)
    verticesOnEdge0[eid(A, 2)] = np.stack([vid(A, 0), vid(W, 1)], axis=1)

    # vertices on cell CCW starting after edge slot 0:  between edge i and i+1
    verticesOnCell0 = np.stack([vid(A, 1), vid(A, 0), vid(W, 1), vid(SW, 0), vid(SE, 1), vid(SE, 0)], axis=1)

    # TRiSK weights, regular hexagon (kite ratio 1/6)
    eoe0 = np.full((nE, 10), -1, dtype=np.int64)
    woe = np.zeros((nE, 10))
    slot_of = {}  # slot of own edge d in owner, and in the neighbour
    for d in range(3):
        e = eid(A, d)
        for side in range(2):
            cell = A if side == 0 else nbr[:, d]
            m0 = d if side == 0 else d + 3
            s = 1.0 if side == 0 else -1.0
            for k in range(1, 6):
                slot = (m0 + k) % 6
                e2 = edgesOnCell0[cell, slot]
                nout = 1.0 if slot < 3 else -1.0
                w = -s * (k / 6.0 - 0.5) * nout * dvEdge[e2] / dcEdge[e]
                j = side * 5 + (k - 1)
                eoe0[e, j] = e2
                woe[e, j] = w

    maxEdges, maxEdges2 = 6, 12
    edgesOnEdge = np.zeros((nE, maxEdges2), dtype=I32)
    weightsOnEdge = np.zeros((nE, maxEdges2), dtype=F64)
    edgesOnEdge[:, :10] = eoe0 + 1
    weightsOnEdge[:, :10] = woe
    edgesOnCell = (edgesOnCell0 + 1).astype(I32)
    edgesOnVertex = (edgesOnVertex0 + 1).astype(I32)
    verticesOnEdge = (verticesOnEdge0 + 1).astype(I32)
    esc, esv = sign_index_fields(cellsOnEdge, verticesOnEdge, nEdgesOnCell, edgesOnCell,
                                 edgesOnVertex, maxEdges, 3)
    return MeshData(
        nCells=nC, nEdges=nE, nVertices=nV, maxEdges=maxEdges, maxEdges2=maxEdges2, vertexDegree=3,
        xCell=xC, yCell=yC, zCell=np.zeros(nC), fCell=np.full(nC, f0),
        areaCell=np.full(nC, math.sqrt(3.0) / 2.0 * dc * dc),
        nEdgesOnCell=nEdgesOnCell, edgesOnCell=edgesOnCell,
        verticesOnCell=(verticesOnCell0 + 1).astype(I32), cellsOnCell=(nbr + 1).astype(I32),
        edgeSignOnCell=esc,
        xEdge=xE, yEdge=yE, zEdge=np.zeros(nE), fEdge=np.full(nE, f0), dvEdge=dvEdge, dcEdge=dcEdge,
        angleEdge=angleEdge, nEdgesOnEdge=np.full(nE, 10, dtype=I32), cellsOnEdge=cellsOnEdge,
        verticesOnEdge=verticesOnEdge, edgesOnEdge=edgesOnEdge, weightsOnEdge=weightsOnEdge,
        xVertex=xV, yVertex=yV, zVertex=np.zeros(nV), fVertex=np.full(nV, f0),
        areaTriangle=np.full(nV, math.sqrt(3.0) / 4.0 * dc * dc),
        edgesOnVertex=edgesOnVertex, cellsOnVertex=(cellsOnVertex0 + 1).astype(I32),
        edgeSignOnVertex=esv, on_sphere=False, is_periodic=True,
        meta={"kind": "planar_hex", "nx": nx, "ny": ny, "dc": dc},
        kiteAreasOnVertex=np.full((nV, 3), math.sqrt(3.0) / 12.0 * dc * dc),      # a third of the triangle each
    )


# --------------------------------------------------------------------------------------------
# icosahedral geodesic Voronoi sphere
# --------------------------------------------------------------------------------------------

def _icosahedron():
    phi = (1.0 + math.sqrt(5.0)) / 2.0
    v = np.array([[-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0],
                  [0, -1, phi], [0, 1, phi], [0, -1, -phi], [0, 1, -phi],
                  [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=F64)
    v /= np.linalg.norm(v, axis=1)[:, None]
    # tilt slightly so that no generator sits exactly on a pole or the date line
    a, b = 0.1234, 0.2345
    Rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    Rz = np.array([[math.cos(b), -math.sin(b), 0], [math.sin(b), math.cos(b), 0], [0, 0, 1]])
    v = v @ (Rz @ Rx).T
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    # make every face counter-clockwise seen from outside
    n = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    flip = np.einsum("ij,ij->i", n, v[f[:, 0]]) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    return v, f


def _geodesic_points(m: int):
    """Vertices + triangles of the frequency-m subdivision; shared points merged by exact integer keys."""
    V, F = _icosahedron()
    ii, jj = np.meshgrid(np.arange(m + 1), np.arange(m + 1), indexing="ij")
    keep = (ii + jj) <= m
    ii, jj = ii[keep], jj[keep]          # lattice (i, j): weights (m-i-j, i, j) on corners (A, B, C)
    npf = ii.size
    lid = -np.ones((m + 1, m + 1), dtype=np.int64)
    lid[ii, jj] = np.arange(npf)
    big = np.int64(m + 1)
    keys = np.empty((20, npf), dtype=np.int64)
    pts = np.empty((20, npf, 3))
    for fidx in range(20):
        A, B, C = F[fidx]
        w = np.stack([m - ii - jj, ii, jj], axis=1)           # (npf, 3) integer weights
        corners = np.array([A, B, C])
        pts[fidx] = (w[:, :, None] * V[corners][None, :, :]).sum(axis=1) / m
        # canonical key: sort the (corner, weight) pairs with non-zero weight by corner id
        cw = np.where(w > 0, corners[None, :], 99)            # 99 marks unused
        order = np.argsort(cw, axis=1, kind="stable")
        cs = np.take_along_axis(cw, order, axis=1)
        ws = np.take_along_axis(w, order, axis=1)
        nz = (w > 0).sum(axis=1)
        interior = nz == 3
        k = ((cs[:, 0] * 100 + cs[:, 1]) * big + ws[:, 0]) * big + np.where(nz >= 2, ws[:, 1], 0)
        # interior points are unique to their face
        k = np.where(interior, np.int64(10) ** 15 + (np.int64(fidx) * npf + np.arange(npf)), k)
        keys[fidx] = k
    uk, first, inv = np.unique(keys.ravel(), return_index=True, return_inverse=True)
    P = pts.reshape(-1, 3)[first]
    P /= np.linalg.norm(P, axis=1)[:, None]
    gid = inv.reshape(20, npf)
    # small triangles
    iu, ju = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    ku = (iu + ju) <= m - 1
    iu, ju = iu[ku], ju[ku]
    idn, jdn = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    kd = (idn + jdn) <= m - 2
    idn, jdn = idn[kd], jdn[kd]
    tris = []
    for fidx in range(20):
        g = gid[fidx]
        up = np.stack([g[lid[iu, ju]], g[lid[iu + 1, ju]], g[lid[iu, ju + 1]]], axis=1)
        dn = np.stack([g[lid[idn + 1, jdn]], g[lid[idn + 1, jdn + 1]], g[lid[idn, jdn + 1]]], axis=1)
        tris.append(up)
        tris.append(dn)
    T = np.concatenate(tris, axis=0)
    return P, T


def _unit(a):
    return a / np.linalg.norm(a, axis=-1)[..., None]


def _arc(a, b):
    """Great-circle angle between unit vectors (atan2 form, accurate for small angles)."""
    cr = np.linalg.norm(np.cross(a, b), axis=-1)
    return np.arctan2(cr, np.einsum("...i,...i->...", a, b))


def _sph_tri_area(a, b, c):
    """Signed spherical triangle area on the unit sphere (positive when a,b,c is CCW from outside)."""
    num = np.einsum("...i,...i->...", a, np.cross(b, c))
    den = 1.0 + np.einsum("...i,...i->...", a, b) + np.einsum("...i,...i->...", b, c) + \
        np.einsum("...i,...i->...", c, a)
    return 2.0 * np.arctan2(num, den)


def _flip_edges(T, nflips, seed):
    """Flip `nflips` well-separated interior edges of a triangulation: each flip turns two 6-valent points into
    5-valent and two into 7-valent ones, i.e. the Voronoi dual gains pentagon/heptagon pairs like a
    variable-resolution SCVT has.  Used to exercise the maxEdges = 7/8 code paths."""
    rng = np.random.default_rng(seed)
    T = T.copy()
    used = set()
    done = 0
    order = rng.permutation(T.shape[0])
    for t0 in order:
        if done >= nflips:
            break
        a, b, c = T[t0]
        # neighbour triangle across edge a-b
        cand = np.nonzero(((T == a).any(1)) & ((T == b).any(1)))[0]
        cand = [t for t in cand if t != t0]
        if not cand:
            continue
        t1 = cand[0]
        d = [x for x in T[t1] if x != a and x != b][0]
        pts = {int(a), int(b), int(c), int(d)}
        if pts & used:
            continue
        T[t0] = (a, d, c)
        T[t1] = (b, c, d)
        # keep flips apart: block the whole 1-ring of the four points
        ring = set(T[(np.isin(T, list(pts))).any(1)].ravel().tolist())
        used |= ring
        done += 1
    return T


def icosahedral_mesh(m: int, radius: float = RADIUS_EARTH, omega: float = OMEGA_EARTH, flips: int = 0,
                     seed: int = 0, stretch: float = 1.0) -> MeshData:
    """Voronoi dual of the frequency-m geodesic triangulation: nCells = 10m^2+2, nEdges = 30m^2,
    nVertices = 20m^2 (sizes of SURVEY.md section 8: m=64 -> 40 962 cells, m=320 -> 1 024 002).
    flips > 0 flips that many edges of the triangulation first (5/7-gon pairs, maxEdges = 7).
    stretch = c > 1 gives a variable-resolution mesh: the generators are moved by the Schmidt transformation
    sin(lat') = (D + sin(lat)) / (1 + D sin(lat)), D = (1 - c^2) / (1 + c^2) (longitude kept): a conformal (Moebius)
    map of the sphere, so circumcircles stay circles and the triangulation stays Delaunay; cell spacing varies by
    c^2 between the poles (c = 4.47: ~3 km to ~52 km at m = 608, BASELINE config 5's "3-60 km")."""
    P, T = _geodesic_points(m)
    if stretch != 1.0:
        c2 = float(stretch) ** 2
        D = (1.0 - c2) / (1.0 + c2)
        z = (D + P[:, 2]) / (1.0 + D * P[:, 2])
        rxy = np.sqrt(np.maximum(1.0 - z * z, 0.0)) / np.maximum(np.hypot(P[:, 0], P[:, 1]), 1e-300)
        P = np.stack([P[:, 0] * rxy, P[:, 1] * rxy, z], axis=1)
        P = _unit(P)
    if flips:
        T = _flip_edges(T, flips, seed)
    nC, nV = P.shape[0], T.shape[0]
    assert nC == 10 * m * m + 2 and nV == 20 * m * m
    # make triangles CCW seen from outside
    nrm = np.cross(P[T[:, 1]] - P[T[:, 0]], P[T[:, 2]] - P[T[:, 0]])
    flip = np.einsum("ij,ij->i", nrm, P[T[:, 0]]) < 0
    T[flip] = T[flip][:, [0, 2, 1]]
    nrm[flip] *= -1.0
    Vx = _unit(nrm)                                        # circumcentres = Voronoi vertices

    # edges from triangle sides; side s of triangle t joins T[t,s], T[t,s+1]; left of a->b is t
    a = T[:, [0, 1, 2]].ravel()
    b = T[:, [1, 2, 0]].ravel()
    tri = np.repeat(np.arange(nV), 3)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = lo.astype(np.int64) * nC + hi
    order = np.argsort(key, kind="stable")
    ks = key[order]
    assert np.all(ks[0::2] == ks[1::2]), "every edge must be shared by exactly two triangles"
    nE = ks.size // 2
    assert nE == 30 * m * m
    h0, h1 = order[0::2], order[1::2]                       # the two half-edges of each edge
    c1 = lo[h0]
    c2 = hi[h0]
    # half-edge h0 runs a->b with triangle on its left.  Normal n points c1->c2; k x n is 90deg CCW.
    # If h0 runs c1->c2 its triangle (left side) is in direction k x n  => it is v2, the other v1.
    fwd = a[h0] == c1
    v_left0, v_left1 = tri[h0], tri[h1]
    v2 = np.where(fwd, v_left0, v_left1)
    v1 = np.where(fwd, v_left1, v_left0)
    cellsOnEdge0 = np.stack([c1, c2], axis=1)
    verticesOnEdge0 = np.stack([v1, v2], axis=1)

    Pc1, Pc2 = P[c1], P[c2]
    Em = _unit(Pc1 + Pc2)                                   # edge point: midpoint of the cell-centre arc
    dcEdge = radius * _arc(Pc1, Pc2)
    dvEdge = radius * _arc(Vx[v1], Vx[v2])
    # unit normal at the edge point (tangent, towards c2) and angleEdge relative to local east
    nvec = Pc2 - Pc1
    nvec -= np.einsum("ij,ij->i", nvec, Em)[:, None] * Em
    nvec = _unit(nvec)
    east = _unit(np.cross(np.array([0.0, 0.0, 1.0])[None, :], Em))
    north = np.cross(Em, east)
    angleEdge = np.arctan2(np.einsum("ij,ij->i", nvec, north), np.einsum("ij,ij->i", nvec, east))
    latE = np.arcsin(np.clip(Em[:, 2], -1, 1))

    # vertex tables
    cellsOnVertex0 = T.copy()                               # CCW
    # edge id for each triangle side
    edge_of_half = np.empty(3 * nV, dtype=np.int64)
    edge_of_half[h0] = np.arange(nE)
    edge_of_half[h1] = np.arange(nE)
    eov = edge_of_half.reshape(nV, 3)                       # side s joins cell s and s+1
    # MPAS lists edgesOnVertex so that edge j is "opposite-ish"; order is irrelevant for curl, keep CCW
    edgesOnVertex0 = eov
    areaTriangle = radius * radius * _sph_tri_area(P[T[:, 0]], P[T[:, 1]], P[T[:, 2]])

    # cell tables: incidences (cell, edge) sorted CCW around the cell
    inc_c = np.concatenate([c1, c2])
    inc_e = np.concatenate([np.arange(nE), np.arange(nE)])
    cnt = np.bincount(inc_c, minlength=nC)
    maxEdges = int(cnt.max())
    assert 6 <= maxEdges <= 8
    # reference direction: towards the incidence with the smallest edge id of that cell
    o = np.lexsort((inc_e, inc_c))
    inc_c, inc_e = inc_c[o], inc_e[o]
    start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    d = Em[inc_e] - P[inc_c]
    d -= np.einsum("ij,ij->i", d, P[inc_c])[:, None] * P[inc_c]
    d0 = d[start][np.repeat(np.arange(nC), cnt)]
    rc = P[inc_c]
    ang = np.arctan2(np.einsum("ij,ij->i", rc, np.cross(d0, d)), np.einsum("ij,ij->i", d0, d))
    ang = np.where(ang < -1e-12, ang + 2 * np.pi, ang)
    o2 = np.lexsort((ang, inc_c))
    inc_c, inc_e = inc_c[o2], inc_e[o2]
    slot = np.arange(inc_c.size) - start[inc_c]
    edgesOnCell0 = -np.ones((nC, maxEdges), dtype=np.int64)
    edgesOnCell0[inc_c, slot] = inc_e
    nEdgesOnCell = cnt.astype(I32)
    # neighbour cell and the vertex *after* each edge slot (between edge i and i+1, CCW)
    e_safe = np.where(edgesOnCell0 >= 0, edgesOnCell0, 0)
    is_c1 = cellsOnEdge0[e_safe, 0] == np.arange(nC)[:, None]
    cellsOnCell0 = np.where(is_c1, cellsOnEdge0[e_safe, 1], cellsOnEdge0[e_safe, 0])
    cellsOnCell0 = np.where(edgesOnCell0 >= 0, cellsOnCell0, -1)
    # walking CCW around c, crossing edge e with c == c1: k x n points CCW => next vertex is v2
    vertAfter0 = np.where(is_c1, verticesOnEdge0[e_safe, 1], verticesOnEdge0[e_safe, 0])
    vertAfter0 = np.where(edgesOnCell0 >= 0, vertAfter0, -1)

    # kite areas: kite(c, v) = quad (c, mid(e_i), v, mid(e_{i+1})) for v between edge i and i+1
    areaCell = np.zeros(nC)
    kiteR = np.zeros((nC, maxEdges))
    for i in range(maxEdges):
        act = i < cnt
        inext = np.where(i + 1 < cnt, i + 1, 0)
        ei = e_safe[:, i]
        en = e_safe[np.arange(nC), inext]
        vv = np.where(act, vertAfter0[:, i], 0)
        kite = _sph_tri_area(P, Em[ei], Vx[vv]) + _sph_tri_area(P, Vx[vv], Em[en])
        kite = np.where(act, kite, 0.0)
        kiteR[:, i] = kite
        areaCell += kite
    kiteAreasOnVertex = np.zeros((nV, 3))
    for i in range(maxEdges):
        act = i < cnt
        cc = np.nonzero(act)[0]
        vv = vertAfter0[cc, i]
        for j in range(3):
            hit = cellsOnVertex0[vv, j] == cc
            kiteAreasOnVertex[vv[hit], j] = kiteR[cc[hit], i] * radius * radius
    kiteR /= areaCell[:, None]
    areaCell *= radius * radius
    assert abs(areaCell.sum() / (4 * np.pi * radius * radius) - 1.0) < (1e-9 if not flips else 1e-2)

    # TRiSK weights (Thuburn et al. 2009 / Ringler et al. 2010), general form of Appendix B
    maxEdges2 = 2 * maxEdges
    edgesOnEdge0 = -np.ones((nE, maxEdges2), dtype=np.int64)
    weightsOnEdge = np.zeros((nE, maxEdges2))
    nEdgesOnEdge = np.zeros(nE, dtype=I32)
    # slot of each edge in each of its two cells
    slot_in = np.zeros((nE, 2), dtype=np.int64)
    for i in range(maxEdges):
        act = i < cnt
        e = edgesOnCell0[act, i]
        cc = np.nonzero(act)[0]
        side = (cellsOnEdge0[e, 0] != cc).astype(np.int64)
        slot_in[e, side] = i
    fill = np.zeros(nE, dtype=np.int64)
    ar = np.arange(nE)
    for side in range(2):
        cell = cellsOnEdge0[:, side]
        n = cnt[cell]
        m0 = slot_in[:, side]
        s = 1.0 if side == 0 else -1.0
        run = np.zeros(nE)
        for k in range(1, maxEdges):
            act = k < n
            run = run + kiteR[cell, (m0 + k - 1) % n]
            sl = (m0 + k) % n
            e2 = edgesOnCell0[cell, sl]
            e2s = np.where(act, e2, 0)
            nout = np.where(cellsOnEdge0[e2s, 0] == cell, 1.0, -1.0)
            w = -s * (run - 0.5) * nout * dvEdge[e2s] / dcEdge
            pos = fill[act]
            edgesOnEdge0[ar[act], pos] = e2[act]
            weightsOnEdge[ar[act], pos] = w[act]
            fill[act] += 1
    nEdgesOnEdge[:] = fill

    edgesOnCell = (edgesOnCell0 + 1).astype(I32)
    cellsOnEdge = (cellsOnEdge0 + 1).astype(I32)
    verticesOnEdge = (verticesOnEdge0 + 1).astype(I32)
    edgesOnVertex = (edgesOnVertex0 + 1).astype(I32)
    esc, esv = sign_index_fields(cellsOnEdge, verticesOnEdge, nEdgesOnCell, edgesOnCell,
                                 edgesOnVertex, maxEdges, 3)
    latC = np.arcsin(np.clip(P[:, 2], -1, 1))
    latV = np.arcsin(np.clip(Vx[:, 2], -1, 1))
    return MeshData(
        nCells=nC, nEdges=nE, nVertices=nV, maxEdges=maxEdges, maxEdges2=maxEdges2, vertexDegree=3,
        xCell=radius * P[:, 0], yCell=radius * P[:, 1], zCell=radius * P[:, 2],
        fCell=2 * omega * np.sin(latC), areaCell=areaCell, nEdgesOnCell=nEdgesOnCell,
        edgesOnCell=edgesOnCell, verticesOnCell=(vertAfter0 + 1).astype(I32),
        cellsOnCell=(cellsOnCell0 + 1).astype(I32), edgeSignOnCell=esc,
        xEdge=radius * Em[:, 0], yEdge=radius * Em[:, 1], zEdge=radius * Em[:, 2],
        fEdge=2 * omega * np.sin(latE), dvEdge=dvEdge, dcEdge=dcEdge, angleEdge=angleEdge,
        nEdgesOnEdge=nEdgesOnEdge, cellsOnEdge=cellsOnEdge, verticesOnEdge=verticesOnEdge,
        edgesOnEdge=(edgesOnEdge0 + 1).astype(I32), weightsOnEdge=weightsOnEdge,
        xVertex=radius * Vx[:, 0], yVertex=radius * Vx[:, 1], zVertex=radius * Vx[:, 2],
        fVertex=2 * omega * np.sin(latV), areaTriangle=areaTriangle,
        edgesOnVertex=edgesOnVertex, cellsOnVertex=(cellsOnVertex0 + 1).astype(I32),
        edgeSignOnVertex=esv, on_sphere=True, sphere_radius=radius, is_periodic=True,
        meta={"kind": "icosahedral", "m": m, "radius": radius}, kiteAreasOnVertex=kiteAreasOnVertex,
    )


# --------------------------------------------------------------------------------------------
# initial states
# --------------------------------------------------------------------------------------------

@dataclass
class IGWParams:
    """Constants of `src/inertialGravityWave.jl:6-19` (lx in km, as there)."""
    g: float = GRAVITY
    f0: float = 1e-4
    npx: float = 2.0
    npy: float = 2.0
    eta0: float = 1.0
    bottom_depth: float = 1000.0
    lx: float = 10000.0

    @property
    def ly(self):
        return math.sqrt(3.0) / 2.0 * self.lx

    @property
    def kx(self):
        return self.npx * 2.0 * math.pi / (self.lx * 1e3)

    @property
    def ky(self):
        return self.npy * 2.0 * math.pi / (self.ly * 1e3)

    @property
    def omega(self):
        return math.sqrt(self.f0 ** 2 + self.g * self.bottom_depth * (self.kx ** 2 + self.ky ** 2))


def igw_mesh(resolution_km: float) -> MeshData:
    """polaris `inertial_gravity_wave` sizing: 200 km -> 50x50."""
    p = IGWParams()
    nx = max(2 * int(0.5 * p.lx / resolution_km + 0.5), 4)
    ny = max(2 * int(0.5 * p.ly * (2.0 / math.sqrt(3.0)) / resolution_km + 0.5), 4)
    return planar_hex_mesh(nx, ny, resolution_km * 1e3, f0=p.f0)


def igw_exact(mesh: MeshData, t: float, p: IGWParams | None = None):
    """exact_ssh / exact_norm_vel of `src/inertialGravityWave.jl:38-63`."""
    p = p or IGWParams()
    ssh = p.eta0 * np.cos(p.kx * mesh.xCell + p.ky * mesh.yCell - p.omega * t)
    ph = p.kx * mesh.xEdge + p.ky * mesh.yEdge - p.omega * t
    fac = p.eta0 * p.g / (p.omega ** 2 - p.f0 ** 2)
    u = fac * (p.omega * p.kx * np.cos(ph) - p.f0 * p.ky * np.sin(ph))
    v = fac * (p.omega * p.ky * np.cos(ph) + p.f0 * p.kx * np.sin(ph))
    un = u * np.cos(mesh.angleEdge) + v * np.sin(mesh.angleEdge)
    return ssh, un


def igw_initial_state(mesh: MeshData, p: IGWParams | None = None):
    """ssh, normalVelocity (nE,1), layerThickness (nC,1), restingThickness (nC,1) at t = 0."""
    p = p or IGWParams()
    ssh, un = igw_exact(mesh, 0.0, p)
    rest = np.full((mesh.nCells, 1), p.bottom_depth)
    h = rest + ssh[:, None]
    return ssh.copy(), un[:, None].copy(), h, rest


def igw_dt(mesh: MeshData) -> float:
    """dt rule of `src/forward/init.jl:118`."""
    mdc = float(np.mean(mesh.dcEdge))
    return math.floor(2 * (mdc / 1e3) * mdc / 200e3)


def sphere_synthetic_state(mesh: MeshData, K: int, seed: int = 20250216, total_depth: float = 4000.0):
    """Synthetic inputs of SURVEY.md section 8(d) for configs 2-5."""
    rng = np.random.default_rng(seed)
    r = mesh.sphere_radius
    lat = np.arcsin(np.clip(mesh.zCell / r, -1, 1))
    lon = np.arctan2(mesh.yCell, mesh.xCell)
    rest = np.full((mesh.nCells, K), total_depth / K)
    h = rest + (1.0 / K) * (np.cos(lat) * np.cos(4 * lon))[:, None]
    u = 0.1 * rng.uniform(-1.0, 1.0, size=(mesh.nEdges, K))
    ssh = h.sum(axis=1) - rest.sum(axis=1)
    dt = 0.2 * float(mesh.dcEdge.min()) / math.sqrt(GRAVITY * total_depth)
    return ssh, u, h, rest, dt
