"""Multi-GPU layer: owner-computes cell partition, one-cell-deep halo, halo exchange per
RK stage overlapped with the interior patches.  One process per GPU.

The reference has no distributed code at all; this is new.  Pure-numpy pieces (partition, local mesh,
exchange lists) need no GPU and are covered by world-size-2 gloo tests; the device pieces (pack / unpack kernels,
patch-range launches, streams) live in libmoka_hip (moka_halo_*, moka_rk4_dist_*).
"""
from __future__ import annotations

import ctypes as C
import types

import numpy as np

from . import api
from . import lib as L

__all__ = ["partition_cells", "build_local", "LocalMesh", "DistributedModel"]


# ---------------------------------------------------------------------------------------------
# partition: recursive coordinate bisection (no METIS in this image; a METIS file can be dropped in as `part`)
# ---------------------------------------------------------------------------------------------
def partition_cells(mesh, nparts: int) -> np.ndarray:
    X = np.stack([mesh.xCell, mesh.yCell, mesh.zCell], axis=1)
    part = np.zeros(mesh.nCells, dtype=np.int32)

    def split(idx, p0, n):
        if n == 1:
            part[idx] = p0
            return
        nl = n // 2
        ext = X[idx].max(0) - X[idx].min(0)
        ax = int(np.argmax(ext))
        k = int(round(idx.size * nl / n))
        order = np.argpartition(X[idx, ax], k - 1 if k > 0 else 0)
        split(idx[order[:k]], p0, nl)
        split(idx[order[k:]], p0 + nl, n - nl)

    split(np.arange(mesh.nCells), 0, int(nparts))
    return part


def edge_cut(mesh, part) -> int:
    """Number of edges whose two cells lie in different parts (the halo surface a partition costs)."""
    c = np.asarray(mesh.cellsOnEdge).reshape(mesh.nEdges, 2) - 1
    ok = (c[:, 0] >= 0) & (c[:, 1] >= 0)
    return int((part[c[ok, 0]] != part[c[ok, 1]]).sum())


def write_graph_info(mesh, path: str):
    """The cell graph in the METIS format MPAS tools use (`graph.info`: "nCells nAdjacencies", then the 1-based neighbour
    cells of every cell): `gpmetis graph.info N` gives `graph.info.part.N`, which read_partition() / the `part=` argument of
    DistributedModel take."""
    c = np.asarray(mesh.cellsOnEdge).reshape(mesh.nEdges, 2)
    ok = (c[:, 0] >= 1) & (c[:, 1] >= 1)
    nbrs = [[] for _ in range(mesh.nCells)]
    for a, b in c[ok]:
        nbrs[a - 1].append(int(b)); nbrs[b - 1].append(int(a))
    with open(path, "w") as f:
        f.write(f"{mesh.nCells} {int(ok.sum())}\n")
        for lst in nbrs:
            f.write(" ".join(str(x) for x in lst) + "\n")


def read_partition(path: str, nCells: int) -> np.ndarray:
    """`graph.info.part.N` (one 0-based part number per cell)."""
    part = np.loadtxt(path, dtype=np.int64).reshape(-1)
    if part.size != nCells or part.min() < 0:
        raise ValueError(f"{path}: expected {nCells} non-negative part numbers, found {part.size}")
    return part.astype(np.int32)


class LocalMesh:
    """A rank's local mesh (owned cells + one ring of halo cells, every edge of those cells) in reference
    conventions, plus the maps needed for the exchange."""
    pass


def _halo_sets(mesh, part, r, rings=1):
    """(halo cells of rank r -- `rings` rings around its cells --, local cell mask, local edge mask, edges rank r must receive,
    the rank that sends each edge).

    An edge without an r-owned cell cannot be computed on r and arrives by exchange.  Its sender is the LOWEST rank
    among the owners of those of its cells that are local on r: every such owner computes the edge (it owns one of
    its cells), and the rule matches the numbering of r's local plan, where a halo-halo edge belongs to the cell of
    the lower class (= the lower neighbour), so that what one neighbour sends is one contiguous range of edges."""
    coe = mesh.cellsOnEdge.astype(np.int64) - 1
    own = part == r
    o1, o2 = own[coe[:, 0]], own[coe[:, 1]]
    ring = np.zeros(mesh.nCells, dtype=bool)
    ring[coe[o1 & ~o2, 1]] = True
    ring[coe[o2 & ~o1, 0]] = True
    for _ in range(int(rings) - 1):                   # the optional nonlinear terms reach two cells deep
        inn = own | ring
        i1, i2 = inn[coe[:, 0]], inn[coe[:, 1]]
        ring[coe[i1 & ~i2, 1]] = True
        ring[coe[i2 & ~i1, 0]] = True
    local_c = own | ring
    l1, l2 = local_c[coe[:, 0]], local_c[coe[:, 1]]
    local_e = l1 | l2
    recv_e = local_e & ~o1 & ~o2                      # no owned cell incident: cannot be computed here
    big = np.iinfo(np.int32).max
    p1 = np.where(l1, part[coe[:, 0]], big)
    p2 = np.where(l2, part[coe[:, 1]], big)
    sender = np.minimum(p1, p2)
    return ring, local_c, local_e, recv_e, sender


def build_local(mesh, part: np.ndarray, rank: int, world: int, rings: int = 1, vertex_fields: bool = False) -> LocalMesh:
    """rings: depth of the halo in cells (1: the reference's linear terms; 2: the optional nonlinear terms, whose stencil --
    kinetic energy and potential vorticity of the cells / vertices around an edge's two cells -- reaches one ring further).
    vertex_fields: carry verticesOnEdge, cellsOnVertex, kiteAreasOnVertex and fVertex too (the nonlinear terms read them)."""
    coe = mesh.cellsOnEdge.astype(np.int64) - 1
    own = part == rank
    ring, local_c, local_e, recv_e, sender = _halo_sets(mesh, part, rank, rings)

    # ---- exchange lists (global ids, sorted: both sides derive the same order) ----
    recv_cells = {q: np.nonzero(ring & (part == q))[0] for q in range(world) if q != rank}
    recv_edges = {q: np.nonzero(recv_e & (sender == q))[0] for q in range(world) if q != rank}
    send_cells, send_edges = {}, {}
    for q in range(world):
        if q == rank:
            continue
        ring_q, _, _, recv_e_q, sender_q = _halo_sets(mesh, part, q, rings)
        send_cells[q] = np.nonzero(ring_q & own)[0]
        send_edges[q] = np.nonzero(recv_e_q & (sender_q == rank))[0]
    nbrs = [q for q in range(world) if q != rank and
            (recv_cells[q].size or recv_edges[q].size or send_cells[q].size or send_edges[q].size)]

    # ---- cell classes: 0 owned & needed elsewhere, 1 owned interior, 2 + i halo cells owned by neighbour i ----
    boundary = np.zeros(mesh.nCells, dtype=bool)
    for q in nbrs:
        boundary[send_cells[q]] = True
        ce = coe[send_edges[q]]                       # a sent edge is computed by the boundary launch: its owned cell(s)
        boundary[ce[own[ce]]] = True
    assert not np.any(boundary & ~own)

    cells_g = np.concatenate([np.nonzero(own)[0], np.nonzero(ring)[0]])
    edges_g = np.nonzero(local_e)[0]
    g2l_c = -np.ones(mesh.nCells, dtype=np.int64)
    g2l_c[cells_g] = np.arange(cells_g.size)
    g2l_e = -np.ones(mesh.nEdges, dtype=np.int64)
    g2l_e[edges_g] = np.arange(edges_g.size)
    # vertices: those whose edges are all local
    eov = mesh.edgesOnVertex.astype(np.int64) - 1
    vert_ok = np.all(g2l_e[eov] >= 0, axis=1)
    verts_g = np.nonzero(vert_ok)[0]
    if verts_g.size == 0:
        raise ValueError("partition too small: no complete dual cell on this rank")

    m = types.SimpleNamespace()
    m.nCells, m.nEdges, m.nVertices = int(cells_g.size), int(edges_g.size), int(verts_g.size)
    m.maxEdges, m.maxEdges2, m.vertexDegree = mesh.maxEdges, mesh.maxEdges2, mesh.vertexDegree
    for n in ("xCell", "yCell", "zCell", "areaCell"):
        setattr(m, n, np.ascontiguousarray(getattr(mesh, n)[cells_g]))
    m.nEdgesOnCell = np.ascontiguousarray(mesh.nEdgesOnCell[cells_g])
    eoc = mesh.edgesOnCell[cells_g].astype(np.int64) - 1
    m.edgesOnCell = np.where(eoc >= 0, g2l_e[np.maximum(eoc, 0)] + 1, 0).astype(np.int32)
    assert np.all(m.edgesOnCell[np.arange(mesh.maxEdges)[None, :] < m.nEdgesOnCell[:, None]] > 0)
    m.edgeSignOnCell = np.ascontiguousarray(mesh.edgeSignOnCell[cells_g])
    lc = g2l_c[coe[edges_g]]                           # (nEl, 2) local cells, -1 outside
    inside = np.where(lc[:, 0] >= 0, lc[:, 0], lc[:, 1])
    lc = np.where(lc >= 0, lc, inside[:, None])       # outer boundary of the halo: both sides -> the inside cell
    m.cellsOnEdge = (lc + 1).astype(np.int32)
    m.verticesOnEdge = np.zeros((m.nEdges, 2), dtype=np.int32)   # not used by the hot path
    eoe = mesh.edgesOnEdge[edges_g].astype(np.int64) - 1
    m.edgesOnEdge = np.where(eoe >= 0, g2l_e[np.maximum(eoe, 0)] + 1, 0).astype(np.int32)   # non-local -> 0 (skipped)
    m.nEdgesOnEdge = np.ascontiguousarray(mesh.nEdgesOnEdge[edges_g])
    for n in ("weightsOnEdge", "dvEdge", "dcEdge", "fEdge"):
        setattr(m, n, np.ascontiguousarray(getattr(mesh, n)[edges_g]))
    m.edgesOnVertex = (g2l_e[eov[verts_g]] + 1).astype(np.int32)
    m.cellsOnVertex = np.zeros((m.nVertices, mesh.vertexDegree), dtype=np.int32)   # 0: let the library derive it
    m.edgeSignOnVertex = np.ascontiguousarray(mesh.edgeSignOnVertex[verts_g])
    m.areaTriangle = np.ascontiguousarray(mesh.areaTriangle[verts_g])
    if vertex_fields:
        # Every vertex whose potential vorticity an owned entity reads (the vertices of the cells up to one ring out) has its
        # three cells and edges inside a two-ring halo.  Rim vertices / edges get a valid local stand-in: what is computed
        # from it is never read by an owned entity.
        g2l_v = -np.ones(mesh.nVertices, dtype=np.int64)
        g2l_v[verts_g] = np.arange(verts_g.size)
        cov = g2l_c[mesh.cellsOnVertex[verts_g].astype(np.int64) - 1]                 # (nVl, 3), -1 = outside
        first = np.where(cov >= 0, cov, cov.max(axis=1, keepdims=True))
        m.cellsOnVertex = (first + 1).astype(np.int32)
        m.kiteAreasOnVertex = np.ascontiguousarray(mesh.kiteAreasOnVertex[verts_g])
        m.fVertex = np.ascontiguousarray(mesh.fVertex[verts_g])
        voe = g2l_v[mesh.verticesOnEdge[edges_g].astype(np.int64) - 1]                # (nEl, 2), -1 = outside
        stand = np.where(voe.max(axis=1, keepdims=True) >= 0, voe.max(axis=1, keepdims=True), 0)
        m.verticesOnEdge = (np.where(voe >= 0, voe, stand) + 1).astype(np.int32)
        lm_complete_v = np.all(cov >= 0, axis=1)
    else:
        m.kiteAreasOnVertex = None

    lm = LocalMesh()
    lm.rank, lm.world, lm.mesh = rank, world, m
    lm.rings, lm.vertex_fields = int(rings), bool(vertex_fields)
    lm.cells_g, lm.edges_g, lm.verts_g = cells_g, edges_g, verts_g
    lm.g2l_c, lm.g2l_e = g2l_c, g2l_e
    lm.n_owned_cells = int(own.sum())
    lm.owned_cell_mask = own[cells_g]
    # an edge is reported by the rank of cellsOnEdge[1] (every edge with an owned cell is computed locally)
    lm.owned_edge_mask = own[coe[edges_g, 0]]
    # a vertex is reported by the owner of its first edge's first cell: all three of its edges are local there
    lm.owned_vert_mask = own[coe[eov[verts_g, 0], 0]]
    nbr_index = {q: i for i, q in enumerate(nbrs)}
    halo_class = np.array([2 + nbr_index[q] if q in nbr_index else 2 for q in part[cells_g]], dtype=np.int32)
    lm.cell_class = np.where(boundary[cells_g], 0, np.where(own[cells_g], 1, halo_class)).astype(np.int32)
    lm.neighbors = nbrs
    # local ids, concatenated in neighbour order, + per-neighbour offsets
    cat = lambda d, g2l: (np.concatenate([g2l[d[q]] for q in nbrs]).astype(np.int32) if nbrs else np.zeros(0, np.int32),
                          np.cumsum([0] + [d[q].size for q in nbrs]))
    lm.send_cells, lm.send_cell_off = cat(send_cells, g2l_c)
    lm.send_edges, lm.send_edge_off = cat(send_edges, g2l_e)
    lm.recv_cells, lm.recv_cell_off = cat(recv_cells, g2l_c)
    lm.recv_edges, lm.recv_edge_off = cat(recv_edges, g2l_e)
    return lm


def message_slices(lm: LocalMesh, K: int, send: bool):
    """Per neighbour: the contiguous slice [h rows | ssh | u rows] of the packed buffer that goes to / comes from it."""
    co, eo = (lm.send_cell_off, lm.send_edge_off) if send else (lm.recv_cell_off, lm.recv_edge_off)
    out = []
    for i, q in enumerate(lm.neighbors):
        a = int(co[i]) * (K + 1) + int(eo[i]) * K
        b = int(co[i + 1]) * (K + 1) + int(eo[i + 1]) * K
        out.append((q, a, b))
    return out


def alltoall_splits(lm: LocalMesh, K: int):
    """(input_split_sizes, output_split_sizes) of the one all_to_all_single that moves every halo message of a stage:
    neighbours are kept in ascending rank order, so the packed buffers ARE the rank-ordered concatenation."""
    assert list(lm.neighbors) == sorted(lm.neighbors)
    ins, outs = [0] * lm.world, [0] * lm.world
    for q, a, b in message_slices(lm, K, True):
        ins[q] = b - a
    for q, a, b in message_slices(lm, K, False):
        outs[q] = b - a
    return ins, outs


def pack_numpy(lm: LocalMesh, K: int, ssh, u, h):
    """The library's send-buffer layout, in numpy (used by the CPU tests)."""
    parts = []
    for i in range(len(lm.neighbors)):
        c = lm.send_cells[lm.send_cell_off[i]:lm.send_cell_off[i + 1]]
        e = lm.send_edges[lm.send_edge_off[i]:lm.send_edge_off[i + 1]]
        parts += [h[c].ravel(), ssh[c], u[e].ravel()]
    return np.concatenate(parts) if parts else np.zeros(0)


def unpack_numpy(lm: LocalMesh, K: int, buf, ssh, u, h):
    pos = 0
    for i in range(len(lm.neighbors)):
        c = lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]
        e = lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]
        h[c] = buf[pos:pos + c.size * K].reshape(c.size, K)
        pos += c.size * K
        ssh[c] = buf[pos:pos + c.size]
        pos += c.size
        u[e] = buf[pos:pos + e.size * K].reshape(e.size, K)
        pos += e.size * K


# ---------------------------------------------------------------------------------------------
# device model
# ---------------------------------------------------------------------------------------------
class DistributedModel:
    """Partitioned shallow-water model: RK4 (and reference Forward-Euler) steps with one halo exchange per stage.

    Halo transports:
      "ipc"       direct: every rank pushes its rows straight into the neighbours' fields (IPC-mapped device memory over
                  xGMI between processes), completion through flag words in shared host memory; one library call per
                  step (moka_rk4_dist_step), no send buffer, no unpack, no collective library;
      "nccl-a2a"  buffered: one all_to_all_single per stage on device buffers (RCCL: a grouped send/recv per neighbour
                  underneath), issued on the library's comm stream so that the interior patches overlap it;
      "nccl"      the same messages as batched P2P ops;  "nccl-default-stream": P2P with full synchronisation;
      "gloo"      buffered, staged through the host (tests: ranks may share one GPU);
      "local"     driven by LocalCluster (all ranks in one process)."""

    def __init__(self, mesh, ssh, u, h, rest, dt, backend, rank, world, ordering=0, patch_cells=0,
                 transport="nccl", part=None, group=None, state_bytes=8, exchange_lists=None, timeout_s=30.0, nonlinear=False,
                 placement_tries=24,
                 visc_del2=0.0):
        """exchange_lists(wants: {rank: obj}) -> {rank: obj}: all-to-all of small Python objects between the ranks
        (default: torch.distributed.all_gather_object on `group`); LocalCluster passes None and calls finish() itself."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.dt, self.backend, self.transport = rank, world, float(dt), backend, transport
        self.group = group                 # process group of the gloo transport / control messages (None = default group)
        self.timeout_s = float(timeout_s)
        K = np.asarray(u).reshape(mesh.nEdges, -1).shape[1]
        self.K, self.state_bytes = K, int(state_bytes)
        self.part = partition_cells(mesh, world) if part is None else np.asarray(part, dtype=np.int32)
        # the optional nonlinear terms: a two-ring halo (their stencil reaches one ring further) carrying the vertex-side fields;
        # the whole local mesh is computed every stage (step_rk4_whole), the halo's share redundantly
        self.nonlinear = bool(nonlinear)
        lm = self.lm = build_local(mesh, self.part, rank, world, rings=2 if nonlinear else 1, vertex_fields=self.nonlinear)
        rest = np.asarray(rest).reshape(mesh.nCells, -1)
        h_mesh = api.HorzMesh(lm.mesh)
        v_mesh = api.VerticalMesh(h_mesh, nVertLevels=K, restingThickness=rest[lm.cells_g], multilayer=True)
        self.mesh = api.Mesh.__new__(api.Mesh)
        self.mesh.HorzMesh, self.mesh.VertMesh, self.mesh.backend = h_mesh, v_mesh, backend
        self.mesh.state_bytes = int(state_bytes)
        self.mesh._h = C.c_void_p()
        desc, keep = L.make_desc(lm.mesh, K, v_mesh.restingThicknessSum, v_mesh.maxLevelEdge.Top, ordering,
                                 patch_cells, cell_class=lm.cell_class, state_bytes=state_bytes)
        if not self.nonlinear:
            desc.cellsOnVertex = None
            desc.verticesOnEdge = None
        L.check(L.lib().moka_mesh_create(backend._h, C.byref(desc), C.byref(self.mesh._h)), backend._h)
        api._own(self.mesh, L.lib().moka_mesh_destroy, self.mesh._h, backend)
        # placement_tries > 1 (default): the library's per-array placement search on this rank's state
        # (moka_state_optimize_placement through api.prognostic_vars_best_placement) -- before the halo exists: its peers
        # address the arrays chosen here, and the library refuses to move them afterwards
        self.placement = {}
        self.Prog = api.prognostic_vars_best_placement(np.asarray(ssh)[lm.cells_g], np.asarray(u).reshape(mesh.nEdges, K)[lm.edges_g],
                                                       np.asarray(h).reshape(mesh.nCells, K)[lm.cells_g], 2, self.mesh,
                                                       tries=placement_tries, report=self.placement)
        self.Diag = api.DiagnosticVars(None, self.mesh, self.Prog._state)
        self.Tend = api.TendencyVars(None, self.mesh, self.Prog._state)
        if self.nonlinear:
            api.set_nonlinear(self.Prog, True, visc_del2=visc_del2)
        # launch ranges and receive order come from the plan: class 0 = boundary patches, 1 = interior, 2 + i = what
        # neighbour i sends, contiguous and in the library's order
        pS, cS, eS = L.class_ranges(self.mesh._h, True)
        self.p_boundary, self.p_owned = int(pS[1]), int(pS[2])
        cperm = np.empty(lm.mesh.nCells, dtype=np.int32)
        eperm = np.empty(lm.mesh.nEdges, dtype=np.int32)
        L.check(L.lib().moka_mesh_permutation(self.mesh._h, L.CELL, L.i32(cperm)), backend._h)
        L.check(L.lib().moka_mesh_permutation(self.mesh._h, L.EDGE, L.i32(eperm)), backend._h)
        rc, re_ = [], []
        for i in range(len(lm.neighbors)):
            c = cperm[cS[2 + i]:cS[3 + i]]
            e = eperm[eS[2 + i]:eS[3 + i]]
            assert np.array_equal(np.sort(c), np.sort(lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]))
            assert np.array_equal(np.sort(e), np.sort(lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]))
            rc.append(c); re_.append(e)
        if lm.neighbors:
            lm.recv_cells, lm.recv_edges = np.concatenate(rc).astype(np.int32), np.concatenate(re_).astype(np.int32)
        # the senders have to list their rows in the same order: tell them (global ids)
        self._wants = {q: (lm.cells_g[rc[i]], lm.edges_g[re_[i]]) for i, q in enumerate(lm.neighbors)}
        self._halo = C.c_void_p()
        self._ready = False
        if exchange_lists is not None or transport != "local":
            self.finish((exchange_lists or self._gather_lists)(self._wants))

    def _gather_lists(self, wants):
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, wants, group=self.group)
        return {q: everyone[q][self.rank] for q in self.lm.neighbors}

    def finish(self, asked):
        """asked[q] = (global cell ids, global edge ids) neighbour q wants from this rank, in the order it wants them."""
        torch, lm, backend, K = self.torch, self.lm, self.backend, self.K
        sc, se = [], []
        for i, q in enumerate(lm.neighbors):
            c, e = lm.g2l_c[np.asarray(asked[q][0], dtype=np.int64)], lm.g2l_e[np.asarray(asked[q][1], dtype=np.int64)]
            assert np.array_equal(np.sort(c), np.sort(lm.send_cells[lm.send_cell_off[i]:lm.send_cell_off[i + 1]]))
            assert np.array_equal(np.sort(e), np.sort(lm.send_edges[lm.send_edge_off[i]:lm.send_edge_off[i + 1]]))
            sc.append(c); se.append(e)
        if lm.neighbors:
            lm.send_cells, lm.send_edges = np.concatenate(sc).astype(np.int32), np.concatenate(se).astype(np.int32)
        self._keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (lm.send_cells, lm.send_edges, lm.recv_cells, lm.recv_edges)]
        offs = [np.ascontiguousarray(a, dtype=np.int64)
                for a in (lm.send_cell_off, lm.send_edge_off, lm.recv_cell_off, lm.recv_edge_off)]
        self._keep += offs
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        L.check(L.lib().moka_halo_create(self.Prog._state._h, len(lm.neighbors),
                                         L.i32(self._keep[0]), i64(offs[0]), L.i32(self._keep[1]), i64(offs[1]),
                                         L.i32(self._keep[2]), i64(offs[2]), L.i32(self._keep[3]), i64(offs[3]),
                                         self.p_boundary, self.p_owned, C.byref(self._halo)), backend._h)
        api._own(self, L.lib().moka_halo_destroy, self._halo, self.Prog._state, self.mesh, backend)
        ns, nr = C.c_int64(), C.c_int64()
        L.check(L.lib().moka_halo_buffer_elems(self._halo, C.byref(ns), C.byref(nr)))
        dev = torch.device("cuda", backend.device)
        real = torch.float32 if self.state_bytes == 4 else torch.float64      # halo messages carry state reals
        self.sendbuf = torch.zeros(max(ns.value, 1), dtype=real, device=dev)
        self.recvbuf = torch.zeros(max(nr.value, 1), dtype=real, device=dev)
        torch.cuda.synchronize(dev)           # the zero fills ran on torch's stream; the library's streams do not order against it
        self.send_slices = message_slices(lm, K, True)
        self.recv_slices = message_slices(lm, K, False)
        self._p2p = None                      # P2POp list of the nccl transport, built once
        self._splits = alltoall_splits(lm, K)
        cs, ms = C.c_void_p(), C.c_void_p()
        L.check(L.lib().moka_ctx_streams(backend._h, C.byref(cs), C.byref(ms)))
        self.comm_stream = torch.cuda.ExternalStream(ms.value, device=dev)
        self.halo_bytes_per_stage = int(self.state_bytes) * (ns.value + nr.value)
        self.direct_available = bool(L.lib().moka_halo_direct_available(self._halo))
        self.connected = False
        self._cb_error = None

        def _cb(user, what, sendbuf, recvbuf):   # the buffered transport as a C callback of moka_rk4_dist_step
            try:
                self._transport()
                return 0
            except Exception as exc:             # noqa: BLE001  (an exception must not cross the C frame)
                self._cb_error = exc
                return 1
        self._cb = L.TRANSPORT_FN(_cb)
        self._ready = True

    def set_overlap(self, mode: int):
        """How an RK4 stage queues its boundary and interior launches (moka_halo_set_overlap): -1 automatic, 0 one behind
        the other on the compute stream, 1 together on two streams.  Same results either way."""
        L.check(L.lib().moka_halo_set_overlap(self._halo, int(mode)), self.backend._h)

    # ---- direct transport: tell the neighbours where to push ----
    def export_peer_info(self, shared: bool):
        """{neighbour rank: bytes} -- what each neighbour needs to push its rows into this rank's fields."""
        out = {}
        for i, q in enumerate(self.lm.neighbors):
            info = L.HaloPeerInfo()
            L.check(L.lib().moka_halo_export(self._halo, i, 1 if shared else 0, C.byref(info)), self.backend._h)
            out[q] = bytes(info)
        return out

    def connect_peers(self, infos, shared: bool):
        """infos[q] = what neighbour q exported FOR this rank."""
        for i, q in enumerate(self.lm.neighbors):
            info = L.HaloPeerInfo.from_buffer_copy(infos[q])
            L.check(L.lib().moka_halo_connect(self._halo, i, C.byref(info), 1 if shared else 0), self.backend._h)
        self.connected = True

    def connect_ipc(self):
        """Multi-process set-up of the direct transport: IPC handles and flag-block names travel over `group`.
        Collective, and failure-safe as one: a rank whose export or connect fails still takes part in every collective of this
        method (with a failure marker), so no rank is left waiting; all ranks then raise together."""
        err = None
        try:
            mine = self.export_peer_info(True)
        except Exception as exc:                     # noqa: BLE001
            mine, err = None, exc
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine, group=self.group)
        bad = [q for q in range(self.world) if everyone[q] is None]
        if not bad:
            try:
                self.connect_peers({q: everyone[q][self.rank] for q in self.lm.neighbors}, True)
            except Exception as exc:                 # noqa: BLE001
                err = exc
        ok = self.torch.tensor([0.0 if (err is not None or bad) else 1.0], dtype=self.torch.float64)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN, group=self.group)      # doubles as the closing barrier
        if float(ok[0]) != 1.0:
            self.connected = False
            raise RuntimeError(f"direct halo transport: set-up failed on rank(s) {bad or '(connect)'}"
                               + (f"; this rank: {err!r}" if err is not None else ""))

    def set_transport(self, name: str):
        """Select the halo transport of the steps that follow.  "ipc" = direct; "ipc-acq" = direct with an explicit
        system-scope acquire in front of every launch that reads received rows (moka_halo_set_acquire); "ipc-smo" = direct with
        the flag handshake enqueued on the streams instead of done by the host thread (moka_halo_set_stream_flags)."""
        self.transport = name
        if getattr(self, "_halo", None):
            L.check(L.lib().moka_halo_set_acquire(self._halo, 1 if name == "ipc-acq" else 0), self.backend._h)
            # "ipc-smo": the direct transport with its handshake enqueued as stream memory operations (experiment:
            # moka_halo_set_stream_flags; raises MokaError where the device or the flag memory does not allow it)
            L.check(L.lib().moka_halo_set_stream_flags(self._halo, 1 if name == "ipc-smo" else 0), self.backend._h)

    def snapshot(self):
        """The local prognostic state (current level, halo rows included) as host arrays."""
        self.sync_device()
        return (self.Prog.ssh[-1].get(), self.Prog.normalVelocity[-1].get(), self.Prog.layerThickness[-1].get())

    def restore(self, snap):
        """Both time levels := a snapshot() (halo rows included: no exchange needed)."""
        for f, a in zip((self.Prog.ssh, self.Prog.normalVelocity, self.Prog.layerThickness), snap):
            f[0].set(a); f[-1].set(a)

    def reset_state(self, ssh, u, h):
        """Both time levels := the given GLOBAL arrays' local rows, halo rows included."""
        lm = self.lm
        u, h = np.asarray(u).reshape(-1, self.K), np.asarray(h).reshape(-1, self.K)
        self.restore((np.asarray(ssh)[lm.cells_g], u[lm.edges_g], h[lm.cells_g]))

    def sync_device(self):
        """Everything this rank has queued on its device is done (the library's streams and torch's)."""
        self.backend.synchronize()
        self.torch.cuda.synchronize()

    # ---- transport of the packed buffers ----
    def _transport(self):
        torch, dist = self.torch, self.dist
        if not self.lm.neighbors:
            return
        if self.transport == "nccl-a2a":
            ins, outs = self._splits
            with torch.cuda.stream(self.comm_stream):     # stream-ordered after the pack kernel; the host does not wait
                dist.all_to_all_single(self.recvbuf[:sum(outs)], self.sendbuf[:sum(ins)], outs, ins)
        elif self.transport == "nccl":
            if self._p2p is None:             # one message per neighbour and direction; the op list is reused
                self._p2p = [dist.P2POp(dist.irecv, self.recvbuf[a:b], q) for q, a, b in self.recv_slices if b > a] + \
                            [dist.P2POp(dist.isend, self.sendbuf[a:b], q) for q, a, b in self.send_slices if b > a]
            with torch.cuda.stream(self.comm_stream):
                for w in dist.batch_isend_irecv(self._p2p):
                    w.wait()                     # stream-ordered: the comm stream waits, the host does not
        elif self.transport == "nccl-default-stream":
            # conservative form: same RCCL P2P, but on torch's current stream with full synchronisation either side
            self.backend.synchronize()
            for w in dist.batch_isend_irecv(
                    [dist.P2POp(dist.irecv, self.recvbuf[a:b], q) for q, a, b in self.recv_slices if b > a] +
                    [dist.P2POp(dist.isend, self.sendbuf[a:b], q) for q, a, b in self.send_slices if b > a]):
                w.wait()
            torch.cuda.synchronize()
        elif self.transport in ("local", "ipc", "ipc-acq", "ipc-smo"):
            raise RuntimeError(f"transport {self.transport!r} has no buffered form")
        elif self.transport != "gloo":
            raise ValueError(f"unknown halo transport {self.transport!r}")
        else:                                    # gloo: through the host
            self.backend.synchronize()
            send_cpu, recv_cpu = self.sendbuf.cpu(), torch.empty_like(self.recvbuf, device="cpu")
            reqs = [dist.irecv(recv_cpu[a:b], q, group=self.group) for q, a, b in self.recv_slices if b > a]
            reqs += [dist.isend(send_cpu[a:b].contiguous(), q, group=self.group) for q, a, b in self.send_slices if b > a]
            for w in reqs:
                w.wait()
            self.recvbuf.copy_(recv_cpu)
            torch.cuda.synchronize()

    def _transport_buffered(self):
        """sendbuf -> the neighbours' recvbuf for the paths that exist in buffered form only (tape recording, the adjoint
        fields): the model's own transport, or host-staged gloo when that is the direct one."""
        t = self.transport
        if t not in ("ipc", "ipc-acq", "ipc-smo"):
            return self._transport()
        self.transport = "gloo"
        try:
            self._transport()
        finally:
            self.transport = t

    def received_bytes(self):
        """The halo rows of the current time level as this rank holds them now, in message order: what a transport has
        to have delivered."""
        self.backend.synchronize()
        hh, ssh, uu = self.Prog.layerThickness[-1].get(), self.Prog.ssh[-1].get(), self.Prog.normalVelocity[-1].get()
        parts = []
        lm = self.lm
        for i in range(len(lm.neighbors)):
            c = lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]
            e = lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]
            parts += [hh[c].ravel(), ssh[c], uu[e].ravel()]
        return np.concatenate(parts) if parts else np.zeros(0)

    def verify_transport(self, trusted="gloo") -> bool:
        """Move the current state's halo with this model's transport and again with `trusted`: True when both deliver
        the same bytes on this rank (callers combine the ranks' answers)."""
        lib = L.lib()
        mine = self.transport
        if mine in ("ipc", "ipc-acq", "ipc-smo"):
            lm = self.lm
            # wipe the halo rows, exchange directly, read them back
            hh, ssh, uu = self.Prog.layerThickness[-1].get(), self.Prog.ssh[-1].get(), self.Prog.normalVelocity[-1].get()
            rc, re_ = lm.recv_cells, lm.recv_edges
            hh[rc], ssh[rc], uu[re_] = -7.0, -7.0, -7.0
            self.Prog.layerThickness[-1].set(hh); self.Prog.ssh[-1].set(ssh); self.Prog.normalVelocity[-1].set(uu)
            self.dist.barrier(group=self.group)
            self.exchange_state()
            self.backend.synchronize()
            got = self.received_bytes()
            hh[rc], ssh[rc], uu[re_] = -9.0, -9.0, -9.0
            self.Prog.layerThickness[-1].set(hh); self.Prog.ssh[-1].set(ssh); self.Prog.normalVelocity[-1].set(uu)
            self.dist.barrier(group=self.group)
            try:
                self.transport = trusted
                self.exchange_state()
            finally:
                self.transport = mine
            self.backend.synchronize()
            return bool(np.array_equal(got, self.received_bytes()))
        L.check(lib.moka_halo_pack(self._halo, 0, self.sendbuf.data_ptr()), self.backend._h)
        self.recvbuf.zero_()
        self.torch.cuda.synchronize()
        self._transport()
        self.backend.synchronize(); self.torch.cuda.synchronize()
        got = self.recvbuf.clone()
        self.recvbuf.zero_()
        self.torch.cuda.synchronize()
        try:
            self.transport = trusted
            self._transport()
        finally:
            self.transport = mine
        self.backend.synchronize(); self.torch.cuda.synchronize()
        return bool(self.torch.equal(got, self.recvbuf))

    def _direct(self):
        return self.transport in ("ipc", "ipc-acq", "ipc-smo") and self.connected

    def exchange_state(self):
        """Halo exchange of the current time level (e.g. after an upload).  Collective: every rank calls it."""
        lib, h = L.lib(), self._halo
        if self._direct():
            self.backend.synchronize()
            self.dist.barrier(group=self.group)          # nobody is still reading the rows about to be overwritten
            L.check(lib.moka_halo_push_begin(h, 0), self.backend._h)
            L.check(lib.moka_halo_push_signal(h), self.backend._h)
            L.check(lib.moka_halo_push_wait(h, self.timeout_s), self.backend._h)
            return
        L.check(lib.moka_halo_pack(h, 0, self.sendbuf.data_ptr()), self.backend._h)
        self._transport()
        L.check(lib.moka_halo_unpack(h, 0, self.recvbuf.data_ptr()), self.backend._h)

    def _check_cb(self, rc):
        if self._cb_error is not None:
            exc, self._cb_error = self._cb_error, None
            raise exc
        L.check(rc, self.backend._h)

    def step_rk4(self):
        """One distributed RK4 step = ONE library call (the stage loop, the launches and -- for the direct transport --
        the flag waits are in C; the buffered transports come back into Python once per stage through a callback)."""
        cb = L.TRANSPORT_FN() if self._direct() or not self.lm.neighbors else self._cb
        self._check_cb(L.lib().moka_rk4_dist_step(self._halo, self.dt, cb, None, self.sendbuf.data_ptr(),
                                                  self.recvbuf.data_ptr(), self.timeout_s))

    def exchange_stats(self, enable=None):
        """moka_halo_stats: enable=True starts recording, enable=False stops; enable=None reads what was recorded, per RK4 step:
        host time waiting for the own push kernel / storing the flags / polling the neighbours' flags, host time of the whole step
        call, device time of the boundary and interior launches."""
        if enable is not None:
            L.check(L.lib().moka_halo_stats_enable(self._halo, 1 if enable else 0), self.backend._h)
            return None
        st = L.HaloStats()
        L.check(L.lib().moka_halo_stats_read(self._halo, C.byref(st)), self.backend._h)
        n = max(int(st.steps), 1)
        return {"steps": int(st.steps), "exchanges": int(st.exchanges),
                "host_step_ms_per_step": st.host_step_ms / n,
                "host_wait_ms_per_step": st.host_wait_ms / n,                       # inside moka_halo_push_wait
                "host_signal_wait_ms_per_step": st.host_signal_wait_ms / n,         # waiting for the own push kernel's event
                "push_to_flag_us": 1e3 * st.host_flag_store_ms / max(int(st.exchanges), 1),   # event -> last flag store, per exchange
                "boundary_launch_ms_per_step": 4.0 * st.boundary_launch_ms / max(int(st.boundary_launches), 1),
                "interior_launch_ms_per_step": 4.0 * st.interior_launch_ms / max(int(st.interior_launches), 1),
                "launches_recorded": int(st.boundary_launches + st.interior_launches)}

    def step_fe(self, flags=L.FE_REFERENCE_COMPAT & ~L.FE_LEVEL1_ONLY):
        """One distributed Forward-Euler step (the reference's live integrator) with the given compat flags."""
        cb = L.TRANSPORT_FN() if self._direct() or not self.lm.neighbors else self._cb
        self._check_cb(L.lib().moka_fe_dist_step(self._halo, self.dt, int(flags), cb, None, self.sendbuf.data_ptr(),
                                                 self.recvbuf.data_ptr(), self.timeout_s))

    def step_rk4_whole(self):
        """An RK4 step with every stage as ONE launch over the whole local mesh (halo entities included, redundantly) and the
        exchange behind it -- no overlap; the form the optional nonlinear terms run in on a partitioned mesh."""
        lib, h, ctx = L.lib(), self._halo, self.backend._h
        L.check(lib.moka_rk4_dist_begin(h, self.dt), ctx)
        for s in (1, 2, 3, 4):
            L.check(lib.moka_rk4_dist_stage(h, s, 2), ctx)
            L.check(lib.moka_halo_pack(h, s, self.sendbuf.data_ptr()), ctx)
            self._transport_buffered()
            L.check(lib.moka_halo_unpack(h, s, self.recvbuf.data_ptr()), ctx)
        L.check(lib.moka_rk4_dist_end(h), ctx)

    def step_rk4_piecewise(self):
        """The same step through the piecewise entry points (17 library calls): kept for tests and host-overhead timing."""
        lib, h, ctx = L.lib(), self._halo, self.backend._h
        L.check(lib.moka_rk4_dist_begin(h, self.dt), ctx)
        for s in (1, 2, 3, 4):
            L.check(lib.moka_rk4_dist_stage(h, s, 0), ctx)            # boundary patches
            L.check(lib.moka_halo_pack(h, s, self.sendbuf.data_ptr()), ctx)
            L.check(lib.moka_rk4_dist_stage(h, s, 1), ctx)            # interior patches overlap the exchange
            self._transport()
            L.check(lib.moka_halo_unpack(h, s, self.recvbuf.data_ptr()), ctx)
        L.check(lib.moka_rk4_dist_end(h), ctx)

    # ---- reverse mode of a partitioned RK4 run: d sum(ssh^2 over ALL ranks' owned cells) / d initial state ----
    def tape(self, capacity_steps: int):
        """A tape on this rank's state.  step_rk4_taped() records, adjoint_gradient() reverses (RK4 only)."""
        self._tape = api.AdjointTape(self.Prog, capacity_steps)
        return self._tape

    def _record(self, slot, what):
        L.check(L.lib().moka_tape_record_rk4(self._tape._h, slot, what), self.backend._h)

    def step_rk4_taped(self):
        """step_rk4 with the four provisional states recorded, halo rows included (piecewise entry points: every stage
        output is complete only after its exchange)."""
        lib, h, ctx = L.lib(), self._halo, self.backend._h
        L.check(lib.moka_rk4_dist_begin(h, self.dt), ctx)
        self._record(0, 0)
        for s in (1, 2, 3, 4):
            L.check(lib.moka_rk4_dist_stage(h, s, 0), ctx)
            L.check(lib.moka_halo_pack(h, s, self.sendbuf.data_ptr()), ctx)
            L.check(lib.moka_rk4_dist_stage(h, s, 1), ctx)
            self._transport_buffered()
            L.check(lib.moka_halo_unpack(h, s, self.recvbuf.data_ptr()), ctx)
            if s <= 3:
                self._record(s, s)
        L.check(lib.moka_rk4_dist_end(h), ctx)
        L.check(lib.moka_tape_commit_rk4(self._tape._h, self.dt), ctx)

    def _adjoint_fields(self, sg):
        fu, fh, fs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().moka_adjoint_rk4_stage_fields(self._tape._h, sg, C.byref(fu), C.byref(fh), C.byref(fs)), self.backend._h)
        return fu, fh, fs

    def adjoint_seed(self):
        L.check(L.lib().moka_adjoint_seed_sum_sq_ssh(self._tape._h), self.backend._h)

    def adjoint_pack(self, sg):
        fu, fh, fs = self._adjoint_fields(sg)
        L.check(L.lib().moka_halo_pack_fields(self._halo, fu, fh, fs, self.sendbuf.data_ptr()), self.backend._h)

    def adjoint_unpack_and_stage(self, sg):
        fu, fh, fs = self._adjoint_fields(sg)
        L.check(L.lib().moka_halo_unpack_fields(self._halo, fu, fh, fs, self.recvbuf.data_ptr()), self.backend._h)
        L.check(L.lib().moka_adjoint_rk4_stage(self._tape._h, sg), self.backend._h)

    def adjoint_gradient(self, nsteps: int, overlap: bool = True):
        """Seeds with d sum(ssh^2) at the current state and reverses the `nsteps` recorded steps; returns owned_gradient().
        overlap=False: the halo rows of the adjoint fields are exchanged in front of every transposed stage, which then runs over
        the whole local mesh.  overlap=True (where the chunk kernels serve the mesh): a stage transposes the boundary class first,
        the rows that produced travel while the interior class is transposed (moka_adjoint_rk4_stage_part)."""
        self.adjoint_seed()
        if overlap and self.adjoint_parts_available():
            self.adjoint_pack(4)                              # the seed's halo rows, once
            self._transport_buffered()
            self._adjoint_unpack(*self._adjoint_fields(4))
            for step in range(nsteps):
                for sg in (4, 3, 2, 1):
                    last = step == nsteps - 1 and sg == 1     # nothing gathers from the final gradient
                    out = self.adjoint_stage_boundary_and_pack(sg, pack=not last)
                    self.adjoint_stage_interior(sg)
                    if not last:
                        self._transport_buffered()
                        self._adjoint_unpack(*out)
            return self.owned_gradient()
        for _ in range(nsteps):
            for sg in (4, 3, 2, 1):
                self.adjoint_pack(sg)
                self._transport_buffered()
                self.adjoint_unpack_and_stage(sg)
        return self.owned_gradient()

    def adjoint_parts_available(self):
        """Can the transposed RK4 stages of this model's tape run class by class (the chunk kernels: even 34 <= K <= 64, hexagon-width
        lists)?  The library says (moka_adjoint_rk4_parts_available)."""
        t = getattr(self, "_tape", None)
        return bool(t is not None and L.lib().moka_adjoint_rk4_parts_available(t._h))

    def _adjoint_unpack(self, fu, fh, fs):
        L.check(L.lib().moka_halo_unpack_fields(self._halo, fu, fh, fs, self.recvbuf.data_ptr()), self.backend._h)

    def adjoint_stage_boundary_and_pack(self, sg, pack=True):
        """Part 0 of the transposed stage `sg` (the boundary class) and, behind it, the pack of the rows it wrote for the next
        transposed stage; returns those arrays (for the unpack once the exchange has arrived)."""
        lib, t = L.lib(), self._tape._h
        L.check(lib.moka_adjoint_rk4_stage_part(t, sg, 0), self.backend._h)
        fu, fh, fs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(lib.moka_adjoint_rk4_stage_out_fields(t, sg, C.byref(fu), C.byref(fh), C.byref(fs)), self.backend._h)
        if pack:
            L.check(lib.moka_halo_pack_fields(self._halo, fu, fh, fs, self.sendbuf.data_ptr()), self.backend._h)
        return fu, fh, fs

    def adjoint_stage_interior(self, sg):
        L.check(L.lib().moka_adjoint_rk4_stage_part(self._tape._h, sg, 1), self.backend._h)

    def owned_gradient(self):
        """(global cell ids, d/d layerThickness), (global edge ids, d/d normalVelocity) of the entities this rank owns."""
        lm, g = self.lm, self._tape.download()
        cm, em = lm.owned_cell_mask, lm.owned_edge_mask
        return (lm.cells_g[cm], g["layerThickness"][cm]), (lm.edges_g[em], g["normalVelocity"][em])

    # ---- the same for Forward-Euler runs (the integrator the reference differentiates, test_Enzyme_end2end.jl) ----
    def step_fe_taped(self, flags=L.FE_REFERENCE_COMPAT & ~L.FE_LEVEL1_ONLY):
        lib, t, ctx = L.lib(), self._tape._h, self.backend._h
        L.check(lib.moka_tape_record_fe(t, int(flags), 0), ctx)
        self.step_fe(flags)
        L.check(lib.moka_tape_record_fe(t, int(flags), 1), ctx)
        L.check(lib.moka_tape_commit_fe(t, self.dt, int(flags)), ctx)

    def _adjoint_fe_fields(self):
        fu, fh, fs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(L.lib().moka_adjoint_fe_step_fields(self._tape._h, C.byref(fu), C.byref(fh), C.byref(fs)), self.backend._h)
        return fu, fh, fs

    def adjoint_fe_pack(self):
        L.check(L.lib().moka_halo_pack_fields(self._halo, *self._adjoint_fe_fields(), self.sendbuf.data_ptr()), self.backend._h)

    def adjoint_fe_unpack_and_step(self):
        L.check(L.lib().moka_halo_unpack_fields(self._halo, *self._adjoint_fe_fields(), self.recvbuf.data_ptr()), self.backend._h)
        L.check(L.lib().moka_adjoint_fe_step(self._tape._h), self.backend._h)

    def adjoint_gradient_fe(self, nsteps: int):
        self.adjoint_seed()
        for _ in range(nsteps):
            self.adjoint_fe_pack()
            self._transport_buffered()
            self.adjoint_fe_unpack_and_step()
        return self.owned_gradient_fe()

    def owned_gradient_fe(self):
        """{name: (global ids, rows)} of d / d (ssh, normalVelocity, layerThickness, carried layerThicknessEdge), owned entities."""
        lm, g = self.lm, self._tape.download()
        cm, em = lm.owned_cell_mask, lm.owned_edge_mask
        return {"ssh": (lm.cells_g[cm], g["ssh"][cm]), "layerThickness": (lm.cells_g[cm], g["layerThickness"][cm]),
                "normalVelocity": (lm.edges_g[em], g["normalVelocity"][em]),
                "layerThicknessEdge": (lm.edges_g[em], g["layerThicknessEdge"][em])}

    def owned_state(self):
        """(global cell ids, ssh, h), (global edge ids, u) of the entities this rank owns."""
        lm = self.lm
        ssh, hh, uu = self.Prog.ssh[-1].get(), self.Prog.layerThickness[-1].get(), self.Prog.normalVelocity[-1].get()
        cm, em = lm.owned_cell_mask, lm.owned_edge_mask
        return (lm.cells_g[cm], ssh[cm], hh[cm]), (lm.edges_g[em], uu[em])

    def owned_diagnostics(self):
        """Owned parts of Diag / Tend: {name: (global ids, values)} (cells, edges and vertices this rank reports)."""
        lm = self.lm
        cm, em, vm = lm.owned_cell_mask, lm.owned_edge_mask, lm.owned_vert_mask
        D, T = self.Diag, self.Tend
        return {"hEdge": (lm.edges_g[em], D.layerThicknessEdge.get()[em]), "F": (lm.edges_g[em], D.thicknessFlux.get()[em]),
                "div": (lm.cells_g[cm], D.velocityDivCell.get()[cm]), "vort": (lm.verts_g[vm], D.relativeVorticity.get()[vm]),
                "tendU": (lm.edges_g[em], T.tendNormalVelocity.get()[em]), "tendH": (lm.cells_g[cm], T.tendLayerThickness.get()[cm])}

    def info(self):
        d = self.mesh.info()
        d.update({"rank_cells_owned": self.lm.n_owned_cells, "rank_cells_local": self.lm.mesh.nCells,
                  "neighbors": len(self.lm.neighbors), "halo_bytes_per_stage": self.halo_bytes_per_stage,
                  "patches_boundary": self.p_boundary, "patches_owned": self.p_owned,
                  "direct_transport_available": self.direct_available})
        return d

    def close(self):
        """Release the device objects in dependency order (tape, halo, state, mesh); the backend stays with the caller."""
        if getattr(self, "_tape", None) is not None:
            self._tape.close()
            self._tape = None
        if getattr(self, "_halo", None):
            api._release(self, L.lib().moka_halo_destroy, self._halo)
            self._halo = C.c_void_p()
        self.Prog._state.close()
        self.mesh.close()


def choose_transport(model, candidates, fallbacks, control_group, log=lambda msg: None, trial_steps=5, check_steps=3):
    """Pick the halo transport of `model` on this node; every rank calls this and all return the same name.

    A candidate qualifies in three phases, each closed by an agreement of all ranks over `control_group` (gloo), so that
    nobody runs ahead into a collective the others will never post: (1) it can be set up on every rank, (2) on every
    rank it moves the current state's halo to exactly the bytes the host-staged gloo exchange delivers, (3) `check_steps`
    full RK4 steps with it leave every rank's local state (halo rows included) equal, bit for bit, to the same steps over
    the host-staged gloo exchange from the same start -- which catches a timing-dependent stale read of peer-written rows
    that a single exchange or a single step can miss.  Any failure on any rank makes ALL ranks drop the candidate, at once:
    a failing rank takes part in the agreement with a 0, it never leaves the others waiting for a timeout.  "ipc" failing
    is followed by "ipc-acq" (the same transport with an explicit system-scope acquire in front of the launches that read
    received rows).  Of the qualifying `candidates` the fastest over `trial_steps` steps is kept (max over ranks);
    otherwise the first qualifying one of `fallbacks`.
    Candidates whose failure mode is a hang rather than an exception (an RCCL collective that never completes) must be
    screened BEFORE this function, in a child process (bench.py: probe_rccl).  Returns (name, {candidate: ms/step}).

    `model` needs: dist, torch, set_transport, direct_available, connected, connect_ipc, verify_transport, step_rk4,
    snapshot, restore, sync_device (DistributedModel; the CPU tests pass a stand-in)."""
    import time
    torch, dist = model.torch, model.dist

    def agree(x, op):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=op, group=control_group)
        return float(t[0])

    def sync_all():
        model.sync_device()
        dist.barrier(group=control_group)

    def phase(cand, what, fn):
        ok = 1.0
        try:
            if fn() is False:
                ok = 0.0
                log(f"halo transport {cand}: {what} failed")
        except Exception as exc:                 # noqa: BLE001
            log(f"halo transport {cand}: {what} raised {exc!r}")
            ok = 0.0
        return agree(ok, dist.ReduceOp.MIN) == 1.0

    start = model.snapshot()                    # every trial starts from, and the function returns with, this state
    reference = []                               # the local state after check_steps steps over gloo (computed on first use)

    def run_steps(name, n):
        model.restore(start)
        model.set_transport(name)
        sync_all()                               # nobody pushes into a state another rank is still restoring
        for _ in range(n):
            model.step_rk4()
        model.sync_device()
        return model.snapshot()

    def steps_match(cand):
        if not reference:
            reference.append(run_steps("gloo", check_steps))
        got = run_steps(cand, check_steps)
        return all(np.array_equal(a, b) for a, b in zip(got, reference[0]))

    def works(cand):
        def setup():
            if cand in ("ipc", "ipc-acq", "ipc-smo") and not model.connected:
                # every rank must be able to go direct BEFORE anyone enters connect_ipc's collectives
                if agree(1.0 if model.direct_available else 0.0, dist.ReduceOp.MIN) != 1.0:
                    return False
                model.connect_ipc()              # collective and failure-safe: raises on every rank or on none
            model.set_transport(cand)
            return True

        ok = (phase(cand, "set-up", setup) and
              phase(cand, "byte comparison with gloo", lambda: model.verify_transport("gloo")) and
              (cand == "gloo" or phase(cand, f"{check_steps} RK4 steps against the same steps over gloo", lambda: steps_match(cand))))
        model.set_transport("gloo")
        return ok

    def trial_ms(cand):
        model.restore(start)
        model.set_transport(cand)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(trial_steps):
            model.step_rk4()
        sync_all()
        return agree((time.perf_counter() - t0) / trial_steps * 1e3, dist.ReduceOp.MAX)

    times = {}
    good = []
    for c in candidates:
        if works(c):
            good.append(c)
        elif c == "ipc" and "ipc-acq" not in candidates and works("ipc-acq"):
            good.append("ipc-acq")
    if good:
        times = {c: trial_ms(c) for c in good}
        cand = min(good, key=lambda c: times[c])
        log(f"qualified halo transports, ms/step over {trial_steps} steps: {times} -> {cand}")
    else:
        for cand in fallbacks:
            if works(cand):
                break
        else:
            raise RuntimeError("no halo transport works on this node")
        log(f"using halo transport {cand}")
    model.restore(start)
    model.set_transport(cand)
    sync_all()
    return cand, times


class LocalCluster:
    """All ranks of a partition inside ONE process on ONE GPU (or on a list of devices): every rank is a
    DistributedModel with its own context (two HIP streams each).

    direct=True (default when the plan allows it): the ranks are connected with the library's direct transport exactly
    as processes would be -- push kernels store into the neighbours' fields, flag words (plain host memory here) complete
    the exchange; only the launch / signal / wait phases of a stage are interleaved over the ranks because one host
    thread drives all of them.  direct=False: the buffered transport with device-to-device copies between the ranks'
    send and receive buffers, ordered by stream events only."""

    def __init__(self, mesh, ssh, u, h, rest, dt, world, device=0, ordering=0, patch_cells=0, state_bytes=8, direct=True,
                 devices=None, overlap=-1, nonlinear=False, visc_del2=0.0, part=None, placement_tries=1):
        import torch
        self.torch, self.world = torch, world
        devs = list(devices) if devices is not None else [device] * world
        self.backends = [api.MokaHIP(devs[r]) for r in range(world)]
        part = partition_cells(mesh, world) if part is None else np.asarray(part, dtype=np.int32)
        self.models = [DistributedModel(mesh, ssh, u, h, rest, dt, self.backends[r], r, world, ordering=ordering,
                                        patch_cells=patch_cells, transport="local", part=part, state_bytes=state_bytes,
                                        nonlinear=nonlinear, visc_del2=visc_del2, placement_tries=placement_tries)
                       for r in range(world)]
        for r, m in enumerate(self.models):        # every rank learns the order its neighbours want their rows in
            m.finish({q: self.models[q]._wants[r] for q in m.lm.neighbors})
        for m in self.models:
            m.set_overlap(overlap)
        self.direct = bool(direct) and all(m.direct_available for m in self.models)
        if self.direct:
            exported = [m.export_peer_info(False) for m in self.models]
            for r, m in enumerate(self.models):
                m.connect_peers({q: exported[q][r] for q in m.lm.neighbors}, False)
        self.events = [torch.cuda.Event() for _ in range(world)]
        # (receiver, a, b) <- (sender, c, d): slices of the packed buffers, matched by construction (message_slices)
        self.moves = []
        for r, mr in enumerate(self.models):
            send_of = {}
            for q in range(world):
                send_of[q] = {dst: (a, b) for dst, a, b in self.models[q].send_slices}
            for src, a, b in mr.recv_slices:
                c, d = send_of[src][r]
                assert b - a == d - c
                if b > a:
                    self.moves.append((r, a, b, src, c, d))

    def _exchange(self, what, pack=True):
        """pack (optional) -> copies on the receivers' comm streams behind the senders' pack -> unpack; all stream-ordered."""
        torch = self.torch
        lib = L.lib()
        if pack:
            for m in self.models:
                L.check(lib.moka_halo_pack(m._halo, what, m.sendbuf.data_ptr()), m.backend._h)
        for r, m in enumerate(self.models):
            self.events[r].record(m.comm_stream)                    # "send posted": the sender's pack is queued before it
        for r, a, b, src, c, d in self.moves:
            mr, ms = self.models[r], self.models[src]
            mr.comm_stream.wait_event(self.events[src])
            with torch.cuda.stream(mr.comm_stream):
                mr.recvbuf[a:b].copy_(ms.sendbuf[c:d], non_blocking=True)
        # a sender may not repack before its data was read: its comm stream waits for the receivers' copies
        done = [torch.cuda.Event() for _ in self.models]
        for r, m in enumerate(self.models):
            done[r].record(m.comm_stream)
        for r, a, b, src, c, d in self.moves:
            self.models[src].comm_stream.wait_event(done[r])
        for m in self.models:
            L.check(lib.moka_halo_unpack(m._halo, what, m.recvbuf.data_ptr()), m.backend._h)

    def _finish_direct(self):
        lib = L.lib()
        for m in self.models:
            L.check(lib.moka_halo_push_signal(m._halo), m.backend._h)
        for m in self.models:
            L.check(lib.moka_halo_push_wait(m._halo, m.timeout_s), m.backend._h)

    def exchange_state(self):
        if not self.direct:
            return self._exchange(0)
        for m in self.models:
            m.backend.synchronize()
        for m in self.models:
            L.check(L.lib().moka_halo_push_begin(m._halo, 0), m.backend._h)
        self._finish_direct()

    def step_rk4(self):
        """One RK4 step on every rank.  With the optional nonlinear terms a stage is two kernels: the preparation pass over the
        boundary and halo patches waits for the previous stage's exchange, the one over the interior patches is queued behind the
        previous stage's interior launch and overlaps that exchange (moka_rk4_dist_stage parts 3 / 4; the order of
        moka_rk4_dist_step)."""
        lib = L.lib()
        nl = bool(getattr(self.models[0].Prog._state, "nonlinear", False))

        def each(fn, *a):
            for m in self.models:
                L.check(fn(m._halo, *a), m.backend._h)
        for m in self.models:
            L.check(lib.moka_rk4_dist_begin(m._halo, m.dt), m.backend._h)
        if nl:
            each(lib.moka_rk4_dist_stage, 1, 4)
        for s in (1, 2, 3, 4):
            if nl:
                each(lib.moka_rk4_dist_stage, s, 3)
            if self.direct:
                if nl:
                    for m in self.models:
                        L.check(lib.moka_rk4_dist_stage(m._halo, s, 0), m.backend._h)
                        L.check(lib.moka_halo_push_begin(m._halo, s), m.backend._h)
                        L.check(lib.moka_rk4_dist_stage(m._halo, s, 1), m.backend._h)
                        if s < 4:
                            L.check(lib.moka_rk4_dist_stage(m._halo, s + 1, 4), m.backend._h)
                else:
                    each(lib.moka_rk4_dist_stage_launch, s)             # boundary, push, interior
                self._finish_direct()
                continue
            each(lib.moka_rk4_dist_stage, s, 0)                          # boundary patches
            for m in self.models:
                L.check(lib.moka_halo_pack(m._halo, s, m.sendbuf.data_ptr()), m.backend._h)
            each(lib.moka_rk4_dist_stage, s, 1)                          # interior overlaps the exchange
            if nl and s < 4:
                each(lib.moka_rk4_dist_stage, s + 1, 4)
            self._exchange(s, pack=False)
        for m in self.models:
            L.check(lib.moka_rk4_dist_end(m._halo), m.backend._h)

    # ---- reverse mode across the ranks (buffered exchange: the adjoint fields are not among the IPC-registered buffers) ----
    def tape(self, capacity_steps: int):
        for m in self.models:
            m.tape(capacity_steps)

    def _move(self):
        """sendbuf -> recvbuf of the neighbours, ordered by stream events (the middle part of _exchange)."""
        torch = self.torch
        for r, m in enumerate(self.models):
            self.events[r].record(m.comm_stream)
        for r, a, b, src, c, d in self.moves:
            mr, ms = self.models[r], self.models[src]
            mr.comm_stream.wait_event(self.events[src])
            with torch.cuda.stream(mr.comm_stream):
                mr.recvbuf[a:b].copy_(ms.sendbuf[c:d], non_blocking=True)
        done = [torch.cuda.Event() for _ in self.models]
        for r, m in enumerate(self.models):
            done[r].record(m.comm_stream)
        for r, a, b, src, c, d in self.moves:
            self.models[src].comm_stream.wait_event(done[r])

    def step_rk4_whole(self):
        """Every stage one launch over each rank's whole local mesh, then the exchange (the nonlinear terms' form)."""
        lib = L.lib()
        for m in self.models:
            L.check(lib.moka_rk4_dist_begin(m._halo, m.dt), m.backend._h)
        for s in (1, 2, 3, 4):
            for m in self.models:
                L.check(lib.moka_rk4_dist_stage(m._halo, s, 2), m.backend._h)
            self._exchange(s)
        for m in self.models:
            L.check(lib.moka_rk4_dist_end(m._halo), m.backend._h)

    def step_rk4_taped(self):
        lib = L.lib()
        for m in self.models:
            L.check(lib.moka_rk4_dist_begin(m._halo, m.dt), m.backend._h)
            m._record(0, 0)
        for s in (1, 2, 3, 4):
            for m in self.models:
                L.check(lib.moka_rk4_dist_stage(m._halo, s, 0), m.backend._h)
            for m in self.models:
                L.check(lib.moka_halo_pack(m._halo, s, m.sendbuf.data_ptr()), m.backend._h)
            for m in self.models:
                L.check(lib.moka_rk4_dist_stage(m._halo, s, 1), m.backend._h)
            self._exchange(s, pack=False)
            if s <= 3:
                for m in self.models:
                    m._record(s, s)
        for m in self.models:
            L.check(lib.moka_rk4_dist_end(m._halo), m.backend._h)
            L.check(lib.moka_tape_commit_rk4(m._tape._h, m.dt), m.backend._h)

    def adjoint_gradient(self, nsteps: int, nCells: int, nEdges: int, K: int, overlap: bool = True):
        """d sum(ssh^2 over the whole mesh) / d (normalVelocity, layerThickness) at the state the tapes started from,
        assembled from the ranks' owned rows.  overlap: see DistributedModel.adjoint_gradient."""
        for m in self.models:
            m.adjoint_seed()
        if overlap and all(m.adjoint_parts_available() for m in self.models):
            for m in self.models:
                m.adjoint_pack(4)
            self._move()
            for m in self.models:
                m._adjoint_unpack(*m._adjoint_fields(4))
            for step in range(nsteps):
                for sg in (4, 3, 2, 1):
                    last = step == nsteps - 1 and sg == 1
                    outs = [m.adjoint_stage_boundary_and_pack(sg, pack=not last) for m in self.models]
                    for m in self.models:
                        m.adjoint_stage_interior(sg)
                    if not last:
                        self._move()
                        for m, out in zip(self.models, outs):
                            m._adjoint_unpack(*out)
        else:
            for _ in range(nsteps):
                for sg in (4, 3, 2, 1):
                    for m in self.models:
                        m.adjoint_pack(sg)
                    self._move()
                    for m in self.models:
                        m.adjoint_unpack_and_stage(sg)
        gu, gh = np.full((nEdges, K), np.nan), np.full((nCells, K), np.nan)
        for m in self.models:
            (cg, hh), (eg, uu) = m.owned_gradient()
            gh[cg], gu[eg] = hh, uu
        return gu, gh

    def step_fe_taped(self, flags=L.FE_REFERENCE_COMPAT & ~L.FE_LEVEL1_ONLY):
        lib = L.lib()
        for m in self.models:
            L.check(lib.moka_tape_record_fe(m._tape._h, int(flags), 0), m.backend._h)
        self.step_fe(flags)
        for m in self.models:
            L.check(lib.moka_tape_record_fe(m._tape._h, int(flags), 1), m.backend._h)
            L.check(lib.moka_tape_commit_fe(m._tape._h, m.dt, int(flags)), m.backend._h)

    def adjoint_gradient_fe(self, nsteps: int, mesh, K: int):
        """d sum(ssh^2 over the whole mesh) / d (ssh, normalVelocity, layerThickness, carried layerThicknessEdge) of a taped
        Forward-Euler run, assembled from the ranks' owned rows."""
        for m in self.models:
            m.adjoint_seed()
        for _ in range(nsteps):
            for m in self.models:
                m.adjoint_fe_pack()
            self._move()
            for m in self.models:
                m.adjoint_fe_unpack_and_step()
        out = {"ssh": np.full(mesh.nCells, np.nan), "layerThickness": np.full((mesh.nCells, K), np.nan),
               "normalVelocity": np.full((mesh.nEdges, K), np.nan), "layerThicknessEdge": np.full((mesh.nEdges, K), np.nan)}
        for m in self.models:
            for name, (ids, rows) in m.owned_gradient_fe().items():
                out[name][ids] = rows
        return out

    def step_fe(self, flags=L.FE_REFERENCE_COMPAT & ~L.FE_LEVEL1_ONLY):
        """The reference's Forward-Euler step on the partition: relativeVorticity first (it reads old-level rows of halo edges,
        which the neighbours' next step overwrites once this rank's push is signalled: csrc/halo.hip), boundary patches,
        exchange of the new level (what = 5), interior patches meanwhile."""
        lib = L.lib()
        for m in self.models:
            L.check(lib.moka_fe_dist_launch(m._halo, m.dt, int(flags), 2), m.backend._h)
            L.check(lib.moka_fe_dist_launch(m._halo, m.dt, int(flags), 0), m.backend._h)
        for m in self.models:
            if self.direct:
                L.check(lib.moka_halo_push_begin(m._halo, 5), m.backend._h)
            else:
                L.check(lib.moka_halo_pack(m._halo, 5, m.sendbuf.data_ptr()), m.backend._h)
        for m in self.models:
            L.check(lib.moka_fe_dist_launch(m._halo, m.dt, int(flags), 1), m.backend._h)
        if self.direct:
            self._finish_direct()
        else:
            self._exchange(5, pack=False)
        for m in self.models:
            L.check(lib.moka_fe_dist_end(m._halo), m.backend._h)

    def gather_owned(self, nCells, nEdges, K):
        """(ssh, u, h) of the whole mesh assembled from the ranks' owned entities (synchronises)."""
        ssh, u, h = np.full(nCells, np.nan), np.full((nEdges, K), np.nan), np.full((nCells, K), np.nan)
        for m in self.models:
            (cg, s, hh), (eg, uu) = m.owned_state()
            ssh[cg], h[cg], u[eg] = s, hh, uu
        return ssh, u, h

    def gather_diagnostics(self, mesh, K):
        """Diag / Tend of the whole mesh assembled from the ranks' owned entities."""
        n = {"hEdge": mesh.nEdges, "F": mesh.nEdges, "div": mesh.nCells, "vort": mesh.nVertices, "tendU": mesh.nEdges,
             "tendH": mesh.nCells}
        out = {k: np.full((v, K), np.nan) for k, v in n.items()}
        for m in self.models:
            for k, (ids, vals) in m.owned_diagnostics().items():
                out[k][ids] = vals
        return out

    def close(self):
        for b in self.backends:
            b.synchronize()
        for m in self.models:
            m.close()
        for b in self.backends:
            b.close()
