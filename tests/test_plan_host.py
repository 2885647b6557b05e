"""Host logic of libmoka_hip (no GPU): the C-ABI library loads and exports every declared symbol,
and the reordered mesh plan (permutations, patches, per-entity records) is a faithful renumbering
of the reference mesh -- checked by evaluating the tendency in numpy FROM THE PLAN RECORDS and
comparing bit-for-bit with the CPU oracle in the caller's numbering."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as orc
from moka_hip import lib as L
from moka_hip import meshgen as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "moka_hip.h")).read()
    declared = set(re.findall(r"\b(moka_[a-z0-9_]+)\s*\(", hdr))
    lib = L.lib()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libmoka_hip.so does not export {name}"
    assert declared == set(L.EXPORTS)
    assert b"gfx950" in lib.moka_version()


def test_ctx_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = L.lib().moka_ctx_create(0, C.byref(h))
    assert rc == L.ERR_NO_DEVICE and not h
    assert b"no CPU fallback" in L.lib().moka_last_error(None)


MESHES = {
    "planar": lambda: mg.planar_hex_mesh(12, 10, 1000.0, f0=1e-4),
    "sphere": lambda: mg.icosahedral_mesh(6),
    "sphere_5_7": lambda: mg.icosahedral_mesh(8, flips=5, seed=2),     # pentagon/heptagon pairs: maxEdges = 7
}


@pytest.fixture(scope="module", params=list(MESHES))
def mesh(request):
    return MESHES[request.param]()


def plan_tendency_numpy(plan, K, u_new, h_new, ssh_new):
    """The fused tendency exactly as kernels.hip evaluates it, from the plan's records."""
    inf = plan.info
    nC, nE, ME, ME2 = inf["nCells"], inf["nEdges"], inf["maxEdgesUsed"], inf["maxEdges2Used"]
    eoc = plan.array("eoc").reshape(nC, ME); coc = plan.array("coc").reshape(nC, ME)
    mltc = plan.array("mltc").reshape(nC, ME); sdv = plan.array("sdv").reshape(nC, ME)
    invA = plan.array("invArea")
    ehdr = plan.array("ehdr").reshape(nE, 4); eoe = plan.array("eoe").reshape(nE, ME2)
    woe = plan.array("woe").reshape(nE, ME2); g = plan.array("gInvDc"); f = plan.array("fEdge")
    k = np.arange(K)[None, :]
    tH = np.zeros((nC, K))
    for i in range(ME):
        e, c = eoc[:, i], coc[:, i]
        act = (e >= 0)[:, None] & (k < mltc[:, i][:, None])
        es, cs = np.where(e >= 0, e, 0), np.where(c >= 0, c, 0)
        hE = 0.5 * (h_new + h_new[cs])
        F = u_new[es] * hE
        tH = np.where(act, tH + F * sdv[:, i][:, None] * invA[:, None], tH)
    tU = np.zeros((nE, K))
    lev = k < ehdr[:, 3][:, None]
    ds = ssh_new[ehdr[:, 1]] - ssh_new[ehdr[:, 0]]
    tU = np.where(lev, tU - (g * ds)[:, None], tU)
    for i in range(ME2):
        x = eoe[:, i]
        xs = np.where(x >= 0, x, 0)
        tU = np.where((x >= 0)[:, None] & lev, tU + woe[:, i][:, None] * u_new[xs] * f[xs][:, None], tU)
    return tU, tH


@pytest.mark.parametrize("ordering", [L.ORDER_NONE, L.ORDER_RCM, L.ORDER_RCB])
@pytest.mark.parametrize("K,P", [(1, 16), (5, 7)])
def test_plan_is_a_faithful_renumbering(mesh, ordering, K, P):
    rng = np.random.default_rng(7)
    rsum = 1000.0 + rng.uniform(0, 1, mesh.nCells)
    plan = L.Plan(mesh, K, resting_thickness_sum=rsum, max_level_edge_top=K, ordering=ordering, patch_cells=P)
    inf = plan.info
    assert inf["ordering"] == ordering and inf["patch_cells"] == P
    cperm, eperm, vperm = (plan.permutation(k) for k in (L.CELL, L.EDGE, L.VERTEX))
    for perm, n in ((cperm, mesh.nCells), (eperm, mesh.nEdges), (vperm, mesh.nVertices)):
        assert np.array_equal(np.sort(perm), np.arange(n))
    if ordering == L.ORDER_NONE:
        assert np.array_equal(cperm, np.arange(mesh.nCells))
    cs, es, vs = plan.patch_ranges()
    assert cs[0] == es[0] == vs[0] == 0
    assert (cs[-1], es[-1], vs[-1]) == (mesh.nCells, mesh.nEdges, mesh.nVertices)
    assert np.all(np.diff(cs) > 0) and np.all(np.diff(cs)[:-1] == P) and np.all(np.diff(es) >= 0)
    # every edge lies in the patch of one of its two cells, and the patches own balanced numbers of edges
    ehdr = plan.array("ehdr").reshape(mesh.nEdges, 4)
    patch_of_edge = np.searchsorted(es, np.arange(mesh.nEdges), side="right") - 1
    assert np.all((ehdr[:, 0] // P == patch_of_edge) | (ehdr[:, 1] // P == patch_of_edge))
    full = np.diff(cs) == P
    own = np.diff(es)[full]
    if own.size > 4 and mesh.maxEdges <= 6:
        assert own.max() - np.median(own) <= max(3, 0.12 * np.median(own)), (own.max(), np.median(own))
    # tendency from the records == oracle in the caller's numbering, bit for bit
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    h = 1000.0 / K + rng.uniform(-1, 1, (mesh.nCells, K))
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rsum, max_level_edge_top=K)
    tu_ref, th_ref, ssh_ref = om.tendencies_clean(u, h)
    tU, tH = plan_tendency_numpy(plan, K, u[eperm], h[cperm], ssh_ref[cperm])
    assert np.array_equal(tU, tu_ref[eperm])
    assert np.array_equal(tH, th_ref[cperm])
    assert np.array_equal(plan.array("rsum"), rsum[cperm])


def test_rcb_patches_are_compact():
    """RCB patches must be compact: the edges a patch touches but does not own stay a small
    fraction, and the bandwidth beats the natural ordering's."""
    mesh = mg.icosahedral_mesh(16)
    plan = L.Plan(mesh, 1, ordering=L.ORDER_RCB, patch_cells=32)
    nC, P = mesh.nCells, 32
    coc = plan.array("coc").reshape(nC, 6)
    patch = np.arange(nC) // P
    nb = np.where(coc >= 0, coc, 0) // P
    outside = ((nb != patch[:, None]) & (coc >= 0)).sum() / (coc >= 0).sum()
    assert outside < 0.30          # a 32-cell hex patch has ~22 of 192 cell-neighbour links leaving it... x2 margin
    rcm = L.Plan(mesh, 1, ordering=L.ORDER_RCM, patch_cells=32)
    none = L.Plan(mesh, 1, ordering=L.ORDER_NONE, patch_cells=32)
    assert rcm.info["cellBandwidth"] < none.info["cellBandwidth"]


def test_plan_rejects_bad_meshes():
    mesh = mg.planar_hex_mesh(4, 4, 1.0)
    bad = mg.planar_hex_mesh(4, 4, 1.0)
    bad.edgesOnCell = bad.edgesOnCell.copy()
    bad.edgesOnCell[3, 2] = mesh.nEdges + 5
    with pytest.raises(L.MokaError, match="edgesOnCell out of range"):
        L.Plan(bad, 1)
    bad2 = mg.planar_hex_mesh(4, 4, 1.0)
    bad2.cellsOnEdge = bad2.cellsOnEdge.copy()
    bad2.cellsOnEdge[0, 1] = 0     # boundary edge: non-periodic meshes are rejected like VertMesh.jl:50
    with pytest.raises(L.MokaError, match="non-periodic"):
        L.Plan(bad2, 1)
    with pytest.raises(L.MokaError, match="nVertLevels"):
        L.Plan(mesh, 0)
    with pytest.raises(L.MokaError, match="unknown ordering"):
        L.Plan(mesh, 1, ordering=9)


def test_zero_entries_in_edges_on_edge_are_skipped():
    """eoe == 0 => continue (horizontal_advection_and_coriolis.jl:67)."""
    mesh = mg.planar_hex_mesh(6, 6, 1000.0, f0=1e-4)
    mesh.edgesOnEdge = mesh.edgesOnEdge.copy()
    mesh.edgesOnEdge[5, 3] = 0
    mesh.edgesOnEdge[9, 0] = 0
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, (mesh.nEdges, 1)); h = 10 + rng.uniform(-1, 1, (mesh.nCells, 1))
    plan = L.Plan(mesh, 1, ordering=L.ORDER_RCB, patch_cells=8)
    cperm, eperm = plan.permutation(L.CELL), plan.permutation(L.EDGE)
    om = orc.OracleMesh(mesh, 1)
    tu, th, ssh = om.tendencies_clean(u, h)
    tU, tH = plan_tendency_numpy(plan, 1, u[eperm], h[cperm], ssh[cperm])
    assert np.array_equal(tU, tu[eperm]) and np.array_equal(tH, th[cperm])


def test_state_bytes_scales_the_gather_records():
    """moka_mesh_desc.stateBytes = 4 (fp32-storage state, config 5): same numbering, byte-offset records for K*4-byte rows."""
    m = mg.icosahedral_mesh(8)
    K = 80
    p8 = L.Plan(m, K, max_level_edge_top=K, patch_cells=12)
    p4 = L.Plan(m, K, max_level_edge_top=K, patch_cells=12, state_bytes=4)
    assert np.array_equal(p8.permutation(L.CELL), p4.permutation(L.CELL))
    assert np.array_equal(p8.permutation(L.EDGE), p4.permutation(L.EDGE))
    c8, c4 = p8.array("cRec").reshape(m.nCells, -1), p4.array("cRec").reshape(m.nCells, -1)
    e8, e4 = p8.array("eRec").reshape(m.nEdges, -1), p4.array("eRec").reshape(m.nEdges, -1)
    ME, ME2 = p8.info["maxEdgesUsed"], p8.info["maxEdges2Used"]
    assert np.array_equal(c8[:, :2 * ME], 2 * c4[:, :2 * ME]) and np.array_equal(c8[:, 2 * ME:], c4[:, 2 * ME:])
    assert np.array_equal(e8[:, :ME2], 2 * e4[:, :ME2]) and np.array_equal(e8[:, ME2:], e4[:, ME2:])
    assert p4.info["patch_cells"] == 12 and L.Plan(m, K, max_level_edge_top=K, state_bytes=4).info["patch_cells"] == 24
    assert L.Plan(m, 60, max_level_edge_top=60).info["patch_cells"] == 16
    with pytest.raises(L.MokaError):
        L.Plan(m, K, state_bytes=2)


@pytest.mark.parametrize("P,groups", [(16, 8), (24, 12), (12, 8)])
def test_edge_ownership_is_levelled_to_the_mean(P, groups):
    """A half-wave group handles one own edge per iteration, so a patch costs ceil(edges / groups) iterations: the plan levels
    edge ownership until (nearly) every full patch owns exactly 3 * P edges on a hexagon mesh -- a patch one edge above
    pays a whole extra iteration (12 % of the patches did before the levelling)."""
    mesh = mg.icosahedral_mesh(40)
    p = L.Plan(mesh, 60, patch_cells=P)
    cs, es, _ = p.patch_ranges()
    nc, ne = np.diff(cs), np.diff(es)
    full = nc == P
    assert ne.sum() == mesh.nEdges
    assert (ne[full] == 3 * P).mean() > 0.95, np.bincount(ne[full])[-6:]
    assert ne.max() <= 3 * P + 1
    assert np.ceil(ne[full] / groups).mean() < np.ceil(3 * P / groups) + 0.02
    p.close()


def test_patch_vertex_lists_of_the_nonlinear_stage_kernel():
    """k_stage_nl5 keeps the potential vorticity of a patch's vertices in LDS: pvList names them, lvoe[e] holds, per edgesOnEdge slot
    and for the edge itself, the patch-local ids (16 bits each) of the slot edge's two vertices.  Check both against verticesOnEdge."""
    mesh = mg.icosahedral_mesh(8)
    K = 60
    plan = L.Plan(mesh, K)
    nE = mesh.nEdges
    pc, pe, pv = plan.patch_ranges()
    pvStart, pvList, lvoe = plan.array("pvStart"), plan.array("pvList"), plan.array("lvoe").reshape(nE, 24)
    eoe = plan.array("eoe").reshape(nE, -1)
    assert eoe.shape[1] == 10 and len(pvStart) == len(pe)
    eperm, vperm = plan.permutation(L.EDGE), plan.permutation(L.VERTEX)
    vinv = np.empty_like(vperm); vinv[vperm] = np.arange(len(vperm), dtype=np.int32)
    voe = vinv[mesh.verticesOnEdge[eperm] - 1]                       # new edge -> its two vertices, new numbering
    most = 0
    for q in range(len(pe) - 1):
        verts = pvList[pvStart[q]:pvStart[q + 1]]
        assert len(set(verts.tolist())) == len(verts) <= 65535
        most = max(most, len(verts))
        for e in range(pe[q], pe[q + 1]):
            assert np.array_equal(verts[lvoe[e, 20:22]], voe[e])
            for i in range(10):
                x = eoe[e, i]
                assert np.array_equal(verts[lvoe[e, 2 * i:2 * i + 2]], voe[x] if x >= 0 else voe[e])
    assert most <= 96          # a 16-cell patch: ~75 vertices (its own edges' and those of the ring of cells around it)
    plan.close()
