"""Multi-rank coverage: partition, local meshes and exchange lists with gloo on CPU
(world size 2 and 3), and the HIP DistributedModel with two ranks sharing one GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from moka_hip import lib as L
from moka_hip import meshgen as mg
from moka_hip import parallel as par

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(nproc, *args, timeout=300, worker="dist_worker.py"):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", worker), *map(str, args)]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK on" in r.stdout
    return r.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_rk4_with_oracle_compute_gloo(world):
    run_workers(world, "cpu")


@pytest.mark.parametrize("world", [2, 3])
def test_transport_selection_agrees_on_failures(world):
    """choose_transport with a candidate that fails on ONE rank in phase 1 (set-up), 2 (bytes) or 3 (whole steps against the
    same steps over gloo): every rank drops it at once and ends on the same transport, nobody waits on a timeout
    (tests/transport_worker.py; gloo, no GPU)."""
    out = run_workers(world, worker="transport_worker.py", timeout=120)
    assert "transport_worker: OK" in out


def test_partition_and_local_mesh_properties():
    mesh = mg.icosahedral_mesh(10)
    for world in (2, 4, 8):
        part = par.partition_cells(mesh, world)
        counts = np.bincount(part, minlength=world)
        assert counts.min() > 0 and counts.max() - counts.min() <= world        # balanced
        sent, recvd = {}, {}
        for r in range(world):
            lm = par.build_local(mesh, part, r, world)
            assert lm.n_owned_cells == counts[r]
            assert np.all(lm.cell_class[lm.owned_cell_mask] <= 1) and np.all(lm.cell_class[~lm.owned_cell_mask] >= 2)
            assert lm.cell_class.max() == 1 + len(lm.neighbors)                 # halo cells: class 2 + neighbour index
            # every halo cell is received, every local edge without an owned cell is received
            assert set(lm.recv_cells.tolist()) == set(np.nonzero(~lm.owned_cell_mask)[0].tolist())
            coe = lm.mesh.cellsOnEdge - 1
            has_owned = lm.owned_cell_mask[coe[:, 0]] | lm.owned_cell_mask[coe[:, 1]]
            assert set(lm.recv_edges.tolist()) == set(np.nonzero(~has_owned)[0].tolist())
            # the local mesh is accepted by the host plan, classes order the patches and no patch straddles a class
            plan = L.Plan(lm.mesh, 3, max_level_edge_top=3, ordering=L.ORDER_RCB, patch_cells=8, cell_class=lm.cell_class)
            cperm = plan.permutation(L.CELL)
            cls = lm.cell_class[cperm]
            assert np.all(np.diff(cls) >= 0)                                     # class-major ordering
            cs, es, _ = plan.patch_ranges()      # the record-staging kernels size their LDS from these maxima
            assert plan.info["maxPatchEdges"] == np.diff(es).max() and plan.info["maxPatchCells"] == np.diff(cs).max()
            pS, cS, eS = plan.class_ranges()
            assert len(pS) == 3 + len(lm.neighbors) and pS[0] == 0 and pS[-1] == plan.info["nPatches"]
            for k in range(len(pS) - 1):
                assert cS[k] == cs[pS[k]] and eS[k] == es[pS[k]]
                assert np.all(cls[cS[k]:cS[k + 1]] == k)
            assert np.diff(cs).max() <= 8
            # launch ranges: patches [0, pB) are computed before the halo leaves, [pB, pO) while it travels, the rest never.
            # Whatever the balancing of edge ownership does, every sent edge must be produced by the boundary launch and
            # every edge with an owned cell by a launched patch.
            pB, pO = int(pS[1]), int(pS[2])
            eperm = plan.permutation(L.EDGE)
            patch_of_edge = np.empty(lm.mesh.nEdges, dtype=np.int64)
            patch_of_edge[eperm] = np.searchsorted(es, np.arange(lm.mesh.nEdges), side="right") - 1
            assert np.all(patch_of_edge[lm.send_edges] < pB)
            assert np.all(patch_of_edge[has_owned] < pO) and np.all(patch_of_edge[~has_owned] >= pO)
            cpos = np.empty(lm.mesh.nCells, dtype=np.int64)
            cpos[cperm] = np.arange(lm.mesh.nCells)
            assert np.all(cpos[lm.send_cells] < cS[1])
            # what neighbour i sends is ONE contiguous range of cells and ONE of edges in the plan's numbering
            epos = np.empty(lm.mesh.nEdges, dtype=np.int64)
            epos[eperm] = np.arange(lm.mesh.nEdges)
            for i in range(len(lm.neighbors)):
                rc = lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]
                re_ = lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]
                assert sorted(cpos[rc].tolist()) == list(range(cS[2 + i], cS[3 + i]))
                assert sorted(epos[re_].tolist()) == list(range(eS[2 + i], eS[3 + i]))
            for i, q in enumerate(lm.neighbors):
                sent[(r, q)] = (lm.cells_g[lm.send_cells[lm.send_cell_off[i]:lm.send_cell_off[i + 1]]],
                                lm.edges_g[lm.send_edges[lm.send_edge_off[i]:lm.send_edge_off[i + 1]]])
                recvd[(q, r)] = (lm.cells_g[lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]],
                                 lm.edges_g[lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]])
        assert set(sent) == set(recvd)
        for k in sent:                                                            # both sides agree, in order
            assert np.array_equal(sent[k][0], recvd[k][0]) and np.array_equal(sent[k][1], recvd[k][1])


@pytest.mark.gpu
@pytest.mark.parametrize("K,variant", [(60, 0), (1, 0), (60, 3)])
def test_distributed_rk4_hip_two_ranks_one_gpu(K, variant):
    run_workers(2, "gpu", K, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("world,K", [(2, 60), (3, 60), (4, 1)])
def test_direct_ipc_transport_between_processes_on_one_gpu(world, K):
    """The direct halo transport as bench.py uses it on a multi-GPU node -- one process per rank, fields mapped with
    hipIpcOpenMemHandle, push kernels storing into the neighbours' memory, flag words in POSIX shared memory -- with the
    ranks sharing GPU 0: RK4 and Forward-Euler steps bit-identical to the single-domain oracle, the transport selection
    (set-up / byte comparison with gloo / step, agreed phase by phase) qualifies it."""
    run_workers(world, "gpu", K, 0, "ipc")


@pytest.mark.gpu
def test_rccl_collectives_on_the_library_comm_stream():
    """What one GPU can show of the RCCL transport: see tests/rccl_worker.py."""
    run_workers(1, worker="rccl_worker.py")


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0])
@pytest.mark.parametrize("world,K,P,nsteps,sbytes,direct", [(2, 60, 0, 4, 8, True), (4, 60, 0, 4, 8, True), (8, 60, 0, 3, 8, True),
                                                            (3, 1, 0, 5, 8, True), (4, 60, 8, 3, 8, True), (5, 80, 0, 2, 8, True),
                                                            (4, 80, 0, 3, 4, True), (3, 60, 0, 2, 4, True),
                                                            (2, 60, 0, 4, 8, False), (8, 60, 0, 3, 8, False), (3, 1, 0, 5, 8, False),
                                                            (4, 80, 0, 3, 4, False)])
def test_stream_ordered_exchange_in_one_process(world, K, P, nsteps, sbytes, direct, overlap):
    """All ranks in one process on one GPU.  direct: the library's direct transport (push kernels store into the neighbours'
    fields, flag words complete the exchange) exactly as between processes; otherwise the buffered transport with
    stream-ordered device copies and no host synchronisation anywhere in the step -- the ordering RCCL gives.  Exercises
    the two-stream / event choreography of the distributed RK4 step (boundary patches, exchange on the comm stream,
    interior on the compute stream): any missing dependency shows up as a mismatch against the single-domain oracle.
    overlap: the boundary launch beside the interior launch on two streams (1) or in front of it on one (0)."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(24)
    rng = np.random.default_rng(17 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 20.0
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h, mixed=sbytes == 4)          # fp32-storage states exchange fp32 halos
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, patch_cells=P, state_bytes=sbytes, direct=direct, overlap=overlap)
    assert cl.direct == direct
    cl.exchange_state()
    for rep in range(3):                      # several rounds: timing-dependent races get more than one chance to show
        for _ in range(nsteps):
            cl.step_rk4()
            ref.step_rk4(dt)
        gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, ref.u[1]), (world, rep)
        assert np.array_equal(gh, ref.h[1]), (world, rep)
        assert np.array_equal(gs, ref.ssh[1]), (world, rep)
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,flags,direct", [(2, 60, 3, True), (4, 60, 0, True), (3, 60, 1, False), (8, 60, 3, True), (3, 1, 7, True),
                                                  (4, 1, 0, False), (5, 34, 2, True), (4, 7, 3, True)])
def test_forward_euler_on_a_partitioned_mesh(world, K, flags, direct):
    """The reference's live integrator (ocn_timestep(..., ForwardEuler), time_integration.jl:150-193) with its quirks
    (stale layerThicknessEdge in the flux, accumulating relativeVorticity, level-1-only kernels when K = 1) on 2-8 ranks:
    every field of Prog, Diag and Tend equals the single-domain oracle bit for bit.  Only the new time level is
    exchanged: the carried layerThicknessEdge of every edge with an owned cell is computed where it is used."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(20)
    rng = np.random.default_rng(29 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 15.0
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, direct=direct)
    cl.exchange_state()
    for step in range(5):
        cl.step_fe(flags)
        ref.step_fe(dt, flags)
        gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1]), step
    d = cl.gather_diagnostics(mesh, K)
    for name, exp in (("hEdge", ref.hEdge), ("F", ref.F), ("div", ref.div), ("vort", ref.vort), ("tendU", ref.tendU),
                      ("tendH", ref.tendH)):
        assert np.array_equal(d[name], exp), name
    # an RK4 step after Forward-Euler steps (and back) keeps the ranks' buffer sets aligned
    cl.step_rk4(); ref.step_rk4(dt)
    cl.step_fe(flags); ref.step_fe(dt, flags)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1])
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,sbytes", [(60, 8), (80, 4)])
def test_forward_euler_with_unbalanced_ranks(K, sbytes):
    """Direct transport, Forward Euler, ranks of very different size (the vertex pass reads old-level rows of halo
    edges, and a small rank that runs a step ahead pushes its next level into exactly those rows).  Rank 0 owns ~85 % of the
    cells, so ranks 1 and 2 finish every step long before rank 0's vertex pass is through; no synchronisation between the
    steps.  Accumulating relativeVorticity would carry a mixed-level read into every later step."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(40)
    rng = np.random.default_rng(97)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt, flags = 10.0, 3
    # a thin slab for rank 1, a thinner one for rank 2, the rest for rank 0
    z = mesh.zCell / np.abs(mesh.zCell).max()
    part = np.where(z > 0.80, 1, np.where(z < -0.88, 2, 0)).astype(np.int32)
    assert np.bincount(part)[0] > 0.8 * mesh.nCells
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h, mixed=sbytes == 4)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, 3, direct=True, part=part, state_bytes=sbytes)
    assert cl.direct
    cl.exchange_state()
    for rep in range(3):
        for _ in range(6):
            cl.step_fe(flags)
            ref.step_fe(dt, flags)
        gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1]), rep
        d = cl.gather_diagnostics(mesh, K)
        assert np.array_equal(d["vort"], ref.vort), rep
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,flags,direct", [(2, 80, 3, True), (4, 60, 0, False), (3, 36, 1, True), (8, 80, 3, True)])
def test_forward_euler_of_fp32_storage_states_on_a_partitioned_mesh(world, K, flags, direct):
    """The reference's live step on fp32-storage states across 2-8 ranks (halo messages carry floats): every array of Prog, Diag
    and Tend equals the storage-emulating single-domain oracle (oracle_step_fe_mixed) bit for bit."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(20)
    rng = np.random.default_rng(61 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 15.0
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h, mixed=True)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, direct=direct, state_bytes=4)
    cl.exchange_state()
    for step in range(4):
        cl.step_fe(flags)
        ref.step_fe(dt, flags)
        gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1]), step
    d = cl.gather_diagnostics(mesh, K)
    for name, exp in (("hEdge", ref.hEdge), ("F", ref.F), ("div", ref.div), ("vort", ref.vort), ("tendU", ref.tendU),
                      ("tendH", ref.tendH)):
        assert np.array_equal(d[name], exp), name
    cl.step_rk4(); ref.step_rk4(dt)                 # RK4, then a Forward-Euler step that carries nothing over
    cl.step_fe(0); ref.step_fe(dt, 0)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, ref.u[1]) and np.array_equal(gh, ref.h[1]) and np.array_equal(gs, ref.ssh[1])
    cl.close()


def test_metis_workflow_files(tmp_path):
    """The cell graph in the METIS format MPAS tools use (graph.info) and a part file read back (graph.info.part.N): the
    route to a METIS partition where gpmetis exists (it does not in this image; recursive coordinate bisection is the
    built-in partitioner).  The file is checked against the mesh: symmetric, every adjacency once per direction."""
    mesh = mg.icosahedral_mesh(8)
    gpath = str(tmp_path / "graph.info")
    par.write_graph_info(mesh, gpath)
    lines = open(gpath).read().splitlines()
    nC, nAdj = (int(x) for x in lines[0].split())
    assert nC == mesh.nCells and nAdj == mesh.nEdges and len(lines) == nC + 1
    nb = [set(int(x) for x in ln.split()) for ln in lines[1:]]
    assert sum(len(x) for x in nb) == 2 * nAdj
    assert all((c + 1) in nb[d - 1] for c in range(nC) for d in nb[c])
    part = par.partition_cells(mesh, 4)
    ppath = gpath + ".part.4"
    np.savetxt(ppath, part, fmt="%d")
    back = par.read_partition(ppath, mesh.nCells)
    assert np.array_equal(back, part)
    # a partition is what it costs: the cut of the bisection equals the number of edges between parts, and every part is
    # within one cell of the mean
    assert par.edge_cut(mesh, part) == int(sum(1 for e in range(mesh.nEdges)
                                               if part[mesh.cellsOnEdge.reshape(-1, 2)[e, 0] - 1] != part[mesh.cellsOnEdge.reshape(-1, 2)[e, 1] - 1]))
    sizes = np.bincount(part)
    assert sizes.max() - sizes.min() <= 1
    with pytest.raises(ValueError):
        par.read_partition(ppath, mesh.nCells + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("world,K,nsteps", [(2, 60, 2), (4, 60, 2), (3, 34, 3), (8, 60, 1), (5, 1, 3)])
def test_reverse_mode_on_a_partitioned_mesh(world, K, nsteps, overlap):
    """d sum(ssh^2) / d initial state of an RK4 run on 2-8 ranks (tests/ on one GPU): every rank tapes its part of the
    distributed steps (halo rows included) and reverses it; the assembled gradient equals the single-domain oracle's
    (OracleAdjointRK4) bit for bit.  overlap=False: the halo rows of the adjoint fields are exchanged before every transposed
    stage, which runs over the whole local mesh; overlap=True: a stage transposes the boundary class, starts the exchange of what
    that produced and transposes the interior class meanwhile (moka_adjoint_rk4_stage_part; K = 1 has no chunk kernels and falls
    back to the first form)."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(20)
    rng = np.random.default_rng(37 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 20.0
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjointRK4(st)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, direct=False)
    cl.exchange_state()
    cl.tape(nsteps)
    for _ in range(nsteps):
        cl.step_rk4_taped()
        adj.step_rk4(dt)
    gs, gu_, gh_ = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu_, st.u[1]) and np.array_equal(gh_, st.h[1]) and np.array_equal(gs, st.ssh[1])
    gU, gH = adj.gradient_sum_sq_ssh()
    assert all(m.adjoint_parts_available() == (K >= 34) for m in cl.models)
    gu, gh = cl.adjoint_gradient(nsteps, mesh.nCells, mesh.nEdges, K, overlap=overlap)
    assert np.array_equal(gu, gU)
    assert np.array_equal(gh, gH)
    assert np.abs(gU).max() > 0 and np.abs(gH).max() > 0
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,flags,nsteps,direct", [(2, 60, 3, 3, True), (4, 60, 0, 2, False), (3, 34, 1, 3, True), (8, 60, 3, 2, True),
                                                         (5, 1, 7, 4, False), (4, 60, 2, 3, True)])
def test_reverse_mode_of_a_partitioned_forward_euler_run(world, K, flags, nsteps, direct):
    """The computation the reference differentiates with Enzyme (test/enzyme/test_Enzyme_end2end.jl): d sum(ssh^2) / d initial
    state of a Forward-Euler run, here on 2-8 ranks with the reference's quirk flags: the adjoints of normalVelocity,
    layerThickness, ssh and of the carried layerThicknessEdge, assembled from the ranks, equal the single-domain oracle adjoint
    (oracle_step_fe_adjoint) bit for bit."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(20)
    rng = np.random.default_rng(43 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 20.0
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h)
    adj = orc.OracleAdjoint(st)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, direct=direct)
    cl.exchange_state()
    cl.tape(nsteps)
    for _ in range(nsteps):
        cl.step_fe_taped(flags)
        adj.step_fe(dt, flags)
    gs, gu_, gh_ = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu_, st.u[1].reshape(mesh.nEdges, K)) and np.array_equal(gh_, st.h[1].reshape(mesh.nCells, K))
    gS, gU, gH, gE = adj.gradient_sum_sq_ssh()
    g = cl.adjoint_gradient_fe(nsteps, mesh, K)
    assert np.array_equal(g["ssh"], gS)
    assert np.array_equal(g["normalVelocity"], np.asarray(gU).reshape(mesh.nEdges, K))
    assert np.array_equal(g["layerThickness"], np.asarray(gH).reshape(mesh.nCells, K))
    assert np.array_equal(g["layerThicknessEdge"], np.asarray(gE).reshape(mesh.nEdges, K))
    assert np.abs(gU).max() > 0
    cl.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,visc,nsteps", [(2, 60, 0.0, 3), (4, 60, 1.0, 2), (3, 34, 0.0, 2), (8, 60, 0.0, 2), (4, 3, 0.0, 3)])
def test_nonlinear_terms_on_a_partitioned_mesh(world, K, visc, nsteps):
    """The optional nonlinear terms (potential-vorticity Coriolis, kinetic-energy gradient, Del2 mixing) on 2-8 ranks: a halo
    two cells deep with the vertex-side fields, every stage one launch over the whole local mesh, the exchange behind it;
    owned rows equal the single-domain restatement (OracleNonlinear; parity unpinned: the reference has no such terms) bit
    for bit.  Also: the same two-ring local meshes under the reference's linear terms."""
    import oracle as orc
    mesh = mg.icosahedral_mesh(20)
    rng = np.random.default_rng(53 + world)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    dt = 20.0
    v = visc * 0.01 * float(mesh.dcEdge.min()) ** 2 / dt
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    nl = orc.OracleNonlinear(om, visc_del2=v) if v else orc.OracleNonlinear(om)
    st = orc.OracleState(om, ssh, u, h)
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, direct=False, nonlinear=True, visc_del2=v)
    assert all(m.lm.rings == 2 for m in cl.models)
    cl.exchange_state()
    parts = all(L.lib().moka_rk4_dist_parts_available(m._halo) for m in cl.models)
    assert parts == (K % 2 == 0 and K <= 64)             # the per-patch kernels serve even K <= 64; K = 3: whole-mesh stages only
    for i in range(nsteps):
        # alternately: every stage as one launch over the whole local mesh with the exchange behind it, and the overlapped form
        # (stage kernel over boundary / interior patches, preparation passes split so that the interior's overlaps the exchange)
        (cl.step_rk4 if parts and i % 2 == 0 else cl.step_rk4_whole)()
        nl.step_rk4(st, dt)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, st.u[1])
    assert np.array_equal(gh, st.h[1])
    assert np.array_equal(gs, st.ssh[1])
    if parts and world <= 4:                               # the same over the direct transport (stores into the neighbours' fields)
        cd = par.LocalCluster(mesh, st.ssh[1], st.u[1], st.h[1], rest, dt, world, direct=True, nonlinear=True, visc_del2=v)
        cd.exchange_state()
        ref = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
        for _ in range(2):
            cd.step_rk4()
            nl.step_rk4(ref, dt)
        ds, du, dh = cd.gather_owned(mesh.nCells, mesh.nEdges, K)
        assert np.array_equal(du, ref.u[1]) and np.array_equal(dh, ref.h[1]) and np.array_equal(ds, ref.ssh[1])
        cd.close()
    # back to the reference's terms on the same (two-ring) local meshes: the ordinary distributed step
    from moka_hip import api as mk
    for m in cl.models:
        mk.set_nonlinear(m.Prog, False)
    lin = orc.OracleState(om, st.ssh[1], st.u[1], st.h[1])
    cl.exchange_state()
    for _ in range(2):
        cl.step_rk4()
        lin.step_rk4(dt)
    gs, gu, gh = cl.gather_owned(mesh.nCells, mesh.nEdges, K)
    assert np.array_equal(gu, lin.u[1]) and np.array_equal(gh, lin.h[1]) and np.array_equal(gs, lin.ssh[1])
    cl.close()


def test_two_ring_local_meshes_for_the_nonlinear_terms():
    """build_local(rings=2, vertex_fields=True): every cell within two rings of an owned cell is local, the exchange lists of
    the ranks match pairwise, and every vertex whose potential vorticity an owned entity can read -- the vertices of the
    owned cells and of their neighbours -- is local with its three cells and three edges (no stand-ins there)."""
    mesh = mg.icosahedral_mesh(10)
    coe = mesh.cellsOnEdge.astype(np.int64) - 1
    nbrs = [set() for _ in range(mesh.nCells)]
    for a, b in coe:
        nbrs[a].add(b); nbrs[b].add(a)
    for world in (2, 5):
        part = par.partition_cells(mesh, world)
        lms = [par.build_local(mesh, part, r, world, rings=2, vertex_fields=True) for r in range(world)]
        for r, lm in enumerate(lms):
            own = set(np.nonzero(part == r)[0].tolist())
            ring1 = set().union(*(nbrs[c] for c in own)) - own
            ring2 = set().union(*(nbrs[c] for c in ring1)) - own - ring1
            assert set(lm.cells_g.tolist()) == own | ring1 | ring2
            assert lm.rings == 2 and lm.mesh.kiteAreasOnVertex is not None
            # what r receives from q is exactly what q sends to r (global ids; the order is agreed later, in finish())
            for i, q in enumerate(lm.neighbors):
                j = lms[q].neighbors.index(r)
                rc = lm.cells_g[lm.recv_cells[lm.recv_cell_off[i]:lm.recv_cell_off[i + 1]]]
                sc = lms[q].cells_g[lms[q].send_cells[lms[q].send_cell_off[j]:lms[q].send_cell_off[j + 1]]]
                assert np.array_equal(np.sort(rc), np.sort(sc))
                re_ = lm.edges_g[lm.recv_edges[lm.recv_edge_off[i]:lm.recv_edge_off[i + 1]]]
                se = lms[q].edges_g[lms[q].send_edges[lms[q].send_edge_off[j]:lms[q].send_edge_off[j + 1]]]
                assert np.array_equal(np.sort(re_), np.sort(se))
            # vertices of the owned cells and of ring 1: local, complete, in the reference's slot order
            g2l_v = -np.ones(mesh.nVertices, dtype=np.int64); g2l_v[lm.verts_g] = np.arange(lm.verts_g.size)
            need = np.nonzero(np.isin(mesh.cellsOnVertex - 1, list(own | ring1)).any(axis=1))[0]
            assert np.all(g2l_v[need] >= 0)
            lv = g2l_v[need]
            assert np.array_equal(lm.cells_g[lm.mesh.cellsOnVertex[lv] - 1], mesh.cellsOnVertex[need] - 1)
            assert np.array_equal(lm.edges_g[lm.mesh.edgesOnVertex[lv] - 1], mesh.edgesOnVertex[need] - 1)
            assert np.array_equal(lm.mesh.kiteAreasOnVertex[lv], mesh.kiteAreasOnVertex[need])
