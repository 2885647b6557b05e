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
            assert np.all(lm.cell_class[lm.owned_cell_mask] <= 1) and np.all(lm.cell_class[~lm.owned_cell_mask] == 2)
            # every halo cell is received, every local edge without an owned cell is received
            assert set(lm.recv_cells.tolist()) == set(np.nonzero(~lm.owned_cell_mask)[0].tolist())
            # the local mesh is accepted by the host plan, classes order the patches
            plan = L.Plan(lm.mesh, 3, max_level_edge_top=3, ordering=L.ORDER_RCB, patch_cells=8, cell_class=lm.cell_class)
            cperm = plan.permutation(L.CELL)
            cls = lm.cell_class[cperm]
            assert np.all(np.diff(cls) >= 0)                                     # class-major ordering
            cs, es, _ = plan.patch_ranges()      # the record-staging kernels size their LDS from these maxima
            assert plan.info["maxPatchEdges"] == np.diff(es).max() and plan.info["maxPatchCells"] == np.diff(cs).max()
            # launch ranges: patches [0, pB) are computed before the halo is packed, [pB, pO) while it travels, the rest never.
            # Whatever the balancing of edge ownership does, every sent edge must be produced by the boundary launch and
            # every edge with an owned cell by a launched patch.
            P = plan.info["patch_cells"]
            nB, nO = int((lm.cell_class == 0).sum()), int((lm.cell_class <= 1).sum())
            pB, pO = -(-nB // P), -(-nO // P)
            eperm = plan.permutation(L.EDGE)
            patch_of_edge = np.empty(lm.mesh.nEdges, dtype=np.int64)
            patch_of_edge[eperm] = np.searchsorted(es, np.arange(lm.mesh.nEdges), side="right") - 1
            assert np.all(patch_of_edge[lm.send_edges] < pB)
            coe = lm.mesh.cellsOnEdge - 1
            has_owned = lm.owned_cell_mask[coe[:, 0]] | lm.owned_cell_mask[coe[:, 1]]
            assert np.all(patch_of_edge[has_owned] < pO)
            cpos = np.empty(lm.mesh.nCells, dtype=np.int64)
            cpos[cperm] = np.arange(lm.mesh.nCells)
            assert np.all(cpos[lm.send_cells] // P < pB)
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
def test_rccl_collectives_on_the_library_comm_stream():
    """What one GPU can show of the RCCL transport: see tests/rccl_worker.py."""
    run_workers(1, worker="rccl_worker.py")


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,P,nsteps,sbytes", [(2, 60, 0, 4, 8), (4, 60, 0, 4, 8), (8, 60, 0, 3, 8), (3, 1, 0, 5, 8), (4, 60, 8, 3, 8),
                                                     (5, 80, 0, 2, 8), (4, 80, 0, 3, 4), (3, 60, 0, 2, 4)])
def test_stream_ordered_exchange_in_one_process(world, K, P, nsteps, sbytes):
    """All ranks in one process on one GPU, halo messages as stream-ordered device copies with no host synchronisation
    anywhere in the step -- the ordering RCCL gives.  Exercises the two-stream / event choreography of the distributed
    RK4 step (boundary patches + pack on the comm stream, interior on the compute stream, unpack overlapping it): any
    missing dependency shows up as a mismatch against the single-domain oracle."""
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
    cl = par.LocalCluster(mesh, ssh, u, h, rest, dt, world, patch_cells=P, state_bytes=sbytes)
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
