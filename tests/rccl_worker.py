"""One-rank RCCL check (launched under torch.distributed.run on a GPU box).  The multi-GPU transport of
DistributedModel issues RCCL operations on the library's own high-priority comm stream, wrapped as a
torch.cuda.ExternalStream, between kernels the library launches on that same stream.  One GPU cannot host two RCCL
ranks, so this covers what one rank can: the RCCL process group comes up next to libmoka_hip, collectives issued under
the wrapped stream are ordered after the library's pack kernel and before its unpack kernel, and the library still
computes correctly afterwards."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import datetime as _dt  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import moka_hip as mk  # noqa: E402
import oracle as orc  # noqa: E402
from moka_hip import meshgen as mg  # noqa: E402
from moka_hip import parallel as par  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_world_size() == 1
    K, dt, nsteps = 60, 20.0, 3
    mesh = mg.icosahedral_mesh(16)
    rng = np.random.default_rng(3)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h)

    backend = mk.MokaHIP(0)
    # a 2-way partition, rank 0's half: its send buffer is packed by the library on the comm stream
    # (there is no rank 1 to say in which order it wants its rows: take the default order of the exchange lists)
    part = par.partition_cells(mesh, 2)
    lm0 = par.build_local(mesh, part, 0, 2)
    asked = {1: (lm0.cells_g[lm0.send_cells], lm0.edges_g[lm0.send_edges])}
    dm = par.DistributedModel(mesh, ssh, u, h, rest, dt, backend, 0, 2, transport="nccl", part=part,
                              exchange_lists=lambda wants: asked)
    lib, ctx = mk.lib.lib(), backend._h
    mk.lib.check(lib.moka_halo_pack(dm._halo, 0, dm.sendbuf.data_ptr()), ctx)
    expect = par.pack_numpy(dm.lm, K, ssh[dm.lm.cells_g], u[dm.lm.edges_g], h[dm.lm.cells_g])
    with torch.cuda.stream(dm.comm_stream):           # RCCL ops ordered after the pack kernel, no host sync before
        gathered = torch.empty_like(dm.sendbuf)
        dist.all_gather_into_tensor(gathered, dm.sendbuf)
        summed = dm.sendbuf.clone()
        dist.all_reduce(summed)
        out = torch.empty_like(dm.sendbuf)
        dist.all_to_all_single(out, dm.sendbuf)
    dm.comm_stream.synchronize()
    n = expect.size
    for name, t in (("all_gather", gathered), ("all_reduce", summed), ("all_to_all", out)):
        assert np.array_equal(t.cpu().numpy()[:n], expect), name
    # the library's own work is unaffected by the process group living next to it
    config = {"time_management": {"config_start_time": _dt.datetime(1, 1, 1), "config_run_duration": _dt.timedelta(hours=1)},
              "time_integration": {"config_dt": _dt.timedelta(seconds=dt), "config_number_of_time_levels": 2},
              "output": {"output_interval": _dt.timedelta(hours=1)}}
    Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, config, backend, multilayer=True)
    mk.run_steps(Prog, mk.RungeKutta4, dt, nsteps)
    for _ in range(nsteps):
        ref.step_rk4(dt)
    assert np.array_equal(Prog.normalVelocity[-1].get(), ref.u[1])
    assert np.array_equal(Prog.layerThickness[-1].get(), ref.h[1])
    dist.barrier()
    dist.destroy_process_group()
    print("OK on 1 RCCL rank")


if __name__ == "__main__":
    main()
