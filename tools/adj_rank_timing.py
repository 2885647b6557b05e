#!/usr/bin/env python3
"""GPU box helper: one rank's share of an N-way partition of config 4, reverse sweep of taped RK4 steps with the halo traffic itself
left out (pack + unpack kernels stand for the exchange): ms per reversed step of
  * the plain form: exchange in front of every transposed stage, the stage over the whole local mesh (halo entities redundantly), and
  * the overlapped form: a stage transposes the boundary class, the rows that produced are packed on the communication stream, the
    interior class is transposed meanwhile (moka_adjoint_rk4_stage_part).
   python tools/adj_rank_timing.py [world=8] [rank=0] [steps=3]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402
from moka_hip import parallel as par       # noqa: E402

world, rank, nsteps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 0), (3, 3)))
mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
part = par.partition_cells(mesh, world)
lm0 = par.build_local(mesh, part, rank, world)
asked = {q: (lm0.cells_g[lm0.send_cells[lm0.send_cell_off[i]:lm0.send_cell_off[i + 1]]],
             lm0.edges_g[lm0.send_edges[lm0.send_edge_off[i]:lm0.send_edge_off[i + 1]]]) for i, q in enumerate(lm0.neighbors)}
m = par.DistributedModel(mesh, ssh, u, h, rest, dts, b, rank, world, transport="local", part=part, exchange_lists=lambda w: asked)
m._transport = lambda: None                # no peers here: what a transport would deliver stays what the receive buffer holds
m._transport_buffered = m._transport
m.tape(nsteps)


def reverse(overlap):
    for _ in range(nsteps):
        m.step_rk4_taped()
    b.synchronize(); t0 = time.perf_counter()
    m.adjoint_seed()
    if overlap:
        m.adjoint_pack(4)
        m._adjoint_unpack(*m._adjoint_fields(4))
        for step in range(nsteps):
            for sg in (4, 3, 2, 1):
                last = step == nsteps - 1 and sg == 1
                out = m.adjoint_stage_boundary_and_pack(sg, pack=not last)
                m.adjoint_stage_interior(sg)
                if not last:
                    m._adjoint_unpack(*out)
    else:
        for _ in range(nsteps):
            for sg in (4, 3, 2, 1):
                m.adjoint_pack(sg)
                m.adjoint_unpack_and_stage(sg)
    b.synchronize()
    return 1e3 * (time.perf_counter() - t0) / nsteps


assert m.adjoint_parts_available()
reverse(False); reverse(True)
tw = min(reverse(False) for _ in range(3))
tp = min(reverse(True) for _ in range(3))
info = m.info()
print(f"reverse RK4 sweep, world {world} rank {rank}: {info['rank_cells_owned']} owned cells of {lm0.mesh.nCells} local, ms per reversed step "
      f"(seed included, {nsteps} steps): exchange then whole-mesh stages {tw:.3f}, boundary class / exchange beside the interior class {tp:.3f}", flush=True)
