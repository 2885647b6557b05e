#!/usr/bin/env python3
"""GPU box helper: one rank's share of an N-way partition of config 4 with the optional NONLINEAR terms (two-ring halo), the halo
traffic itself left out (pack + unpack kernels stand for the exchange): ms per RK4 step of
  * the plain form: every stage one launch over the whole local mesh (halo patches redundantly), exchange behind it, and
  * the overlapped form of moka_rk4_dist_step: stage kernel over the boundary patches -> exchange starts -> stage kernel over the
    interior patches -> preparation pass of the NEXT stage over the interior patches (owned rows only) -> exchange has arrived ->
    preparation pass over the boundary and halo patches.
   python tools/nl_rank_timing.py [world=8] [rank=0]
Under `rocprofv3 --kernel-trace` the trace of the last steps gives the timeline (tools/nl_timeline.py <results.db>)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402
from moka_hip import parallel as par       # noqa: E402

world, rank = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 0)))
mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
part = par.partition_cells(mesh, world)
lm0 = par.build_local(mesh, part, rank, world, rings=2, vertex_fields=True)
asked = {q: (lm0.cells_g[lm0.send_cells[lm0.send_cell_off[i]:lm0.send_cell_off[i + 1]]],
             lm0.edges_g[lm0.send_edges[lm0.send_edge_off[i]:lm0.send_edge_off[i + 1]]]) for i, q in enumerate(lm0.neighbors)}
m = par.DistributedModel(mesh, ssh, u, h, rest, dts, b, rank, world, transport="local", part=part, exchange_lists=lambda w: asked,
                         nonlinear=True)
lib, H = L.lib(), m._halo
assert lib.moka_rk4_dist_parts_available(H)


def ck(rc):
    L.check(rc, b._h)


def step_whole():
    ck(lib.moka_rk4_dist_begin(H, m.dt))
    for s in (1, 2, 3, 4):
        ck(lib.moka_rk4_dist_stage(H, s, 2))
        ck(lib.moka_halo_pack(H, s, m.sendbuf.data_ptr()))
        ck(lib.moka_halo_unpack(H, s, m.recvbuf.data_ptr()))
    ck(lib.moka_rk4_dist_end(H))


def step_parts():
    ck(lib.moka_rk4_dist_begin(H, m.dt))
    ck(lib.moka_rk4_dist_stage(H, 1, 4))
    for s in (1, 2, 3, 4):
        ck(lib.moka_rk4_dist_stage(H, s, 3))
        ck(lib.moka_rk4_dist_stage(H, s, 0))
        ck(lib.moka_halo_pack(H, s, m.sendbuf.data_ptr()))
        ck(lib.moka_rk4_dist_stage(H, s, 1))
        if s < 4:
            ck(lib.moka_rk4_dist_stage(H, s + 1, 4))
        ck(lib.moka_halo_unpack(H, s, m.recvbuf.data_ptr()))
    ck(lib.moka_rk4_dist_end(H))


def timed(fn, N=30):
    for _ in range(5):
        fn()
    b.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        fn()
    b.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


info = m.info()
tw, tp = timed(step_whole), timed(step_parts)
print(f"nonlinear terms, world {world} rank {rank}: {info['rank_cells_owned']} owned cells of {lm0.mesh.nCells} local (two-ring halo), "
      f"{info['patches_boundary']} boundary / {info['patches_owned']} owned patches, halo {info['halo_bytes_per_stage'] / 1e6:.2f} MB/stage; "
      f"ms per RK4 step: whole-mesh stages {tw:.3f}, boundary / interior parts {tp:.3f}", flush=True)
step_parts(); step_parts()
b.synchronize()
