#!/usr/bin/env python3
"""GPU box helper: one rank's share of an N-way partition of config 4 on this GPU, distributed step loop with the halo
traffic itself left out: the per-rank GPU time an N-GPU run cannot beat -- for the stage launches alone, with the one
gather kernel per stage the direct transport adds (its push kernel does the same reads; the stores go over xGMI), and
with the pack + unpack pair of the buffered transports.
   python tools/rank_timing.py [world=8] [rank=0] [overlap=-1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402
from moka_hip import parallel as par       # noqa: E402

world, rank, overlap = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 8), (2, 0), (3, -1)))
mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
part = par.partition_cells(mesh, world)
lm0 = par.build_local(mesh, part, rank, world)
# no peers here: the rows go out in the default order of the exchange lists
asked = {q: (lm0.cells_g[lm0.send_cells[lm0.send_cell_off[i]:lm0.send_cell_off[i + 1]]],
             lm0.edges_g[lm0.send_edges[lm0.send_edge_off[i]:lm0.send_edge_off[i + 1]]]) for i, q in enumerate(lm0.neighbors)}
m = par.DistributedModel(mesh, ssh, u, h, rest, dts, b, rank, world, transport="local", part=part, exchange_lists=lambda w: asked)
lib = L.lib()
L.check(lib.moka_halo_set_overlap(m._halo, overlap), b._h)     # -1 automatic, 0 / 1: boundary launch beside the interior launch


def step(pack, unpack):
    L.check(lib.moka_rk4_dist_begin(m._halo, m.dt), b._h)
    for s in (1, 2, 3, 4):
        L.check(lib.moka_rk4_dist_stage(m._halo, s, 0), b._h)
        if pack:
            L.check(lib.moka_halo_pack(m._halo, s, m.sendbuf.data_ptr()), b._h)
        L.check(lib.moka_rk4_dist_stage(m._halo, s, 1), b._h)
        if unpack:
            L.check(lib.moka_halo_unpack(m._halo, s, m.recvbuf.data_ptr()), b._h)
    L.check(lib.moka_rk4_dist_end(m._halo), b._h)


def timed(pack, unpack, N=50):
    for _ in range(10):
        step(pack, unpack)
    b.synchronize(); t0 = time.perf_counter()
    for _ in range(N):
        step(pack, unpack)
    b.synchronize()
    return 1e3 * (time.perf_counter() - t0) / N


info = m.info()
t_stage, t_direct, t_buf = timed(False, False), timed(True, False), timed(True, True)
print(f"world {world} rank {rank} overlap {overlap}: {info['rank_cells_owned']} owned cells, {info['patches_boundary']} boundary / "
      f"{info['patches_owned']} owned patches, halo {info['halo_bytes_per_stage'] / 1e6:.2f} MB/stage, ms per step: "
      f"boundary + interior launches only {t_stage:.3f}; + one gather kernel per stage (= the direct transport's push) {t_direct:.3f}; "
      f"+ unpack (buffered transport without the copies) {t_buf:.3f}")
