#!/usr/bin/env python3
"""GPU box helper: one rank's share of an N-way partition of config 4 on this GPU, distributed step loop with the halo
copies left out (pack / unpack kernels and all stream dependencies kept): the per-rank GPU time an N-GPU run cannot beat.
   python tools/rank_timing.py [world=8] [rank=0]"""
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
m = par.DistributedModel(mesh, ssh, u, h, rest, dts, b, rank, world, transport="local")
lib = L.lib()


def step():
    L.check(lib.moka_rk4_dist_begin(m._halo, m.dt), b._h)
    for s in (1, 2, 3, 4):
        L.check(lib.moka_rk4_dist_stage(m._halo, s, 0), b._h)
        L.check(lib.moka_halo_pack(m._halo, s, m.sendbuf.data_ptr()), b._h)
        L.check(lib.moka_rk4_dist_stage(m._halo, s, 1), b._h)
        L.check(lib.moka_halo_unpack(m._halo, s, m.recvbuf.data_ptr()), b._h)
    L.check(lib.moka_rk4_dist_end(m._halo), b._h)


for _ in range(10):
    step()
b.synchronize(); t0 = time.perf_counter()
N = 50
for _ in range(N):
    step()
b.synchronize(); t1 = time.perf_counter()
info = m.info()
print(f"world {world} rank {rank}: {info['rank_cells_owned']} owned cells, {info['patches_boundary']} boundary / "
      f"{info['patches_owned']} owned patches, halo {info['halo_bytes_per_stage'] / 1e6:.2f} MB/stage: "
      f"{1e3 * (t1 - t0) / N:.3f} ms per step without the copies")
