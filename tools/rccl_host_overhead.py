"""Host-side cost of the per-stage RCCL call of the distributed step (one rank; run under torch.distributed.run
--nproc-per-node 1 on a GPU box).  Times the host cost of (a) one all_to_all_single with split sizes, (b) one
all_gather / all_reduce for scale, issued on the library's comm stream, and (c) the library calls of one distributed
RK4 step without transport."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import moka_hip as mk  # noqa: E402
from moka_hip import meshgen as mg, parallel as mp  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
mesh = mg.icosahedral_mesh(32); K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
model = mp.DistributedModel(mesh, ssh, u, h, rest, dts, b, 0, 8, transport="nccl")
n = model.sendbuf.numel()
out = torch.empty_like(model.sendbuf)
N = 300


def timeit(label, fn):
    for _ in range(20):
        fn()
    model.comm_stream.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t1 = time.perf_counter()
    model.comm_stream.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: host {1e6 * (t1 - t0) / N:.1f} us per call, with drain {1e6 * (t2 - t0) / N:.1f} us ({n * 8 / 1e6:.2f} MB)")


def a2a():
    with torch.cuda.stream(model.comm_stream):
        dist.all_to_all_single(out, model.sendbuf, [n], [n])


def a2a_async():
    with torch.cuda.stream(model.comm_stream):
        w = dist.all_to_all_single(out, model.sendbuf, [n], [n], async_op=True)
        w.wait()


def ar():
    with torch.cuda.stream(model.comm_stream):
        dist.all_reduce(out)


timeit("all_to_all_single (sync op)", a2a)
timeit("all_to_all_single (async_op + wait)", a2a_async)
timeit("all_reduce", ar)
lib, hh, ctx = mk.lib.lib(), model._halo, b._h


def stage_calls():
    mk.lib.check(lib.moka_rk4_dist_begin(hh, model.dt), ctx)
    for s in (1, 2, 3, 4):
        mk.lib.check(lib.moka_rk4_dist_stage(hh, s, 0), ctx)
        mk.lib.check(lib.moka_halo_pack(hh, s, model.sendbuf.data_ptr()), ctx)
        mk.lib.check(lib.moka_rk4_dist_stage(hh, s, 1), ctx)
        mk.lib.check(lib.moka_halo_unpack(hh, s, model.recvbuf.data_ptr()), ctx)
    mk.lib.check(lib.moka_rk4_dist_end(hh), ctx)


def t_step():
    for _ in range(20):
        stage_calls()
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        stage_calls()
    t1 = time.perf_counter()
    b.synchronize()
    t2 = time.perf_counter()
    print(f"library calls of one distributed RK4 step (rank 0 of 8 on a {mesh.nCells}-cell mesh, no transport): "
          f"host {1e6 * (t1 - t0) / N:.1f} us, with drain {1e6 * (t2 - t0) / N:.1f} us")


t_step()
dist.destroy_process_group()
