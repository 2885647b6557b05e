#!/usr/bin/env python3
"""GPU box helper (experiment): the direct transport's handshake done by the host thread ("ipc": wait for the
push event, store the flags, poll the neighbours' flags) against the same handshake ENQUEUED as stream memory operations ("ipc-smo":
hipStreamWriteValue64 behind the push kernel, hipStreamWaitValue64 in front of the next boundary launch).  Several processes on the
one GPU of the box, exactly as bench.py --gpus N starts them:

    timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 \\
        tools/smo_probe.py [m=128] [K=60]

Both candidates go through the transport selection's three phases (set-up, one exchange byte for byte against gloo, three whole
steps bit for bit against the same steps over gloo); then 20 steps of each are timed and the library's exchange statistics read."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import numpy as np                          # noqa: E402
import torch                                # noqa: E402,F401
import torch.distributed as dist            # noqa: E402

import moka_hip as mk                       # noqa: E402
from moka_hip import meshgen as mg          # noqa: E402
from moka_hip import parallel as par        # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mesh = mg.icosahedral_mesh(m)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
backend = mk.MokaHIP(0)
model = par.DistributedModel(mesh, ssh, u, h, rest, dts, backend, rank, world, transport="gloo", placement_tries=1)
model.exchange_state()
msgs = []
log = (lambda s: (msgs.append(s), print(f"[rank {rank}] {s}", flush=True))) if rank == 0 else msgs.append
name, times = par.choose_transport(model, ("ipc", "ipc-smo"), ("gloo",), None, log, trial_steps=20)
if rank == 0:
    print(f"qualified: {times} -> {name}", flush=True)
for cand in times:
    model.set_transport(cand)
    for _ in range(3):
        model.step_rk4()
    model.sync_device(); dist.barrier()
    model.exchange_stats(True)
    t0 = time.perf_counter()
    for _ in range(20):
        model.step_rk4()
    t_enq = time.perf_counter() - t0           # host time to get 20 steps queued / handshaken
    model.sync_device(); dist.barrier()
    wall = time.perf_counter() - t0
    st = model.exchange_stats()
    model.exchange_stats(False)
    rows = [None] * world
    dist.all_gather_object(rows, {"host_call_ms": t_enq / 20 * 1e3, "wall_ms": wall / 20 * 1e3, **st})
    if rank == 0:
        print(f"== {cand}: {world} ranks on one GPU, m={m} K={K} ==")
        for r, d in enumerate(rows):
            print(f"  rank {r}: wall {d['wall_ms']:.3f} ms/step, host inside the step calls {d['host_call_ms']:.3f} ms/step "
                  f"(wait for own push {d['host_signal_wait_ms_per_step']:.3f}, poll {d['host_wait_ms_per_step']:.3f}, event->flag "
                  f"{d['push_to_flag_us']:.2f} us), boundary launches {d['boundary_launch_ms_per_step']:.3f} ms, interior {d['interior_launch_ms_per_step']:.3f} ms", flush=True)
model.set_transport("gloo")
dist.barrier()
model.close()
dist.destroy_process_group()
