"""Worker of test_transport_selection_agrees_on_failures (gloo, CPU only, launched under torch.distributed.run).

moka_hip.parallel.choose_transport is driven with a stand-in for DistributedModel whose "device" is a numpy state and whose
exchange is gloo: the selection logic -- phases closed by an agreement of all ranks, a failing rank never leaving the others
in a collective or waiting for a timeout -- needs no GPU.  The stand-in's candidates fail in scripted ways:
  "setup-fails-on-1"   raises during set-up on rank 1 only                       (phase 1)
  "no-direct-on-1"     "ipc" while rank 1 reports direct_available = False: nobody may enter connect_ipc (phase 1)
  "bytes-differ-on-0"  delivers other bytes than gloo on rank 0 only             (phase 2)
  "stale-on-1"         passes set-up and the byte comparison, but its steps leave a different state on rank 1 only (phase 3)
  "slow", "fast"       work; "fast" must win the timing
Every rank must return the same name, and quickly: the whole run is bounded by a timeout far below any collective timeout.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from moka_hip import parallel as par  # noqa: E402


class FakeModel:
    def __init__(self, rank, world, direct_available=True):
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.transport = "gloo"
        self.direct_available = direct_available
        self.connected = False
        self.connect_calls = 0
        self.state = np.arange(8, dtype=np.float64) + rank

    # --- what choose_transport needs ---
    def set_transport(self, name):
        if name == "setup-fails-on-1" and self.rank == 1:
            raise RuntimeError("scripted set-up failure")
        self.transport = name

    def connect_ipc(self):
        self.connect_calls += 1
        t = torch.ones(1)
        dist.all_reduce(t)                      # a collective: every rank must be here, or this hangs
        self.connected = True

    def verify_transport(self, trusted="gloo"):
        return not (self.transport == "bytes-differ-on-0" and self.rank == 0)

    def step_rk4(self):
        # one "step": every rank adds its neighbour's first element (an exchange over gloo), the scripted candidate reads a stale value
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        send, recv = torch.tensor([self.state[0]]), torch.zeros(1, dtype=torch.float64)
        reqs = [dist.irecv(recv, prv), dist.isend(send, nxt)]
        for w in reqs:
            w.wait()
        got = float(recv[0])
        if self.transport == "stale-on-1" and self.rank == 1:
            got -= 1.0
        self.state = self.state * 0.5 + got
        if self.transport == "slow":
            time.sleep(0.02)

    def snapshot(self):
        return (self.state.copy(),)

    def restore(self, snap):
        self.state = snap[0].copy()

    def sync_device(self):
        pass


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    msgs = []
    t0 = time.time()

    m = FakeModel(rank, world)
    s0 = m.snapshot()[0]
    name, times = par.choose_transport(m, ("setup-fails-on-1", "bytes-differ-on-0", "stale-on-1"), ("gloo",), None, msgs.append)
    assert name == "gloo" and times == {}, (name, times)
    assert np.array_equal(m.snapshot()[0], s0), "the selection must hand the state back as it found it"
    assert m.transport == "gloo"

    # a working pair: the faster one is kept, the broken one in between is dropped by everybody
    name, times = par.choose_transport(m, ("slow", "stale-on-1", "fast"), ("gloo",), None, msgs.append, trial_steps=3)
    assert name == "fast" and set(times) == {"slow", "fast"} and times["fast"] < times["slow"], (name, times)

    # "ipc" while one rank cannot go direct: nobody enters connect_ipc, "ipc-acq" is not tried into a hang either
    m2 = FakeModel(rank, world, direct_available=(rank != 1))
    name, _ = par.choose_transport(m2, ("ipc",), ("gloo",), None, msgs.append)
    assert name == "gloo" and m2.connect_calls == 0, (name, m2.connect_calls)

    # "ipc" that sets up everywhere qualifies (the stand-in's steps do not depend on the name)
    m3 = FakeModel(rank, world)
    name, times = par.choose_transport(m3, ("ipc",), ("gloo",), None, msgs.append, trial_steps=1)
    assert name == "ipc" and m3.connect_calls == 1, (name, times)

    # nothing qualifies at all
    try:
        par.choose_transport(m, ("setup-fails-on-1",), ("bytes-differ-on-0",), None, msgs.append)
        raise AssertionError("no transport should have qualified")
    except RuntimeError:
        pass

    elapsed = time.time() - t0
    assert elapsed < 20.0, f"the selection waited for something: {elapsed:.1f} s"
    # the rank that saw a failure says why; the others just agree
    if rank == 1:
        assert any("setup-fails-on-1" in x and "raised" in x for x in msgs), msgs
        assert any("stale-on-1" in x and "failed" in x for x in msgs), msgs
    if rank == 0:
        assert any("bytes-differ-on-0" in x and "failed" in x for x in msgs), msgs
    ok = torch.ones(1)
    dist.all_reduce(ok)
    if rank == 0:
        print(f"transport_worker: OK on {world} ranks in {elapsed:.1f} s")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
