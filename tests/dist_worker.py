"""Worker for the multi-rank tests (launched under torch.distributed.run).

mode cpu : validates the partition, the local meshes and the exchange lists with the CPU ORACLE as each rank's
           compute (gloo, no GPU): the distributed RK4 -- and the reference's Forward-Euler step with its stale
           layerThicknessEdge, which needs no exchange of its own -- must reproduce the single-domain oracle bit for bit.
mode gpu : the same check for the HIP path (DistributedModel); ranks share GPU 0.  argv: K variant transport, transport =
           gloo (buffered, host-staged) or ipc (direct: IPC-mapped fields, flag words in shared memory -- the ranks are
           separate processes exactly as under bench.py, only the GPU is shared).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle as orc  # noqa: E402
from moka_hip import lib as L  # noqa: E402
from moka_hip import meshgen as mg  # noqa: E402
from moka_hip import parallel as par  # noqa: E402


def exchange_numpy(lm, K, fields, dist):
    """fields = (ssh, u, h) local arrays; halo rows are overwritten with the owners' values (gloo).
    Same buffer layout and message slices as the device path (par.pack_numpy / message_slices)."""
    ssh, u, h = fields
    sendbuf = torch.from_numpy(par.pack_numpy(lm, K, ssh, u, h))
    recvbuf = torch.zeros(lm.recv_cells.size * (K + 1) + lm.recv_edges.size * K, dtype=torch.float64)
    reqs = [dist.irecv(recvbuf[a:b], q) for q, a, b in par.message_slices(lm, K, False) if b > a]
    reqs += [dist.isend(sendbuf[a:b].contiguous(), q) for q, a, b in par.message_slices(lm, K, True) if b > a]
    for w in reqs:
        w.wait()
    par.unpack_numpy(lm, K, recvbuf.numpy(), ssh, u, h)


def exchange_numpy_a2a(lm, K, fields, dist):
    """The same exchange as one all_to_all_single with the split sizes of the "nccl-a2a" transport."""
    ssh, u, h = fields
    ins, outs = par.alltoall_splits(lm, K)
    sendbuf = torch.from_numpy(par.pack_numpy(lm, K, ssh, u, h))
    recvbuf = torch.zeros(sum(outs), dtype=torch.float64)
    assert sendbuf.numel() == sum(ins)
    dist.all_to_all_single(recvbuf, sendbuf, outs, ins)
    par.unpack_numpy(lm, K, recvbuf.numpy(), ssh, u, h)


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    K, nsteps, dt = (4, 3, 30.0) if mode == "cpu" else (int(sys.argv[2]) if len(sys.argv) > 2 else 60, 3, 20.0)
    mesh = mg.icosahedral_mesh(8 if mode == "cpu" else 16)
    rng = np.random.default_rng(5)
    rest = np.full((mesh.nCells, K), 1000.0 / K) + rng.uniform(0, 0.1, (mesh.nCells, K))
    h = rest + rng.uniform(-1, 1, (mesh.nCells, K))
    u = rng.uniform(-1, 1, (mesh.nEdges, K))
    ssh = h.sum(1) - rest.sum(1)
    # single-domain oracle
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    ref = orc.OracleState(om, ssh, u, h)
    for _ in range(nsteps):
        ref.step_rk4(dt)

    part = par.partition_cells(mesh, world)
    lm = par.build_local(mesh, part, rank, world)
    assert world > 3 or set(lm.neighbors) == set(range(world)) - {rank}
    cm, em = lm.owned_cell_mask, lm.owned_edge_mask
    # every cell is owned exactly once, every edge exactly once
    owned_c = torch.zeros(mesh.nCells, dtype=torch.int32); owned_c[lm.cells_g[cm]] = 1
    owned_e = torch.zeros(mesh.nEdges, dtype=torch.int32); owned_e[lm.edges_g[em]] = 1
    dist.all_reduce(owned_c); dist.all_reduce(owned_e)
    assert int(owned_c.min()) == 1 and int(owned_c.max()) == 1 and int(owned_e.min()) == 1 and int(owned_e.max()) == 1

    if mode == "cpu":
        oml = orc.OracleMesh(lm.mesh, K, resting_thickness_sum=rest.sum(1)[lm.cells_g], max_level_edge_top=K)
        # local tendency of entities with an owned cell == global tendency, bit for bit
        tu_g, th_g, _ = om.tendencies_clean(u, h)
        tu_l, th_l, _ = oml.tendencies_clean(u[lm.edges_g], h[lm.cells_g])
        assert np.array_equal(th_l[cm], th_g[lm.cells_g[cm]]) and np.array_equal(tu_l[em], tu_g[lm.edges_g[em]])
        # distributed RK4 (time_integration.jl:61-148) with a halo exchange after every stage
        cu, ch, cssh = u[lm.edges_g].copy(), h[lm.cells_g].copy(), ssh[lm.cells_g].copy()
        rs = rest.sum(1)[lm.cells_g]
        for _ in range(nsteps):
            a, b = [dt / 2., dt / 2., dt], [dt / 6., dt / 3., dt / 3., dt / 6.]
            pu, ph = cu.copy(), ch.copy()
            nu, nh = cu.copy(), ch.copy()
            for s in range(4):
                tu, th, _ = oml.tendencies_clean(pu, ph)
                if s < 3:
                    pu, ph = cu + a[s] * tu, ch + a[s] * th
                    pssh = oml.update_ssh(ph)
                    (exchange_numpy if s % 2 else exchange_numpy_a2a)(lm, K, (pssh, pu, ph), dist)   # both forms
                nu, nh = nu + b[s] * tu, nh + b[s] * th
            cssh = oml.update_ssh(nh)
            exchange_numpy(lm, K, (cssh, nu, nh), dist)
            cu, ch = nu, nh
        got = (cssh, cu, ch)
        # the reference's Forward-Euler step on the partition (time_integration.jl:150-193), quirks included: only the
        # new level is exchanged; the carried layerThicknessEdge of every edge with an owned cell is local
        for flags in (3, 0):
            gref = orc.OracleState(om, ssh, u, h)
            loc = orc.OracleState(oml, ssh[lm.cells_g], u[lm.edges_g], h[lm.cells_g])
            vm = lm.owned_vert_mask
            for _ in range(4):
                gref.step_fe(dt, flags)
                loc.step_fe(dt, flags)
                exchange_numpy(lm, K, (loc.ssh[1], loc.u[1], loc.h[1]), dist)
                for name, mask, ids in (("u", em, lm.edges_g), ("h", cm, lm.cells_g), ("ssh", cm, lm.cells_g)):
                    assert np.array_equal(getattr(loc, name)[1][mask], getattr(gref, name)[1][ids[mask]]), (flags, name)
                for name, mask, ids in (("hEdge", em, lm.edges_g), ("F", em, lm.edges_g), ("tendU", em, lm.edges_g),
                                        ("div", cm, lm.cells_g), ("tendH", cm, lm.cells_g), ("vort", vm, lm.verts_g)):
                    assert np.array_equal(getattr(loc, name)[mask], getattr(gref, name)[ids[mask]]), (flags, name)
    else:
        import moka_hip as mk
        backend = mk.MokaHIP(0)
        variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        transport = sys.argv[4] if len(sys.argv) > 4 else "gloo"
        backend.set_kernel_variant(variant)
        model = par.DistributedModel(mesh, ssh, u, h, rest, dt, backend, rank, world, transport=transport, part=part)
        assert model.p_boundary <= model.p_owned <= model.mesh.info()["nPatches"]
        assert model.direct_available
        # the transport selection bench.py runs before timing (on a model of its own: the trials advance the state):
        # broken candidates are dropped by agreement of the ranks, phase by phase
        probe = par.DistributedModel(mesh, ssh, u, h, rest, dt, backend, rank, world, transport="gloo", part=part)
        probe.exchange_state()
        msgs = []
        assert par.choose_transport(probe, ("bogus", "gloo"), ("gloo",), None, msgs.append, trial_steps=1)[0] == "gloo"
        assert par.choose_transport(probe, ("bogus",), ("bogus2", "gloo"), None, msgs.append)[0] == "gloo"
        assert sum("bogus" in m for m in msgs) == 3, msgs
        try:
            par.choose_transport(probe, (), ("bogus",), None)
            raise AssertionError("no transport should have qualified")
        except RuntimeError:
            pass
        if transport == "ipc":
            # the direct transport qualifies (set-up, byte comparison with gloo, a step) and is kept over gloo when faster
            name, times = par.choose_transport(probe, ("ipc", "gloo"), ("gloo",), None, msgs.append, trial_steps=2)
            assert "ipc" in times and "gloo" in times, (name, times, msgs)
            model.connect_ipc()
            model.exchange_state()
            assert model.verify_transport("gloo")
            # the caller chooses: a CONNECTED model whose transport is "gloo" runs its steps through the buffered callback, four
            # times per RK4 step and once per Forward-Euler step -- the library does not override it with the direct exchange
            # (it did, so the buffered candidates of the selection were never exercised)
            calls = []
            real_transport = model._transport
            model._transport = lambda: (calls.append(1), real_transport())[1]
            before = model.snapshot()
            model.set_transport("gloo")
            model.step_rk4()
            assert len(calls) == 4, calls
            model.step_fe(3 if K > 1 else 7)
            assert len(calls) == 5, calls
            via_gloo = model.snapshot()
            model.restore(before)
            model.set_transport("ipc-acq")          # the direct form with the explicit acquire: same results, no callback
            dist.barrier()
            model.step_rk4()
            model.step_fe(3 if K > 1 else 7)
            assert len(calls) == 5, calls
            assert all(np.array_equal(a, b) for a, b in zip(model.snapshot(), via_gloo)), "ipc-acq vs gloo"
            model._transport = real_transport
            model.set_transport("ipc")
            dist.barrier()
            model.Prog.ssh[-1].set(ssh[lm.cells_g]); model.Prog.layerThickness[-1].set(h[lm.cells_g])
            model.Prog.normalVelocity[-1].set(u[lm.edges_g])
            for f in (model.Prog.ssh, model.Prog.layerThickness, model.Prog.normalVelocity):
                f[0].set(f[-1].get())
            model.exchange_state()
        probe.close()
        for _ in range(nsteps):
            model.step_rk4()
        got = (model.Prog.ssh[-1].get(), model.Prog.normalVelocity[-1].get(), model.Prog.layerThickness[-1].get())
        if K % 2 == 0 or K == 1:
            # the reference's Forward-Euler step on the partition, continuing from the RK4 state
            flags = 7 if K == 1 else 3
            for _ in range(3):
                model.step_fe(flags)
                ref.step_fe(dt, flags)
            diag = model.owned_diagnostics()
            for name, exp in (("hEdge", ref.hEdge), ("F", ref.F), ("div", ref.div), ("vort", ref.vort), ("tendU", ref.tendU),
                              ("tendH", ref.tendH)):
                ids, vals = diag[name]
                assert np.array_equal(vals, exp[ids]), name
            got = (model.Prog.ssh[-1].get(), model.Prog.normalVelocity[-1].get(), model.Prog.layerThickness[-1].get())
        if K % 2 == 0 and transport != "nccl":
            # reverse mode between processes: two taped RK4 steps from the state reached so far, d sum(ssh^2) / d that state;
            # every rank's owned rows against the single-domain oracle adjoint
            model.exchange_state()
            st2 = orc.OracleState(om, ref.ssh[1], ref.u[1], ref.h[1])
            adj = orc.OracleAdjointRK4(st2)
            model.tape(2)
            for _ in range(2):
                model.step_rk4_taped()
                adj.step_rk4(dt)
            gU, gH = adj.gradient_sum_sq_ssh()
            (cg, gh), (eg, gu) = model.adjoint_gradient(2)
            assert np.array_equal(gh, gH[cg]) and np.array_equal(gu, gU[eg]), "partitioned reverse mode"
            assert np.abs(gH).max() > 0
            for _ in range(2):
                ref.step_rk4(dt)
            # and of a Forward-Euler run with the reference's stale layerThicknessEdge and accumulating vorticity
            st3 = orc.OracleState(om, ref.ssh[1], ref.u[1], ref.h[1])
            st3.hEdge[...] = 0.0; ref.hEdge[...] = 0.0; ref.vort[...] = 0.0
            model.Diag.layerThicknessEdge.set(np.zeros((lm.mesh.nEdges, K)))
            model.Diag.relativeVorticity.set(np.zeros((lm.mesh.nVertices, K)))
            afe = orc.OracleAdjoint(st3)
            model._tape.close()
            model.tape(2)
            for _ in range(2):
                model.step_fe_taped(3)
                afe.step_fe(dt, 3)
                ref.step_fe(dt, 3)
            gS, gU, gH, gE = afe.gradient_sum_sq_ssh()
            g = model.adjoint_gradient_fe(2)
            for name, exp in (("ssh", gS), ("normalVelocity", np.asarray(gU).reshape(mesh.nEdges, K)),
                              ("layerThickness", np.asarray(gH).reshape(mesh.nCells, K)),
                              ("layerThicknessEdge", np.asarray(gE).reshape(mesh.nEdges, K))):
                ids, rows = g[name]
                assert np.array_equal(rows, exp[ids]), "partitioned reverse mode (FE) " + name
            got = (model.Prog.ssh[-1].get(), model.Prog.normalVelocity[-1].get(), model.Prog.layerThickness[-1].get())
        if K % 2 == 0 and transport != "nccl":
            # the optional nonlinear terms between processes: a model of its own (two-ring halo, whole-mesh stages)
            nlm = par.DistributedModel(mesh, ssh, u, h, rest, dt, backend, rank, world, transport="gloo", part=part, nonlinear=True)
            nlm.exchange_state()
            onl, stn = orc.OracleNonlinear(om), orc.OracleState(om, ssh, u, h)
            for i in range(3):
                # one library call per step (stage kernel over boundary / interior patches, the interior's preparation pass under the
                # exchange) and the plain form (every stage one launch over the whole local mesh, exchange behind it), alternately
                parts = bool(L.lib().moka_rk4_dist_parts_available(nlm._halo))       # per-patch kernels: even 34 <= K <= 64
                (nlm.step_rk4 if parts and i != 1 else nlm.step_rk4_whole)()
                onl.step_rk4(stn, dt)
            (cg, s_, hh), (eg, uu) = nlm.owned_state()
            assert np.array_equal(hh, stn.h[1][cg]) and np.array_equal(uu, stn.u[1][eg]) and np.array_equal(s_, stn.ssh[1][cg]), "nonlinear"
            if transport == "ipc" and parts:
                # ... and over the direct transport: the neighbours' push kernels store into this rank's IPC-mapped fields while its
                # interior stage kernel and the next stage's interior preparation pass run
                nlm.connect_ipc()
                nlm.set_transport("ipc")
                dist.barrier()
                for _ in range(2):
                    nlm.step_rk4()
                    onl.step_rk4(stn, dt)
                (cg, s_, hh), (eg, uu) = nlm.owned_state()
                assert np.array_equal(hh, stn.h[1][cg]) and np.array_equal(uu, stn.u[1][eg]) and np.array_equal(s_, stn.ssh[1][cg]), "nonlinear over ipc"
            dist.barrier()
            nlm.close()
        dist.barrier()               # nobody pushes into fields that are about to be freed
        model.close()
    assert np.array_equal(got[0][cm], ref.ssh[1][lm.cells_g[cm]]), "ssh"
    assert np.array_equal(got[2][cm], ref.h[1][lm.cells_g[cm]]), "layerThickness"
    assert np.array_equal(got[1][em], ref.u[1][lm.edges_g[em]]), "normalVelocity"
    # after the final exchange the halo holds the owners' values too
    assert np.array_equal(got[2], ref.h[1][lm.cells_g]) and np.array_equal(got[0], ref.ssh[1][lm.cells_g])
    ok = torch.ones(1)
    dist.all_reduce(ok)
    if rank == 0:
        print(f"dist_worker {mode}: OK on {world} ranks ({mesh.nCells} cells x {K} layers, {nsteps} RK4 steps)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
