#!/usr/bin/env python3
"""bench.py -- cell-updates/s per RK4 step + achieved HBM GB/s of the fused stage kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one full 4-stage RK4 step of the shallow-water dycore over the whole synthetic mesh
(BASELINE.json metric).  Default workload = BASELINE config 4: ~1 M-cell (1 024 002) icosahedral
sphere x 60 layers, fp64, inputs resident in HBM before the timed region starts.
For N > 1 the driver launches this file under torch.distributed.run (one rank per GPU); the mesh is
partitioned across ranks (strong scaling) and halos are exchanged over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import datetime as dt
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL between processes needs on this driver stack

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "mpas-ocean.jl_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # measured float4 copy ceiling, same guide

WORKLOADS = {
    # name: (icosahedral frequency m, layers[, state bytes, Schmidt stretch])  -- SURVEY.md section 8 size table
    "config4_1M_x60": (320, 60),
    "config3_41k_x60": (64, 60),
    "exp_1M_x64": (320, 64),          # experiment: 512-byte rows (cache-line aligned)
    "config2_41k_x1": (64, 1),
    "small_10k_x60": (32, 60),
    # BASELINE config 5: variable-resolution (~3-52 km) 3 696 642-cell sphere x 80 layers, fp32 state / fp64 arithmetic
    "config5_3.7M_x80_f32": (608, 80, 4, 4.47),
    "small_41k_x80_f32": (64, 80, 4, 4.47),
}


def algorithmic_bytes(nC, nE, K, S=8, I=4):
    """the one formula for algorithmic bytes (the contract formula: 5 streams per RK stage 1-3,
    3 for stage 4 = 18 per step)."""
    nnzEE = 10 * nE - 60
    nnzEC = 2 * nE
    b_mesh = nE * (2 * I + 3 * 8 + 2 * I) + nnzEE * (I + 8) + nnzEC * 2 * I + nC * (I + 8 + 8)
    b_tend = 2 * S * K * (nE + nC) + b_mesh
    b_step = 18 * S * K * (nE + nC) + 4 * b_mesh
    return b_mesh, b_tend, b_step


# State streams (one stream = S*K*(nE+nC) bytes) the stage-s launch has to move AT LEAST, aliasing taken into account:
# stage 1: Provis == Curr == the current level and New starts from it -> read 1, write Provis' and New = 3 (the contract
# formula counts 5); stages 2, 3: read Provis, Curr, New, write Provis', New' = 5; stage 4: read Provis, New, write New = 3.
STAGE_MIN_STREAMS = (3, 5, 5, 3)
STAGE_CONTRACT_STREAMS = (5, 5, 5, 3)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def get_mesh(m, stretch=1.0):
    from moka_hip import meshgen as mg
    t0 = time.time()
    mesh = mg.icosahedral_mesh(m, stretch=stretch)
    log(f"[bench] mesh m={m}: {mesh.nCells} cells, {mesh.nEdges} edges built in {time.time() - t0:.1f}s")
    return mesh


def host_cores():
    """Threads this process may really use: affinity mask, cgroup quota, and the 16-core share a one-GPU box
    is given (oversubscribing the 256 hardware threads it *shows* made the OpenMP leg slower than 1 thread)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MOKA_BENCH_THREADS", "16"))))


def cpu_baseline(mesh, K, ssh, u, h, rest, dts, budget_s=25.0, mixed=False):
    """Oracle (C restatement of the reference loop nests) timed on this box's host cores: clean RK4 step,
    all cores (OpenMP), on the same mesh when one step fits the budget, else on a smaller sphere."""
    import oracle as orc
    cores = host_cores()
    orc.set_threads(cores)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h, mixed=mixed)
    t0 = time.time()
    st.step_rk4(dts)                   # warm-up (page faults) and a size probe
    t_probe = time.time() - t0
    n = max(1, min(5, int(budget_s / max(t_probe, 1e-3)) - 1))
    if t_probe > budget_s:
        n = 0
    t0 = time.time()
    for _ in range(n):
        st.step_rk4(dts)
    t = (time.time() - t0) / n if n else t_probe
    out = {"value": mesh.nCells * K / t, "unit": "cell-updates/s", "cores": cores, "kind": "port",
           "sample": f"{max(n, 1)} RK4 step(s) on {mesh.nCells} cells x {K} layers, "
                     f"oracle/moka_oracle.c with OpenMP on {cores} host threads (the box's CPU share for one GPU)",
           "ms_per_step": t * 1e3}
    if not mixed:
        # the reference's live path: reference_compat Forward-Euler steps (time_integration.jl:150-193), same threads
        fe = orc.OracleState(om, ssh, u, h)
        flags = orc.FE_REFERENCE_COMPAT if K == 1 else (orc.FE_REFERENCE_COMPAT & ~4)
        fe.step_fe(dts, flags)
        t0 = time.time()
        nf = 0
        while nf < 8 and time.time() - t0 < budget_s / 3:
            fe.step_fe(dts, flags)
            nf += 1
        out["forward_euler_compat"] = {"value": mesh.nCells * K / ((time.time() - t0) / nf), "unit": "cell-updates/s",
                                       "sample": f"{nf} reference_compat FE steps, same mesh and threads"}
    orc.set_threads(1)
    return out


def cpu_baseline_1t(mesh, K, ssh, u, h, rest, dts, budget_s=12.0, mixed=False):
    """Single-thread figure (what `julia mpas_ocean.jl` gives with JULIA_NUM_THREADS=1) on the SAME mesh as the GPU line:
    one warm-up step is the size probe; then as many steps as fit the budget (at least one)."""
    import oracle as orc
    orc.set_threads(1)
    om = orc.OracleMesh(mesh, K, resting_thickness_sum=rest.sum(1), max_level_edge_top=K)
    st = orc.OracleState(om, ssh, u, h, mixed=mixed)
    t0 = time.time()
    st.step_rk4(dts)
    t_probe = time.time() - t0
    n = max(1, min(5, int(budget_s / max(t_probe, 1e-3))))
    t0 = time.time()
    for _ in range(n):
        st.step_rk4(dts)
    t = (time.time() - t0) / n
    return {"value": mesh.nCells * K / t, "unit": "cell-updates/s", "cores": 1, "kind": "port",
            "sample": f"{n} RK4 step(s) on {mesh.nCells} cells x {K} layers (the GPU line's mesh), 1 thread",
            "ms_per_step": t * 1e3}


def rccl_probe_child(backend="nccl"):
    """Child process of probe_rccl: bring RCCL up between the ranks and move a few bytes the two ways the halo transports
    do (all_to_all_single with uneven splits, batched isend / irecv).  Exit code 0 = it works on this node.
    (backend = "gloo" runs the same exchange on CPU tensors: the rendezvous logic can then be tested without a GPU.)"""
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        ndev = max(torch.cuda.device_count(), 1)
        if ndev < world:
            print(f"[rccl-probe] {world} ranks on {ndev} device(s): RCCL needs one device per rank", file=sys.stderr)
            sys.exit(3)
        torch.cuda.set_device(local % ndev)
        dev = torch.device("cuda", local % ndev)
        dist.init_process_group("nccl", device_id=dev, timeout=dt.timedelta(seconds=60))
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", timeout=dt.timedelta(seconds=60))
    ins = [(rank + q) % 3 + 1 if q != rank else 0 for q in range(world)]
    outs = [(q + rank) % 3 + 1 if q != rank else 0 for q in range(world)]
    send = torch.full((sum(ins),), float(rank), device=dev, dtype=torch.float64)
    recv = torch.zeros(sum(outs), device=dev, dtype=torch.float64)
    dist.all_to_all_single(recv, send, outs, ins)
    exp = torch.cat([torch.full((outs[q],), float(q), dtype=torch.float64) for q in range(world)])
    ok = torch.equal(recv.cpu(), exp)
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    a, b = torch.full((5,), float(rank), device=dev), torch.zeros(5, device=dev)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, prv), dist.P2POp(dist.isend, a, nxt)]):
        w.wait()
    if backend == "nccl":
        torch.cuda.synchronize()
    ok = ok and bool((b.cpu() == float(prv)).all())
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 4)


def probe_rccl(timeout_s=120, backend="nccl"):
    """Does RCCL work between the ranks of this launch?  Answered by a CHILD process per rank (its own rendezvous port),
    started before this process has touched the GPU: a wedged RCCL collective does not raise, it hangs until a watchdog
    kills the process -- so the risk is taken by a process whose only job is to take it.  True = every step ran."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 23)
    env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
    # under torch.distributed.run the ranks are told to use the AGENT's store at MASTER_PORT; the children rendezvous among
    # themselves on another port, where child rank 0 has to host the store
    for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
        env.pop(k)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-probe", backend], env=env, timeout=timeout_s,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            log(f"[bench] RCCL probe failed (exit {r.returncode}): {r.stdout.strip().splitlines()[-1:] }")
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        log(f"[bench] RCCL probe did not finish within {timeout_s} s (killed)")
        return False


def main():
    if "--rccl-probe" in sys.argv:
        rccl_probe_child(sys.argv[sys.argv.index("--rccl-probe") + 1] if len(sys.argv) > sys.argv.index("--rccl-probe") + 1 else "nccl")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config4_1M_x60", choices=list(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (moka_set_kernel_variant): 0 auto, 11 default stage kernel, 4 column kernel, 3 generic index kernel")
    ap.add_argument("--patch-cells", type=int, default=0)
    ap.add_argument("--ordering", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--tend-iters", type=int, default=20)
    ap.add_argument("--transport", default="auto", choices=["auto", "ipc", "nccl", "nccl-a2a", "nccl-p2p", "nccl-default-stream", "gloo"],
                    help="halo transport for N > 1: auto = the fastest of those that qualify on this node -- ipc (direct stores "
                         "into the neighbours' IPC-mapped fields over xGMI) and the RCCL forms (nccl-a2a, nccl-p2p); nccl = the RCCL "
                         "forms only; gloo = host-staged (rehearsal on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    import torch
    import moka_hip as mk
    from moka_hip import meshgen as mg

    ndev = max(torch.cuda.device_count(), 1)            # counting devices does not initialise the GPU
    device_index = local_rank % ndev            # rehearsals may put several ranks on one GPU (ipc and gloo transports)
    gloo_group = None
    rccl_ok = False
    if world > 1:
        import torch.distributed as dist
        want_rccl = args.transport in ("auto", "nccl", "nccl-a2a", "nccl-p2p", "nccl-default-stream")
        if want_rccl and ndev >= world:
            rccl_ok = probe_rccl()               # in a child process, before this one touches the GPU
        torch.cuda.set_device(device_index)
        if rccl_ok:
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index),
                                        timeout=dt.timedelta(seconds=300))
                gloo_group = dist.new_group(backend="gloo")     # control messages + last-resort halo transport
            except Exception as exc:                 # noqa: BLE001
                log(f"[bench] rank {rank}: RCCL process group unavailable ({str(exc).splitlines()[0]})")
                rccl_ok = False
                try:
                    dist.destroy_process_group()
                except Exception:                    # noqa: BLE001
                    pass
        if not rccl_ok:
            dist.init_process_group("gloo")
            if args.transport not in ("auto", "ipc", "gloo"):
                log(f"[bench] rank {rank}: RCCL is not usable here; --transport {args.transport} falls back to auto (ipc, gloo)")
                args.transport = "auto"
        # every rank must have reached the same verdict about RCCL
        t = torch.tensor([1.0 if rccl_ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=gloo_group)
        if rccl_ok and float(t[0]) != 1.0:
            log(f"[bench] rank {rank}: another rank has no RCCL: this launch cannot continue consistently")
            sys.exit(5)
    m, K, sbytes, stretch = (tuple(WORKLOADS[args.workload]) + (8, 1.0))[:4]
    mesh = get_mesh(m, stretch)
    ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
    cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
           "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}

    backend = mk.MokaHIP(device_index)
    if args.variant:
        backend.set_kernel_variant(args.variant)

    transport_trials = {}
    if world > 1:
        from moka_hip import parallel as mp
        t0 = time.time()
        model = mp.DistributedModel(mesh, ssh, u, h, rest, dts, backend, rank, world, ordering=args.ordering,
                                    patch_cells=args.patch_cells, transport="gloo", group=gloo_group,
                                    state_bytes=sbytes)
        log(f"[bench] rank {rank}: partition + local plan + upload: {time.time() - t0:.1f}s  {model.info()}")
        # Choose the halo transport on this node.  Every candidate must, on every rank, (1) set up, (2) deliver exactly the
        # bytes the host-staged gloo exchange delivers for the same state, (3) run a step; the ranks agree after each phase
        # over the gloo group so nobody is left waiting in a collective.  The fastest qualifying candidate is kept.
        rccl_forms = ("nccl-a2a", "nccl") if rccl_ok else ()
        cands = {"auto": ("ipc",) + rccl_forms, "nccl": rccl_forms, "ipc": ("ipc",), "nccl-a2a": rccl_forms[:1],
                 "nccl-p2p": rccl_forms[1:], "nccl-default-stream": (), "gloo": ()}[args.transport]
        fallbacks = (("nccl-default-stream",) if rccl_ok else ()) + ("gloo",)
        model.transport = "gloo"
        model.exchange_state()
        cand, times = mp.choose_transport(model, cands, fallbacks, gloo_group, lambda msg: log(f"[bench] rank {rank}: {msg}"))
        model.transport = cand
        # the trials advanced the state: start the timed run from the initial state again
        lm = model.lm
        for f, a in ((model.Prog.ssh, ssh[lm.cells_g]), (model.Prog.normalVelocity, u[lm.edges_g]),
                     (model.Prog.layerThickness, h[lm.cells_g])):
            f[0].set(a); f[-1].set(a)
        model.exchange_state()
        args.transport = {"nccl": "nccl-p2p"}.get(cand, cand)
        transport_trials = times
        step = model.step_rk4
        sync = backend.synchronize
        info = model.info()
    else:
        t0 = time.time()
        Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, backend, multilayer=True,
                                                           ordering=args.ordering, patch_cells=args.patch_cells,
                                                           state_bytes=sbytes)
        log(f"[bench] plan + upload: {time.time() - t0:.1f}s")
        info = Setup.mesh.info()
        step = lambda: mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4)  # noqa: E731
        sync = backend.synchronize

    def barrier():
        if world > 1:
            dist.barrier(group=gloo_group)

    import gc
    for _ in range(args.warmup):
        step()
    gc.collect()
    gc.disable()          # a 24 ms host-side pause was seen once inside a 136 ms timed region: the collector is the one
                          # source of such pauses this process controls
    sync(); torch.cuda.synchronize(); barrier()
    backend.timer_start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ev_ms = backend.timer_stop()
    sync(); torch.cuda.synchronize(); barrier()
    t1 = time.perf_counter()
    gc.enable()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=gloo_group)
        elapsed, ev_ms = float(tt[0]), float(tt[1])
    ms_per_step = elapsed / args.steps * 1e3
    value = mesh.nCells * K / (elapsed / args.steps)

    b_mesh, b_tend, b_step = algorithmic_bytes(mesh.nCells, mesh.nEdges, K, S=sbytes)
    stream_bytes = sbytes * K * (mesh.nEdges + mesh.nCells)
    # dominant kernel = the fused RK-stage kernel (k_stage_rec2c): 4 launches per step.  Average launch duration from HIP
    # events on the library's compute stream over the timed region (the four launches of a step run back to back on that
    # stream, nothing else does); algorithmic bytes per launch = B_step / 4 (contract formula).
    launches = 4 * args.steps
    avg_launch_ms = ev_ms / launches
    per_rank_bytes = b_step / 4 / world
    achieved = per_rank_bytes / (avg_launch_ms * 1e-3) / 1e9
    # the same with the bytes the four launches really have to move (stage 1 aliases Provis = Curr = New): 16 streams
    min_step_bytes = (sum(STAGE_MIN_STREAMS) * stream_bytes + 4 * b_mesh) / world
    achieved_min = min_step_bytes / 4 / (avg_launch_ms * 1e-3) / 1e9
    # HBM traffic per launch from PMC counters: measured by tools/profile.sh in separate rocprofv3 passes (a process cannot
    # read them about itself), so it is quoted only when that profile was taken with THIS configuration, with its source
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile) and world == 1:
        try:
            rec = json.load(open(tfile)).get(args.workload, {})
            same = all(rec.get("config", {}).get(k) == v for k, v in
                       (("patch_cells", info.get("patch_cells")), ("ordering", info.get("ordering")),
                        ("kernel_variant", args.variant)))
            if same and rec.get("stage_bytes_per_launch"):
                traffic = rec["stage_bytes_per_launch"]
                traffic_source = {"file": "profiles/pmc_traffic.json", "profiled_at_commit": rec.get("commit"),
                                  "note": "FETCH_SIZE x2 (gfx950 rule) + WRITE_SIZE, separate --pmc passes; not measured in this run"}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "k_stage_rec2c (fused TRiSK tendency + RK4 stage update), mean of the 4 launches of a step",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": per_rank_bytes, "avg_launch_ms": avg_launch_ms,
                "launches_timed": launches, "formula": "contract: 18 state streams per step (5, 5, 5, 3) + 4 B_mesh",
                "frac_kernel_minimum_bytes": achieved_min / HBM_PEAK_GBS,
                "kernel_minimum_note": "16 state streams per step (3, 5, 5, 3): stage 1 aliases Provis = Curr = New"}
    if world == 1:
        # per-stage launch durations: a second pass of the same steps with one HIP event between the launches
        nrec = max(5, min(args.steps, 20))
        backend.stage_timing(True)
        for _ in range(nrec):
            step()
        ms4, nst = backend.stage_timing_read()
        backend.stage_timing(False)
        per_stage = []
        for sidx, ms in enumerate(ms4):
            bc = STAGE_CONTRACT_STREAMS[sidx] * stream_bytes + b_mesh
            bm = STAGE_MIN_STREAMS[sidx] * stream_bytes + b_mesh
            per_stage.append({"stage": sidx + 1, "kernel_mode": (1, 2, 2, 3)[sidx], "ms": ms,
                              "bytes_contract": bc, "frac_contract": bc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "bytes_minimum": bm, "frac_minimum": bm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        roofline["per_stage"] = per_stage
        roofline["per_stage_steps"] = nst
        roofline["per_stage_sum_ms"] = sum(ms4)

    out = {"metric": "cell-updates/sec per RK4 step", "value": value, "unit": "cell-updates/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64" if sbytes == 8 else "f64 arithmetic on f32-stored state", "data": "synthetic",
           "config": {"workload": args.workload,
                      "mesh": f"icosahedral m={m}" + (f", Schmidt stretch {stretch}" if stretch != 1.0 else ""), "nCells": mesh.nCells,
                      "nEdges": mesh.nEdges, "nVertLevels": K, "integrator": "RK4", "dt_s": dts,
                      "ordering": info.get("ordering"), "patch_cells": info.get("patch_cells"),
                      "kernel_variant": args.variant,
                      "parallelism": "single GPU" if world == 1 else
                      f"mesh partitioned over {world} GPUs (RCB), 1-deep halo exchanged per RK stage over {args.transport}, "
                      f"overlapped with the interior patches; halo {info.get('halo_bytes_per_stage', 0) / 1e6:.1f} MB/stage/rank",
                      **({"halo_transport": args.transport, "halo_transport_trials_ms_per_step": transport_trials,
                          "rccl_usable": rccl_ok} if world > 1 else {})},
           "roofline": roofline}

    if world == 1:
        # the pure tendency kernel (north_star's 40 % target is quoted on it): B_tend / t
        for _ in range(3):
            mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
        backend.synchronize()
        backend.timer_start()
        for _ in range(args.tend_iters):
            mk.computeTendency(Setup.mesh, Diag, Prog, Tend)
        tms = backend.timer_stop() / args.tend_iters
        out["tendency_kernel"] = {"avg_launch_ms": tms, "algorithmic_bytes": b_tend,
                                  "achieved_GBs": b_tend / (tms * 1e-3) / 1e9,
                                  "frac_of_peak": b_tend / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if True:
            # the reference's live integrator: reference_compat Forward-Euler step (one fused launch), for the record
            fe_flags = mk.REFERENCE_COMPAT if K == 1 else (mk.REFERENCE_COMPAT & ~4)
            if sbytes == 4:     # an fp32-storage state has no DiagnosticVars to carry over right after RK4 steps
                mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=0)
            for _ in range(2):
                mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=fe_flags)
            backend.synchronize()
            backend.timer_start()
            for _ in range(args.tend_iters):
                mk.ocn_timestep(dts, Prog, Diag, Tend, Setup, mk.ForwardEuler, flags=fe_flags)
            fms = backend.timer_stop() / args.tend_iters
            out["forward_euler_compat"] = {"ms_per_step": fms, "value": mesh.nCells * K / (fms * 1e-3),
                                           "unit": "cell-updates/s", "note": "moka_step_fe, reference_compat flags, all levels"}
    if rank == 0 and world == 1 and not args.no_cpu:       # the CPU leg belongs to the N = 1 line only
        t0 = time.time()
        mixed = sbytes == 4
        if mesh.nCells * K > 1.5e8:        # bounded sample: the same workload family on a quarter of the cells
            from moka_hip import meshgen as mg
            cm = get_mesh(m // 2, stretch)
            cssh, cu, ch, crest, cdts = mg.sphere_synthetic_state(cm, K)
            out["cpu_baseline"] = cpu_baseline(cm, K, cssh, cu, ch, crest, cdts, mixed=mixed)
            out["cpu_baseline_1t"] = cpu_baseline_1t(cm, K, cssh, cu, ch, crest, cdts, mixed=mixed)
        else:
            out["cpu_baseline"] = cpu_baseline(mesh, K, ssh, u, h, rest, dts, mixed=mixed)
            out["cpu_baseline_1t"] = cpu_baseline_1t(mesh, K, ssh, u, h, rest, dts, mixed=mixed)
        out["cpu_baseline"]["host"] = _cpu_model()
        log(f"[bench] cpu baseline legs: {time.time() - t0:.1f}s")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(group=gloo_group)
        model.close()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return f"{line.split(':', 1)[1].strip()} x{os.cpu_count()}"
    except Exception:
        pass
    return f"unknown x{os.cpu_count()}"


if __name__ == "__main__":
    main()
