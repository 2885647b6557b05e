"""bench.py host-side contract: the algorithmic-bytes formula is SURVEY.md section 8(d)'s, sizes are BASELINE's."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_match_survey_numbers():
    nC, nE, K = 1024002, 3072000, 60                      # config 4 (SURVEY section 8 size table)
    b_mesh, b_tend, b_step = bench.algorithmic_bytes(nC, nE, K)
    assert abs(b_mesh / 1e9 - 0.56) < 0.01                # "B_mesh = 0.56 GB"
    assert abs(b_tend / 1e9 - 4.49) < 0.01                # "B_tend = 3.93 + 0.56 = 4.49 GB"
    assert abs(b_step / 1e9 - 37.6) < 0.1                 # "B_step = 37.6 GB"
    assert b_tend == 2 * 8 * K * (nE + nC) + b_mesh and b_step == 18 * 8 * K * (nE + nC) + 4 * b_mesh


def test_default_workload_is_config4():
    m, K = bench.WORKLOADS["config4_1M_x60"]
    assert 10 * m * m + 2 == 1024002 and K == 60
    assert bench.HBM_PEAK_GBS == 8000.0


def test_rccl_probe_rendezvous_in_child_processes():
    """bench.py asks a CHILD process per rank whether RCCL works (a wedged collective hangs rather than raises).  The
    children rendezvous on their own port, outside the launcher's agent store.  Run that logic under torch.distributed.run
    with the probe's exchange on gloo / CPU tensors: two ranks, both must report success."""
    import subprocess
    import sys
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = ("import importlib.util, os, sys; spec = importlib.util.spec_from_file_location('bench', %r); "
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b); "
            "ok = b.probe_rccl(timeout_s=90, backend='gloo'); print('PROBE', os.environ['RANK'], ok); sys.exit(0 if ok else 1)"
            % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "--no-python", sys.executable, "-c", code],
                       capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "PROBE 0 True" in r.stdout and "PROBE 1 True" in r.stdout


def test_sysfs_and_statistics_helpers(tmp_path):
    """The plain-file readers of the calibration record (no GPU involved) and the per-step statistics."""
    assert bench._dpm_current("0: 500Mhz\n1: 2394Mhz *\n2: 2400Mhz\n") == 2394.0
    assert bench._dpm_current("S: 94Mhz *\n0: 500Mhz\n1: 2400Mhz") == 94.0
    assert bench._dpm_current("0: 2000Mhz *") == 2000.0
    assert bench._dpm_current(None) is None and bench._dpm_current("garbage") is None
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "power1_input").write_text("554000000\n")
    assert bench._power_watts(str(tmp_path)) == 554.0
    assert bench._power_watts(str(tmp_path / "nothing")) is None
    d = bench.device_sysfs("ffff:ff:1f.7")                # not a device of this machine: reported, not raised
    assert d["pci_bus_id"] == "ffff:ff:1f.7" and "note" in d
    st = bench.step_stats([7.0, 6.8, 6.9, 9.5, 6.85])
    assert st["median"] == 6.9 and st["min"] == 6.8 and st["max"] == 9.5 and st["n"] == 5
    assert abs(st["mean"] - 7.41) < 1e-12


def _run_bench(args, nproc=1, timeout=420):
    """bench.py as the driver launches it (a child process; torch.distributed.run for N > 1): the JSON line it prints."""
    import json
    import socket
    import subprocess
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_on_one_gpu_small_workload():
    """The driver's contract on a small workload: one JSON line with the metric, the roofline (with its calibration) and the
    CPU baseline; value = cells x layers / median step time."""
    b = _run_bench(["--workload", "small_10k_x60", "--steps", "6", "--warmup", "2", "--no-config5", "--tend-iters", "3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "calibration", "step_ms"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["ranks"] == 1 and b["rehearsal"] is False and b["steps"] == 6 and b["dtype"] == "f64"
    assert b["config"]["workload"] == "small_10k_x60" and b["vs_baseline"] is None and b["higher_is_better"] is True
    nC, K = b["config"]["nCells"], b["config"]["nVertLevels"]
    assert abs(b["value"] - nC * K / (b["step_ms"]["median"] * 1e-3)) <= 1e-6 * b["value"]
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert b["calibration"]["copy_GBs_before"] > 1000 and b["calibration"]["streams5_GBs_before"] > 1000
    assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1 and b["cpu_baseline"]["value"] > 0
    # round 4: medians everywhere a single stray launch could move a mean, and the line says how many strays there were
    assert b["step_ms"]["strays"] >= 0 and r["stray_steps"] == b["step_ms"]["strays"] and r["frac_at_median_step"] > 0
    t = b["tendency_kernel"]
    assert t["avg_launch_ms"] == t["launch_ms"]["median"] and t["launch_ms"]["n"] == 3 and t["launch_ms"]["min"] <= t["avg_launch_ms"] <= t["launch_ms"]["max"]
    for k in ("under_load", "placement", "rk4_13_streams", "forward_euler_compat"):
        assert k in b, k
    assert b["forward_euler_compat"]["arrays_pending_after_a_step"] == 1 and b["rk4_13_streams"]["ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_is_labelled_a_rehearsal():
    """bench.py --gpus 2 the way the driver starts it, with both ranks on the one GPU of this box: the transport selection must
    qualify a transport (bitwise against the host-staged exchange), and the line must say what it is -- one distinct device, two
    ranks, a rehearsal -- instead of claiming two GPUs."""
    b = _run_bench(["--workload", "small_10k_x60", "--steps", "4", "--warmup", "2"], nproc=2)
    assert b["ranks"] == 2 and b["n_gpus"] == 1 and b["n_devices_visible"] == 1 and b["ranks_per_device"] == 2 and b["rehearsal"] is True
    assert b["scaling"] == "strong" and b["config"]["halo_transport"] in ("ipc", "ipc-acq", "gloo", "nccl-a2a", "nccl-p2p")
    assert b["config"]["halo_transport"] in b["config"]["halo_transport_trials_ms_per_step"]
    assert b["value"] > 0 and "REHEARSAL" in b["config"]["parallelism"]
    # the line says where the time of a distributed step goes: per rank, from the library's own statistics
    ex = b["exchange"]
    for k in ("host_wait_ms_per_step", "host_signal_wait_ms_per_step", "push_to_flag_us", "host_step_ms_per_step",
              "boundary_launch_ms_per_step", "interior_launch_ms_per_step", "rank_share_ms_per_step"):
        assert len(ex[k]) == 2 and all(v >= 0.0 for v in ex[k]), k
    assert all(v > 0.0 for v in ex["interior_launch_ms_per_step"]) and ex["max_rank_share_ms"] == max(ex["rank_share_ms_per_step"])
    # (two ranks on ONE device: each rank's launches wait for the other's, so the bound itself means nothing here -- it is there)
    assert ex["t1_whole_mesh_on_rank0_ms"] > 0 and ex["bound_from_share"] > 0 and ex["measured_speedup_vs_t1"] > 0
    if b["config"]["halo_transport"].startswith("ipc"):
        assert all(v > 0.0 for v in ex["host_step_ms_per_step"])
