"""bench.py host-side contract: the algorithmic-bytes formula is SURVEY.md section 8(d)'s, sizes are BASELINE's."""
import importlib.util
import os

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
