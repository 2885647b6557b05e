#!/usr/bin/env python3
"""GPU box helper: what clock and power does the device run at DURING the stage launches?  (bench.py's sysfs reads happen between
timed regions.)  A thread samples pp_dpm_sclk / fclk / mclk and power1_average every few ms while RK4 steps (then tendency launches)
run back to back for ~2 s each.   python3 tools/clock_probe.py [workload=config5_3.7M_x80_f32]"""
import datetime as dt
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import bench                               # noqa: E402
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "config5_3.7M_x80_f32"
m, K, sbytes, stretch = (tuple(bench.WORKLOADS[name]) + (8, 1.0))[:4]
mesh = mg.icosahedral_mesh(m, stretch=stretch)
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
backend = mk.MokaHIP(0)
cfg = {"time_management": {"config_start_time": dt.datetime(1, 1, 1), "config_run_duration": dt.timedelta(hours=1)},
       "time_integration": {"config_dt": dt.timedelta(seconds=dts), "config_number_of_time_levels": 2}}
Setup, Diag, Tend, Prog = mk.ocn_init_from_arrays(mesh, ssh, u, h, rest, cfg, backend, multilayer=True, state_bytes=sbytes)
pci = backend.pci_bus_id()
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        s = bench.device_sysfs(pci)
        samples.append((time.time(), s.get("sclk_mhz"), s.get("fclk_mhz"), s.get("mclk_mhz"), s.get("power_w")))
        time.sleep(0.004)


def phase(tag, fn, seconds=2.0):
    samples.clear()
    stop.clear()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(10):
            fn()
        backend.synchronize()
        n += 10
    el = time.time() - t0
    stop.set()
    th.join()
    a = np.array([[x if x is not None else np.nan for x in s[1:]] for s in samples[len(samples) // 4:]], dtype=float)
    print(f"{tag}: {el / n * 1e3:.3f} ms per call; sclk {np.nanmean(a[:, 0]):.0f} MHz (min {np.nanmin(a[:, 0]):.0f}, max {np.nanmax(a[:, 0]):.0f}), "
          f"fclk {np.nanmean(a[:, 1]):.0f}, mclk {np.nanmean(a[:, 2]):.0f}, power {np.nanmean(a[:, 3]):.0f} W (max {np.nanmax(a[:, 3]):.0f}); {len(a)} samples", flush=True)


phase("idle", lambda: time.sleep(0.01), 1.0)
phase("RK4 steps", lambda: mk.ocn_timestep(Prog, Diag, Tend, Setup, mk.RungeKutta4))
phase("tendency launches", lambda: mk.computeTendency(Setup.mesh, Diag, Prog, Tend))
cal = backend.bw_probe(4 << 30, 3)
phase("copy probe", lambda: backend.bw_probe(4 << 30, 1), 2.0)
print("copy probe:", cal.get("copy_GBs"))
