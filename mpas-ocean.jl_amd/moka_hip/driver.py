"""`python -m moka_hip.driver config.yml` -- the reference's driver (src/driver/mpas_ocean.jl:20-61) on the MokaHIP
backend: init from the YAML configuration and the MPAS files it names, alarm-driven Forward-Euler run loop, output
file at the end of the simulation."""
from __future__ import annotations

import os
import sys

import numpy as np

from . import api as mk
from .timemanager import period_seconds


def ocn_run(config_fp, device: int = 0, method=mk.ForwardEuler):
    print("Setting the backend...")
    backend = mk.MokaHIP(device)                                       # mpas_ocean.jl:28 (was CUDABackend())
    Setup, Diag, Tend, Prog = mk.ocn_init(config_fp, backend=backend)
    print("Initialized the model")
    clock, simulationAlarm, outputAlarm = mk.ocn_init_alarms(Setup)
    print("Initialized the clock.")
    timestep = np.array([period_seconds(Setup.timeManager.timeStep)])  # KA.zeros(backend, Float64, (1,)); timestep[1] = dt
    mk.ocn_run_loop(timestep, Prog, Diag, Tend, Setup, method, clock, simulationAlarm, outputAlarm)
    out = mk.write_netcdf(Setup, Diag, Prog)                           # i/o only at the end of the simulation (:44)
    print("Moka.jl ran on GPU")
    print(clock.currTime)
    return out


if __name__ == "__main__":
    # python -m moka_hip.driver config.yml [--integrator fe|rk4] [--device N]
    # (the reference's driver takes the config file only and always steps with ForwardEuler, mpas_ocean.jl:40,56-60)
    args = sys.argv[1:]
    opts = {"--integrator": "fe", "--device": "0"}
    while len(args) >= 3 and args[-2] in opts:
        opts[args[-2]] = args[-1]
        args = args[:-2]
    if len(args) == 1 and os.path.isfile(args[0]) and opts["--integrator"] in ("fe", "rk4") and opts["--device"].isdigit():
        ocn_run(args[0], device=int(opts["--device"]), method=mk.RungeKutta4 if opts["--integrator"] == "rk4" else mk.ForwardEuler)
    else:
        raise SystemExit("yaml config file invalid")                   # mpas_ocean.jl:58
