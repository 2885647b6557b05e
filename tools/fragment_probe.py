#!/usr/bin/env python3
"""GPU box helper: does deliberately FRAGMENTED device memory give the fast placement?  Spacer blocks are allocated, every other one freed, and
the states then have to be pieced together from the holes.   python tools/fragment_probe.py <spacer MiB, 0 = none> [nblocks=600] [nstates=4]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
import torch                               # noqa: E402
import moka_hip as mk                      # noqa: E402
from moka_hip import lib as L              # noqa: E402
from moka_hip import meshgen as mg         # noqa: E402

spacer, nblocks, nstates = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 64), (2, 600), (3, 4)))
mesh = mg.icosahedral_mesh(320)
K = 60
ssh, u, h, rest, dts = mg.sphere_synthetic_state(mesh, K)
b = mk.MokaHIP(0)
lib = L.lib()
hm = mk.HorzMesh(mesh)
vm = mk.VerticalMesh(hm, nVertLevels=K, restingThickness=rest, multilayer=True)
M = mk.Mesh(hm, vm, backend=b)
keep = []
if spacer:
    blocks = [torch.empty(spacer << 20, dtype=torch.uint8, device="cuda:0") for _ in range(nblocks)]
    keep = blocks[0::2]
    del blocks
    torch.cuda.empty_cache()               # the freed half goes back to the driver: holes of `spacer` MiB between the kept blocks
    torch.cuda.synchronize()


def time_state(st):
    for _ in range(3):
        L.check(lib.moka_step_rk4(st._h, dts), b._h)
    b.stage_timing(True)
    b.marks_reset(); b.mark()
    for _ in range(8):
        L.check(lib.moka_step_rk4(st._h, dts), b._h)
        b.mark()
    ms = sorted(b.marks_read())
    st4, _ = b.stage_timing_read()
    b.stage_timing(False)
    return ms[len(ms) // 2], st4


states = []
for i in range(nstates):
    Prog = mk.PrognosticVars(ssh, u, h, 2, M)
    states.append(Prog)
    med, st4 = time_state(Prog._state)
    print(f"spacer {spacer:4d} MiB x {nblocks // 2 if spacer else 0} kept: state {i}: {med:.3f} ms/step  stages " + " ".join(f"{x:.3f}" for x in st4), flush=True)
