#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS figures of the library's kernels from the compiler's own metadata (no GPU needed).

    python3 tools/kernel_regs.py [pattern]        # default pattern: rec2c

Compiles csrc/kernels.hip device-only to assembly (hipcc --cuda-device-only -S) and prints one line per kernel whose
demangled name contains `pattern`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "rec2c"
src = sys.argv[2] if len(sys.argv) > 2 else "kernels.hip"
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
                           "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "mpas-ocean.jl_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    txt = open(out).read()
rows = []
for b in txt.split("- .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    name = g("name")
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except FileNotFoundError:
        dem = name
    if pat in dem:
        rows.append((dem.split("(")[0], int(g("vgpr_count")), int(g("sgpr_count")), int(g("vgpr_spill_count")),
                     int(g("group_segment_fixed_size"))))
for r in sorted(rows):
    print(f"{r[0]:75s} vgpr {r[1]:4d}  sgpr {r[2]:4d}  spill {r[3]:4d}  static LDS {r[4]}")
