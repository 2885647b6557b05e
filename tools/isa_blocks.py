#!/usr/bin/env python3
"""Basic-block instruction mix of one kernel from hipcc's assembly output (no GPU needed).

    python3 tools/isa_blocks.py k.s <mangled-name-substring> [min_instr]

Prints one line per basic block: label, instruction counts by class (fp64 VALU, conversions, other VALU, SALU, LDS,
VMEM loads / stores, waits), so the hot loops of a kernel can be read off and compared between variants."""
import re
import sys
from collections import Counter, OrderedDict

path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].strip() == "s_endpgm")


def cls(op):
    if op.startswith("v_cvt"):
        return "cvt"
    if re.match(r"v_(fma|mul|add|sub|max|min|div|rcp|trig|fract|ldexp|cmp\w*)_f64", op) or op.endswith("_f64"):
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if re.match(r"(global|buffer|flat|scratch)_(load|atomic)", op):
        return "vld"
    if re.match(r"(global|buffer|flat|scratch)_store", op):
        return "vst"
    return "other"


blocks = OrderedDict()
cur = "entry"
blocks[cur] = Counter()
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        cur = m.group(1)
        blocks[cur] = Counter()
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    op = s.split()[0]
    blocks[cur][cls(op)] += 1
    if op.startswith("s_cbranch") or op == "s_branch":
        blocks[cur]["br:" + s.split()[-1]] += 0
tot = Counter()
for k, c in blocks.items():
    n = sum(v for kk, v in c.items() if not kk.startswith("br:"))
    for kk, v in c.items():
        if not kk.startswith("br:"):
            tot[kk] += v
    if n >= minn:
        tgt = ",".join(kk[3:] for kk in c if kk.startswith("br:"))
        print(f"{k:14s} n={n:4d}  f64={c['f64']:4d} cvt={c['cvt']:3d} valu={c['valu']:4d} salu={c['salu']:3d} lds={c['lds']:3d} "
              f"vld={c['vld']:3d} vst={c['vst']:3d} wait={c['wait']:2d}  -> {tgt}")
print("total", dict(tot))
