#!/usr/bin/env python3
"""Where can a re-fetched row come from?  Reuse distances of the stage kernel's gathers, from the host plan (no GPU).

VERDICT r02 item 6 asks where the rows a patch gathers from its neighbours (fetched by their owner patch as well) are served
from: the XCD's L2, the Infinity Cache (MALL) or HBM.  gfx950 exposes no memory-side (MALL hit / DRAM) counter to rocprofv3
(`rocprofv3 -L`: the TCC_EA0_* counters stop at the L2 <-> fabric interface; TCC_EA0_RDREQ_DRAM means "destined for the memory
controller as opposed to GMI / IO", it is counted before the Infinity Cache), so this is answered from the launch order the
kernel fixes and the sizes of the two caches:

  * a launch gives XCD x the contiguous patch range [x * chunk, (x + 1) * chunk) in index order (patch_of_block), about 128
    patches of an XCD are resident at a time (32 CUs x 4 workgroups), and all 8 XCDs advance together;
  * a row gathered by patch p and owned by patch q is fetched twice from beyond L2 unless the two touches are close: the
    distance between them, in bytes that pass through the caches in between, is about |p - q| patches x the bytes one patch
    moves -- through ONE L2 (4 MiB) when p and q are on the same XCD, x 8 XCDs through the shared Infinity Cache (256 MiB;
    MI355X_MICROARCH.md: a line stays resident while everything moved between two uses fits in about 256 MiB).

Prints, for every gathered (patch, row) pair of the normalVelocity and layerThickness gathers: the share owned by the patch
itself (LDS / L1), the shares whose owner is within the L2 window, within the Infinity-Cache window, and beyond (HBM again).
The byte windows are UPPER bounds for the L2: about 128 patches of an XCD are in flight at once, at arbitrary phases of their
~19 us lives, while an L2 turns over in ~6 us of its XCD's traffic -- so a second line ("timed L2 estimate") applies a simple
timing model: patch p starts at p / 128 lifetimes, stages its own rows at once, gathers at a phase uniform in [0.1, 1] of its
life; a gather hits L2 when the other touch of the row is less than tau = (L2 size) / (XCD traffic rate) away.  That estimate
is what the measured FETCH_SIZE can be compared with (profiles/pmc_traffic.json): the tendency launch fetches ~1.3 GB of
such rows again through the fabric out of 2.4 GB gathered.

    python3 tools/reuse_distance.py [m=320] [K=60] [P=0 (default)]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mpas-ocean.jl_amd"))
from moka_hip import lib as L            # noqa: E402
from moka_hip import meshgen as mg       # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 320
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sbytes = int(sys.argv[4]) if len(sys.argv) > 4 else 8
mesh = mg.icosahedral_mesh(m)
plan = L.Plan(mesh, K, max_level_edge_top=K, ordering=L.ORDER_DEFAULT, patch_cells=P, state_bytes=sbytes)
info = plan.info
nP = info["nPatches"]
cs, es, vs = plan.patch_ranges()
eoc, coc, eoe = plan.array("eoc"), plan.array("coc"), plan.array("eoe")
ME, ME2 = info["maxEdgesUsed"], info["maxEdges2Used"]
nC, nE = info["nCells"], info["nEdges"]
eoc, coc, eoe = eoc.reshape(nC, ME), coc.reshape(nC, ME), eoe.reshape(nE, ME2)
patch_of_cell = np.repeat(np.arange(nP), np.diff(cs))
patch_of_edge = np.repeat(np.arange(nP), np.diff(es))
rowB = K * sbytes
chunk = (nP + 7) // 8
# argv[5] = "hilbert": PREDICTION for another launch order -- inside every XCD chunk the patches are
# visited along a Hilbert curve through their centroids (chunk projected onto its two principal axes) instead of in the
# RCB order of the plan; everything below then measures index distances in that order.  Nothing on the GPU changes.
order_name = "RCB (the plan's)"
if len(sys.argv) > 5 and sys.argv[5] == "hilbert":
    order_name = "Hilbert inside every XCD chunk"
    cperm = plan.permutation(L.CELL) if hasattr(plan, "permutation") else None
    if cperm is None:
        raise SystemExit("this Plan binding has no permutation(): cannot place patches")
    xyz = np.stack([mesh.xCell, mesh.yCell, mesh.zCell], 1)[cperm]          # library order
    cen = np.add.reduceat(xyz, cs[:-1], axis=0) / np.diff(cs)[:, None]

    def hilbert_index(ix, iy, bits):
        """Hilbert curve index of integer grid points (vectorised xy2d)."""
        d = np.zeros(ix.shape, dtype=np.int64)
        x, y = ix.astype(np.int64).copy(), iy.astype(np.int64).copy()
        sft = bits - 1
        while sft >= 0:
            s_ = 1 << sft
            rx = (x & s_) > 0
            ry = (y & s_) > 0
            d += s_ * s_ * ((3 * rx.astype(np.int64)) ^ ry.astype(np.int64))
            swap = ~ry
            flip = swap & rx
            x = np.where(flip, s_ - 1 - x, x); y = np.where(flip, s_ - 1 - y, y)      # (coordinates taken modulo the current cell)
            x, y = np.where(swap, y, x), np.where(swap, x, y)
            x &= s_ - 1; y &= s_ - 1
            sft -= 1
        return d

    new_pos = np.empty(nP, dtype=np.int64)
    for xcd in range(8):
        a, b = xcd * chunk, min((xcd + 1) * chunk, nP)
        if a >= b:
            continue
        pts = cen[a:b] - cen[a:b].mean(0)
        _, _, vt = np.linalg.svd(pts, full_matrices=False)
        uv = pts @ vt[:2].T
        bits = 10
        g = ((uv - uv.min(0)) / np.maximum(np.ptp(uv, axis=0), 1e-30) * ((1 << bits) - 1)).astype(np.int64)
        hidx = hilbert_index(g[:, 0], g[:, 1], bits)
        new_pos[a:b][np.argsort(hidx, kind="stable")] = np.arange(a, b)
    relabel = new_pos            # old patch index -> position in the Hilbert launch order
else:
    relabel = np.arange(nP)


def distinct_pairs(patch, row, valid):
    """distinct (gathering patch, gathered row) pairs: a patch's repeated gathers of one row hit L1 / L2 within microseconds"""
    key = patch[valid].astype(np.int64) * (max(nC, nE) + 1) + row[valid]
    key = np.unique(key)
    return key // (max(nC, nE) + 1), key % (max(nC, nE) + 1)


# normalVelocity rows: the cell loop gathers eoc (6 per cell), the edge loop eoe (10 per edge)
p1, r1 = distinct_pairs(np.repeat(patch_of_cell, ME), eoc.ravel(), eoc.ravel() >= 0)
p2, r2 = distinct_pairs(np.repeat(patch_of_edge, ME2), eoe.ravel(), eoe.ravel() >= 0)
pu, ru = distinct_pairs(np.concatenate([p1, p2]), np.concatenate([r1, r2]), np.ones(p1.size + p2.size, bool))
# layerThickness rows: the cell loop gathers the cells across (coc)
ph, rh = distinct_pairs(np.repeat(patch_of_cell, ME), coc.ravel(), coc.ravel() >= 0)

# bytes one patch moves through its XCD's L2 (tendency launch; the RK stages move about twice that)
own_rows = np.diff(es).mean() + np.diff(cs).mean()
halo_u = (patch_of_edge[ru] != pu).sum() / nP
halo_h = (patch_of_cell[rh] != ph).sum() / nP
per_patch = {"tendency": (2 * own_rows + halo_u + halo_h) * rowB + 14e3, "rk_stage": (5 * own_rows + halo_u + halo_h) * rowB + 14e3}
L2, MALL = 4 << 20, 256 << 20
print(f"mesh m={m}: {nC} cells, {nE} edges, K={K}, {rowB}-byte rows, P={info['patch_cells']}, {nP} patches, {chunk} per XCD; launch order: {order_name}")
print(f"per patch: {np.diff(es).mean():.1f} own u rows, {np.diff(cs).mean():.1f} own h rows, {halo_u:.1f} u rows and {halo_h:.1f} h rows "
      f"of other patches; it moves {per_patch['tendency'] / 1e3:.0f} KB (tendency launch) / {per_patch['rk_stage'] / 1e3:.0f} KB (RK stages 2-3)")
for launch, bpp in per_patch.items():
    w_l2 = L2 / bpp                 # patches of ONE XCD whose traffic fits its L2
    w_mall = MALL / (8 * bpp)       # patches (index distance on one XCD) whose traffic -- times 8 XCDs -- fits the Infinity Cache
    print(f"\n{launch}: L2 window {w_l2:.0f} patches, Infinity-Cache window {w_mall:.0f} patches of index distance")
    tot_bytes = 0.0
    acc = {"own": 0.0, "l2": 0.0, "mall": 0.0, "hbm": 0.0}
    for name, pp, rr, owner in (("normalVelocity", pu, ru, patch_of_edge), ("layerThickness", ph, rh, patch_of_cell)):
        q = relabel[owner[rr]]
        pp = relabel[pp]
        d = np.abs(pp - q)
        same_xcd = (pp // chunk) == (q // chunk)
        own = d == 0
        in_l2 = ~own & same_xcd & (d <= w_l2)
        # a row first touched on another XCD, or further back than the L2 holds, is still in the Infinity Cache when the
        # traffic of all 8 XCDs between the two touches fits it: same position inside the chunks (XCDs advance together)
        dpos = np.abs(pp % chunk - q % chunk)
        in_mall = ~own & ~in_l2 & (dpos <= w_mall)
        hbm = ~own & ~in_l2 & ~in_mall
        n = pp.size
        print(f"  {name:15s} {n / nP:6.1f} distinct rows per patch: own {own.mean():.3f}, owner within the L2 window {in_l2.mean():.3f}, "
              f"within the Infinity-Cache window {in_mall.mean():.3f}, beyond (HBM again) {hbm.mean():.4f}")
        for k, msk in (("own", own), ("l2", in_l2), ("mall", in_mall), ("hbm", hbm)):
            acc[k] += msk.sum() * rowB
        tot_bytes += n * rowB
    # timed L2 estimate (see the module docstring): lifetime T of a workgroup and L2 turnover tau, both in units of T
    T_us = {"tendency": 19.0, "rk_stage": 31.0}[launch]             # nPatches / 1024 resident x T = launch time (1.18 / 1.96 ms)
    tau = (L2 / (bpp * 128 / (T_us * 1e-6))) / (T_us * 1e-6)         # L2 bytes / (bytes per second through one XCD), in lifetimes
    est_l2 = 0.0
    for pp, rr, owner in ((pu, ru, patch_of_edge), (ph, rh, patch_of_cell)):
        q = relabel[owner[rr]]
        pp = relabel[pp]
        other = (q != pp) & ((pp // chunk) == (q // chunk))
        lag = (q[other] - pp[other]) / 128.0                        # start of the owner relative to the gatherer, in lifetimes
        # P(|phase - lag| < tau), phase ~ U(0.1, 1)
        lo, hi = np.maximum(lag - tau, 0.1), np.minimum(lag + tau, 1.0)
        est_l2 += (np.clip(hi - lo, 0.0, None) / 0.9).sum() * rowB
    refetch = acc["l2"] + acc["mall"] + acc["hbm"]
    print(f"  timed L2 estimate: workgroup life {T_us:.0f} us, L2 turnover {tau * T_us:.1f} us -> {est_l2 / 1e9:.2f} GB of the {refetch / 1e9:.2f} GB hit the L2, "
          f"{(refetch - est_l2) / 1e9:.2f} GB cross the fabric again (measured for the tendency launch: ~1.3 GB)")
    print(f"  of those, at most {acc['hbm'] / 1e9:.3f} GB ({acc['hbm'] / refetch:.1%} of the gathered rows) have an owner beyond the Infinity-Cache window: "
          f"the re-fetched rows are served by the Infinity Cache, not by HBM")
    print(f"  rows of other patches per launch: {refetch / 1e9:.2f} GB gathered; upper bounds by distance: {acc['l2'] / 1e9:.2f} GB can still be "
          f"in the L2, {acc['mall'] / 1e9:.2f} GB in the Infinity Cache, {acc['hbm'] / 1e9:.3f} GB only in HBM")
plan.close()
