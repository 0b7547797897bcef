#!/usr/bin/env python3
"""CPU-side census of the line-sum's work on the C3 workload, with the nodal kernel's geometry (16-row tiles, near zone
+- 2 rows): per tile and layer, how many candidate lines, how many reach the tile, how many are row-level members (full:
every row of the tile a far row; partial), how many rows go point by point (near zone / window edge / Weideman band).
Pure NumPy on the oracle's line parameters; no GPU.   python tests/count_classes.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref
from radtxfr_amd import synthetic

ROWS, NEAR = 16, 2
TILE = 64 * ROWS
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
atm = synthetic.c3_atmosphere(32)
N = 5500000
step = 5500.0 / (N - 1)
nu = full["nu"]
ic = np.rint((nu - 500.0) / step).astype(np.int64)
for k in (0, 8, 16, 24, 31):
    T, p = atm["Ts"][k], atm["Ps"][k] / 101325.0
    P = cpu_ref.line_params(full, T, p)
    W = np.maximum(50 * P["Gamma0"], 50 * P["GammaD"])
    lo = np.clip(np.ceil((nu - W - 500.0) / step), 0, N).astype(np.int64)
    hi = np.clip(np.floor((nu + W - 500.0) / step) + 1, 0, N).astype(np.int64)
    i0 = np.rint((nu + P["Shift0"] - 500.0) / step).astype(np.int64)
    cte = np.sqrt(np.log(2)) / P["GammaD"]
    y = P["Gamma0"] * cte
    zw = np.where(y < 15, np.ceil((15 - y) / (step * cte)) + 2, 0).astype(np.int64)
    maxhw = int(np.max(np.ceil(W[hi > lo] / step))) + 2
    c = dict(cand=0, reach=0, full=0, part=0, far_rows=0, pp_rows=0, edge_rows=0, band_rows=0, entries=0, edge_only=0)
    nt = 0
    for t in range(100, (N + TILE - 1) // TILE, 97):
        ia, ib = t * TILE, min((t + 1) * TILE, N)
        c["cand"] += int(((ic >= ia - maxhw) & (ic <= ib - 1 + maxhw)).sum())
        idx = np.nonzero((hi > ia) & (lo < ib))[0]
        rows = np.arange(ROWS)[None, :]
        lo_t, hi_t = np.maximum(lo[idx] - ia, 0)[:, None], np.minimum(hi[idx] - ia, ib - ia)[:, None]
        reach = (rows * 64 < hi_t) & (rows * 64 + 64 > lo_t)
        inside = (rows * 64 >= lo_t) & (rows * 64 + 64 <= hi_t)
        rc = ((i0[idx] - ia) >> 6)[:, None]
        zl, zh = ((i0[idx] - zw[idx] - ia) >> 6)[:, None], ((i0[idx] + zw[idx] - ia) >> 6)[:, None]
        has_band = (zw[idx] > 0)[:, None]
        band = has_band & (rows >= zl) & (rows <= zh)
        near = (rows >= np.minimum(rc - NEAR, zl)) & (rows <= np.maximum(rc + NEAR, zh))
        far = inside & ~near
        pp = inside & near & ~band
        edge = reach & ~inside & ~band
        bd = reach & band
        ent = (pp | edge | bd).any(1)
        c["reach"] += idx.size
        c["full"] += int(far.all(1).sum())
        c["part"] += int((far.any(1) & ~far.all(1)).sum())
        c["far_rows"] += int(far.sum()); c["pp_rows"] += int(pp.sum()); c["edge_rows"] += int(edge.sum()); c["band_rows"] += int(bd.sum())
        c["entries"] += int(ent.sum())
        c["edge_only"] += int((ent & ~pp.any(1) & ~bd.any(1)).sum())
        nt += 1
    print("layer %2d (p = %.3f atm), per tile: %s" % (k, p, "  ".join("%s %.1f" % (a, b / nt) for a, b in c.items())))
