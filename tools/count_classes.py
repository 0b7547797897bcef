#!/usr/bin/env python3
"""CPU-side census of the line-sum's work on the C3 workload: for a few layers, how many (line, tile) pairs fall in
each level of the nodal kernel (tile level / row level / point-by-point entries), how many rows of each kind, and how
full the 8-member groups are. Pure NumPy on the oracle's line parameters; no GPU. python tools/count_classes.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref
from radtxfr_amd import synthetic

ROWS, NEAR = 20, 3
TILE = 64 * ROWS
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
atm = synthetic.c3_atmosphere(32)
N = 5500000
step = 5500.0 / (N - 1)
tot = {}
for k in (0, 8, 16, 24, 31):
    T, p = atm["Ts"][k], atm["Ps"][k] / 101325.0
    P = cpu_ref.line_params(full, T, p)
    nu = full["nu"]
    W = np.maximum(50 * P["Gamma0"], 50 * P["GammaD"])
    lo = np.clip(np.ceil((nu - W - 500.0) / step), 0, N).astype(np.int64)
    hi = np.clip(np.floor((nu + W - 500.0) / step) + 1, 0, N).astype(np.int64)
    i0 = np.rint((nu + P["Shift0"] - 500.0) / step).astype(np.int64)
    cte = np.sqrt(np.log(2)) / P["GammaD"]
    y = P["Gamma0"] * cte
    zw = np.where(y < 15, np.ceil((15 - y) / (step * cte)) + 2, 0).astype(np.int64)
    ok = hi > lo
    c = dict(pairs_reach=0, T=0, R=0, R_rows=0, P=0, pp_rows=0, band_rows=0, edge_rows=0, groupsT=0, groupsR=0, cand=0, rounds=0)
    t_first, t_last = lo // TILE, (hi - 1) // TILE
    maxhw = int(np.max(np.ceil(W[ok] / step))) + 2
    ic = np.rint((nu - 500.0) / step).astype(np.int64)
    ntile = (N + TILE - 1) // TILE
    # sample every 16th tile
    for t in range(0, ntile, 16):
        ia, ib = t * TILE, min((t + 1) * TILE, N)
        cand = np.nonzero((ic >= ia - maxhw) & (ic <= ib - 1 + maxhw))[0]
        c["cand"] += cand.size
        c["rounds"] += (cand.size + 255) // 256
        m = cand[ok[cand] & (hi[cand] > ia) & (lo[cand] < ib)]
        c["pairs_reach"] += m.size
        dlo, dhi = lo[m] - ia, hi[m] - ia
        nt = ib - ia
        lo_t, hi_t = np.maximum(dlo, 0), np.minimum(dhi, nt)
        r_lo, r_hi = lo_t >> 6, (hi_t + 63) >> 6
        c0 = (lo_t + 63) >> 6
        c1 = np.where(dhi < nt, hi_t >> 6, r_hi)
        rc = (i0[m] - ia) >> 6
        zl, zh = i0[m] - zw[m] - ia, i0[m] + zw[m] - ia
        hasz = (zw[m] > 0) & (zh >= 0) & (zl < TILE)
        z0 = np.where(hasz, np.maximum(zl, 0) >> 6, ROWS)
        z1 = np.where(hasz, np.minimum(zh >> 6, ROWS - 1), -1)
        n0 = np.minimum(rc - NEAR, zl >> 6)
        n1 = np.maximum(rc + NEAR, zh >> 6)
        is_t = (lo[m] <= ia) & (dhi >= nt) & ((n1 < 0) | (n0 >= ROWS))
        rows = np.arange(ROWS)[None, :]
        reach = (rows >= r_lo[:, None]) & (rows < r_hi[:, None])
        inn = (rows >= c0[:, None]) & (rows < c1[:, None])
        near = (rows >= n0[:, None]) & (rows <= n1[:, None])
        band = (rows >= z0[:, None]) & (rows <= z1[:, None])
        far = inn & ~near & ~is_t[:, None]
        bd = reach & band
        pp = ((reach & near) | (reach & ~inn)) & ~band
        isR = far.any(1)
        isP = (pp | bd).any(1)
        c["T"] += int(is_t.sum()); c["R"] += int(isR.sum()); c["R_rows"] += int(far.sum()); c["P"] += int(isP.sum())
        c["pp_rows"] += int(pp.sum()); c["band_rows"] += int(bd.sum()); c["edge_rows"] += int((reach & ~inn & ~band).sum())
        # groups per wave-round: members among the wave's lanes (slot = rng.x + wave + 4 lane + 256 round)
        pos = np.searchsorted(cand, m)  # index within candidate list
        for w in range(4):
            for rd in range((cand.size + 255) // 256):
                sel = (pos % 4 == w) & (pos // 256 == rd)
                c["groupsT"] += (int(is_t[sel].sum()) + 7) // 8
                c["groupsR"] += (int(isR[sel].sum()) + 7) // 8
                fr = far[sel & isR]
                for g0 in range(0, fr.shape[0], 8):
                    c["union_rows"] = c.get("union_rows", 0) + int(fr[g0:g0 + 8].any(0).sum())
    scale = 16
    print(f"layer {k:2d} (p={p:.3f} atm): per tile: cand {c['cand']*1.0/ (ntile/16):.0f} reach {c['pairs_reach']/(ntile/16):.0f} "
          f"T {c['T']/(ntile/16):.1f} R {c['R']/(ntile/16):.1f} (rows/R {c['R_rows']/max(c['R'],1):.1f}) P {c['P']/(ntile/16):.1f} "
          f"pp_rows {c['pp_rows']/(ntile/16):.1f} (edge {c['edge_rows']/(ntile/16):.1f}) band_rows {c['band_rows']/(ntile/16):.1f} "
          f"groupsT {c['groupsT']/(ntile/16):.1f} groupsR {c['groupsR']/(ntile/16):.1f} (union rows/group {c.get('union_rows',0)/max(c['groupsR'],1):.1f}) rounds {c['rounds']/(ntile/16):.2f}")
    for kk, v in c.items():
        tot[kk] = tot.get(kk, 0) + v * scale * 32 / 5
print("C3 totals (x32 layers, extrapolated):", {k: f"{v:.3g}" for k, v in tot.items()})
