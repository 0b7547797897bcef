#!/usr/bin/env python3
"""Balance of the wavenumber shards across ranks, measured on ONE GPU by running every rank's shard in turn (C3: 5.5 M
wavenumbers x 32 layers, each shard with the line subset its rank would hold).

1. calibration: the step (prologue + line-sum + TUD) timed on 32 equal contiguous chunks; a non-negative least-squares
   fit of the four per-tile features of engine.tile_costs (tiles, lines in reach, line centres, Weideman band rows) to
   those times -> the coefficients engine.TILE_COST should hold.
2. for N = 2, 4, 8: per-rank step time with equal tile counts and with the cost-weighted cut (dist.tud_shard_plan):
   max / mean is what an N-GPU step loses to imbalance.

    python tools/shard_balance.py [--table clustered]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, dist as rdist, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--table", default="uniform", choices=["uniform", "clustered"])
ap.add_argument("--chunks", type=int, default=32)
args = ap.parse_args()
N, NL = 5500000, 32
if args.table == "clustered":
    full = synthetic.synth_clustered_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
else:
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(NL)
p_atm = a["Ps"] / 101325.0
tile = int(_lib.load().rtx_voigt_tile_points())
g_full = engine.Grid(500.0, 6000.0, N)
step = g_full.step
reach = engine.max_wing_cm(full, a["Ts"], p_atm) + step
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def time_shard(off, n):
    if n == 0:
        return 0.0
    grid = g_full.shard(off, n)
    sub = synthetic.subset_table(full, grid.x_at(0) - reach, grid.x_at(n - 1) + reach)
    lines = engine.LineTable(sub)
    run = engine.TudRunner(lines, grid, a["Zs"], n_layers=NL)
    for _ in range(2):
        run.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(4):
            run.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / 4)
    lines.close()
    return float(np.median(ts))


feat_names = ("tile", "reach", "centre", "band_row")
F = np.stack([engine.tile_costs(full, 500.0, step, N, a["Ts"], p_atm, tile, coef={k: (1.0 if k == f else 0.0) for k in feat_names})
              for f in feat_names], axis=1)  # [n_tiles][4]
offs = rdist.tile_aligned_bounds(N, args.chunks, tile)
t_chunk = np.array([time_shard(int(offs[c]), int(offs[c + 1] - offs[c])) for c in range(args.chunks)])
Fc = np.stack([F[offs[c] // tile:(offs[c + 1] + tile - 1) // tile].sum(axis=0) for c in range(args.chunks)])
print("chunk times [ms]:", " ".join("%.3f" % t for t in t_chunk))
print("sum of chunk times %.3f ms" % t_chunk.sum())
from scipy.optimize import nnls
# a chunk's time = fixed launch cost + its tiles' cost: fit the fixed part too, then drop it (it does not move a cut)
A = np.concatenate([Fc, np.ones((args.chunks, 1))], axis=1)
scale = A.max(axis=0)
c, res = nnls(A / scale, t_chunk)
c = c / scale
fit = A @ c
print("fit residual: rms %.4f ms, max %.4f ms of mean %.3f ms" % (np.sqrt(np.mean((fit - t_chunk) ** 2)), np.max(np.abs(fit - t_chunk)), t_chunk.mean()))
norm = c[1] / 30.0 if c[1] > 0 else 1.0
fitted = {k: float(v / norm) for k, v in zip(feat_names, c[:4])}
print("fitted coefficients (scaled so that reach = 30):", {k: round(v, 2) for k, v in fitted.items()}, " fixed per launch %.3f ms" % c[4])
print("built-in engine.TILE_COST:", engine.TILE_COST)
print("(the fit is ill-conditioned -- equal chunks have equal tile counts, and centres / band rows rise together with the wavenumber --")
print(" so its coefficients move from run to run; the built-in ones were picked from the first fit and are the ones checked below)")
for world in (2, 4, 8):
    cuts = {"equal tiles": rdist.tile_aligned_bounds(N, world, tile),
            "weighted (engine.TILE_COST)": rdist.tile_aligned_bounds(N, world, tile, F @ np.array([engine.TILE_COST[k] for k in feat_names]))}
    for name, o in cuts.items():
        t = np.array([time_shard(int(o[r]), int(o[r + 1] - o[r])) for r in range(world)])
        print(f"N={world} {name:28s}: per-rank step [ms] " + " ".join("%.3f" % v for v in t) +
              f"  max/mean {t.max() / t.mean():.3f}  (slowest rank {t.max():.3f} ms -> {N * NL / t.max() / 1e-3:.3e} points/s before the all-gather)")
