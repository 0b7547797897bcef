import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from radtxfr_amd import _lib, dist as rdist, engine, synthetic
N, NL = 5500000, 32
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(NL); p_atm = a["Ps"] / 101325.0
tile = int(_lib.load().rtx_voigt_tile_points())
g_full = engine.Grid(500.0, 6000.0, N); step = g_full.step
reach = engine.max_wing_cm(full, a["Ts"], p_atm) + step
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def time_shard(off, n):
    grid = g_full.shard(off, n)
    sub = synthetic.subset_table(full, grid.x_at(0) - reach, grid.x_at(n - 1) + reach)
    lines = engine.LineTable(sub)
    pipes = engine.TudPipelines(lines, grid, a["Zs"], n_layers=NL, n_pipes=3)
    for _ in range(6): pipes.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); ev[0].record()
        for _ in range(24): pipes.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
        torch.cuda.synchronize(); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / 24)
    pipes.close(); lines.close()
    return float(np.median(ts))
names = ("tile", "reach", "centre", "band_row")
F = np.stack([engine.tile_costs(full, 500.0, step, N, a["Ts"], p_atm, tile, coef={k: (1.0 if k == f else 0.0) for k in names}) for f in names], axis=1)
for world in (8, 4):
    for coef in ((100, 30, 0, 24), (100, 30, 0, 40), (100, 30, 0, 60), (300, 30, 0, 40), (0, 30, 0, 40)):
        o = rdist.tile_aligned_bounds(N, world, tile, F @ np.array(coef, dtype=float))
        t = np.array([time_shard(int(o[r]), int(o[r + 1] - o[r])) for r in range(world)])
        print(f"N={world} coef {coef}: per-rank (3 pipelines) [ms] " + " ".join("%.3f" % v for v in t) + f"  max/mean {t.max()/t.mean():.3f} slowest {t.max():.3f} longest shard {int(np.diff(o).max())}", flush=True)
