#!/usr/bin/env python3
"""Fuzz of the N-invariance claim: random tile-aligned shards of the C3 axis, each computed from a random SUPERSET of the lines
in reach of it, must reproduce the full-table, full-grid optical depths, tau, L-up and L-down bit for bit -- uniform and
clustered tables (hot-tile split), 32 layers and the 66-layer column (Doppler-dominated upper layers).
    python tools/fuzz_shards.py [--trials 12]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=12)
args = ap.parse_args()
tile = int(_lib.load().rtx_voigt_tile_points())
rng = np.random.default_rng(2026)
bad = 0
for table_kind in ("uniform", "clustered"):
    full = (synthetic.synth_clustered_table if table_kind == "clustered" else synthetic.synth_line_table)(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    for nl, n in ((32, 5500000), (66, 2750000)):
        A = synthetic.load_standard_atmosphere()[:nl]
        a = dict(Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6 * (1e-3 if nl == 66 else 1.0), MFs_ID=np.array([1, 2]))
        grid = engine.Grid(500.0, 6000.0, n)
        lines = engine.LineTable(full)
        OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
        tau, Lu, Ld, _ = engine.tud(OD, grid, a["Ts"], a["Zs"])
        torch.cuda.synchronize()
        reach = engine.max_wing_cm(full, a["Ts"], a["Ps"] / 101325.0) + grid.step
        n_tiles = (n + tile - 1) // tile
        heads = np.argsort(np.histogram(full["nu"], bins=np.linspace(500, 6000, n_tiles + 1))[0])[-3:]  # busiest tiles
        for trial in range(args.trials):
            t0 = int(heads[trial % 3]) - int(rng.integers(0, 6)) if trial < 6 else int(rng.integers(0, n_tiles - 1))
            t0 = max(0, min(t0, n_tiles - 2))
            nt = int(rng.integers(1, 40))
            off, ln = t0 * tile, min(nt * tile, n - t0 * tile)
            sh = grid.shard(off, ln)
            extra = float(rng.uniform(0.0, 30.0))
            sub = synthetic.subset_table(full, sh.x_at(0) - reach - extra, sh.x_at(ln - 1) + reach + extra * float(rng.random()))
            ls = engine.LineTable(sub)
            OD_s = engine.optical_depths(ls, sh, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
            t_s, u_s, d_s, _ = engine.tud(OD_s, sh, a["Ts"], a["Zs"])
            ok = (torch.equal(OD_s, OD[:, off:off + ln]) and torch.equal(t_s[0], tau[0, off:off + ln]) and torch.equal(u_s[0], Lu[0, off:off + ln])
                  and torch.equal(d_s, Ld[off:off + ln]))
            if not ok:
                bad += 1
                print(f"MISMATCH {table_kind} {nl} layers: tiles [{t0}, +{nt}) lines {sub['nu'].size}: max |dOD| {float((OD_s - OD[:, off:off + ln]).abs().max()):.3e}", flush=True)
            ls.close()
        print(f"{table_kind} table, {nl} layers x {n} points: {args.trials} random tile-aligned shards with random line supersets: {'all bit-identical' if bad == 0 else 'MISMATCHES so far: %d' % bad}", flush=True)
        lines.close()
        del OD, tau, Lu, Ld
sys.exit(1 if bad else 0)
