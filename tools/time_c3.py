#!/usr/bin/env python3
"""Developer timing of the C3 stages (HIP events on the launch stream), for A/B-testing builds:
    RADTXFR_LIB=build/libp4.so python tools/time_c3.py [--reps 5] [--n 5500000]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--n", type=int, default=5500000)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--n-alt", type=int, default=1, help="sensor altitudes (the reference's main caller asks for 9)")
ap.add_argument("--shard", type=str, default="", help="R/N: time rank R's wavenumber shard of N, with the FULL line table")
ap.add_argument("--mf-scale", type=float, default=1.0, help="scale the mixing ratios (1e-4: an optically thin column, tau ~ 0.5-1 between lines)")
ap.add_argument("--table", default="uniform", choices=["uniform", "clustered"], help="clustered: band heads of 3000 lines per 0.5 cm^-1, combs, empty stretches (synthetic.synth_clustered_table)")
args = ap.parse_args()
lib = _lib.load()
full = (synthetic.synth_clustered_table if args.table == "clustered" else synthetic.synth_line_table)(synthetic.SEED_C3, 100000, 475.0, 6025.0)
A = synthetic.load_standard_atmosphere()[:args.layers]
atm = dict(Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6 * args.mf_scale, MFs_ID=np.array([1, 2]))
lines = engine.LineTable(full)
grid = engine.Grid(500.0, 6000.0, args.n)
if args.shard:
    from radtxfr_amd import dist
    r, nw = (int(t) for t in args.shard.split("/"))
    off, cnt, _ = dist.shard_bounds(args.n, nw, r)
    grid = grid.shard(off, cnt)  # (full line table: what dist.hsi_cube_from_atmosphere hands every rank)
T, Z = atm["Ts"], atm["Zs"]
w, p_atm = engine.layer_weights_od(lines.species, T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])
qr, mass = engine.species_factors(lines.species, T)
OD = torch.empty((args.layers, grid.n), dtype=torch.float32, device="cuda")
alts = (500,) if args.n_alt == 1 else tuple(np.linspace(Z[2], Z[-1], args.n_alt))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tv, tt = [], []
for it in range(args.reps + 2):
    ev[0].record()
    engine.voigt_sum(lines, grid, T, p_atm, w, out_f32=OD, qratio=qr, mass=mass)
    ev[1].record()
    tau, Lu, Ld, _ = engine.tud(OD, grid, T, Z, Altitudes=alts)
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tv.append(ev[0].elapsed_time(ev[1])); tt.append(ev[1].elapsed_time(ev[2]))
print("%s tile=%d  prep+voigt %.3f ms (min %.3f)  tud %.3f ms  checksum OD %.6e tau %.6e Lu %.6e Ld %.6e" % (
    os.path.basename(_lib.LIB_PATH), lib.rtx_voigt_tile_points(), np.median(tv), np.min(tv), np.median(tt),
    float(OD.double().sum()), float(tau.double().sum()), float(Lu.double().sum()), float(Ld.double().sum())))
# (the drop-in rt.compute_TUD / compute_TUD_batch calls are timed by tools/time_dropin.py)
