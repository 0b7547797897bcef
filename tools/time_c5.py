#!/usr/bin/env python3
"""Config C5 at full size: 256x256-pixel HSI cube x 256 MAKO bands (resFactor=2) from monochromatic
tau/La/Ld on the MAKO span of the C3 grid (570 k wavenumbers). python tools/time_c5.py [--reps 5]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, sensor, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
_lib.load()
dev = torch.device("cuda")
full = engine.Grid(500.0, 6000.0, 5500000)
i0, i1 = int((755.0 - 500.0) / full.step), int((1325.0 - 500.0) / full.step)
grid = full.shard(i0, i1 - i0)
X = grid.axis()
f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
tau, La, Ld = f32(0.5 + 0.45 * np.sin(X / 13.0)), f32(2.0 + np.cos(X / 29.0)), f32(4.0 + 2.0 * np.sin(X / 7.0))
Xe, em = synthetic.synth_emissivities(n_emis=2000)
sc = synthetic.synth_scene()
E = f32(em[:, sc["end_idx"]])
kidx, frac, T = torch.as_tensor(sc["kidx"], device=dev), f32(sc["frac"]), torch.as_tensor(sc["T"], device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
t = []
for it in range(args.reps + 1):
    ev[0].record()
    xo, cube = sensor.hsi_cube(grid, tau, La, Ld, Xe, E, kidx, frac, T, resFactor=2)
    ev[1].record()
    torch.cuda.synchronize()
    if it:
        t.append(ev[0].elapsed_time(ev[1]))
ms = float(np.median(t))
print(f"C5 cube {tuple(cube.shape)} from nX={grid.n}: {ms:.3f} ms (incl. host-side band set-up) -> "
      f"{cube.numel()/ms/1e-3:.3e} band*pixel values/s, {grid.n*cube.shape[1]/ms/1e-3:.3e} equivalent wavenumber*pixel points/s; "
      f"finite {bool(torch.isfinite(cube).all())} checksum {float(cube.double().sum()):.6e}")

# ---- end to end (round 2): line table + atmosphere -> TUD on the MAKO span -> cube, the driver the 8-GPU run uses
from radtxfr_amd import dist as rdist
import time
table = synthetic.subset_table(synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0), 740.0, 1340.0)
a = synthetic.c3_atmosphere(32)
lines = engine.LineTable(table)
te = []
for it in range(args.reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xo2, cube2 = rdist.hsi_cube_from_atmosphere(755.0, 1325.0, 0.001, lines, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"] * 3e-3, a["MFs_ID"],
                                                Xe, E, kidx, frac, T, resFactor=2)
    torch.cuda.synchronize()
    if it:
        te.append((time.perf_counter() - t0) * 1e3)
print(f"C5 end to end on one GPU (dist.hsi_cube_from_atmosphere: prologue + line-sum + TUD on 570 000 wavenumbers x 32 layers, band moments, "
      f"pixel cube {tuple(cube2.shape)}): {np.median(te):.2f} ms wall per scene; finite {bool(torch.isfinite(cube2).all())} checksum {float(cube2.double().sum()):.6e}")
lines.close()
