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
