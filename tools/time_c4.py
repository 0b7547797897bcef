#!/usr/bin/env python3
"""Config C4 at full size: 2000 emissivities x the MAKO span of the C3 grid (560 k wavenumbers)
x triangle ILS -> (128, 2000). Times the three streaming kernels with HIP events and prints their
algorithmic HBM rates.   python tools/time_c4.py [--nE 2000] [--reps 5]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, sensor, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--nE", type=int, default=2000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
_lib.load()
dev = torch.device("cuda")
# MAKO span on the C3 grid: 500-6000 cm^-1, 5.5 M points -> indices covering 755..1325 cm^-1
full = engine.Grid(500.0, 6000.0, 5500000)
i0, i1 = int((755.0 - 500.0) / full.step), int((1325.0 - 500.0) / full.step)
grid = full.shard(i0, i1 - i0)
X = grid.axis()
f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
tau, La, Ld = f32(0.5 + 0.45 * np.sin(X / 13.0)), f32(2.0 + np.cos(X / 29.0)), f32(4.0 + 2.0 * np.sin(X / 7.0))
Xe, em = synthetic.synth_emissivities(n_emis=args.nE)
em_d = f32(em)
nX, nE = grid.n, args.nE
X_d = torch.as_tensor(X, device=dev)
Ts_d = torch.as_tensor(np.array([287.87]), device=dev)
X_out, centre, sigma = sensor.mako_bands(X[0], X[-1])
c_d, s_d = torch.as_tensor(centre, device=dev), torch.as_tensor(sigma, device=dev)
col = lambda v: v.reshape(-1, 1).contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
t = [[], [], []]
for it in range(args.reps + 1):
    ev[0].record()
    em_hi = sensor.interp_knots(grid, Xe, em_d)
    ev[1].record()
    L, _ = engine.apparent_radiance(X_d, em_hi, Ts_d, col(tau), col(La), col(Ld))
    ev[2].record()
    out = engine.ils(0, L.reshape(nX, nE), c_d, s_d, grid=grid)
    ev[3].record()
    torch.cuda.synchronize()
    if it:
        for k in range(3):
            t[k].append(ev[k].elapsed_time(ev[k + 1]))
    del em_hi, L
tf = []
for it in range(args.reps + 1):
    ev[0].record()
    xo_f, out_f = sensor.band_radiance_fused(grid, tau, La, Ld, Xe, em_d, 287.87)
    ev[1].record()
    torch.cuda.synchronize()
    if it:
        tf.append(ev[0].elapsed_time(ev[1]))
err = float(((out_f - out).abs() / out.abs().clamp_min(1e-3 * float(out.abs().max()))).max())
print(f"C4 fused (band_moments + band_mix, incl. host-side band/knot uploads): {np.median(tf):.3f} ms -> "
      f"{nX*nE/np.median(tf)/1e-3:.3e} spectrum points/s; max rel diff vs unfused {err:.2e}")
ms = [float(np.median(v)) for v in t]
B = 4.0 * nX * nE
support = float(np.sum(2 * sigma) / grid.step)  # grid points under all triangles
print(f"C4 nX={nX} nE={nE} nB={X_out.size}: interp {ms[0]:.3f} ms ({B/ms[0]/1e6:.0f} GB/s written), "
      f"radiance {ms[1]:.3f} ms ({2*B/ms[1]/1e6:.0f} GB/s r+w), "
      f"ils {ms[2]:.3f} ms (input once {B/ms[2]/1e6:.0f} GB/s; points under triangles {support/nX:.2f}x -> {support*4*nE/ms[2]/1e6:.0f} GB/s touched), "
      f"total {sum(ms):.3f} ms -> {nX*nE/sum(ms)/1e-3:.3e} spectrum points/s; checksum {float(out.double().sum()):.6e}")
