#!/usr/bin/env python3
"""Host-visible cost of the drop-in calls at C3 size (5.5 M wavenumbers x 32 layers): rt.compute_TUD per call (NumPy float64
out: device widening + pinned device-to-host copies included) and rt.compute_TUD_batch per atmosphere (copies of atmosphere k
overlapping the kernels of k+1; with reduce= only the reduced spectra cross PCIe). python tools/time_dropin.py [--mf-scale 1e-3]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import engine, synthetic
from radtxfr_amd import radiative_transfer as rt

ap = argparse.ArgumentParser()
ap.add_argument("--mf-scale", type=float, default=1.0)
ap.add_argument("--batch", type=int, default=8)
args = ap.parse_args()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
a["MFs_VAL"] = a["MFs_VAL"] * args.mf_scale
lines = engine.LineTable(full)  # device-resident table handed to the shim: no per-call fingerprinting of 100 000 rows
kw = dict(DVOUT=0.001, line_table=lines, Altitudes=np.asarray([500]), **a)
ts = []
for it in range(7):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    X, tau, Lu, Ld = rt.compute_TUD(500.0, 6000.0, **kw)
    ts.append((time.perf_counter() - t0) * 1e3)
    del X, tau, Lu, Ld
print("rt.compute_TUD(500, 6000, DVOUT=0.001) per call [ms]:", " ".join("%.1f" % t for t in ts), "-> median of the last 5: %.1f ms" % np.median(ts[2:]))
# the same with the table given as a column dict (the shim fingerprints its columns on every call and keeps the device copy)
kw_d = dict(kw, line_table=full)
ts = []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    X, tau, Lu, Ld = rt.compute_TUD(500.0, 6000.0, **kw_d)
    ts.append((time.perf_counter() - t0) * 1e3)
    del X, tau, Lu, Ld
print("  ... with line_table = column dict, per call [ms]:", " ".join("%.1f" % t for t in ts), "-> median of the last 4: %.1f ms" % np.median(ts[2:]))
rng = np.random.default_rng(0)
atms = [dict(Ts=a["Ts"] + rng.normal(0, 1.0, 32)) for _ in range(args.batch)]
for label, red in (("full spectra", None), ("reduceResolution dX=0.25 on the device", dict(dX=0.25))):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = rt.compute_TUD_batch(500.0, 6000.0, atms, reduce=red, **kw)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"rt.compute_TUD_batch, {args.batch} atmospheres, {label}: {dt / args.batch:.2f} ms per atmosphere "
          f"({5.5e6 * 32 * args.batch / dt / 1e-3:.3e} points/s), output points per spectrum {res[0][0].size}")
    del res
lines.close()
