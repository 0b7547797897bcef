#!/usr/bin/env python3
"""Cross-section generation at the reference caller's own settings (misc/RT_gen_AbsXS_files.py:15-18, 86-92):
absorptionCoefficient_SDVoigt, WavenumberStep 0.0025, WavenumberRange [400, 7100], WavenumberWingHW 350 -- 2.68 M points,
wings of +-25 ... 38 cm^-1 at one atmosphere. Times afit_xs.cross_section_grid per (T, p) state for a synthetic table with the
C3 line density (18 lines per cm^-1), without speed-dependence columns (fp32 Voigt line-sum) and with SD_air (fp64 path).
    python tools/time_xs.py [--states 4]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import afit_xs, hapi, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--states", type=int, default=4)
ap.add_argument("--lines", type=int, default=122000)
args = ap.parse_args()
X = np.linspace(400.0, 7100.0, int(round(6700.0 / 0.0025)) + 1)
tbl = dict(synthetic.synth_line_table(2016, args.lines, 360.0, 7140.0))
T = np.linspace(220.0, 320.0, args.states)
for label, sd in (("no SD columns (Voigt line-sum, fp32)", False), ("SD_air 0.05-0.2 (speed-dependent Voigt, fp64)", True)):
    t = dict(tbl)
    if sd:
        t["SD_air"] = np.round(np.random.default_rng(5).uniform(0.05, 0.2, args.lines), 3)
    hapi.storage2cache_from_columns("xs", t)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xs = afit_xs.cross_section_grid("xs", T, [1.0], X, WavenumberWingHW=350.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    # the device part alone (prologue + line-sum of all states in one launch), HIP events
    from radtxfr_amd import engine
    lines = hapi._device_table(["xs"])
    grid = engine.Grid.from_axis(X)
    w = np.array([[hapi.abundance(*mi) / hapi.abundance(*mi)] * args.states for mi in lines.species])
    dev = torch.empty((args.states, grid.n), dtype=torch.float64, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        ev[0].record()
        engine.voigt_sum(lines, grid, T, np.ones(args.states), w, out_f64=dev, omega_wing_hw=350.0, scale=2.0 ** 70, profile=3 if sd else 0)
        ev[1].record()
        torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / args.states)
    print(f"{label}: {X.size} points x {args.lines} lines, {args.states} states at 1 atm, HW 350: device {np.median(ts):.2f} ms per (T, p) state "
          f"(HIP events, prologue + line-sum); {dt * 1e3 / args.states:.1f} ms per state through afit_xs.cross_section_grid incl. the "
          f"device-to-host copy of {X.size * 8 / 1e6:.0f} MB per state; checksum {xs.sum():.6e}", flush=True)
