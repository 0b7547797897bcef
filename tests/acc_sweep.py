"""Accuracy sweep at full C3 size: the 32-layer TUD on the whole 500-6000 cm^-1 grid, compared with the oracle on random
1500-point windows (optical depth of every layer, tau, L-up, L-down). python tests/acc_sweep.py [--windows 6]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import engine, synthetic
from oracle import cpu_ref as ref

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=6)
ap.add_argument("--mf-scale", type=float, default=1.0)
args = ap.parse_args()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
a["MFs_VAL"] = a["MFs_VAL"] * args.mf_scale
lines = engine.LineTable(full)
grid = engine.Grid(500.0, 6000.0, 5500000)
OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
tau, Lu, Ld, _ = engine.tud(OD, grid, a["Ts"], a["Zs"])
X = grid.axis()
rel = lambda x, r: float(np.max(np.abs(x - r) / np.maximum(np.abs(r), 1e-3 * np.max(np.abs(r)) + 1e-300)))
rng = np.random.default_rng(7)
worst = [0, 0, 0, 0]
for w in range(args.windows):
    i0 = int(rng.integers(0, 5500000 - 1500)) if w else 5500000 - 1500  # the first window is the top of the grid
    Xw = X[i0:i0 + 1500]
    sub = synthetic.subset_table(full, Xw[0] - 12.0, Xw[-1] + 12.0)
    ODr = np.stack([ref.layer_od(sub, Xw, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"]) for k in range(32)], 1)
    tr, ur, dr = ref.tud_from_od(Xw, ODr, a["Ts"], a["Zs"])
    e = [rel(OD[:, i0:i0 + 1500].T.double().cpu().numpy(), ODr), float(np.max(np.abs(tau[0, i0:i0 + 1500].double().cpu().numpy() - tr))),
         rel(Lu[0, i0:i0 + 1500].double().cpu().numpy(), ur), rel(Ld[i0:i0 + 1500].double().cpu().numpy(), dr)]
    worst = [max(p, q) for p, q in zip(worst, e)]
    print(f"window {Xw[0]:9.3f}-{Xw[-1]:9.3f} cm^-1: OD {e[0]:.2e}  |dtau| {e[1]:.2e}  Lu {e[2]:.2e}  Ld {e[3]:.2e}", flush=True)
print("worst: OD %.2e  |dtau| %.2e  Lu %.2e  Ld %.2e" % tuple(worst))
