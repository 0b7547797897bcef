import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from radtxfr_amd import synthetic
from radtxfr_amd import radiative_transfer as rt
from oracle import cpu_ref
def rel(x, r): return float(np.max(np.abs(x - r) / np.maximum(np.abs(r), 1e-3 * np.max(np.abs(r)))))
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
for lo, hi, scale in ((1000.0, 1004.0, 1e-3), (2380.0, 2381.0, 3e-4), (700.0, 702.0, 1e-2)):
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    a = synthetic.c3_atmosphere(32); a["MFs_VAL"] = a["MFs_VAL"] * scale
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, Altitudes=np.asarray([500]), **a)
    Xr, tr, ur, dr, ODr = cpu_ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"], return_layers=True)
    od = np.stack([rt.compute_OD(lo, hi, DVOUT=0.001, line_table=sub, T=a["Ts"][k], P=a["Ps"][k], PL=a["PLs"][k], MF_VAL=a["MFs_VAL"][k], MF_ID=a["MFs_ID"])[1] for k in (0, 31)], 1)
    print(os.environ.get("RADTXFR_VOIGT_KERNEL", "scatter"), (lo, hi, scale), "OD", rel(od, ODr[:, [0, 31]]), "tau", float(np.max(np.abs(tau - tr))), "Lu", rel(Lu, ur), "Ld", rel(Ld, dr), "tau range", tr.min(), tr.max())
