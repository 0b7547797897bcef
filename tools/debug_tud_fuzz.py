import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from radtxfr_amd import engine
from oracle import cpu_ref as ref
rng = np.random.default_rng(20261011)
for trial in range(14):
    nL = int(rng.integers(1, 71)); n = int(rng.integers(700, 3000))
    Z = np.sort(rng.uniform(0.0, 60.0, nL)); T = rng.uniform(190.0, 310.0, nL)
    lo = float(rng.uniform(500.0, 5500.0)); grid = engine.Grid(lo, lo + 2.0, n); X = grid.axis()
    scale = 10.0 ** rng.uniform(-6.0, 2.0, nL)
    shape = 1.0 + 0.9 * np.sin(rng.uniform(5, 60) * X + rng.uniform(0, 6))
    for _ in range(4):
        c, w = rng.uniform(X[0], X[-1]), 10.0 ** rng.uniform(-3.0, -1.0)
        shape = shape + rng.uniform(10, 3000) * w * w / ((X - c) ** 2 + w * w)
    OD = (scale[:, None] * shape[None, :]).astype(np.float32)
    if trial % 3 == 0: OD[:, rng.integers(0, n, 20)] = 0.0
    nalt = int(rng.integers(1, 4)); alts = rng.uniform(-1.0, 70.0, nalt)
    if trial % 4 == 1: alts = np.array([500.0])
    theta = float(rng.choice([0.0, 0.3, 1.1])); nA = int(rng.choice([1, 2, 7, 30, 33, 40])); ret_od = bool(trial % 5 == 2)
    if trial != 3: continue
    tau, Lu, Ld, _ = engine.tud(torch.as_tensor(OD, device="cuda"), grid, T, Z, Altitudes=alts, theta_r=theta, N_angle=nA, returnOD=ret_od)
    tr, ur, dr = ref.tud_from_od(X, OD.astype(np.float64).T, T, Z, Altitudes=alts, theta_r=theta, N_angle=nA, returnOD=ret_od)
    g = Ld.double().cpu().numpy()
    err = np.abs(g - dr) / np.maximum(np.abs(dr), 1e-3 * np.abs(dr).max())
    i = int(np.argmax(err))
    print("lo", lo, "nL", nL, "alts", alts, "Z", Z, "T", T)
    print("max err", err.max(), "at", i, "X", X[i], "gpu", g[i], "ref", dr[i], "OD col", OD[:, i], "scale", scale)
    print("planck per layer", [float(ref.planckian(np.array([X[i]]), t)[0]) for t in T])
    print("err pct", np.percentile(err, [50, 90, 99, 100]))
