#!/usr/bin/env python3
"""Timing build only (tools/build_variant.sh cab256 "-DCUBE_ABLATE=256", RADTXFR_LIB=build/cab256.so): phase stamps of
band_basis_moments_kernel per band (100 MHz ticks -> us)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, sensor, synthetic
lib = _lib.load()
dev = torch.device("cuda")
full = engine.Grid(500.0, 6000.0, 5500000)
i0, i1 = int((755.0 - 500.0) / full.step), int((1325.0 - 500.0) / full.step)
grid = full.shard(i0, i1 - i0)
X = grid.axis()
f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
tau, La, Ld = f32(0.5 + 0.45 * np.sin(X / 13.0)), f32(2.0 + np.cos(X / 29.0)), f32(4.0 + 2.0 * np.sin(X / 7.0))
Xe, em = synthetic.synth_emissivities(n_emis=10)
Q = 4
plan = sensor._cube_plan(grid, Xe, 2, None, Q, None, dev)
nB, nk = plan["X_out"].size, len(Xe)
N, Cb = torch.empty(nB, dtype=torch.float32, device=dev), torch.empty(nB, dtype=torch.float32, device=dev)
M = torch.empty((Q + 1, nB, nk), dtype=torch.float32, device=dev)
jr = torch.empty((nB, 2), dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
for it in range(3):
    _lib.check(lib.rtx_band_basis_moments(0, grid.byref(), p(tau), p(La), p(Ld), p(plan["Xk_d"]), nk, nB, p(plan["c_d"]), p(plan["s_d"]), Q,
                                          plan["coef32"].ctypes.data_as(C.c_void_p), 1.0, p(N), p(Cb), p(M[Q]), p(M), p(jr), None))
    torch.cuda.synchronize()
t1, t2, t3, ns = N.cpu().numpy() / 100, Cb.cpu().numpy() / 100, jr.cpu().numpy()[:, 0] / 100, jr.cpu().numpy()[:, 1]
print("per band [us]: to end of knot count  min/med/max %.2f %.2f %.2f" % (t1.min(), np.median(t1), t1.max()))
print("               to end of first tasks min/med/max %.2f %.2f %.2f" % (t2.min(), np.median(t2), t2.max()))
print("               to end of kernel      min/med/max %.2f %.2f %.2f" % (t3.min(), np.median(t3), t3.max()))
print("intervals per band min/med/max", ns.min(), np.median(ns), ns.max())
o = np.argsort(t3)[-5:]
print("slowest bands:", [(int(b), float(t1[b]), float(t2[b]), float(t3[b]), int(ns[b])) for b in o])
