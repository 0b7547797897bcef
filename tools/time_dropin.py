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

import gc
slow_gc, _gc_t = [], [0.0]


def _gc_cb(phase, info):  # an unexplained 79 ms call in round 2's record: log every collection that takes more than 1 ms
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    elif (time.perf_counter() - _gc_t[0]) > 1e-3:
        slow_gc.append((time.perf_counter() - _gc_t[0]) * 1e3)


gc.callbacks.append(_gc_cb)
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
# ... and as a named table of the hapi cache (storage2cache_from_columns: numeric columns kept as ndarrays)
from radtxfr_amd import hapi
hapi.storage2cache_from_columns("dropin", full)
kw_n = dict(kw, line_table="dropin")
ts = []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    X, tau, Lu, Ld = rt.compute_TUD(500.0, 6000.0, **kw_n)
    ts.append((time.perf_counter() - t0) * 1e3)
    del X, tau, Lu, Ld
print("  ... with line_table = name of a cached table, per call [ms]:", " ".join("%.1f" % t for t in ts), "-> median of the last 4: %.1f ms" % np.median(ts[2:]))
# a caller that KEEPS its results: the pinned zero-copy blocks are capped, later results land in pageable memory
keep, ts = [], []
from radtxfr_amd import _hostio
for it in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep.append(rt.compute_TUD(500.0, 6000.0, **kw))
    ts.append((time.perf_counter() - t0) * 1e3)
print("  ... results kept by the caller, per call [ms]:", " ".join("%.1f" % t for t in ts), "; pinned bytes lent out: %.0f MB of a %.0f MB cap"
      % (_hostio.pinned_lent_bytes() / 1e6, _hostio.PINNED_RESULT_CAP / 1e6))
del keep
gc.collect()
# the pieces of the pageable path on their own: float32 device rows -> pinned ring (PCIe), pinned -> fresh pageable float64
dev_rows = [torch.rand((1, 5500000), dtype=torch.float32, device="cuda") for _ in range(3)]
st = _hostio.Staging(depth=2)
t_dma, t_wid = [], []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tk = st.stage(dev_rows)
    tk[3].synchronize()
    t1 = time.perf_counter()
    out = st.collect(tk)
    t2 = time.perf_counter()
    t_dma.append((t1 - t0) * 1e3)
    t_wid.append((t2 - t1) * 1e3)
print("pageable path, one C3 result: 66 MB float32 over PCIe into the pinned ring %.2f ms (%.1f GB/s); widening into fresh pageable float64 "
      "(132 MB, %d host threads) %.2f ms" % (np.median(t_dma[2:]), 66.0 / np.median(t_dma[2:]), _hostio._threads()._max_workers, np.median(t_wid[2:])))
del dev_rows, out
rng = np.random.default_rng(0)
atms = [dict(Ts=a["Ts"] + rng.normal(0, 1.0, 32)) for _ in range(args.batch)]
for label, extra in (("full spectra, float64", dict()), ("full spectra, float32 out", dict(out_dtype=np.float32)),
                     ("full spectra, float64, devices=[0, 0]", dict(devices=[0, 0])),
                     ("reduceResolution dX=0.25 on the device", dict(reduce=dict(dX=0.25))),
                     ("reduceResolution dX=0.25, devices=[0, 0]", dict(reduce=dict(dX=0.25), devices=[0, 0]))):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = rt.compute_TUD_batch(500.0, 6000.0, atms, **extra, **kw)
        dt = (time.perf_counter() - t0) * 1e3
    print(f"rt.compute_TUD_batch, {args.batch} atmospheres, {label}: {dt / args.batch:.2f} ms per atmosphere "
          f"({5.5e6 * 32 * args.batch / dt / 1e-3:.3e} points/s), output points per spectrum {res[0][0].size}")
    del res
if slow_gc:
    print("garbage collections longer than 1 ms during this run [ms]:", " ".join("%.1f" % t for t in slow_gc))
lines.close()
