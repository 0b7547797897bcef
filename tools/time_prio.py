#!/usr/bin/env python3
"""Experiment: does running the small kernels of a step (fp64 prologue, tile ranges, TUD) on a HIGH-PRIORITY stream next to the
line-sum of another atmosphere (normal priority) hide more of them than plain round-robin pipelines do?
Per pipeline: records + OD + outputs of its own; per step: hi stream: rtx_line_prep -> event -> lo stream: rtx_voigt_sum ->
event -> hi stream: rtx_tud.   python tools/time_prio.py [--pipes 2] [--steps 32]"""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=32)
args = ap.parse_args()
lib = _lib.load()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
lines = engine.LineTable(full)
grid = engine.Grid(500.0, 6000.0, 5500000)
T, Z = a["Ts"], a["Zs"]
w, p_atm = engine.layer_weights_od(lines.species, T, a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
q, m = engine.species_factors(lines.species, T, weight=w)
keep = [np.ascontiguousarray(x, dtype=np.float64) for x in (T, p_atm, q, w, m)]
kp = [k.ctypes.data_as(C.c_void_p) for k in keep]
vp = C.c_void_p
lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (lowest, highest):", lo_p, hi_p)
for P, prio in ((1, False), (2, False), (3, False), (2, True), (3, True), (4, True)):
    pipes = []
    for _ in range(P):
        plan = engine.VoigtPlan(lines, 32, grid.n)
        OD = torch.empty((32, grid.n), dtype=torch.float32, device="cuda")
        out = [torch.empty((1, grid.n), dtype=torch.float32, device="cuda") for _ in range(2)] + [torch.empty((grid.n,), dtype=torch.float32, device="cuda")]
        s_lo = torch.cuda.Stream()
        s_hi = torch.cuda.Stream(priority=-1) if prio else s_lo
        pipes.append((plan, OD, out, s_lo, s_hi))
    mask = np.ones((1, 32), dtype=np.uint8)
    mu = np.ones(1)
    def step(k):
        plan, OD, out, s_lo, s_hi = pipes[k % P]
        with torch.cuda.stream(s_hi):
            if prio:
                s_hi.wait_stream(s_lo)  # the records are free once the previous line-sum on this pipeline is done
            _lib.check(lib.rtx_line_prep(plan._h, lines._h, grid.byref(), 32, *kp, 1.0, 0.0, 0.0, 50.0, 0.0, 1.0, vp(s_hi.cuda_stream)))
        if prio:
            s_lo.wait_stream(s_hi)
        with torch.cuda.stream(s_lo):
            _lib.check(lib.rtx_voigt_sum(plan._h, grid.byref(), 32, vp(OD.data_ptr()), None, grid.n, vp(s_lo.cuda_stream)))
        if prio:
            s_hi.wait_stream(s_lo)
        with torch.cuda.stream(s_hi):
            _lib.check(lib.rtx_tud(vp(OD.data_ptr()), grid.n, grid.byref(), 32, kp[0], 1, mask.ctypes.data_as(vp), 1, mu.ctypes.data_as(vp), 32, 30, 0,
                                   vp(out[0].data_ptr()), vp(out[1].data_ptr()), vp(out[2].data_ptr()), None, grid.n, vp(s_hi.cuda_stream)))
    for k in range(2 * P):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"{P} pipeline(s), small kernels on a high-priority stream: {prio}: {dt:.3f} ms per atmosphere; tau checksum {float(pipes[0][2][0].double().sum()):.9e}", flush=True)
    for pl in pipes:
        pl[0].close()
    del pipes
lines.close()
