#!/usr/bin/env python3
"""Throughput of a stream of atmospheres at C3 size with 1, 2 or 3 independent pipelines (each: its own stream, per-(line,
layer) records, optical-depth buffer and outputs; atmosphere k runs on pipeline k mod P). One pipeline runs the four kernels
of a step back to back; with two, the fp64 prologue and the HBM-bound TUD pass of one atmosphere can share the chip with the
VALU-bound line-sum of the next.   python tools/time_pipeline.py [--steps 24]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--pipes", default="1,2,3,1,2", help="pipeline counts to time, in this order")
ap.add_argument("--shard", default="", help="R/N: rank R's cost-weighted tile-aligned shard of N (with its line subset), as bench.py --gpus N runs it")
args = ap.parse_args()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
grid = engine.Grid(500.0, 6000.0, 5500000)
n_pts = 5500000
if args.shard:
    from radtxfr_amd import dist as rdist
    r, nw = (int(t) for t in args.shard.split("/"))
    offs, reach = rdist.tud_shard_plan(full, 500.0, 6000.0, 5500000, a["Ts"], a["Ps"], nw)
    grid = grid.shard(int(offs[r]), int(offs[r + 1] - offs[r]))
    full = synthetic.subset_table(full, grid.x_at(0) - reach, grid.x_at(grid.n - 1) + reach)
    n_pts = grid.n
    print(f"shard {r} of {nw}: {grid.n} points, {full['nu'].size} lines")
lines = engine.LineTable(full)
for P in [int(v) for v in args.pipes.split(",")]:
    streams = [torch.cuda.Stream() for _ in range(P)]
    runs = []
    for s in streams:
        with torch.cuda.stream(s):
            runs.append(engine.TudRunner(lines, grid, a["Zs"], n_layers=32, plan=engine.VoigtPlan(lines, 32, grid.n)))
    def go(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % P]):
                runs[k % P].run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    go(2 * P)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(args.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps * 1e3
    chk = [float(r.tau.double().sum()) for r in runs]
    print(f"{P} pipeline(s): {dt:.3f} ms per atmosphere ({n_pts * 32 / dt / 1e-3:.3e} points/s); tau checksums {chk}", flush=True)
    for r in runs:
        r.plan.close()
    del runs
lines.close()
