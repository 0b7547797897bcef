#!/usr/bin/env python3
"""Where do the rare 80 ms compute_TUD calls come from? The body of rt.compute_TUD at C3 size, 60 times, with HIP events
between the stages (kernels | widening | device-to-host copy) and the allocator's state, printing every call slower than
10 ms. python tools/trace_dropin.py [--new-runner 0|1]"""
import argparse, gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _hostio, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--new-runner", type=int, default=1, help="1: a fresh TudRunner (fresh OD / output tensors) per call, as rt.compute_TUD does")
ap.add_argument("--calls", type=int, default=60)
args = ap.parse_args()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
lines = engine.LineTable(full)
grid = engine.Grid(500.0, 6000.0, 5500000)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
run = None
gcs = []
gc.callbacks.append(lambda ph, info: gcs.append((ph, info["generation"], time.perf_counter())))
for it in range(args.calls):
    res0 = torch.cuda.memory_reserved()
    n_gc = len(gcs)
    t0 = time.perf_counter()
    if run is None or args.new_runner:
        run = engine.TudRunner(lines, grid, a["Zs"], n_layers=32)
    ev[0].record()
    tau, Lu, Ld = run.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    ev[1].record()
    t1 = time.perf_counter()
    got = _hostio.rows_to_pinned_f64([tau, Lu, Ld[None, :]])
    ev[2].record()
    t2 = time.perf_counter()
    got[1].synchronize()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    tot = (t3 - t0) * 1e3
    if tot > 10.0 or it < 3:
        print("call %2d: %.1f ms  host: enqueue kernels %.2f, enqueue copy %.2f, wait %.2f | GPU: kernels %.2f ms, widen+copy %.2f ms | "
              "torch reserved %+d MB, gc events %s" % (it, tot, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, ev[0].elapsed_time(ev[1]),
                                                       ev[1].elapsed_time(ev[2]), (torch.cuda.memory_reserved() - res0) >> 20,
                                                       [(p, g) for p, g, _ in gcs[n_gc:]]), flush=True)
    del got, tau, Lu, Ld
print("done")
lines.close()
